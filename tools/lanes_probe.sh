ROOT=${GRAFT_REPO_ROOT:-/root/repo}
# the MI355_TUNE_* overrides are only compiled into the tune build (csrc/Makefile, `make tune`)
export MI355_IMGFILTER_LIB=${MI355_IMGFILTER_LIB:-${GRAFT_REPO_ROOT:-/root/repo}/tools/lib/libmi355_imgfilter_tune.so}
run() { env "$1" python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --filter $2 $3 $4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 $3 $4 $1', round(d['roofline']['achieved']), d['checksum'])"; }
run MI355_TUNE_LANES_OUT=0 sobel
for r in 1 2; do for v in 0 60 52 48 44 40; do run MI355_TUNE_LANES_OUT=$v sobel; done; for v in 0 56 48; do run MI355_TUNE_LANES_OUT=$v gauss; done; for v in 0 56 48; do run MI355_TUNE_LANES_OUT=$v gauss --k 3; done; done
