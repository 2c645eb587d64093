ROOT=${GRAFT_REPO_ROOT:-/root/repo}
# the MI355_TUNE_* overrides are only compiled into the tune build (csrc/Makefile, `make tune`)
export MI355_IMGFILTER_LIB=${MI355_IMGFILTER_LIB:-${GRAFT_REPO_ROOT:-/root/repo}/tools/lib/libmi355_imgfilter_tune.so}
run() { env "$1" python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --filter sobel $2 $3 $4 $5 $6 $7 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sobel $2 $3 $4 $5 $6 $7 $1', round(d['roofline']['achieved']), d['checksum'])"; }
for a in "" "--frames 8 --steps 200" "--frames 1 --steps 300" "--width 1920 --height 1080" "--width 640 --height 512 --frames 8000" "--width 1000 --height 1000 --frames 2000"; do
  for r in 1 2; do run MI355_TUNE_SOBEL_STRIP=0 $a; run MI355_TUNE_SOBEL_STRIP=1 $a; done
done
