ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() { env "$1" python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --filter $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 $1', round(d['roofline']['achieved']), d['checksum'])"; }
python3 $ROOT/bench.py --no-cpu-baseline --filter gauss 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('box: gauss', round(d['roofline']['achieved']), 'copy ceiling', round(d['roofline']['copy_ceiling_GBs']))"
for r in 1 2; do for v in 0 4 8 16 32; do run MI355_TUNE_GRAY_STRIP=$v gray; done; for v in 0 4 8; do run MI355_TUNE_GRAY_STRIP=$v gray1; done; done
