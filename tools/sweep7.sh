#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f GB/s  %7.3f ms  value %.0f' % (d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['value']))"; }
for sw in "20 3" "50 10" "100 20" "200 50" "500 100" "1000 200" "2000 200"; do set -- $sw; echo -n "steps=$1 warmup=$2: "; run --steps $1 --warmup $2; done
echo -n "single phase 135, steps 500: "; MI355_TUNE_BAND_ROWS=135 MI355_TUNE_TAIL_FRAC=0 run --steps 500 --warmup 100
echo -n "frames 256 steps 200: "; run --frames 256 --steps 200 --warmup 20
for f in gray sobel pipeline; do echo -n "$f steps 500: "; run --filter $f --steps 500 --warmup 100; done
