#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ceiling "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f GB/s  %7.3f ms' % (d['roofline']['achieved'], d['roofline']['avg_launch_ms']))"; }
echo -n "default (128 / 40 / 0.2): "; run
echo -n "single phase 135: "; MI355_TUNE_BAND_ROWS=135 MI355_TUNE_TAIL_FRAC=0 run
for tf in 0.1 0.2 0.3; do for tr in 24 40 64; do echo -n "big=128 tail_rows=$tr tail_frac=$tf: "; MI355_TUNE_TAIL_ROWS=$tr MI355_TUNE_TAIL_FRAC=$tf run; done; done
echo -n "default again: "; run
echo -n "frames 256 default: "; run --frames 256
