#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f GB/s  %7.3f ms' % (d['roofline']['achieved'], d['roofline']['avg_launch_ms']))"; }
echo -n "default: "; run
for big in 96 128 160 216; do for tr in 32 48; do for tf in 0.1 0.2; do echo -n "big=$big tail_rows=$tr frac=$tf: "; MI355_TUNE_BAND_ROWS=$big MI355_TUNE_TAIL_ROWS=$tr MI355_TUNE_TAIL_FRAC=$tf run; done; done; done
echo -n "single 128: "; MI355_TUNE_TAIL_FRAC=0 run
echo -n "default: "; run
