#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f GB/s  %7.3f ms  %.0f Mpx/s' % (d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['value']))"; }
for big in 8 12 16 20 24 32 40 48; do echo -n "sobel big=$big tail=0: "; MI355_TUNE_BAND_ROWS=$big MI355_TUNE_TAIL_FRAC=0 run --filter sobel; done
for b in 65536 262144 1048576; do echo -n "gray blocks=$b: "; MI355_TUNE_GRAY_BLOCKS=$b run --filter gray; done
for big in 48 64 96; do echo -n "gauss big=$big tail=0.1: "; MI355_TUNE_BAND_ROWS=$big run; done
echo -n "gauss k=3: "; run --k 3; echo -n "gauss k=7: "; run --k 7
