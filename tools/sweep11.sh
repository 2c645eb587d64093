#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --frames 64 --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f GB/s  %7.3f ms  %.0f Mpx/s' % (d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['value']))"; }
for k in 9 11 13 17 25 31; do echo -n "gauss k=$k (sigma=k/3): "; run --k $k --sigma $(python3 -c "print($k/3.0)"); done
echo -n "gauss k=5 exact mode: "; run --mode exact
echo -n "gauss k=17 exact mode: "; run --k 17 --sigma 6 --mode exact
echo -n "pipeline k=17: "; run --filter pipeline --k 17 --sigma 6
echo -n "gauss 1023x819 k=5 frames 2048 (tile path, odd width): "; run --width 1023 --height 819 --frames 2048
echo -n "sobel 1023x819 frames 2048 (tile path): "; run --filter sobel --width 1023 --height 819 --frames 2048
