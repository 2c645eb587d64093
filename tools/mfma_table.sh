#!/bin/bash
# tools/mfma_table.sh — same box: the matrix-core Gaussian (--impl mfma) beside the library's VALU kernels (--impl auto)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
row() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --no-side-figures --pool-candidates 1 --steps 20 --warmup 5 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-58s %6.0f GB/s  %5.1f %%  %7.3f ms  parity max %s mism %.2e' % ('$*', r['achieved'], 100*r['frac'], r['avg_launch_ms'], d['parity']['max_abs_diff'], d['parity']['mismatch_frac']))"; }
for k in "5 1.5" "7 2.0" "9 2.5" "11 3.0" "13 3.3" "17 6.0"; do
  set -- $k
  row --filter gauss --k $1 --sigma $2 --frames 64 --impl auto
  row --filter gauss --k $1 --sigma $2 --frames 64 --impl mfma
done
row --filter gauss --k 17 --sigma 6 --frames 256 --impl mfma
row --filter gauss --k 17 --sigma 6 --frames 64 --impl mfma --random-alpha
row --filter gauss --k 17 --sigma 6 --frames 64 --impl auto --random-alpha
row --filter gauss --k 17 --sigma 6 --frames 256 --width 1920 --height 1080 --impl mfma
