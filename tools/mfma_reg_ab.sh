#!/bin/bash
# A/B of the two matrix-core Gaussian skeletons on one box: LDS-staged (gauss_mfma.hip) vs register-only (gauss_mfma_reg.hip)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export MI355_IMGFILTER_LIB=$ROOT/tools/lib/libmi355_imgfilter_tune.so
row() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --no-side-figures --pool-candidates 1 --steps 20 --warmup 5 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s %-58s %6.0f GB/s  %5.1f %%  %7.3f ms  parity max %s mism %.2e' % ('$TAG', '$*', r['achieved'], 100*r['frac'], r['avg_launch_ms'], d['parity']['max_abs_diff'], d['parity']['mismatch_frac']))"; }
for v in lds reg; do
  if [ $v = lds ]; then export MI355_MFMA_LDS=1; else unset MI355_MFMA_LDS; fi
  TAG=$v
  row --filter gauss --k 17 --sigma 6 --frames 64 --impl mfma
  row --filter gauss --k 17 --sigma 6 --frames 256 --impl mfma
  row --filter gauss --k 11 --sigma 3 --frames 256 --impl mfma
  row --filter gauss --k 5 --frames 256 --impl mfma
  row --filter gauss --k 17 --sigma 6 --frames 256 --impl mfma --random-alpha
  row --filter gauss --k 17 --sigma 6 --width 1920 --height 1080 --frames 256 --impl mfma
done
