#!/bin/bash
# A/B on one box: matrix-core Gaussian with register staging (gauss_mfma_reg.hip, the default) vs LDS-DMA staging
# (gauss_mfma_dma.hip, MI355_MFMA_DMA=1); tuning build
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export MI355_IMGFILTER_LIB=$ROOT/tools/lib/libmi355_imgfilter_tune.so
row() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --no-side-figures --pool-candidates 1 --steps 30 --warmup 10 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-4s %-62s %6.0f GB/s  %5.1f %%  %7.3f ms  parity max %s' % ('$TAG', '$*', r['achieved'], 100*r['frac'], r['avg_launch_ms'], d['parity']['max_abs_diff']))"; }
for rep in 1 2; do
for v in reg dma; do
  if [ $v = dma ]; then export MI355_MFMA_DMA=1; else unset MI355_MFMA_DMA; fi
  TAG=$v
  row --filter gauss --k 17 --sigma 6 --frames 256
  row --filter gauss --k 17 --sigma 6 --frames 256 --random-alpha
  row --filter gauss --k 7 --sigma 2 --frames 256
  row --filter gauss --k 17 --sigma 6 --width 1920 --height 1080 --frames 256
  row --filter gauss --k 17 --sigma 6 --frames 8 --steps 100
done; done
