#!/bin/bash
# tools/content_probe.sh — how much the "exact by exception" kernels depend on frame content: hash noise (mode 0, the bench
# workload), gradient + noise (mode 1), flat frames (mode 2: every window constant, every pixel takes the exception path)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
row() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --no-side-figures --pool-candidates 1 --steps 20 --warmup 5 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-64s %6.0f GB/s  %5.1f %%  %7.3f ms  parity max %s' % ('$*', r['achieved'], 100*r['frac'], r['avg_launch_ms'], d['parity']['max_abs_diff']))"; }
for m in 0 3; do
  row --filter pipeline --synth-mode $m
  row --filter gray --synth-mode $m
  row --filter sobel --synth-mode $m
done
