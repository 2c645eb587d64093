#!/bin/bash
# tools/content_probe.sh — how much each kernel's rate depends on frame content (256 x 4K frames, one box):
#   mode 0 hash noise (the bench workload) | mode 1 gradient + noise | mode 2 flat 64 x 64 patches (every window constant:
#   the exact-by-exception kernels' table path) | mode 3 gray noise (r = g = b: the luminance's ambiguous case on every pixel)
#   photo: the reference's own test photographs (decoded pixels, tests/golden) tiled to 4K — Tulips (colour), Artemis (near-gray)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
row() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --no-side-figures --pool-candidates 1 --steps 20 --warmup 5 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-100s %6.0f GB/s  %5.1f %%  %7.3f ms  parity max %s' % ('$*'.replace('$ROOT/tests/golden/',''), r['achieved'], 100*r['frac'], r['avg_launch_ms'], d['parity']['max_abs_diff']))"; }
for f in "gauss" "gauss --mode exact" "sobel" "pipeline" "gray"; do
  for c in "--synth-mode 0" "--synth-mode 1" "--synth-mode 2" "--synth-mode 3" "--photo $ROOT/tests/golden/tulips_medium640_rgb.png" "--photo $ROOT/tests/golden/ref_images/Artemis_medium640_rgb.png"; do
    row --filter $f $c
  done
done
