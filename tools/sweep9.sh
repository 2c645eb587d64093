#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f GB/s  %7.3f ms  %.0f Mpx/s' % (d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['value']))"; }
echo -n "gauss default: "; run
for f in sobel pipeline; do
  echo -n "$f default: "; run --filter $f
  for big in 32 64 128 216; do for tf in 0 0.1; do echo -n "$f big=$big tail_frac=$tf: "; MI355_TUNE_BAND_ROWS=$big MI355_TUNE_TAIL_FRAC=$tf run --filter $f; done; done
done
echo -n "gauss frames=64 steps=200: "; run --frames 64 --steps 200 --warmup 40
echo -n "gauss 1080p frames=1024: "; run --width 1920 --height 1080 --frames 1024
echo -n "gray: "; run --filter gray
echo -n "gray1: "; run --filter gray1
