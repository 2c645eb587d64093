#!/bin/bash
# A/B of the fused pipeline with 4 (pipe_slide.hip) and 8 (pipe_slide8.hip) pixels per lane; tuning build:
# MI355_PIPE8=1 forces the 8-pixel kernel wherever it applies, MI355_PIPE8=0 forbids it
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export MI355_IMGFILTER_LIB=$ROOT/tools/lib/libmi355_imgfilter_tune.so
row() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --no-side-figures --pool-candidates 1 --warmup 5 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-6s %-62s %6.0f GB/s  %5.1f %%  parity %s  %s' % ('$TAG', '$*', r['achieved'], 100*r['frac'], d['parity']['max_abs_diff'], d['checksum']))"; }
for v in px4 px8; do
  if [ $v = px8 ]; then export MI355_PIPE8=1; else export MI355_PIPE8=0; fi
  TAG=$v
  row --filter pipeline --steps 30
  row --filter pipeline --k 3 --sigma 0.8 --steps 30
  row --filter pipeline --width 1920 --height 1080 --frames 1024 --steps 30
  row --filter pipeline --frames 128 --steps 40
  row --filter pipeline --frames 64 --steps 60
  row --filter pipeline --frames 32 --steps 100
  row --filter pipeline --frames 8 --steps 200
  row --filter pipeline --frames 1 --steps 300
  row --filter pipeline --width 640 --height 512 --frames 4096 --steps 30
done
