#!/bin/bash
# tools/table.sh — one-box table of every kernel's steady-state rate, bench.py per row
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
row() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-44s %6.0f GB/s  %5.1f %%  %8.0f Mpx/s  %6.3f ms' % ('$*', r['achieved'], 100*r['frac'], d['value'], r['avg_launch_ms']))"; }
python3 $ROOT/bench.py --no-cpu-baseline --filter gray 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('box: torch D2D copy ceiling %.0f GB/s' % d['roofline']['copy_ceiling_GBs'])"
row --filter gray
row --filter gray1
row --filter gauss --k 3
row --filter gauss
row --filter gauss --random-alpha
row --filter gauss --k 7
row --filter gauss --k 9
row --filter gauss --k 11 --frames 64
row --filter gauss --k 13 --frames 64
row --filter gauss --k 17 --frames 64
row --filter sobel
row --filter pipeline --k 3
row --filter pipeline
row --filter pipeline --k 7
row --filter gauss --width 1023 --height 819 --frames 2048
row --filter sobel --width 1023 --height 819 --frames 2048
row --filter pipeline --width 1023 --height 819 --frames 2048
row --filter gauss --frames 1 --steps 300
row --filter gauss --frames 8 --steps 200
row --filter gauss --frames 64
