#!/bin/bash
# band-height sweep of pipe_slide8.hip (tuning build) beside pipe_slide.hip on the same box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export MI355_IMGFILTER_LIB=$ROOT/tools/lib/libmi355_imgfilter_tune.so
row() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --no-side-figures --no-parity --pool-candidates 1 --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-5s rows=%-3s %-40s %6.0f GB/s  %5.1f %%' % ('$TAG', '$ROWS', '$*', r['achieved'], 100*r['frac']))"; }
TAG=px4; ROWS=-; unset MI355_PIPE8; row --filter pipeline; row --filter pipeline --k 3 --sigma 0.8; row --filter pipeline --frames 16 --steps 100
export MI355_PIPE8=1; TAG=px8
for ROWS in 48 72 96 120 144 216; do export MI355_TUNE_PIPE8_ROWS=$ROWS; row --filter pipeline; done
for ROWS in 16 32 48 72; do export MI355_TUNE_PIPE8_ROWS=$ROWS; row --filter pipeline --k 3 --sigma 0.8; done
for ROWS in 24 48 72; do export MI355_TUNE_PIPE8_ROWS=$ROWS; row --filter pipeline --frames 16 --steps 100; done
TAG=px4; ROWS=-; unset MI355_PIPE8; row --filter pipeline
