#!/bin/bash
# strip-width sweep of the sliding-window Gaussian (tuning build): do line-aligned store spans pay for the extra waves?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export MI355_IMGFILTER_LIB=$ROOT/tools/lib/libmi355_imgfilter_tune.so
row() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --no-side-figures --pool-candidates 1 --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('lanes=%-3s %-40s %6.0f GB/s  %5.1f %%' % ('$L', '$*', r['achieved'], 100*r['frac']))"; }
for rep in 1 2; do for L in 60 56 48 40 32; do export MI355_TUNE_LANES_OUT=$L; row --filter gauss; done; done
for L in 60 56 48; do export MI355_TUNE_LANES_OUT=$L; row --filter gauss --k 3 --sigma 0.8; row --filter gauss --mode exact; done
