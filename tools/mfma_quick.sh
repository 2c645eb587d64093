#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
row() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --no-side-figures --pool-candidates 1 --steps 20 --warmup 5 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-58s %6.0f GB/s  %5.1f %%  %7.3f ms  parity max %s mism %.2e' % ('$*', r['achieved'], 100*r['frac'], r['avg_launch_ms'], d['parity']['max_abs_diff'], d['parity']['mismatch_frac']))"; }
row --filter gauss --k 17 --sigma 6 --frames 64 --impl mfma
row --filter gauss --k 17 --sigma 6 --frames 256 --impl mfma
row --filter gauss --k 9 --sigma 2.5 --frames 64 --impl mfma
