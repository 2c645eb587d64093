#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ceiling "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f GB/s  %7.3f ms' % (d['roofline']['achieved'], d['roofline']['avg_launch_ms']))"; }
for b in 128 135 68 90 108 128 135; do echo -n "gauss band_rows=$b: "; MI355_TUNE_BAND_ROWS=$b run; done
for f in 64 128 256; do echo -n "gauss frames=$f: "; run --frames $f; done
