#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ceiling "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f GB/s  %7.3f ms' % (d['roofline']['achieved'], d['roofline']['avg_launch_ms']))"; }
for b in 2048 4096 8192 16384 32768 65536; do echo -n "gray blocks=$b: "; MI355_TUNE_GRAY_BLOCKS=$b run --filter gray; done
for b in 2048 16384; do echo -n "gray1 blocks=$b: "; MI355_TUNE_GRAY_BLOCKS=$b run --filter gray1; done
