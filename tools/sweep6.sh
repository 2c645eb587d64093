#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ceiling "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f GB/s  %7.3f ms' % (d['roofline']['achieved'], d['roofline']['avg_launch_ms']))"; }
echo -n "frames 64: "; run
echo -n "frames 64 alloc 256: "; run --alloc-frames 256
echo -n "frames 64 alloc 512: "; run --alloc-frames 512
echo -n "frames 256: "; run --frames 256
echo -n "frames 64 steps 100: "; run --steps 100
echo -n "frames 16 steps 100: "; run --frames 16 --steps 100
echo -n "frames 16 alloc 256 steps 100: "; run --frames 16 --steps 100 --alloc-frames 256
