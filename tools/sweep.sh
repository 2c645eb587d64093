#!/bin/bash
# tools/sweep.sh — quick A/B of tuning knobs through bench.py (each line: knob value -> GB/s)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ceiling "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f GB/s  %7.3f ms' % (d['roofline']['achieved'], d['roofline']['avg_launch_ms']))"; }
for b in 32 64 128 256 540 2160; do echo -n "band_rows=$b: "; MI355_TUNE_BAND_ROWS=$b run; done
for f in 16 32 64 128; do echo -n "frames=$f: "; run --frames $f; done
echo -n "1080p frames=256: "; run --width 1920 --height 1080 --frames 256
for k in 3 7 9; do echo -n "k=$k: "; run --k $k; done
echo -n "synth mode 1: "; run --synth-mode 1
