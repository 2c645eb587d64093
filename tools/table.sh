#!/bin/bash
# tools/table.sh — one-box table of every kernel's steady-state rate, bench.py per row
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
row() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --no-side-figures "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; p=d.get('parity',{}); print('%-52s %6.0f GB/s  %5.1f %%  %8.0f Mpx/s  %6.3f ms  parity max|d| %s' % ('$*', r['achieved'], 100*r['frac'], d['value'], r['avg_launch_ms'], p.get('max_abs_diff')))"; }
python3 $ROOT/bench.py --no-cpu-baseline --no-parity --filter gray 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('box: streaming ceiling %.0f GB/s (%s)' % (d['roofline']['copy_ceiling_GBs'], d['roofline']['copy_ceiling_kernel']))"
row --filter gray
row --filter gray1
row --filter gauss --k 3
row --filter gauss
row --filter gauss --random-alpha
row --filter gauss --const-alpha 128
row --filter gauss --k 7 --sigma 2.0
row --filter gauss --k 9 --sigma 2.5
row --filter gauss --k 11 --sigma 3.0 --frames 64
row --filter gauss --k 13 --sigma 3.3 --frames 64
row --filter gauss --k 17 --sigma 6 --frames 64
row --filter sobel
# mid-size Sobel launches (8 x 10^7 .. 2^28 pixels: the halo-lane kernel with rows in lock-step), steps scaled to >= 100 ms
row --filter sobel --frames 16 --steps 800 --warmup 80
row --filter sobel --frames 32 --steps 400 --warmup 40
row --filter sobel --width 1920 --height 1080 --frames 64 --steps 800 --warmup 80
row --filter pipeline --k 3
row --filter pipeline
row --filter pipeline --k 7
row --filter gauss --width 1023 --height 819 --frames 2048
row --filter sobel --width 1023 --height 819 --frames 2048
row --filter pipeline --width 1023 --height 819 --frames 2048
row --filter gauss --frames 1 --steps 300
row --filter gauss --frames 8 --steps 200
row --filter gauss --frames 64
# what one rank of an 8-GPU strong-scaling job runs (BASELINE config 5: 64 x 4K through the pipeline), steps scaled to >= 200 ms
row --filter pipeline --frames 64 --steps 400 --warmup 40
row --filter pipeline --frames 128 --steps 200 --warmup 20
row --filter gauss --frames 32 --steps 600 --warmup 60
# BASELINE.json config 2 and friends: 1080p frames (1024 frames = the 4K batches' byte count)
row --filter gauss --width 1920 --height 1080 --frames 1024
row --filter gauss --width 1920 --height 1080 --frames 1024 --random-alpha
row --filter sobel --width 1920 --height 1080 --frames 1024
row --filter pipeline --width 1920 --height 1080 --frames 1024
row --filter gauss --width 1920 --height 1080 --frames 1
row --filter gauss --k 17 --sigma 6 --width 1920 --height 1080 --frames 256
# config 5, N = 1 leg: 512 x 4K through the fused pipeline
row --filter pipeline --total-frames 512
# the matrix-core Gaussian forced at small k, and the VALU kernels forced where AUTO (k >= 9) no longer takes them
row --filter gauss --k 5 --frames 64 --impl mfma
row --filter gauss --k 7 --sigma 2.0 --impl mfma
row --filter gauss --k 9 --sigma 2.5 --impl valu
row --filter gauss --k 11 --sigma 3.0 --frames 64 --impl valu
row --filter gauss --k 17 --sigma 6 --frames 64 --impl valu
row --filter gauss --k 17 --sigma 6 --frames 256
row --filter gauss --k 17 --sigma 6 --frames 256 --random-alpha
# EXACT mode (bit-identical to the CPU path): exact-by-exception sliding kernel (k = 3, 5), tiled kernel (k >= 7)
row --filter gauss --mode exact --k 3 --sigma 0.8
row --filter gauss --mode exact
row --filter gauss --mode exact --random-alpha
row --filter gauss --mode exact --frames 64 --impl tile
row --filter gauss --mode exact --k 7 --sigma 2.0 --frames 64
