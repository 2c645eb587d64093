ROOT=${GRAFT_REPO_ROOT:-/root/repo}
row() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --no-side-figures --pool-candidates 1 --steps 20 --warmup 5 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-75s %6.0f GB/s  %5.1f %%  parity max %s mism %.2e' % ('$*', r['achieved'], 100*r['frac'], d['parity']['max_abs_diff'], d['parity']['mismatch_frac']))"; }
for k in "7 --sigma 2.0" "9 --sigma 2.5"; do for ra in "" "--random-alpha"; do for impl in tile_skip auto mfma; do
  [ $impl = tile_skip ] && continue
  row --filter gauss --k $k --frames 256 --impl $impl $ra
done; done; done
row --filter gauss --k 9 --sigma 2.5 --width 1920 --height 1080 --frames 256 --impl auto
row --filter gauss --k 9 --sigma 2.5 --width 1920 --height 1080 --frames 256 --impl mfma
row --filter gauss --k 9 --sigma 2.5 --frames 8 --steps 100 --impl auto
row --filter gauss --k 9 --sigma 2.5 --frames 8 --steps 100 --impl mfma
row --filter gauss --k 9 --sigma 2.5 --frames 1 --steps 200 --impl auto
row --filter gauss --k 9 --sigma 2.5 --frames 1 --steps 200 --impl mfma
