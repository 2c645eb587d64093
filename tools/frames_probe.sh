ROOT=${GRAFT_REPO_ROOT:-/root/repo}
row() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-44s %6.0f GB/s  %8.0f Mpx/s  %6.3f ms' % ('$*', r['achieved'], d['value'], r['avg_launch_ms']))"; }
for r in 1 2; do
for f in 64 128 192 256 384 512; do row --filter gauss --frames $f; done
for f in 64 128 256 512; do row --filter gray --frames $f; done
for f in 128 256 512; do row --filter gauss --frames $f --steps 200; done
done
