ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() { local t0=$(date +%s.%N); python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling "$@" 2>/tmp/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['roofline']['achieved']), d.get('pool_placement'))" || tail -3 /tmp/err.txt; echo "   wall $(echo "$(date +%s.%N) - $t0" | bc) s"; }
for r in 1 2 3 4 5; do
  run --filter gauss --pool-gap-gb 0
  run --filter gauss
done
