# same-box comparison of the plain allocation and the placement search, fresh process each
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() { local t0=$SECONDS; python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling "$@" 2>/tmp/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['roofline']['achieved']), d.get('pool_placement'))" || tail -3 /tmp/err.txt; echo "    wall $((SECONDS - t0)) s"; }
for r in 1 2 3; do run --filter gauss; run --filter gauss --pool-candidates 1; done
for f in sobel pipeline gray; do run --filter $f; run --filter $f --pool-candidates 1; done
