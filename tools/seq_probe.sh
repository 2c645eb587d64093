run() { python3 bench.py --no-cpu-baseline --no-ceiling "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['roofline']['achieved']), d['ms_per_step'])"; }
run --filter gauss
run --filter gauss
run --filter gauss --steps 400
run --filter gauss --steps 2000
sleep 30
run --filter gauss
sleep 5
run --filter gauss
run --filter pipeline
sleep 30
run --filter pipeline
rocm-smi --showclocks --showpower 2>/dev/null | grep -v "^$" | head -20
