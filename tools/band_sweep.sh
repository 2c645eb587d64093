#!/bin/bash
# the MI355_TUNE_* overrides are only compiled into the tune build (csrc/Makefile, `make tune`)
export MI355_IMGFILTER_LIB=${MI355_IMGFILTER_LIB:-${GRAFT_REPO_ROOT:-/root/repo}/tools/lib/libmi355_imgfilter_tune.so}
# tools/band_sweep.sh "<bench args>" v1 v2 ... : two alternating rounds of MI355_TUNE_BAND_ROWS values (tail phase off)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=$1; shift
run() { env MI355_TUNE_TAIL_FRAC=0 "$1" python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ARGS $1', round(d['roofline']['achieved']))"; }
python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ARGS default', round(d['roofline']['achieved']))"
for r in 1 2; do for v in "$@"; do run MI355_TUNE_BAND_ROWS=$v; done; done
