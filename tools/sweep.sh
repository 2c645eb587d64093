#!/bin/bash
# the MI355_TUNE_* overrides are only compiled into the tune build (csrc/Makefile, `make tune`)
export MI355_IMGFILTER_LIB=${MI355_IMGFILTER_LIB:-${GRAFT_REPO_ROOT:-/root/repo}/tools/lib/libmi355_imgfilter_tune.so}
# tools/sweep.sh — A/B of one tuning knob through bench.py on the GPU box.
#   tools/sweep.sh <filter> <ENV_VAR> <v1> <v2> ... [-- extra bench args]
# e.g. tools/sweep.sh gauss MI355_TUNE_BAND_ROWS 96 128 216 -- --frames 64
# Knobs (read once per process, tuning only): MI355_TUNE_BAND_ROWS, MI355_TUNE_TAIL_ROWS,
# MI355_TUNE_TAIL_FRAC (sliding-window band plan, slide_common.hpp), MI355_TUNE_GRAY_BLOCKS (gray.hip).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
F=$1; VAR=$2; shift 2
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "${1:-}" = "--" ] && shift
run() { python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --filter $F "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f GB/s  frac %.3f  %7.3f ms  %.0f Mpx/s' % (d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['value']))"; }
echo -n "$F default: "; run "$@"
for v in "${VALS[@]}"; do echo -n "$F $VAR=$v: "; env $VAR=$v python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --filter $F "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f GB/s  frac %.3f  %7.3f ms  %.0f Mpx/s' % (d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['value']))"; done
