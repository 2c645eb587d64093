#!/bin/bash
# tools/prof.sh <tag> [bench args...] — rocprofv3 passes for one bench command on the GPU box:
#   pass 1: --kernel-trace --stats (per-kernel durations)
#   pass 2..: --pmc counter groups, each in its own run (never combined with tracing)
# Summaries land in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the trace pass runs bench.py's DEFAULT step/warmup counts so its per-kernel average can be laid beside
# bench.py's own HIP-event figure; the counter passes use fewer steps (counters serialise dispatches)
TRACE_ARGS="--no-cpu-baseline --no-ceiling --no-side-figures $*"
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-ceiling --no-parity --no-side-figures $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $TRACE_ARGS > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/trace.log
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 $ROOT/bench.py $ARGS > $OUT/pmc$i.log 2>&1
  echo "pmc$i rc=$?" >> $OUT/pmc$i.log
done
python3 $ROOT/tools/prof_summary.py $OUT "$*" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
