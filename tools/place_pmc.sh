#!/bin/bash
# tools/place_pmc.sh — counter passes of tools/place_pmc.py (one process per counter group, each process draws its
# own placements; within a process the candidates' counters are compared with their own durations).
# Summary: tools/place_pmc_summary.py gpurun_out/place > profiles/r02_place_pmc.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/place
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# a plain run first (no profiler): the spread of this box
python3 $ROOT/tools/place_pmc.py --tag plain --out $OUT > $OUT/plain.log 2>&1
i=0
for grp in \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_64B_sum" \
  "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUBBLE_sum TCC_REQ_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
  "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $ROOT/tools/place_pmc.py --tag g$i --out $OUT > $OUT/g$i.log 2>&1
  echo "g$i rc=$?" >> $OUT/g$i.log
done
python3 $ROOT/tools/place_pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
