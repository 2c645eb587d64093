#!/bin/bash
# tools/pmc_probe.sh <tag> <bench args...> — wider counter sweep for ONE bench command (where does a kernel wait?):
# issue / wait cycles by instruction class, instruction cache, vector-memory FIFOs, L1 / L2 / EA write path.
# Counter passes only (never combined with tracing); output: gpurun_out/<tag>/probe.txt (mean per dispatch of the main kernel)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-ceiling --no-parity --no-side-figures $*"
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
  "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_STALL SQC_DCACHE_MISSES SQC_ICACHE_BUSY_CYCLES" \
  "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_WRITE_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
  "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS TD_TC_STALL_sum TD_TD_BUSY_sum" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/probe$i -- python3 $ROOT/bench.py $ARGS > $OUT/probe$i.log 2>&1
  echo "probe$i rc=$?" >> $OUT/probe$i.log
done
python3 - "$OUT" > $OUT/probe.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/probe*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
main = collections.Counter()
for (k, c), v in acc.items():
    main[k] += len(v)
for (k, c), v in sorted(acc.items()):
    if k.startswith("void mi355") or "mi355::" in k:
        if "synth" in k or "checksum" in k:
            continue
        print("%-62s %-36s n=%3d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
PY
cat $OUT/probe.txt
