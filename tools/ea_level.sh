#!/bin/bash
# tools/ea_level.sh <tag> [bench args] — memory-side occupancy of one kernel: requests and summed outstanding-request
# levels at the L2's memory interface (TCC_EA0_*): LEVEL / REQ = average latency in cycles, LEVEL / busy cycles = requests
# in flight.  Two counter passes (reads, writes), never combined with tracing.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-ceiling --no-parity --no-side-figures --pool-candidates 1 $*"
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_32B_sum GRBM_GUI_ACTIVE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_TAG_STALL_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 $ROOT/bench.py $ARGS > $OUT/pmc$i.log 2>&1
  echo "pmc$i rc=$?" >> $OUT/pmc$i.log
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "synth" in k or "checksum" in k or "rocclr" in k or "at::" in k:
            continue
        agg[(k[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print("%-60s %-34s n=%3d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
PY
