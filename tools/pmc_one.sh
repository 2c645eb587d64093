#!/bin/bash
# tools/pmc_one.sh <tag> "<counters>" <bench args...> — ONE counter pass for one bench command (few counters: the derived
# *_sum counters of TCP / TCC take many hardware slots and a crowded pass can time out); prints the main kernel's means
set -u
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-ceiling --no-parity --no-side-figures $*"
timeout -k 10 150 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pass -- python3 $ROOT/bench.py $ARGS > $OUT/pass.log 2>&1
echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pass/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "mi355" in k and "synth" not in k and "checksum" not in k:
            acc[(k[:70], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-72s %-40s n=%3d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
PY
