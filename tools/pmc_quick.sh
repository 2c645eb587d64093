#!/bin/bash
# tools/pmc_quick.sh <tag> <counters...> -- <bench args>: one PMC pass, prints per-kernel means
TAG=$1; shift
CNT=""; while [ "$1" != "--" ]; do CNT="$CNT $1"; shift; done; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc $CNT --output-format csv -d $OUT/pmc -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ceiling "$@" > $OUT/log 2>&1
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if any(t in k for t in ("gauss","sobel","gray","pipe_")):
            agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in agg.items():
    print(k, " ".join("%s=%.4g"%(c,sum(v)/len(v)) for c,v in cs.items()))
PY
