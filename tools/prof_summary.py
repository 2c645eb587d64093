#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by tools/prof.sh into one text summary:
per-kernel count / average duration from the kernel trace, and per-kernel counter sums per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel trace (per kernel: dispatches, avg / min / max duration in us) ==")
for f in find("trace/**/*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    agg = defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-90s n=%4d avg=%10.2f min=%10.2f max=%10.2f" % (k[:90], len(v), sum(v) / len(v), min(v), max(v)))
    # the timed region of bench.py = the LAST `steps` dispatches of the dominant kernel (before them come the
    # placement probes and the warm-up launches); bench.py's own HIP-event figure is in trace.log
    import json
    import re
    log = os.path.join(out, "trace.log")
    m = re.search(r'\{"metric".*\}', open(log).read()) if os.path.exists(log) else None
    if m and agg:
        line = json.loads(m.group(0))
        steps = int(line["steps"])
        dom = max(agg.items(), key=lambda kv: sum(kv[1]))
        if len(dom[1]) >= steps:
            tail = dom[1][-steps:]
            print("timed region: last %d dispatches of %s: avg=%.2f us (rocprofv3 trace)  |  bench.py HIP events, same "
                  "process: %.2f us" % (steps, dom[0][:60], sum(tail) / len(tail), 1e3 * line["roofline"]["avg_launch_ms"]))
    for r in rows[:1]:
        print("columns:", ",".join(r.keys()))
    regs = {}
    for r in rows:
        regs[r["Kernel_Name"]] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
                                  r.get("Workgroup_Size"), r.get("Grid_Size"))
    for k, v in regs.items():
        print("  %-80s vgpr=%s sgpr=%s lds=%s wg=%s grid=%s" % (k[:80], *v))
for f in find("trace/**/*kernel_stats.csv"):
    print("== kernel stats ==")
    print(open(f).read())

print("== PMC (per kernel: mean counter value per dispatch) ==")
for f in find("pmc*/**/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    agg = defaultdict(lambda: defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if any(t in k for t in ("gauss", "sobel", "gray", "pipe_")):
            for c, v in cs.items():
                print("%-60s %-24s n=%3d mean=%.6g" % (k[:60], c, len(v), sum(v) / len(v)))

# HBM traffic per launch of the dominant filter kernel, corrected as MI355X_MICROARCH.md "HBM" prescribes:
# FETCH_SIZE (KiB) counts a wide coalesced 16 B/lane read at exactly half on gfx950 -> x2; WRITE_SIZE (KiB)
# reads exact for 16 B/lane streaming stores.
import json
fetch = write = None
kname = None
for f in find("pmc*/**/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(t in k for t in ("gauss", "sobel", "gray", "pipe_")):
            kname = k
            if r["Counter_Name"] == "FETCH_SIZE":
                fetch = (fetch or []) + [float(r["Counter_Value"])]
            if r["Counter_Name"] == "WRITE_SIZE":
                write = (write or []) + [float(r["Counter_Value"])]
if fetch and write:
    fb = 2.0 * 1024.0 * sum(fetch) / len(fetch)
    wb = 1024.0 * sum(write) / len(write)
    rec = {"kernel": kname, "fetch_size_kib_mean": sum(fetch) / len(fetch), "write_size_kib_mean": sum(write) / len(write),
           "hbm_read_bytes_per_launch": fb, "hbm_write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
           "correction": "read = 2 x FETCH_SIZE x 1024 (gfx950 half-count of wide coalesced reads), write = WRITE_SIZE x 1024",
           "bench_args": sys.argv[2] if len(sys.argv) > 2 else ""}
    # the version of the kernel's source these counters belong to: bench.py reports the figure only while it matches
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    rec["source_sha"] = bench.kernel_source_hash(kname)
    print("== traffic ==")
    print(json.dumps(rec, indent=1))
    json.dump(rec, open(os.path.join(out, "traffic.json"), "w"), indent=1)
