#!/usr/bin/env python3
"""Condense tools/place_pmc.sh: per counter group (= one process = one draw of placements) a table of
candidate pool -> kernel duration and counters per launch, plus the Pearson correlation of each counter with
the duration over the candidates."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
KEYS = ("gauss", "sobel", "gray", "pipe_")


def corr(a, b):
    n = len(a)
    if n < 3:
        return float("nan")
    ma, mb = sum(a) / n, sum(b) / n
    va = sum((x - ma) ** 2 for x in a)
    vb = sum((y - mb) ** 2 for y in b)
    if va == 0 or vb == 0:
        return float("nan")
    return sum((x - ma) * (y - mb) for x, y in zip(a, b)) / (va * vb) ** 0.5


p = os.path.join(out, "place_pmc_plain.json")
if os.path.exists(p):
    r = json.load(open(p))
    print("plain run (no profiler), HIP-event ms per launch per candidate:",
          " ".join("%.3f" % m for m in r["hip_event_ms_per_launch"]))
for gdir in sorted(glob.glob(os.path.join(out, "g[0-9]*"))):
    if not os.path.isdir(gdir):
        continue
    tag = os.path.basename(gdir)
    jf = os.path.join(out, "place_pmc_%s.json" % tag)
    if not os.path.exists(jf):
        print("== %s: no result (see %s.log)" % (tag, tag))
        continue
    rec = json.load(open(jf))
    L = rec["launches_per_candidate"]
    nc = len(rec["candidate_addr"])
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(gdir, "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in KEYS)]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        for d, r in enumerate(rows[: nc * L]):
            dur[d // L].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    ctr = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(gdir, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in KEYS)]
        per = defaultdict(list)
        for r in rows:
            per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for name, vals in per.items():
            vals.sort()
            for d, (_, v) in enumerate(vals[: nc * L]):
                ctr[name][d // L].append(v)
    names = sorted(ctr)
    print("== %s  (input %s)" % (tag, rec["input_addr"]))
    print("%-4s %-16s %10s %10s " % ("cand", "addr", "event_ms", "trace_us") + " ".join("%26s" % n[:26] for n in names))
    dmean = []
    for c in range(nc):
        d = dur.get(c, [])
        # the first launch on a candidate also pays first-touch effects: use the later ones
        dm = sum(d[1:]) / max(1, len(d[1:])) if len(d) > 1 else (d[0] if d else float("nan"))
        dmean.append(dm)
        cols = []
        for n in names:
            v = ctr[n].get(c, [])
            cols.append("%26.6g" % (sum(v[1:]) / max(1, len(v[1:])) if len(v) > 1 else (v[0] if v else float("nan"))))
        print("%-4d %-16s %10.3f %10.1f " % (c, rec["candidate_addr"][c], rec["hip_event_ms_per_launch"][c], dm) + " ".join(cols))
    for n in names:
        v = [sum(ctr[n][c][1:]) / max(1, len(ctr[n][c][1:])) if len(ctr[n].get(c, [])) > 1 else float("nan") for c in range(nc)]
        print("   corr(duration, %s) = %.3f   (min %.6g, max %.6g)" % (n, corr(dmean, v), min(v), max(v)))
