#!/usr/bin/env python3
"""Placement experiment under rocprofv3: in ONE process allocate the input pool and NC candidate
output pools side by side, then launch the 4K Gaussian on every candidate L times.  Dispatch number d of the filter
kernel belongs to candidate d // L, so the per-dispatch rows of `rocprofv3 --kernel-trace --pmc ...` can be laid
beside the HIP-event time of each candidate that this script prints itself.

    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d <dir> -- python3 tools/place_pmc.py --tag <t>
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cands", type=int, default=8)
ap.add_argument("--launches", type=int, default=4)
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--filter", default="gauss")
ap.add_argument("--tag", default="place")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out"))
args = ap.parse_args()

pkg = entry.load_package()
filt = {"gauss": pkg.FILTER_GAUSS, "gray": pkg.FILTER_GRAY, "sobel": pkg.FILTER_SOBEL,
        "pipeline": pkg.FILTER_PIPELINE}[args.filter]
bpp = pkg.imgfilter.OUT_BPP[filt]
w, h, F = 3840, 2160, args.frames
with pkg.Context(0) as ctx:
    d_in = ctx.alloc(w * h * 4 * F)
    ctx.synth_dev(d_in, w, h, F, first_frame=0, seed=0x5EED, mode=0)
    cands = [ctx.alloc(w * h * bpp * F) for _ in range(args.cands)]
    ms = []
    for c in cands:
        ctx.timer_begin()
        for _ in range(args.launches):
            ctx.filter_dev(filt, d_in, c, w, h, F, 5, 1.5)
        ms.append(ctx.timer_end() / args.launches)
    rec = {"tag": args.tag, "filter": args.filter, "launches_per_candidate": args.launches,
           "candidate_addr": ["0x%x" % c for c in cands], "input_addr": "0x%x" % d_in,
           "hip_event_ms_per_launch": ms}
    os.makedirs(args.out, exist_ok=True)
    with open(os.path.join(args.out, "place_pmc_%s.json" % args.tag), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))
    for c in cands:
        ctx.free(c)
    ctx.free(d_in)
