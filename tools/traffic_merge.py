#!/usr/bin/env python3
"""tools/traffic_merge.py <gpurun_out/tag> [...] — merges the traffic.json tools/prof.sh wrote for each tag into
profiles/pmc_traffic.json under the key bench.py looks up (<filter>_<w>x<h>_f<frames>_k<k>), with the kernel's source
hash (bench.kernel_source_hash) so that the figure is dropped as soon as the kernel changes."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
pmc = json.load(open(path)) if os.path.exists(path) else {}
for tag_dir in sys.argv[1:]:
    rec = json.load(open(os.path.join(tag_dir, "traffic.json")))
    args = bench.parse(rec.get("bench_args", "").split())
    key = "%s_%dx%d_f%d_k%d" % (args.filter, args.width, args.height, args.total_frames or args.frames, args.k)
    rec["source_sha"] = bench.kernel_source_hash(rec["kernel"])
    rec["source"] = "tools/prof.sh %s" % os.path.basename(os.path.normpath(tag_dir))
    pmc[key] = rec
    print(key, rec["kernel"][:70], "%.4g B" % rec["hbm_bytes_per_launch"], rec["source_sha"])
json.dump(pmc, open(path, "w"), indent=1)
