#!/usr/bin/env python3
"""Benchmark harness with the reference's own shape: for every image, N iterations of the
GPU path through the host-buffer API and N of the CPU path, the mean absolute error between the two, and a
`results.csv` with the reference's exact header (RT/src/FileHandler.cpp:28) plus the two derived columns of
its plotting script (src/GaussianBlur/results/visualisation.py:67,78: Speedup, operation_speedup), so the rows
can be laid beside the reference's published CSVs (src/*/results/*_sorted_results.csv).

    python tools/harness.py --method GAUSSIAN --iterations 100 --images tests/golden --out results.csv

The CPU column is the oracle (test/measurement infrastructure, the restatement of the reference's CPU path);
the GPU column is the product.  Images: .png/.jpg/.ppm via PIL (decode is outside the hot path).
"""
import argparse
import csv
import datetime
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HEADER = ["Timestamp", "Image", "Resolution", "Num_Iterations", "avg_CPU_Time_ms", "avg_OpenCL_Time_ms",
          "avg_OpenCL_kernel_ms", "avg_OpenCL_kernel_write_ms", "avg_OpenCL_kernel_read_ms",
          "avg_OpenCL_kernel_operation_ms", "Error_MAE"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--method", default="GAUSSIAN", choices=["GRAYSCALE", "EDGE", "GAUSSIAN"])
    ap.add_argument("--iterations", type=int, default=100)  # NUMBER_OF_ITERATIONS of the reference apps
    ap.add_argument("--images", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--k", type=int, default=5)             # GAUSSIAN_KERNEL_SIZE, GaussianBlur.cpp:15
    ap.add_argument("--sigma", type=float, default=1.5)     # GAUSSIAN_SIGMA, GaussianBlur.cpp:16
    ap.add_argument("--cpu-iterations", type=int, default=3)
    ap.add_argument("--out", default="results.csv")
    args = ap.parse_args()
    from PIL import Image

    pkg = entry.load_package()
    oracle = entry.load_oracle()
    ctx = pkg.Context(0)
    rows = []
    paths = sorted(p for ext in ("png", "jpg", "ppm") for p in glob.glob(os.path.join(args.images, "*." + ext)))
    for path in paths:
        rgb = np.asarray(Image.open(path).convert("RGB"))
        h, w, _ = rgb.shape
        rgba = np.ascontiguousarray(np.dstack([rgb, np.full((h, w), 255, np.uint8)]))
        if args.method == "GRAYSCALE":
            gpu = lambda: ctx.single("gray", rgba)
            cpu = lambda: oracle.gray_rgba(rgba)
        elif args.method == "EDGE":
            gpu = lambda: ctx.single("sobel", rgba)
            cpu = lambda: oracle.sobel_rgba(rgba)
        else:
            gpu = lambda: ctx.single("gauss", rgba, args.k, args.sigma)
            cpu = lambda: oracle.gauss_rgba(rgba, args.k, args.sigma)
        gpu()
        tot = wr = kern = rd = 0.0
        for _ in range(args.iterations):
            t0 = time.perf_counter()
            out, prof = gpu()
            tot += (time.perf_counter() - t0) * 1e3
            wr += (prof[1] - prof[0]) * 1e-6
            kern += (prof[3] - prof[2]) * 1e-6
            rd += (prof[5] - prof[4]) * 1e-6
        n = args.iterations
        t0 = time.perf_counter()
        for _ in range(args.cpu_iterations):
            ref = cpu()
        cpu_ms = (time.perf_counter() - t0) * 1e3 / args.cpu_iterations
        mae = float(np.mean(np.abs(out.astype(np.int16) - ref.astype(np.int16))))
        rows.append([datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S"), os.path.basename(path), "%dx%d" % (w, h), n,
                     cpu_ms, tot / n, kern / n, wr / n, rd / n, (wr + kern + rd) / n, mae])
    # <out>: the reference's file, byte for byte its header and its 11 columns (RT/src/FileHandler.cpp:28-32);
    # <out>_derived.csv: the same rows plus the two columns its plotting script derives
    # (src/GaussianBlur/results/visualisation.py:67,78: Speedup, operation_speedup)
    with open(args.out, "w", newline="") as f:
        f.write(", ".join(HEADER) + "\n")
        for r in rows:
            f.write(", ".join(str(v) for v in r) + "\n")
    derived = os.path.splitext(args.out)[0] + "_derived.csv"
    with open(derived, "w", newline="") as f:
        f.write(", ".join(HEADER) + ", Speedup, operation_speedup\n")
        for r in rows:
            f.write(", ".join(str(v) for v in r) + ", %g, %g\n" % (r[4] / r[5], r[4] / r[9]))
    print(open(derived).read())
    # <out>_vs_published.csv: where an image is one of the reference's own test images (tests/golden/ref_images/<name>_rgb.png
    # = decoded pixels of images/<name>.jpg), the row the reference published for it on its Linux box
    # (src/*/results/Linux_100_*_sorted_results.csv, committed as tests/golden/published_mae.json) beside this run's
    pub_path = os.path.join(ROOT, "tests", "golden", "published_mae.json")
    if os.path.exists(pub_path):
        import json
        pub = json.load(open(pub_path))[{"GRAYSCALE": "gray", "EDGE": "sobel", "GAUSSIAN": "gauss"}[args.method]]["Linux"]
        side = os.path.splitext(args.out)[0] + "_vs_published.csv"
        cols = ["avg_CPU_Time_ms", "avg_OpenCL_Time_ms", "avg_OpenCL_kernel_operation_ms", "Error_MAE"]
        with open(side, "w", newline="") as f:
            f.write("Image, Resolution, " + ", ".join("here_" + c for c in cols) + ", " +
                    ", ".join("reference_Linux_" + c for c in cols) + ", end_to_end_ratio_reference_over_here\n")
            for r in rows:
                name = {"tulips_medium640_rgb": "Tulips_medium640"}.get(os.path.splitext(r[1])[0],
                                                                      os.path.splitext(r[1])[0].replace("_rgb", ""))
                if name not in pub or pub[name]["resolution"] != r[2]:
                    continue
                here = [r[4], r[5], r[9], r[10]]
                ref = [float(pub[name][c]) for c in cols]
                f.write("%s, %s, " % (name, r[2]) + ", ".join("%g" % v for v in here + ref) + ", %g\n" % (ref[1] / here[1]))
        print(open(side).read())


if __name__ == "__main__":
    main()
