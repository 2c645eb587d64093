#!/usr/bin/env python3
"""tools/latency_probe.py — per-call latency of the host-buffer entry points (what Controller::PerformCL* costs per frame) at
the reference's own image sizes and at camera sizes: wall time per call and the split the six profiling timestamps give
(write / kernel / read), pageable host memory as the reference uses.  GPU box only."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
rng = np.random.default_rng(1)
with pkg.Context(0) as ctx:
    print("%-12s %-10s %9s %9s %9s %9s" % ("size", "filter", "call us", "write us", "kernel us", "read us"))
    for (w, h) in [(75, 75), (240, 192), (640, 480), (640, 512), (1023, 819), (1280, 720), (1920, 1080), (3840, 2160)]:
        x = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        x[..., 3] = 255
        for name, fn in (("gray", lambda: ctx.gray(x, profile=True)), ("sobel", lambda: ctx.sobel(x, profile=True)),
                         ("gauss k5", lambda: ctx.gauss(x, 5, 1.5, profile=True)),
                         ("gauss k17", lambda: ctx.gauss(x, 17, 6.0, profile=True))):
            for _ in range(5):
                fn()
            n = 50
            t0 = time.perf_counter()
            prof = None
            for _ in range(n):
                _, prof = fn()
            dt = (time.perf_counter() - t0) / n * 1e6
            print("%-12s %-10s %9.1f %9.1f %9.1f %9.1f" % ("%dx%d" % (w, h), name, dt, (prof[1] - prof[0]) / 1e3,
                                                       (prof[3] - prof[2]) / 1e3, (prof[5] - prof[4]) / 1e3))
