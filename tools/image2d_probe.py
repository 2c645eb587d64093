#!/usr/bin/env python3
"""tools/image2d_probe.py — what the image2d_t mode (mi355_image2d_rgba8) costs per frame: the
kernel time between the reference's kernel-start / kernel-end timestamps and the whole synchronous call, for the
product build (LDS-tiled Gaussian, 16 B/lane gray / Sobel) and — in the tuning build with MI355_IMAGE2D_PLAIN=1 — the
one-thread-per-pixel kernels they replaced; the buffer-mode call on the same frame beside them.

    python3 tools/image2d_probe.py [--plain]      (--plain: run as a second process; the override is read once)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    plain = "--plain" in sys.argv
    pkg = entry.load_package()
    lib = None
    if plain:
        os.environ["MI355_IMAGE2D_PLAIN"] = "1"
        lib = pkg.imgfilter.load_library(os.path.join(entry.ROOT, "tools", "lib", "libmi355_imgfilter_tune.so"))
    ctx = pkg.Context(0, lib=lib)
    rng = np.random.default_rng(1)
    label = "per-pixel kernels (round 2)" if plain else "tiled / vector kernels"
    for (w, h) in ((3840, 2160), (1920, 1080), (1023, 819), (640, 512)):
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        rows = []
        for name, filt, k, sigma in (("gray", pkg.FILTER_GRAY, 0, 0.0), ("sobel", pkg.FILTER_SOBEL, 0, 0.0),
                                     ("gauss k=5", pkg.FILTER_GAUSS, 5, 1.5), ("gauss k=17", pkg.FILTER_GAUSS, 17, 6.0)):
            for _ in range(3):
                ctx.image2d(filt, img, k, sigma)
            kern, call = [], []
            for _ in range(10):
                _, p = ctx.image2d(filt, img, k, sigma)
                kern.append((p[3] - p[2]) * 1e-3)
                call.append((p[5] - p[0]) * 1e-3)
            rows.append("%s kernel %.0f us, call %.0f us" % (name, np.median(kern), np.median(call)))
        if not plain:
            for name, fn in (("buffer-mode gauss k=17", lambda: ctx.single("gauss", img, 17, 6.0)),
                             ("buffer-mode gauss k=5", lambda: ctx.single("gauss", img, 5, 1.5))):
                for _ in range(3):
                    fn()
                kern = []
                for _ in range(10):
                    _, p = fn()
                    kern.append((p[3] - p[2]) * 1e-3)
                rows.append("%s kernel %.0f us" % (name, np.median(kern)))
        print("%dx%d  [%s]  " % (w, h, label) + "; ".join(rows), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
