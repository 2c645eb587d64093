#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer API (not bench.py's `value`).

Compares, for 4K RGBA frames held in HOST memory:
  per-frame   mi355_gauss_rgba8 per frame, pageable memory — what Controller::PerformCLGaussianBlur does
              (the reference's call shape: write, kernel, read, three waits, RT/src/Controller.cpp:615-744)
  batched     mi355_filter_batched: one H2D, one launch, one D2H, pageable memory
  streamed    mi355_filter_stream: chunks, three stages in flight, pageable and pinned memory
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    pkg = entry.load_package()
    ctx = pkg.Context(0)
    w, h, n = 3840, 2160, 32
    k, sigma = 5, 1.5
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
    px = n * w * h
    res = {"frames": n, "width": w, "height": h}

    ctx.gauss(frames[0], k, sigma)  # warm up pools
    t0 = time.perf_counter()
    prof = None
    for f in range(n):
        _, prof = ctx.single("gauss", frames[f], k, sigma)
    dt = time.perf_counter() - t0
    res["per_frame_pageable"] = {"Mpix_s": px / dt / 1e6, "ms_per_frame": dt / n * 1e3,
                                 "write_ms": (prof[1] - prof[0]) * 1e-6, "kernel_ms": (prof[3] - prof[2]) * 1e-6,
                                 "read_ms": (prof[5] - prof[4]) * 1e-6}

    ctx.gauss(frames, k, sigma)
    t0 = time.perf_counter()
    ctx.gauss(frames, k, sigma)
    dt = time.perf_counter() - t0
    res["batched_pageable"] = {"Mpix_s": px / dt / 1e6, "ms_per_frame": dt / n * 1e3}

    out = np.empty_like(frames)
    ctx.stream(pkg.FILTER_GAUSS, frames, out=out, k=k, sigma=sigma)
    _, ms = ctx.stream(pkg.FILTER_GAUSS, frames, out=out, k=k, sigma=sigma)
    res["streamed_pageable"] = {"Mpix_s": px / (ms * 1e-3) / 1e6, "ms_per_frame": ms / n}

    pin_in = ctx.pinned_empty(frames.shape)
    pin_out = ctx.pinned_empty(frames.shape)
    pin_in[...] = frames
    for chunk in (1, 2, 4, 8):
        ctx.stream(pkg.FILTER_GAUSS, pin_in, out=pin_out, k=k, sigma=sigma, chunk_frames=chunk)
        _, ms = ctx.stream(pkg.FILTER_GAUSS, pin_in, out=pin_out, k=k, sigma=sigma, chunk_frames=chunk)
        res["streamed_pinned_chunk%d" % chunk] = {"Mpix_s": px / (ms * 1e-3) / 1e6, "ms_per_frame": ms / n,
                                                   "GB_s_each_way": px * 4 / (ms * 1e-3) / 1e9}
    assert np.array_equal(pin_out, out)
    for filt, name in ((pkg.FILTER_PIPELINE, "pipeline"), (pkg.FILTER_SOBEL, "sobel")):
        o1 = ctx.pinned_empty((n, h, w))
        ctx.stream(filt, pin_in, out=o1, k=k, sigma=sigma, chunk_frames=2)
        _, ms = ctx.stream(filt, pin_in, out=o1, k=k, sigma=sigma, chunk_frames=2)
        res["streamed_pinned_%s" % name] = {"Mpix_s": px / (ms * 1e-3) / 1e6, "ms_per_frame": ms / n}
        ctx.pinned_free(o1)
    # 3-byte BGR frames (cv::imread layout): 25 % fewer bytes over PCIe, BGR2RGBA done on the device
    ctx.set_input_format(pkg.INPUT_BGR)
    pin_bgr = ctx.pinned_empty((n, h, w, 3))
    pin_bgr[...] = frames[..., 2::-1]
    for filt, name, shape in ((pkg.FILTER_GAUSS, "gauss", (n, h, w, 4)), (pkg.FILTER_PIPELINE, "pipeline", (n, h, w))):
        o1 = ctx.pinned_empty(shape)
        ctx.stream(filt, pin_bgr, out=o1, k=k, sigma=sigma, chunk_frames=2)
        _, ms = ctx.stream(filt, pin_bgr, out=o1, k=k, sigma=sigma, chunk_frames=2)
        res["streamed_pinned_bgr_%s" % name] = {"Mpix_s": px / (ms * 1e-3) / 1e6, "ms_per_frame": ms / n}
        if name == "gauss":
            assert np.array_equal(o1[..., :3], pin_out[..., :3])  # alpha differs: the test frames' A is random
        ctx.pinned_free(o1)
    ctx.set_input_format(pkg.INPUT_RGBA)
    ctx.pinned_free(pin_bgr)
    ctx.pinned_free(pin_in)
    ctx.pinned_free(pin_out)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
