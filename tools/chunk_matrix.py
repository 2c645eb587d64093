#!/usr/bin/env python3
"""tools/chunk_matrix.py — is the placement effect a property of the OUTPUT chunk alone, or of the
pairing of input and output chunk?  One input pool and one output pool of 8 x 32 4K frames; the Gaussian is timed on
every (input slice i, output slice j) pair of 32-frame slices (~1 GiB each)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    pkg = entry.load_package()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.current_stream(dev)
    ctx = pkg.Context(0, stream=stream.cuda_stream)
    w, h, S, N = 3840, 2160, 32, 8
    per = S * w * h * 4
    d_in = torch.empty((N * S, h, w, 4), dtype=torch.uint8, device=dev)
    d_out = torch.empty((N * S, h, w, 4), dtype=torch.uint8, device=dev)
    ctx.synth_dev(d_in.data_ptr(), w, h, N * S)
    filt = pkg.FILTER_GAUSS if len(sys.argv) < 2 else {"gauss": pkg.FILTER_GAUSS, "gray": pkg.FILTER_GRAY}[sys.argv[1]]
    for _ in range(30):
        ctx.filter_dev(filt, d_in.data_ptr(), d_out.data_ptr(), w, h, N * S, 5, 1.5)
    print("rows = input slice, columns = output slice; TB/s (8 B/px), 32 x 4K frames per launch, 20 launches each")
    extra = [x for x in sys.argv[2:]]   # more builds of the library: their column profile on the SAME pools
    if extra:
        ctxs = [("B", ctx)] + [(os.path.basename(x), pkg.Context(0, stream=stream.cuda_stream, lib=pkg.imgfilter.load_library(os.path.join(ROOT, x)))) for x in extra]
        print("input slice 0 -> output slice j, per build (same pools):")
        for name, c in ctxs:
            row = []
            for j in range(N):
                a, b = d_in.data_ptr(), d_out.data_ptr() + j * per
                for _ in range(5):
                    c.filter_dev(filt, a, b, w, h, S, 5, 1.5)
                torch.cuda.synchronize(dev)
                c.timer_begin()
                for _ in range(20):
                    c.filter_dev(filt, a, b, w, h, S, 5, 1.5)
                ms = c.timer_end() / 20
                row.append(8 * S * w * h / (ms * 1e-3) / 1e12)
            print("%-20s " % name + " ".join("%.3f" % r for r in row), flush=True)
        return
    for i in range(N):
        row = []
        for j in range(N):
            a, b = d_in.data_ptr() + i * per, d_out.data_ptr() + j * per
            for _ in range(5):
                ctx.filter_dev(filt, a, b, w, h, S, 5, 1.5)
            torch.cuda.synchronize(dev)
            ctx.timer_begin()
            for _ in range(20):
                ctx.filter_dev(filt, a, b, w, h, S, 5, 1.5)
            ms = ctx.timer_end() / 20
            row.append(8 * S * w * h / (ms * 1e-3) / 1e12)
        print("in %d: " % i + " ".join("%.3f" % r for r in row), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
