#!/usr/bin/env python3
"""tools/graph_probe.py — are the device-resident calls capturable into a hipGraph, and what does replay buy?

A real-time caller (the reference's webcam loop, RT/RealtimeImageProcessing.cpp: one small frame at a time) is bound by
launch latency, not by the kernels.  mi355_filter_dev makes no allocation and no synchronisation once the (k, sigma)
table and the scratch buffers of a size exist, so a chain of calls on the context's stream can be captured once and
replayed: this script captures gauss 5x5 -> sobel, the fused pipeline and a 17x17 Gaussian on one frame, checks the
replayed bytes against the eager ones, and times both (host wall clock around N repetitions + one synchronise).

    python3 tools/graph_probe.py [--width 640 --height 480] [--reps 2000]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--reps", type=int, default=2000)
    a = ap.parse_args()
    import torch
    pkg = entry.load_package()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    s = torch.cuda.Stream(dev)
    w, h = a.width, a.height
    with torch.cuda.stream(s):
        ctx = pkg.Context(0, stream=s.cuda_stream)
        frame = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev)
        frame[..., 3] = 255
        o_gauss = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
        o_sobel = torch.zeros((h, w), dtype=torch.uint8, device=dev)
        o_pipe = torch.zeros((h, w), dtype=torch.uint8, device=dev)
        o_g17 = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)

        def chain():
            ctx.filter_dev(pkg.FILTER_GAUSS, frame.data_ptr(), o_gauss.data_ptr(), w, h, 1, 5, 1.5)
            ctx.filter_dev(pkg.FILTER_SOBEL, o_gauss.data_ptr(), o_sobel.data_ptr(), w, h, 1)
            ctx.filter_dev(pkg.FILTER_PIPELINE, frame.data_ptr(), o_pipe.data_ptr(), w, h, 1, 5, 1.5)
            ctx.filter_dev(pkg.FILTER_GAUSS, frame.data_ptr(), o_g17.data_ptr(), w, h, 1, 17, 6.0)

        chain()  # first use: tables installed, scratch sized (synchronises; not capturable)
        s.synchronize()
        eager = [t.clone() for t in (o_gauss, o_sobel, o_pipe, o_g17)]
        for t in (o_gauss, o_sobel, o_pipe, o_g17):
            t.zero_()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            chain()
        g.replay()
        s.synchronize()
        same = all(torch.equal(x, y) for x, y in zip(eager, (o_gauss, o_sobel, o_pipe, o_g17)))
        print("%dx%d: 4 calls (5 kernels) captured; replayed bytes == eager bytes: %s" % (w, h, same))
        # new content through the same graph
        frame.copy_(torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev))
        g.replay()
        s.synchronize()
        replay2 = [t.clone() for t in (o_gauss, o_sobel, o_pipe, o_g17)]
        chain()
        s.synchronize()
        same2 = all(torch.equal(x, y) for x, y in zip(replay2, (o_gauss, o_sobel, o_pipe, o_g17)))
        print("new frame content through the same graph == eager: %s" % same2)

        def timed(fn):
            for _ in range(50):
                fn()
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                fn()
            s.synchronize()
            return (time.perf_counter() - t0) / a.reps * 1e6

        t_eager = timed(chain)
        t_graph = timed(g.replay)
        print("per chain of 4 calls: eager %.1f us, graph replay %.1f us (%.2fx)" % (t_eager, t_graph, t_eager / t_graph))
        ctx.close()
    return 0 if (same and same2) else 1


if __name__ == "__main__":
    sys.exit(main())
