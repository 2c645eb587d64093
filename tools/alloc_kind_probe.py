#!/usr/bin/env python3
"""tools/alloc_kind_probe.py — does the KIND of device allocation the output pool comes from change the streaming rate?
(the physical placement of the output pool decides up to 8 %.)  Several pools of each kind —
hipMalloc, hipExtMallocWithFlags(fine-grained / uncached), hipMallocAsync from the default mempool — are allocated side by
side and the 4K Gaussian is timed into each."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    pkg = entry.load_package()
    hip = ctypes.CDLL("libamdhip64.so")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.current_stream(dev)
    ctx = pkg.Context(0, stream=stream.cuda_stream)
    w, h, F = 3840, 2160, 256
    nbytes = F * w * h * 4
    d_in = torch.empty((F, h, w, 4), dtype=torch.uint8, device=dev)
    ctx.synth_dev(d_in.data_ptr(), w, h, F)

    def rate(ptr):
        for _ in range(6):
            ctx.filter_dev(pkg.FILTER_GAUSS, d_in.data_ptr(), ptr, w, h, F, 5, 1.5)
        torch.cuda.synchronize(dev)
        ctx.timer_begin()
        for _ in range(10):
            ctx.filter_dev(pkg.FILTER_GAUSS, d_in.data_ptr(), ptr, w, h, F, 5, 1.5)
        ms = ctx.timer_end() / 10
        return 8 * F * w * h / (ms * 1e-3) / 1e12

    kinds = [("hipMalloc", None), ("ext fine-grained (0x1)", 0x1), ("ext uncached (0x3)", 0x3), ("hipMallocAsync", "async"),
             ("hipMalloc again", None)]
    for name, flag in kinds:
        ptrs, rates = [], []
        for _ in range(4):
            p = ctypes.c_void_p()
            if flag is None:
                rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes))
            elif flag == "async":
                rc = hip.hipMallocAsync(ctypes.byref(p), ctypes.c_size_t(nbytes), ctypes.c_void_p(stream.cuda_stream))
            else:
                rc = hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(nbytes), ctypes.c_uint(flag))
            if rc != 0 or not p.value:
                print("%-26s allocation failed rc=%d" % (name, rc), flush=True)
                break
            ptrs.append(p)
            rates.append(rate(p.value))
        print("%-26s %s TB/s" % (name, " ".join("%.3f" % r for r in rates)), flush=True)
        torch.cuda.synchronize(dev)
        for p in ptrs:
            if flag == "async":
                hip.hipFreeAsync(p, ctypes.c_void_p(stream.cuda_stream))
            else:
                hip.hipFree(p)
        torch.cuda.synchronize(dev)
    ctx.close()


if __name__ == "__main__":
    main()
