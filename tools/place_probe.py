"""Where in the 288 GB do the buffers sit, and does it matter?  One process; the two 8.5 GB buffers of the headline
workload are allocated after dummies of different sizes (which pushes them to other physical regions)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry

pkg = entry.load_package()
dev = torch.device("cuda", 0)
ctx = pkg.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
w, h, F = 3840, 2160, 256
nb = F * h * w * 4


def measure(a, b, filt=pkg.FILTER_GAUSS, steps=40):
    for _ in range(8):
        ctx.filter_dev(filt, a, b, w, h, F, 5, 1.5)
    torch.cuda.synchronize()
    ctx.timer_begin()
    for _ in range(steps):
        ctx.filter_dev(filt, a, b, w, h, F, 5, 1.5)
    return 2 * nb / (ctx.timer_end() / steps) / 1e6


def trial(tag, pre_gb, mid_gb):
    pre = torch.empty(int(pre_gb * 2**30), dtype=torch.uint8, device=dev) if pre_gb else None
    a = torch.empty(nb, dtype=torch.uint8, device=dev)
    mid = torch.empty(int(mid_gb * 2**30), dtype=torch.uint8, device=dev) if mid_gb else None
    b = torch.empty(nb, dtype=torch.uint8, device=dev)
    ctx.synth_dev(a.data_ptr(), w, h, F)
    g = measure(a.data_ptr(), b.data_ptr())
    y = measure(a.data_ptr(), b.data_ptr(), pkg.FILTER_GRAY)
    print("%-34s gauss %5.0f GB/s   gray %5.0f GB/s" % (tag, g, y), flush=True)
    del pre, a, mid, b
    torch.cuda.empty_cache()


def trial2(tag, mid_gb, free_mid):
    a = torch.empty(nb, dtype=torch.uint8, device=dev)
    mid = torch.empty(int(mid_gb * 2**30), dtype=torch.uint8, device=dev) if mid_gb else None
    b = torch.empty(nb, dtype=torch.uint8, device=dev)
    if free_mid:
        del mid
        mid = None
        torch.cuda.empty_cache()
    ctx.synth_dev(a.data_ptr(), w, h, F)
    g = measure(a.data_ptr(), b.data_ptr())
    y = measure(a.data_ptr(), b.data_ptr(), pkg.FILTER_GRAY)
    print("%-34s gauss %5.0f GB/s   gray %5.0f GB/s   in=%x out=%x" % (tag, g, y, a.data_ptr(), b.data_ptr()), flush=True)
    del a, mid, b
    torch.cuda.empty_cache()


for rep in range(4):
    trial2("plain", 0, False)
    trial2("128 GB spacer, freed", 128, True)
    trial2("100 GB spacer, freed", 100, True)
    trial2("144 GB spacer, freed", 144, True)
