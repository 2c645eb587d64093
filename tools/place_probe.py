"""Where in the 288 GB do the two frame pools sit, and does it matter?  One process; the output pool (or both) is
allocated behind spacers of different sizes; every placement is timed with the headline kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry

pkg = entry.load_package()
dev = torch.device("cuda", 0)
ctx = pkg.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
w, h, F = 3840, 2160, 256
nb = F * h * w * 4


def measure(a, b, filt=pkg.FILTER_GAUSS, steps=12):
    for _ in range(4):
        ctx.filter_dev(filt, a, b, w, h, F, 5, 1.5)
    torch.cuda.synchronize()
    ctx.timer_begin()
    for _ in range(steps):
        ctx.filter_dev(filt, a, b, w, h, F, 5, 1.5)
    return 2 * nb / (ctx.timer_end() / steps) / 1e6


def place(pre_gb, mid_gb):
    pre = torch.empty(int(pre_gb * 2**30), dtype=torch.uint8, device=dev) if pre_gb else None
    a = torch.empty(nb, dtype=torch.uint8, device=dev)
    mid = torch.empty(int(mid_gb * 2**30), dtype=torch.uint8, device=dev) if mid_gb else None
    b = torch.empty(nb, dtype=torch.uint8, device=dev)
    del pre, mid
    torch.cuda.empty_cache()
    ctx.synth_dev(a.data_ptr(), w, h, F)
    g = measure(a.data_ptr(), b.data_ptr())
    del a, b
    torch.cuda.empty_cache()
    return g


print("rows: GB before the input pool; columns: GB between the pools")
gaps = [0, 16, 32, 48, 64, 80, 100, 128, 160, 200]
print("      " + " ".join("%5d" % g for g in gaps))
for pre in (0, 32, 64, 100):
    row = []
    for mid in gaps:
        if pre + mid + 20 > 260:
            row.append("    -")
            continue
        row.append("%5.0f" % place(pre, mid))
    print("%5d " % pre + " ".join(row), flush=True)
