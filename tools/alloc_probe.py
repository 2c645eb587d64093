"""Does the ~8 % bimodality of one command come from where its buffers land?  One process, the same kernel,
buffers re-allocated / skewed between measurements."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry

pkg = entry.load_package()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
ctx = pkg.Context(0, stream=stream.cuda_stream)
w, h, F = 3840, 2160, 256
nb = F * h * w * 4


def measure(d_in, d_out, steps=40):
    for _ in range(8):
        ctx.filter_dev(pkg.FILTER_GAUSS, d_in, d_out, w, h, F, 5, 1.5)
    torch.cuda.synchronize()
    ctx.timer_begin()
    for _ in range(steps):
        ctx.filter_dev(pkg.FILTER_GAUSS, d_in, d_out, w, h, F, 5, 1.5)
    ms = ctx.timer_end() / steps
    return 2 * nb / ms / 1e6


print("fresh torch allocations (empty_cache between):")
for i in range(6):
    a = torch.empty(nb, dtype=torch.uint8, device=dev)
    b = torch.empty(nb, dtype=torch.uint8, device=dev)
    ctx.synth_dev(a.data_ptr(), w, h, F)
    print("  in=%x out=%x  %.0f GB/s  again %.0f" % (a.data_ptr(), b.data_ptr(), measure(a.data_ptr(), b.data_ptr()),
                                                  measure(a.data_ptr(), b.data_ptr())), flush=True)
    del a, b
    torch.cuda.empty_cache()

print("one arena per trial (in | out back to back), fresh each time:")
for i in range(8):
    arena = torch.empty(2 * nb, dtype=torch.uint8, device=dev)
    base = arena.data_ptr()
    ctx.synth_dev(base, w, h, F)
    print("  arena=%x  %.0f GB/s  again %.0f" % (base, measure(base, base + nb), measure(base, base + nb)), flush=True)
    del arena
    torch.cuda.empty_cache()
print("separate allocations again:")
for i in range(6):
    a = torch.empty(nb, dtype=torch.uint8, device=dev)
    b = torch.empty(nb, dtype=torch.uint8, device=dev)
    ctx.synth_dev(a.data_ptr(), w, h, F)
    print("  %.0f GB/s" % measure(a.data_ptr(), b.data_ptr()), flush=True)
    del a, b
    torch.cuda.empty_cache()
