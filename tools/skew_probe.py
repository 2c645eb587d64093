"""In a SLOW placement (see place_probe.py), does moving the output relative to the input inside one arena help?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry

pkg = entry.load_package()
dev = torch.device("cuda", 0)
ctx = pkg.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
w, h, F = 3840, 2160, 256
nb = F * h * w * 4


def measure(a, b, filt=pkg.FILTER_GAUSS, steps=30):
    for _ in range(6):
        ctx.filter_dev(filt, a, b, w, h, F, 5, 1.5)
    torch.cuda.synchronize()
    ctx.timer_begin()
    for _ in range(steps):
        ctx.filter_dev(filt, a, b, w, h, F, 5, 1.5)
    return 2 * nb / (ctx.timer_end() / steps) / 1e6


SKEWS = [0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 32 << 20, 1 << 30, 3 << 30]
for trial in range(10):
    arena = torch.empty(2 * nb + (4 << 30), dtype=torch.uint8, device=dev)
    base = arena.data_ptr()
    ctx.synth_dev(base, w, h, F)
    g0 = measure(base, base + nb)
    line = "trial %d: skew 0 -> %.0f" % (trial, g0)
    if g0 < 5750:
        line += "  SLOW; skews: " + " ".join("%s:%.0f" % (("%dK" % (s >> 10)) if s < (1 << 20) else ("%dM" % (s >> 20)), measure(base, base + nb + s)) for s in SKEWS[1:])
    print(line, flush=True)
    del arena
    torch.cuda.empty_cache()
