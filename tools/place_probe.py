"""Does the physical placement of the two frame pools matter, and of which one?  One process: 4 candidate input pools
and 6 candidate output pools allocated side by side (in that order), every pair timed with the headline kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry

pkg = entry.load_package()
dev = torch.device("cuda", 0)
ctx = pkg.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
w, h, F = 3840, 2160, 256
nb = F * h * w * 4
filt = {"gauss": pkg.FILTER_GAUSS, "gray": pkg.FILTER_GRAY}[sys.argv[1] if len(sys.argv) > 1 else "gauss"]


def measure(a, b, steps=10):
    for _ in range(4):
        ctx.filter_dev(filt, a, b, w, h, F, 5, 1.5)
    torch.cuda.synchronize()
    ctx.timer_begin()
    for _ in range(steps):
        ctx.filter_dev(filt, a, b, w, h, F, 5, 1.5)
    return 2 * nb / (ctx.timer_end() / steps) / 1e6


ins = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(4)]
outs = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(6)]
for t in ins:
    ctx.synth_dev(t.data_ptr(), w, h, F)
measure(ins[0].data_ptr(), outs[0].data_ptr(), 30)  # clocks up
print("rows: input pool #, columns: output pool # (allocation order: in0..in3, out0..out5); GB/s")
for i, a in enumerate(ins):
    print("in%d  " % i + " ".join("%5.0f" % measure(a.data_ptr(), b.data_ptr()) for b in outs), flush=True)
