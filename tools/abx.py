#!/usr/bin/env python3
"""tools/abx.py — same-process A/B timing of several builds of libmi355_imgfilter.so on the SAME buffers.

Physical placement of the frame pools decides up to 8 % of a streaming kernel's rate on this chip and changes with
every allocation, so builds compared across processes need many rounds.  Here every build is
loaded into one process (each .so path is its own library instance), gets its own context on the same stream, and
the builds are timed alternately on one input pool and one output pool.

    python3 tools/abx.py --libs B,ab/lib_x.so,ab/lib_y.so [--rounds 3] [--launches 12] -- <workload> [-- <workload> ...]
    workload: --filter gauss --k 5 --frames 256 --width 3840 --height 2160 [--random-alpha | --alpha-const 128 |
              --alpha-split] [--synth-mode N] [--mode exact] [--impl valu]
"B" = the in-tree product build, "T" = the in-tree tuning build.  Prints one line per (workload, build): median
TB/s over the rounds, all rounds, and the output checksum (builds that should agree bit for bit must print the same).
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

ALGO_BPP = {"gauss": 8, "gray": 8, "gray1": 5, "sobel": 5, "pipeline": 5}
FILTER_ID = {"gray": 0, "gray1": 1, "gauss": 2, "sobel": 3, "pipeline": 4}


def workload_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--filter", default="gauss")
    p.add_argument("--width", type=int, default=3840)
    p.add_argument("--height", type=int, default=2160)
    p.add_argument("--frames", type=int, default=256)
    p.add_argument("--k", type=int, default=5)
    p.add_argument("--sigma", type=float, default=1.5)
    p.add_argument("--mode", default="fast")
    p.add_argument("--impl", default="auto")
    p.add_argument("--synth-mode", type=int, default=0)
    p.add_argument("--random-alpha", action="store_true")
    p.add_argument("--alpha-const", type=int, default=-1, help="every alpha byte = this value")
    p.add_argument("--alpha-split", action="store_true", help="alpha 255 in the top half of each frame, 128 below, "
                   "and a 200 x 200 block of 64 in the middle: piecewise-constant alpha (a matte)")
    p.add_argument("--photo", default="")
    p.add_argument("--then-alpha", type=int, default=-1, help="after the rounds, overwrite alpha with this constant IN THE SAME "
                   "buffers and time the builds again (same placement for both measurements)")
    return p


def main():
    argv = sys.argv[1:]
    head, sets, cur = [], [], None
    for a in argv:
        if a == "--":
            if cur is not None:
                sets.append(cur)
            cur = []
        elif cur is None:
            head.append(a)
        else:
            cur.append(a)
    if cur:
        sets.append(cur)
    hp = argparse.ArgumentParser()
    hp.add_argument("--libs", default="B")
    hp.add_argument("--rounds", type=int, default=3)
    hp.add_argument("--launches", type=int, default=12)
    hp.add_argument("--warm", type=int, default=4)
    hargs = hp.parse_args(head)
    if not sets:
        sets = [[]]

    import torch
    pkg = entry.load_package()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.current_stream(dev)
    libdir = os.path.join(entry.ROOT, "tools", "lib")
    builds = []
    import shutil
    import tempfile
    for name in hargs.libs.split(","):
        tune = {}
        if name.startswith("T:") or name.startswith("T@"):
            # T:BAND_ROWS=48:LANES_OUT=56 -> a private copy of the tuning build with MI355_TUNE_* set;
            # T@ab/lib_x.so:BAND_ROWS=48 -> the same for a tuning build made by VARIANT_TUNE=1 tools/build_variant.sh
            src = os.path.join(libdir, "libmi355_imgfilter_tune.so")
            spec = name[2:]
            if name[1] == "@":
                src, _, spec = spec.partition(":")
                src = src if os.path.isabs(src) else os.path.join(ROOT, src)
            tune = dict(kv.split("=") for kv in spec.split(":")) if spec else {}
            path = os.path.join(tempfile.mkdtemp(prefix="abx_"), "libtune_%s.so" % name[2:].replace(":", "_").replace("=", "").replace("/", "_"))
            shutil.copy(src, path)
        else:
            path = {"B": None, "T": os.path.join(libdir, "libmi355_imgfilter_tune.so")}.get(name, name)
        if path is not None and not os.path.isabs(path):
            path = os.path.join(ROOT, path)
        for k, v in tune.items():
            os.environ["MI355_TUNE_" + k] = v
        lib = pkg.imgfilter.load_library(path) if path else pkg.load_library()
        label = name if path is None or name[0] == "T" else os.path.basename(path).replace("lib_", "").replace(".so", "")
        ctx = pkg.Context(0, stream=stream.cuda_stream, lib=lib)
        if name[0] == "T" and name[1:2] in (":", "@"):
            # the overrides are read once, at the first launch of each kernel family: make those launches now
            t_in = torch.zeros((1, 64, 256, 4), dtype=torch.uint8, device=dev)
            t_out = torch.zeros((1, 64, 256, 4), dtype=torch.uint8, device=dev)
            for f, kk in ((0, 0), (1, 0), (2, 3), (2, 5), (2, 7), (2, 17), (3, 0), (4, 3), (4, 5), (4, 7)):
                ctx.filter_dev(f, t_in.data_ptr(), t_out.data_ptr(), 256, 64, 1, kk, 1.5)
            torch.cuda.synchronize(dev)
            for k in tune:
                del os.environ["MI355_TUNE_" + k]
        builds.append((label, ctx))

    wp = workload_parser()
    for s in sets:
        a = wp.parse_args(s)
        w, h, F = a.width, a.height, a.frames
        filt = FILTER_ID[a.filter]
        out_bpp = pkg.imgfilter.OUT_BPP[filt]
        d_in = torch.empty((F, h, w, 4), dtype=torch.uint8, device=dev)
        d_out = torch.empty((F, h, w, out_bpp), dtype=torch.uint8, device=dev)
        builds[0][1].synth_dev(d_in.data_ptr(), w, h, F, first_frame=0, seed=0x5EED, mode=a.synth_mode)
        if a.photo:
            import numpy as np
            from PIL import Image
            img = np.asarray(Image.open(a.photo).convert("RGB"))
            reps = (-(-h // img.shape[0]) + 1, -(-w // img.shape[1]) + 1, 1)
            big = torch.from_numpy(np.ascontiguousarray(np.tile(img, reps))).to(dev)
            for f in range(F):
                oy, ox = (7 * f) % img.shape[0], (13 * f) % img.shape[1]
                d_in[f, :, :, :3] = big[oy:oy + h, ox:ox + w]
            d_in[..., 3] = 255
        if a.random_alpha:
            d_in[..., 3] = torch.randint(0, 256, (F, h, w), dtype=torch.uint8, device=dev)
        if a.alpha_const >= 0:
            d_in[..., 3] = a.alpha_const
        if a.alpha_split:
            d_in[:, h // 2:, :, 3] = 128
            d_in[:, h // 2 - 100:h // 2 + 100, w // 2 - 100:w // 2 + 100, 3] = 64
        torch.cuda.synchronize(dev)
        for _, ctx in builds:
            ctx.set_gauss_mode(pkg.GAUSS_EXACT if a.mode == "exact" else pkg.GAUSS_FAST)
            ctx.set_impl({"auto": pkg.IMPL_AUTO, "tile": pkg.IMPL_TILE, "mfma": pkg.IMPL_MFMA, "valu": pkg.IMPL_VALU}[a.impl])
        algo = ALGO_BPP[a.filter] * F * w * h
        res = {label: [] for label, _ in builds}
        sums = {}
        # clocks up before the first timed round
        for _ in range(20):
            builds[0][1].filter_dev(filt, d_in.data_ptr(), d_out.data_ptr(), w, h, F, a.k, a.sigma)
        for _ in range(hargs.rounds):
            for label, ctx in builds:
                for _ in range(hargs.warm):
                    ctx.filter_dev(filt, d_in.data_ptr(), d_out.data_ptr(), w, h, F, a.k, a.sigma)
                torch.cuda.synchronize(dev)
                ctx.timer_begin()
                for _ in range(hargs.launches):
                    ctx.filter_dev(filt, d_in.data_ptr(), d_out.data_ptr(), w, h, F, a.k, a.sigma)
                ms = ctx.timer_end() / hargs.launches
                res[label].append(algo / (ms * 1e-3) / 1e12)
                if label not in sums:
                    sums[label] = "%016x" % ctx.checksum_dev(d_out.data_ptr(), d_out.numel())
        if a.then_alpha >= 0:
            d_in[..., 3] = a.then_alpha
            torch.cuda.synchronize(dev)
            res2 = {label: [] for label, _ in builds}
            for _ in range(hargs.rounds):
                for label, ctx in builds:
                    for _ in range(hargs.warm):
                        ctx.filter_dev(filt, d_in.data_ptr(), d_out.data_ptr(), w, h, F, a.k, a.sigma)
                    torch.cuda.synchronize(dev)
                    ctx.timer_begin()
                    for _ in range(hargs.launches):
                        ctx.filter_dev(filt, d_in.data_ptr(), d_out.data_ptr(), w, h, F, a.k, a.sigma)
                    res2[label].append(algo / (ctx.timer_end() / hargs.launches * 1e-3) / 1e12)
            for label in res2:
                res[label + " | same buffers, alpha=%d" % a.then_alpha] = res2[label]
                sums[label + " | same buffers, alpha=%d" % a.then_alpha] = "-"
            builds_print = [(l, None) for l in res]
        else:
            builds_print = builds
        print("== " + (" ".join(s) or "(default workload)"), flush=True)
        for label, _ in builds_print:
            v = res[label]
            print("  %-44s median %.3f TB/s (%.3f of 8)   rounds %s   checksum %s" %
                  (label, statistics.median(v), statistics.median(v) / 8.0, " ".join("%.3f" % x for x in v), sums[label]),
                  flush=True)
        del d_in, d_out
        torch.cuda.empty_cache()
    for _, ctx in builds:
        ctx.close()


if __name__ == "__main__":
    main()
