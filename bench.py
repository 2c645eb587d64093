#!/usr/bin/env python3
"""bench.py — headline measurement: Mpixels/s of the Gaussian 5x5 (sigma 1.5) on 4K RGBA frames.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU.  A "step" is one pass of the hot path (one batched kernel launch through the
C-ABI, mi355_filter_dev) over this rank's batch of synthetic frames, which are generated on the GPU
and resident in HBM before the timed region.  Frames shard by index (rank r owns frames
[r*F, (r+1)*F)), no pixel ever crosses GPUs; the only collective on the data path's setup is one RCCL
broadcast of the coefficient table from rank 0.  Weak scaling: F frames per GPU at every N.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")

# algorithmic bytes per pixel: what the kernel must read + write, halo re-reads,
# LDS traffic and cache hits not counted
ALGO_BPP = {"gauss": 8, "gray": 8, "gray1": 5, "sobel": 5, "pipeline": 5}
FILTER_ID = {"gray": 0, "gray1": 1, "gauss": 2, "sobel": 3, "pipeline": 4}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    # defaults: 60 launches x ~3 ms.  The chip needs some tens of ms of continuous work to settle its clocks
    # (measured: the same kernel reads 4.6 TB/s over a 19 ms run and 5.2-5.4 TB/s over 50-800 ms runs)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--filter", default="gauss", choices=sorted(FILTER_ID))
    p.add_argument("--width", type=int, default=3840)
    p.add_argument("--height", type=int, default=2160)
    p.add_argument("--frames", type=int, default=256, help="frames per GPU per step (>=32: working set must "
                   "exceed the 256 MiB Infinity Cache so the kernel streams from HBM; 256 x 4K = 8.5 GB in + "
                   "8.5 GB out of the 288 GB)")
    p.add_argument("--k", type=int, default=5)
    p.add_argument("--sigma", type=float, default=1.5)
    p.add_argument("--mode", default="fast", choices=["fast", "exact"])
    p.add_argument("--synth-mode", type=int, default=0, help="0 = hash noise, 1 = gradient + noise")
    p.add_argument("--random-alpha", action="store_true", help="overwrite the frames' alpha (255 by definition "
                   "of the synthetic frames, as after cvtColor BGR2RGBA) with noise: measures the Gaussian's "
                   "general 4-channel path instead of its opaque fast path")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the CPU baseline leg")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-ceiling", action="store_true", help="skip the device-copy ceiling measurement")
    p.add_argument("--pool-candidates", type=int, default=12,
                   help="output pools allocated and probed before the warm-up, the fastest is kept (1 = plain allocation)")
    p.add_argument("--alloc-frames", type=int, default=0, help="experiment: size the device buffers for this "
                   "many frames (>= --frames) but process only --frames of them")
    return p.parse_args()


def cpu_baseline(args, oracle):
    """The reference's CPU path (our C restatement, oracle/), timed on this box's host cores on a
    bounded sample of the same workload.  Test/measurement infrastructure: never on the product path."""
    if args.filter not in ("gauss", "pipeline", "sobel", "gray", "gray1"):
        return None
    w, h = args.width, args.height
    fn = {
        "gauss": lambda f, t: oracle.gauss_rgba(f, args.k, args.sigma, threads=t),
        "pipeline": lambda f, t: oracle.pipeline_rgba(f, args.k, args.sigma),
        "sobel": lambda f, t: oracle.sobel_rgba(f),
        "gray": lambda f, t: oracle.gray_rgba(f),
        "gray1": lambda f, t: oracle.gray_rgba_1ch(f),
    }[args.filter]
    frames = oracle.synth_rgba(w, h, 2, first_frame=0, seed=0x5EED, mode=args.synth_mode)
    fn(frames[0][: min(h, 64)], 1)  # warm the code path
    n, t0 = 0, time.perf_counter()
    while True:
        fn(frames[n % 2], 1)
        n += 1
        el = time.perf_counter() - t0
        if el >= args.cpu_seconds or n >= 64:
            break
    out = {"value": n * w * h / el / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": "port",
           "sample": "%d x %dx%d synthetic frame(s), %s k=%d sigma=%g, 1 thread (the reference CPU path is "
                     "single-threaded), %.1f s" % (n, w, h, args.filter, args.k, args.sigma, el)}
    if args.filter == "gauss":
        cores = oracle.max_threads()
        fn(frames[0], cores)
        m, t0 = 0, time.perf_counter()
        while True:
            fn(frames[m % 2], cores)
            m += 1
            el = time.perf_counter() - t0
            if el >= args.cpu_seconds / 2 or m >= 256:
                break
        out["all_cores"] = {"value": m * w * h / el / 1e6, "cores": cores,
                            "sample": "%d frames, OpenMP rows, %.1f s" % (m, el)}
    return out


def main():
    args = parse()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU fallback for the hot path")

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    pkg = entry.load_package()
    # launch on torch's current stream so torch's allocator and the barrier below order with the kernels
    stream = torch.cuda.current_stream(dev)
    ctx = pkg.Context(local_rank, stream=stream.cuda_stream)
    ctx.set_gauss_mode(pkg.GAUSS_EXACT if args.mode == "exact" else pkg.GAUSS_FAST)

    w, h, F = args.width, args.height, args.frames
    filt = FILTER_ID[args.filter]
    out_bpp = pkg.imgfilter.OUT_BPP[filt]
    FA = max(F, args.alloc_frames)
    d_in = torch.empty((FA, h, w, 4), dtype=torch.uint8, device=dev)[:F]
    first_frame = rank * F
    ctx.synth_dev(d_in.data_ptr(), w, h, F, first_frame=first_frame, seed=0x5EED, mode=args.synth_mode)
    if args.random_alpha:
        d_in[..., 3] = torch.randint(0, 256, (F, h, w), dtype=torch.uint8, device=dev)

    # Pool placement: where the two frame pools land physically decides up to 8 % of the
    # streaming rate on this chip, and an allocation cannot be steered, only re-drawn.  So several candidate output
    # pools are allocated side by side (all alive at once, hence all in different places), each is probed with a
    # handful of launches of the very filter to be measured, the fastest is kept and the others are freed.
    # This is set-up: it happens before the warm-up, outside the timed region, and its probes are reported.
    def probe(out_t, launches=8):
        for _ in range(4):
            ctx.filter_dev(filt, d_in.data_ptr(), out_t.data_ptr(), w, h, F, args.k, args.sigma)
        torch.cuda.synchronize(dev)
        ctx.timer_begin()
        for _ in range(launches):
            ctx.filter_dev(filt, d_in.data_ptr(), out_t.data_ptr(), w, h, F, args.k, args.sigma)
        return ctx.timer_end() / launches

    pool_probes = []
    out_bytes = FA * h * w * out_bpp
    free_b, _total = torch.cuda.mem_get_info(dev)
    ncand = max(1, min(args.pool_candidates, int((free_b - (24 << 30)) // max(out_bytes, 1))))
    # candidates are added one at a time and stay allocated while the search runs
    cands = []
    for _ in range(ncand):
        cands.append(torch.empty((FA, h, w, out_bpp), dtype=torch.uint8, device=dev))
        if ncand == 1:
            break
        if not pool_probes:              # clocks up before the first probe, or it reads slow for the wrong reason
            for _ in range(20):
                ctx.filter_dev(filt, d_in.data_ptr(), cands[0].data_ptr(), w, h, F, args.k, args.sigma)
        pool_probes.append(round(probe(cands[-1]), 4))
        # there are more than two states (gray on one box: 6.08 / 6.31 / 6.72 / 6.79 TB/s over six pools), so the
        # search only stops early when it has seen enough pools AND holds one clearly out of the slow state
        if len(pool_probes) >= 6 and min(pool_probes) < 0.955 * max(pool_probes):
            break
    keep = min(range(len(pool_probes)), key=lambda i: pool_probes[i]) if pool_probes else 0
    d_out_full = cands[keep]
    del cands
    torch.cuda.empty_cache()
    d_out = d_out_full[:F]

    # coefficient table: rank 0 generates, RCCL broadcasts over xGMI, every rank installs the same bytes
    if args.filter in ("gauss", "pipeline"):
        table = torch.zeros(args.k * args.k, dtype=torch.float32, device=dev)
        if rank == 0:
            table.copy_(torch.from_numpy(pkg.gauss_weights(args.k, args.sigma).reshape(-1)))
        if dist is not None:
            dist.broadcast(table, src=0)
        ctx.set_gauss_weights(args.k, args.sigma, table.cpu().numpy())

    def step():
        ctx.filter_dev(filt, d_in.data_ptr(), d_out.data_ptr(), w, h, F, args.k, args.sigma)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # on-box streaming ceiling for the same byte count: a device-to-device copy of the input batch
    # (reads 4 B/px, writes 4 B/px = the Gaussian's algorithmic traffic), timed with the same events
    copy_gbs = None
    if rank == 0 and not args.no_ceiling:
        scratch = torch.empty_like(d_in)
        for _ in range(2):
            scratch.copy_(d_in)
        torch.cuda.synchronize(dev)
        ctx.timer_begin()
        for _ in range(5):
            scratch.copy_(d_in)
        copy_ms = ctx.timer_end() / 5
        copy_gbs = 2 * d_in.numel() / (copy_ms * 1e-3) / 1e9
        del scratch
        torch.cuda.empty_cache()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    ctx.timer_begin()           # hipEventRecord on the stream the kernels are launched on
    for _ in range(args.steps):
        step()
    kernel_ms = ctx.timer_end()  # hipEventRecord + hipEventSynchronize + hipEventElapsedTime
    barrier()
    elapsed = time.perf_counter() - t0

    # one launch per step: average launch duration from the HIP events around the timed region
    avg_launch_ms = kernel_ms / args.steps
    checksum = ctx.checksum_dev(d_out.data_ptr(), d_out.numel(), index_base=first_frame * (w * h * out_bpp // 4))

    t_max, ms_max, ck_sum = elapsed, avg_launch_ms, checksum
    if dist is not None:
        t = torch.tensor([elapsed, avg_launch_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_max, ms_max = float(t[0]), float(t[1])
        # 64-bit modular sum of per-rank checksums, carried as two 32-bit halves in int64
        c = torch.tensor([checksum & 0xFFFFFFFF, checksum >> 32], dtype=torch.int64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        ck_sum = (int(c[0]) + (int(c[1]) << 32)) & 0xFFFFFFFFFFFFFFFF

    if rank == 0:
        px_per_launch = F * w * h
        total_px = world * px_per_launch * args.steps
        algo_bytes = ALGO_BPP[args.filter] * px_per_launch
        achieved = algo_bytes / (avg_launch_ms * 1e-3) / 1e9
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                key = "%s_%dx%d_f%d_k%d" % (args.filter, w, h, F, args.k)
                if key in pmc:
                    traffic = pmc[key]["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        line = {
            "metric": "Mpixels/s (Gaussian 5x5, 4K RGBA)" if (args.filter, args.k, w, h) == ("gauss", 5, 3840, 2160)
                      else "Mpixels/s (%s k=%d, %dx%d RGBA)" % (args.filter, args.k, w, h),
            "value": total_px / t_max / 1e6,
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": t_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 in/out, f32 accumulate" if args.filter in ("gauss", "pipeline") else "u8 in/out, f64 luminance",
            "data": "synthetic (device-generated counter-hash frames, resident in HBM before timing)",
            "config": {"workload": "%s k=%d sigma=%g, %dx%d RGBA, %d frames/GPU/step, mode=%s" %
                                   (args.filter, args.k, args.sigma, w, h, F, args.mode),
                       "frames_per_gpu": F, "width": w, "height": h, "parallelism": "frames sharded x%d" % world,
                       "alpha": "random" if args.random_alpha else "255 (opaque frames, as after cvtColor BGR2RGBA)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         # the north star words the target as "HBM-read roofline": the 4 B/px input stream alone
                         "achieved_read_only_GBs": 4 * px_per_launch / (avg_launch_ms * 1e-3) / 1e9,
                         "copy_ceiling_GBs": copy_gbs,
                         "frac_of_copy_ceiling": (achieved / copy_gbs) if copy_gbs else None,
                         "avg_launch_ms": avg_launch_ms, "avg_launch_ms_max_over_ranks": ms_max,
                         "kernel": "see profiles/ (rocprofv3 --kernel-trace --stats of this command)"},
            "checksum": "%016x" % ck_sum,
            "pool_placement": pool_probes,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, entry.load_oracle())
        print(json.dumps(line), flush=True)

    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
