"""GPU suite, part 2 (-m gpu): BASELINE.json's configurations at their full sizes, whole frames against the oracle.

  config 2  1920x1080 RGBA Gaussian k=5 sigma=1.5                      test_config2_1080p_gaussian
  config 3  3840x2160 RGBA Sobel                                        test_4k_whole_frames
  config 4  3840x2160 fused pipeline                                    test_4k_whole_frames
  config 5  512 x 4K pipeline batch (the N=1 leg; sharding: test_multi_rank.py)   test_config5_512_frame_pipeline_batch
  headline  the launch bench.py times: 4K Gaussian batch through mi355_filter_dev  test_big_batch_gaussian

Bars as everywhere: bit-exact for gray / Sobel / EXACT Gaussian / pipeline, |d| <= 1 LSB for the FAST Gaussian.
The CPU restatement runs row-parallel (oracle.gauss_rgba(threads=N)): a 4K frame costs a fraction of a second.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import __graft_entry__ as entry
from conftest import rand_rgba

pytestmark = pytest.mark.gpu

W4K, H4K = 3840, 2160


def _threads(oracle):
    return max(1, min(oracle.max_threads(), 32))


def _absdiff(a, b):
    return np.abs(a.astype(np.int16) - b.astype(np.int16))


# ---- config 2 ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["synth0", "synth1", "alpha_noise"])
def test_config2_1080p_gaussian(ctx, pkg, oracle, case):
    """BASELINE.json config 2: 1080p RGBA Gaussian 5x5 sigma 1.5 (the reference's own Gaussian defaults,
    src/GaussianBlur/GaussianBlur.cpp:15-16; CPU loop :234-261).  Whole frame: EXACT bit-exact, FAST <= 1 LSB."""
    w, h = 1920, 1080
    if case == "alpha_noise":
        frame = rand_rgba(h, w, seed=1080, alpha=None)
    else:
        frame = oracle.synth_rgba(w, h, 1, first_frame=2, mode=int(case[-1]))[0]
    ref = oracle.gauss_rgba(frame, 5, 1.5, threads=_threads(oracle))
    ctx.set_gauss_mode(pkg.GAUSS_EXACT)
    try:
        assert np.array_equal(ctx.gauss(frame, 5, 1.5), ref)
    finally:
        ctx.set_gauss_mode(pkg.GAUSS_FAST)
    fast = ctx.gauss(frame, 5, 1.5)
    d = _absdiff(fast, ref)
    assert d.max() <= 1 and (d != 0).mean() < 0.01
    # the same frame inside a batch (other band plan: more work items) and through the device-resident entry
    batch = np.stack([frame, frame[::-1].copy(), frame])
    got = ctx.gauss(batch, 5, 1.5)
    assert np.array_equal(got[0], fast) and np.array_equal(got[2], fast)
    assert _absdiff(got[1], oracle.gauss_rgba(batch[1], 5, 1.5, threads=_threads(oracle))).max() <= 1


def test_config2_1080p_other_filters(ctx, pkg, oracle):
    w, h = 1920, 1080
    frame = oracle.synth_rgba(w, h, 1, first_frame=5, mode=1)[0]
    assert np.array_equal(ctx.sobel(frame), oracle.sobel_rgba(frame))
    assert np.array_equal(ctx.gray(frame), oracle.gray_rgba(frame))
    assert np.array_equal(ctx.pipeline(frame, 5, 1.5), oracle.pipeline_rgba(frame, 5, 1.5))


# ---- 4K whole frames (configs 3, 4 and the headline's frame size) ---------------------------------------------------
@pytest.mark.parametrize("mode", [0, 1])
def test_4k_whole_frames(ctx, pkg, oracle, mode):
    frame = oracle.synth_rgba(W4K, H4K, 1, first_frame=7, mode=mode)[0]
    t = _threads(oracle)
    ref_g = oracle.gauss_rgba(frame, 5, 1.5, threads=t)
    d = _absdiff(ctx.gauss(frame, 5, 1.5), ref_g)
    assert d.max() <= 1 and (d != 0).mean() < 0.01
    assert np.array_equal(ctx.sobel(frame), oracle.sobel_rgba(frame))          # config 3
    assert np.array_equal(ctx.gray1(frame), oracle.gray_rgba_1ch(frame))
    ref_p = oracle.pipeline_rgba(frame, 5, 1.5)                                  # config 4
    assert np.array_equal(ctx.pipeline(frame, 5, 1.5), ref_p)
    ctx.set_gauss_mode(pkg.GAUSS_EXACT)
    try:
        assert np.array_equal(ctx.pipeline(frame, 5, 1.5), ref_p)
        assert np.array_equal(ctx.gauss(frame[:540], 5, 1.5), oracle.gauss_rgba(frame[:540], 5, 1.5, threads=t))
    finally:
        ctx.set_gauss_mode(pkg.GAUSS_FAST)


def test_4k_whole_frame_reference_default_gaussian(ctx, pkg, oracle):
    """k = 17, sigma = 6: what RealtimeImageProcessing actually runs (include/ProgramHandler.hpp:9)."""
    frame = oracle.synth_rgba(W4K, H4K, 1, first_frame=17, mode=1)[0]
    ref = oracle.gauss_rgba(frame, 17, 6.0, threads=_threads(oracle))
    d = _absdiff(ctx.gauss(frame, 17, 6.0), ref)
    assert d.max() <= 1
    noisy = rand_rgba(1080, 1920, seed=17, alpha=None)
    d = _absdiff(ctx.gauss(noisy, 17, 6.0), oracle.gauss_rgba(noisy, 17, 6.0, threads=_threads(oracle)))
    assert d.max() <= 1


# ---- the launch the benchmark times -----------------------------------------------------------------------------
def _dev_batch(ctx, n, mode=0):
    d_in = ctx.alloc(W4K * H4K * 4 * n)
    ctx.synth_dev(d_in, W4K, H4K, n, first_frame=0, seed=0x5EED, mode=mode)
    return d_in


def _frame_from_dev(ctx, d_buf, f, bpp):
    out = np.empty((H4K, W4K, 4) if bpp == 4 else (H4K, W4K), np.uint8)
    ctx.d2h(out, d_buf + f * W4K * H4K * bpp)
    return out


def test_big_batch_gaussian(ctx, pkg, oracle):
    """bench.py's own launch shape: a batch of 4K frames through mi355_filter_dev (k = 5, FAST, opaque frames: 24-row
    bands, odd bands walking up, the 3-channel pass).  64 frames as one call == 8 calls of 8 frames (checksums add
    over the word index); first, last and two seeded-random frames, whole, within 1 LSB of the oracle; and the
    general 4-channel path on the same batch with its alpha replaced."""
    import bench
    n, per = 64, W4K * H4K
    d_in = _dev_batch(ctx, n)
    d_out = ctx.alloc(per * 4 * n)
    ctx.filter_dev(pkg.FILTER_GAUSS, d_in, d_out, W4K, H4K, n, 5, 1.5)
    whole = ctx.checksum_dev(d_out, per * 4 * n)
    t = _threads(oracle)
    for f in bench.sample_frame_ids(n):
        frame = oracle.synth_rgba(W4K, H4K, 1, first_frame=f, seed=0x5EED, mode=0)[0]
        assert np.array_equal(_frame_from_dev(ctx, d_in, f, 4), frame)
        d = _absdiff(_frame_from_dev(ctx, d_out, f, 4), oracle.gauss_rgba(frame, 5, 1.5, threads=t))
        assert d.max() <= 1 and (d != 0).mean() < 0.01, f
    d_part = ctx.alloc(per * 4 * 8)
    parts = 0
    for f in range(0, n, 8):
        ctx.filter_dev(pkg.FILTER_GAUSS, d_in + f * per * 4, d_part, W4K, H4K, 8, 5, 1.5)
        parts += ctx.checksum_dev(d_part, per * 4 * 8, index_base=bench.checksum_index_base(f, W4K, H4K, 4))
    assert parts % (1 << 64) == whole
    ctx.free(d_part)
    # general path: one frame of the batch gets noise alpha -> that frame's bands fall back, the others do not
    noisy = rand_rgba(H4K, W4K, seed=5, alpha=None)
    ctx.h2d(d_in + 3 * per * 4, noisy)
    ctx.filter_dev(pkg.FILTER_GAUSS, d_in, d_out, W4K, H4K, n, 5, 1.5)
    assert _absdiff(_frame_from_dev(ctx, d_out, 3, 4), oracle.gauss_rgba(noisy, 5, 1.5, threads=t)).max() <= 1
    f4 = oracle.synth_rgba(W4K, H4K, 1, first_frame=4, seed=0x5EED, mode=0)[0]
    assert _absdiff(_frame_from_dev(ctx, d_out, 4, 4), oracle.gauss_rgba(f4, 5, 1.5, threads=t)).max() <= 1
    ctx.free(d_out)
    ctx.free(d_in)


def test_config5_512_frame_pipeline_batch(ctx, pkg, oracle):
    """BASELINE.json config 5, the N = 1 leg: 512 x 4K frames through the fused pipeline in ONE launch (17 GB in,
    4.2 GB out).  Its checksum equals the sum over 8 launches of 64 frames (= what 8 ranks produce,
    bench.shard_range), and the first, the last and two seeded-random frames equal the chained oracle."""
    import bench
    n, per = 512, W4K * H4K
    d_in = _dev_batch(ctx, n)
    d_out = ctx.alloc(per * n)
    ctx.filter_dev(pkg.FILTER_PIPELINE, d_in, d_out, W4K, H4K, n, 5, 1.5)
    whole = ctx.checksum_dev(d_out, per * n)
    for f in bench.sample_frame_ids(n):
        frame = oracle.synth_rgba(W4K, H4K, 1, first_frame=f, seed=0x5EED, mode=0)[0]
        assert np.array_equal(_frame_from_dev(ctx, d_out, f, 1), oracle.pipeline_rgba(frame, 5, 1.5)), f
    d_part = ctx.alloc(per * 64)
    parts = 0
    for r in range(8):
        first, cnt = bench.shard_range(r, 8, 0, n)
        assert (first, cnt) == (64 * r, 64)
        ctx.filter_dev(pkg.FILTER_PIPELINE, d_in + first * per * 4, d_part, W4K, H4K, cnt, 5, 1.5)
        parts += ctx.checksum_dev(d_part, per * cnt, index_base=bench.checksum_index_base(first, W4K, H4K, 1))
    assert parts % (1 << 64) == whole
    ctx.free(d_part)
    ctx.free(d_out)
    ctx.free(d_in)


def test_bench_line_carries_parity_roofline_and_cpu_baseline(tmp_path):
    """bench.py end to end on a small batch: one JSON line, parity sampled against the oracle inside the run."""
    root = entry.ROOT
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--frames", "8", "--steps", "3", "--warmup", "1",
                          "--pool-candidates", "2", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["metric"] == "Mpixels/s (Gaussian 5x5, 4K RGBA)" and line["n_gpus"] == 1 and line["scaling"] == "weak"
    assert line["parity"]["ok"] and line["parity"]["frames"] == 4 and line["parity"]["max_abs_diff"] <= 1
    r = line["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and r["copy_ceiling_GBs"] > 0 and "frac_plain_alloc" in r
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1
    assert 0 < line["general_path"]["frac"] < 1
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--filter", "pipeline", "--total-frames", "8",
                          "--steps", "2", "--warmup", "1", "--pool-candidates", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["scaling"] == "strong" and line["config"]["total_frames"] == 8 and line["parity"]["ok"]
    assert line["parity"]["max_abs_diff"] == 0
    # photographic content (the reference's Artemis image tiled to the frame size): same checker, same bar
    photo = os.path.join(root, "tests", "golden", "ref_images", "Artemis_medium640_rgb.png")
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--filter", "pipeline", "--frames", "6", "--width", "1920",
                          "--height", "1080", "--photo", photo, "--steps", "2", "--warmup", "1", "--pool-candidates", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["data"].startswith("photographic") and line["parity"]["ok"] and line["parity"]["max_abs_diff"] == 0


def test_bench_multi_rank_code_path_on_one_gpu():
    """The code an 8-GPU run executes — self-spawned ranks, process group, sharded synthesis, table broadcast and
    install, per-rank placement search, barriers, checksum / time reduction, rank-0 parity sample — run for real with
    two and three ranks that share this box's one GPU (`--rehearse-one-gpu`: gloo instead of RCCL, which refuses two
    ranks on one device; a rehearsal, not a measurement).  Frames shard by index, so the job checksum must equal the
    one-rank checksum of the same frames, under weak and under strong scaling."""
    root = entry.ROOT
    common = ["--steps", "2", "--warmup", "1", "--pool-candidates", "2", "--no-cpu-baseline", "--no-ceiling",
              "--no-side-figures", "--width", "1920", "--height", "1080"]

    def run(extra):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common + extra, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, (extra, r.stderr[-3000:])
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout  # rank 0 alone prints
        return json.loads(lines[0])

    one = run(["--gpus", "1", "--frames", "12"])
    weak = run(["--gpus", "2", "--frames", "6", "--rehearse-one-gpu"])
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak" and weak["rehearsal_one_gpu"] and weak["parity"]["ok"]
    assert weak["config"]["total_frames"] == 12 and weak["checksum"] == one["checksum"]
    strong = run(["--gpus", "3", "--total-frames", "12", "--rehearse-one-gpu", "--filter", "pipeline"])
    one_p = run(["--gpus", "1", "--total-frames", "12", "--filter", "pipeline"])
    assert strong["n_gpus"] == 3 and strong["scaling"] == "strong" and strong["parity"]["max_abs_diff"] == 0
    assert strong["config"]["total_frames"] == 12 and strong["checksum"] == one_p["checksum"]


# ---- mi355_group_*: the product-level sharded batch (SURVEY.md §8b "Threading", §8e) -------------------------------
@pytest.mark.parametrize("members", [1, 2, 3])
def test_group_shards_a_host_batch_over_its_members(pkg, oracle, members):
    """mi355_group_filter_batched with 1, 2 and 3 members bound to this box's one GPU (own context, own stream, own
    host thread each): contiguous frame ranges, every filter equal to the oracle frame by frame (FAST Gaussian within
    1 LSB) and equal, byte for byte, to what one context computes for the whole batch."""
    n, h, w = 7, 120, 256
    frames = oracle.synth_rgba(w, h, n, first_frame=30, mode=1)
    frames[2, ..., 3] = 128                                           # one frame with another (constant) alpha
    with pkg.Group([0] * members) as g, pkg.Context(0) as one:
        assert g.size == members
        for filt, k, sigma, ref in (
                (pkg.FILTER_GRAY, 0, 0.0, lambda f: oracle.gray_rgba(f)),
                (pkg.FILTER_SOBEL, 0, 0.0, lambda f: oracle.sobel_rgba(f)),
                (pkg.FILTER_PIPELINE, 5, 1.5, lambda f: oracle.pipeline_rgba(f, 5, 1.5)),
                (pkg.FILTER_GAUSS, 5, 1.5, lambda f: oracle.gauss_rgba(f, 5, 1.5)),
                (pkg.FILTER_GAUSS, 17, 6.0, lambda f: oracle.gauss_rgba(f, 17, 6.0))):
            got, ms = g.filter_batched(filt, frames, k, sigma)
            assert ms > 0 and all(g.member_status(m) == 0 for m in range(members))
            whole = one._host(filt, frames, k, sigma)
            if k != 17:   # (k = 17: AUTO picks the matrix-core or the VALU kernel by launch size — both within 1 LSB of
                assert np.array_equal(got, whole), (filt, k)   # the CPU path, not bit-identical to each other)
            for f in range(n):
                d = _absdiff(got[f], ref(frames[f]).reshape(got[f].shape))
                assert d.max() <= (1 if filt == pkg.FILTER_GAUSS else 0), (filt, k, f)
        # EXACT mode and an installed table reach every member
        g.set_gauss_mode(pkg.GAUSS_EXACT)
        got, _ = g.filter_batched(pkg.FILTER_GAUSS, frames, 5, 1.5)
        assert all(np.array_equal(got[f], oracle.gauss_rgba(frames[f], 5, 1.5)) for f in range(n))
        g.set_gauss_mode(pkg.GAUSS_FAST)
        box = np.full((3, 3), 1.0 / 9, np.float32)
        g.set_gauss_weights(3, 4.0, box)
        got, _ = g.filter_batched(pkg.FILTER_GAUSS, frames, 3, 4.0)
        assert all(_absdiff(got[f], oracle.gauss_rgba(frames[f], 3, weights=box)).max() <= 1 for f in range(n))
        # fewer frames than members: the idle members report OK
        got, _ = g.filter_batched(pkg.FILTER_SOBEL, frames[:1])
        assert np.array_equal(got[0], oracle.sobel_rgba(frames[0]))
        # a bad request comes back as the failing member's code, nothing hangs
        with pytest.raises(pkg.Mi355Error):
            g.filter_batched(pkg.FILTER_GAUSS, frames, 4, 1.5)


def test_group_device_resident_call_adds_up_to_the_single_context_checksum(pkg, oracle):
    """mi355_group_filter_dev: every member owns its shard in device memory (bench.py's layout, inside one process);
    the per-member checksums, keyed by the global word index, add up to the checksum of one context's whole batch."""
    import bench
    n, w, h, members = 10, 640, 360, 2
    with pkg.Group([0, 0]) as g, pkg.Context(0) as one:
        d_all = one.alloc(n * w * h * 4)
        d_all_out = one.alloc(n * w * h)
        one.synth_dev(d_all, w, h, n, first_frame=0, seed=0x5EED, mode=0)
        one.filter_dev(pkg.FILTER_PIPELINE, d_all, d_all_out, w, h, n, 5, 1.5)
        whole = one.checksum_dev(d_all_out, n * w * h)
        d_in, d_out, counts, total = [], [], [], 0
        for m in range(members):
            first, cnt = pkg.group_shard(m, members, n)
            c = g.member(m)
            d_in.append(c.alloc(cnt * w * h * 4))
            d_out.append(c.alloc(cnt * w * h))
            c.synth_dev(d_in[-1], w, h, cnt, first_frame=first, seed=0x5EED, mode=0)
            c.sync()
            counts.append(cnt)
        g.filter_dev(pkg.FILTER_PIPELINE, d_in, d_out, w, h, counts, 5, 1.5)
        for m in range(members):
            first, cnt = pkg.group_shard(m, members, n)
            total += g.member(m).checksum_dev(d_out[m], cnt * w * h, index_base=bench.checksum_index_base(first, w, h, 1))
        assert total % (1 << 64) == whole
        f0 = np.empty((h, w), np.uint8)
        g.member(1).d2h(f0, d_out[1])
        first1, _ = pkg.group_shard(1, members, n)
        assert np.array_equal(f0, oracle.pipeline_rgba(oracle.synth_rgba(w, h, 1, first_frame=first1)[0], 5, 1.5))
        for m in range(members):
            g.member(m).free(d_in[m])
            g.member(m).free(d_out[m])
        one.free(d_all)
        one.free(d_all_out)


def test_bench_rccl_branch_with_one_rank():
    """bench.py's RCCL branch, for real, on this box's one GPU: `--force-dist` with a torchrun-style environment
    (set here, before the child starts: no re-exec of a GPU process) makes init_process_group("nccl", device_id=...),
    the device-tensor broadcast of the coefficient table, both all-reduces, the per-rank all-gather and the barriers
    execute on RCCL with world size 1.  Same frames, same kernel: the checksum equals the plain run's."""
    import bench
    root = entry.ROOT
    common = ["--steps", "2", "--warmup", "1", "--pool-candidates", "2", "--no-cpu-baseline", "--no-ceiling",
              "--no-side-figures", "--width", "1920", "--height", "1080", "--frames", "8", "--gpus", "1"]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(bench.free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common + ["--force-dist"], capture_output=True,
                       text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][0])
    assert line["collectives"].startswith("nccl"), line["collectives"]
    assert line["n_gpus"] == 1 and line["parity"]["ok"]
    assert len(line["per_rank"]) == 1 and line["per_rank"][0]["rank"] == 0 and line["per_rank"][0]["frames"] == 8
    assert abs(line["per_rank"][0]["avg_launch_ms"] - line["roofline"]["avg_launch_ms"]) < 1e-9
    assert line["per_rank"][0]["pool_candidates"] == 2
    plain = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True,
                           timeout=900)
    assert plain.returncode == 0, plain.stderr[-3000:]
    pl = json.loads([ln for ln in plain.stdout.strip().splitlines() if ln.startswith("{")][0])
    assert pl["collectives"].startswith("none") and pl["checksum"] == line["checksum"]


# ---- mi355_ctx_set_gauss_weights: tables the separable kernels must not touch --------------------------------------
def test_external_tables_are_applied_as_given(ctx, pkg, oracle):
    """FAST mode + an installed table that is not w (x) w: the library must apply the 2-D table tap by tap (the
    reference kernel's semantics, RT/kernel/gaussian_base.cl:23-44; CPU loop GaussianBlur.cpp:234-261), not its
    rank-1 factor.  Checked bit-exact against oracle.gauss_rgba(weights=table)."""
    img = rand_rgba(61, 132, seed=12, alpha=None)
    k = 5
    base = oracle.gauss_weights(k, 1.5)
    rng = np.random.default_rng(3)
    nonsep = (base * rng.uniform(0.5, 1.5, (k, k))).astype(np.float32)
    nonsep /= nonsep.sum()
    asym = base.copy()
    asym[0, :] *= 2.0
    asym = (asym / asym.sum()).astype(np.float32)
    v = np.array([-0.1, 0.2, 0.8, 0.2, -0.1], np.float32)
    neg = np.outer(v, v).astype(np.float32)                       # separable, but with negative lobes
    box = np.full((k, k), 1.0 / 25, np.float32)                   # separable and fine: FAST path, 1 LSB
    ctx.set_gauss_mode(pkg.GAUSS_FAST)
    for name, table, sigma in (("nonsep", nonsep, 7.25), ("asym", asym, 7.5), ("neg", neg, 7.75)):
        ctx.set_gauss_weights(k, sigma, table)
        ref = oracle.gauss_rgba(img, k, weights=table)
        assert np.array_equal(ctx.gauss(img, k, sigma), ref), name
        assert np.array_equal(ctx.pipeline(img, k, sigma), oracle.pipeline_rgba(img, k, weights=table)), name
        frames = np.stack([img, img[::-1].copy()])
        assert np.array_equal(ctx.gauss(frames, k, sigma)[1], oracle.gauss_rgba(frames[1], k, weights=table)), name
    ctx.set_gauss_weights(k, 8.0, box)
    assert _absdiff(ctx.gauss(img, k, 8.0), oracle.gauss_rgba(img, k, weights=box)).max() <= 1
    bad = base.copy()
    bad[2, 2] = np.nan
    with pytest.raises(pkg.Mi355Error):
        ctx.set_gauss_weights(k, 9.0, bad)
    # the generated key is untouched by all of the above
    assert _absdiff(ctx.gauss(img, 5, 1.5), oracle.gauss_rgba(img, 5, 1.5)).max() <= 1


@pytest.mark.parametrize("k", [5, 9])
def test_asymmetric_separable_factor_keeps_its_orientation(ctx, pkg, oracle, k):
    """A rank-1 table u (x) u with an ASYMMETRIC u passes separable_factor() and goes to the FAST separable kernels:
    gauss_slide (IMPL_VALU), the matrix-core kernel (IMPL_MFMA: its banded B operand indexes the taps backwards) and
    whatever AUTO picks must all apply the taps in the table's orientation (correlation, as GaussianBlur.cpp:243-256
    does), also in bands that walk upward.  The fused pipeline's sliding-window kernels need a symmetric factor, so it
    must take the tiled path."""
    u = np.array([0.05, 0.15, 0.4, 0.25, 0.15] if k == 5 else [0.02, 0.03, 0.05, 0.1, 0.3, 0.2, 0.15, 0.1, 0.05], np.float32)
    table = np.outer(u, u).astype(np.float32)
    sigma = 11.0 + k
    ctx.set_gauss_mode(pkg.GAUSS_FAST)
    ctx.set_gauss_weights(k, sigma, table)
    frames = rand_rgba(260, 256, seed=77 + k, alpha=None, n=2)   # several bands per strip (odd ones walk up)
    frames[1, ..., 3] = 255
    ref = np.stack([oracle.gauss_rgba(f, k, weights=table) for f in frames])
    for impl in (pkg.IMPL_AUTO, pkg.IMPL_VALU, pkg.IMPL_MFMA, pkg.IMPL_TILE):
        ctx.set_impl(impl)
        assert _absdiff(ctx.gauss(frames, k, sigma), ref).max() <= 1, impl
    ctx.set_impl(pkg.IMPL_TILE)
    tiled = ctx.pipeline(frames, k, sigma)
    ctx.set_impl(pkg.IMPL_AUTO)
    assert np.array_equal(ctx.pipeline(frames, k, sigma), tiled)


def test_installed_tables_survive_the_cache_bound(pkg, oracle):
    """An installed table cannot be regenerated, so it is never evicted: after 17 other keys the installed key still gives
    the installed table (a miss would silently compute the default Gaussian for that sigma)."""
    img = rand_rgba(24, 40, seed=5)
    v = np.array([0.05, 0.1, 0.7, 0.1, 0.05], np.float32)
    box = np.outer(v, v).astype(np.float32)     # a peaked table under a sigma whose own Gaussian is nearly flat
    with pkg.Context(0) as c:
        c.set_gauss_weights(5, 8.0, box)
        want = c.gauss(img, 5, 8.0)
        assert _absdiff(want, oracle.gauss_rgba(img, 5, weights=box)).max() <= 1
        assert _absdiff(want, oracle.gauss_rgba(img, 5, 8.0)).max() > 8      # the default table differs visibly
        for i in range(20):
            c.gauss(img, 5, 0.7 + 0.05 * i)
        assert np.array_equal(c.gauss(img, 5, 8.0), want)
        # the cap on installed tables is an explicit error, not an eviction
        with pytest.raises(pkg.Mi355Error):
            for i in range(70):
                c.set_gauss_weights(3, 20.0 + i, np.full((3, 3), 1.0 / 9, np.float32))
        assert np.array_equal(c.gauss(img, 5, 8.0), want)


def test_coefficient_cache_is_bounded(ctx, pkg, oracle):
    """More distinct (k, sigma) keys than the context keeps (16): old tables are evicted and regenerated on demand."""
    img = rand_rgba(20, 36, seed=2)
    sigmas = [0.7 + 0.05 * i for i in range(24)]
    first = [ctx.gauss(img, 5, s) for s in sigmas]
    for s, want in zip(sigmas, first):
        assert np.array_equal(ctx.gauss(img, 5, s), want)
    assert _absdiff(first[3], oracle.gauss_rgba(img, 5, sigmas[3])).max() <= 1


def test_in_place_and_overlapping_device_calls_are_rejected(ctx, pkg, oracle):
    w, h, n = 64, 32, 2
    frames = oracle.synth_rgba(w, h, n)
    d = ctx.alloc(frames.nbytes * 2)
    ctx.h2d(d, frames)
    for filt in (pkg.FILTER_GRAY, pkg.FILTER_GAUSS, pkg.FILTER_SOBEL, pkg.FILTER_PIPELINE, pkg.FILTER_GRAY1):
        with pytest.raises(pkg.Mi355Error):
            ctx.filter_dev(filt, d, d, w, h, n, 5, 1.5)                      # in place
        with pytest.raises(pkg.Mi355Error):
            ctx.filter_dev(filt, d, d + frames.nbytes - 16, w, h, n, 5, 1.5)  # output starts inside the input
        with pytest.raises(pkg.Mi355Error):
            ctx.filter_dev(filt, d + 64, d, w, h, n, 5, 1.5)                 # input starts inside the output
    ctx.filter_dev(pkg.FILTER_SOBEL, d, d + frames.nbytes, w, h, n)          # adjacent is fine
    got = np.empty((n, h, w), np.uint8)
    ctx.d2h(got, d + frames.nbytes)
    assert np.array_equal(got[1], oracle.sobel_rgba(frames[1]))
    back = np.empty_like(frames)
    ctx.d2h(back, d)
    assert np.array_equal(back, frames)                                       # rejected calls touched nothing
    with pytest.raises(pkg.Mi355Error):
        ctx.stream_copy_dev(d + 16, d, frames.nbytes)
    ctx.stream_copy_dev(d + frames.nbytes, d, frames.nbytes)
    ctx.d2h(back, d + frames.nbytes)
    assert np.array_equal(back, frames)
    ctx.free(d)


def test_stream_copy_odd_sizes(ctx):
    rng = np.random.default_rng(8)
    for nbytes in (1, 15, 16, 17, 4099, 1 << 20, (1 << 20) + 7):
        src = rng.integers(0, 256, nbytes, dtype=np.uint8)
        d = ctx.alloc(2 * nbytes + 64)
        ctx.h2d(d, src)
        dst = d + ((nbytes + 31) // 16) * 16
        ctx.stream_copy_dev(dst, d, nbytes)
        back = np.empty(nbytes, np.uint8)
        ctx.d2h(back, dst)
        assert np.array_equal(back, src), nbytes
        ctx.free(d)


# ---- f2: the streamed host path against the ORACLE (not against another HIP path) ----------------------------------
@pytest.mark.parametrize("pinned", [False, True])
def test_streamed_host_path_matches_oracle(ctx, pkg, oracle, pinned):
    """mi355_filter_stream replaces the reference's write / wait / kernel / wait / read / wait per frame
    (RT/src/Controller.cpp:646-652,712-744): all four filters, pageable and pinned memory, several chunks with a
    ragged last one, compared directly with the CPU restatement."""
    frames = oracle.synth_rgba(500, 131, 7, first_frame=3, mode=1)
    src = frames
    if pinned:
        src = ctx.pinned_empty(frames.shape)
        src[...] = frames
    try:
        for filt, name in ((pkg.FILTER_GRAY, "gray"), (pkg.FILTER_GAUSS, "gauss"), (pkg.FILTER_SOBEL, "sobel"),
                           (pkg.FILTER_PIPELINE, "pipeline"), (pkg.FILTER_GRAY1, "gray1")):
            dst = None
            if pinned:
                dst = ctx.pinned_empty(frames.shape if pkg.imgfilter.OUT_BPP[filt] == 4 else frames.shape[:3])
            got, ms = ctx.stream(filt, src, out=dst, k=5, sigma=1.5, chunk_frames=3)
            assert ms > 0
            for f in range(frames.shape[0]):
                if name == "gauss":
                    assert _absdiff(got[f], oracle.gauss_rgba(frames[f], 5, 1.5)).max() <= 1
                else:
                    ref = {"gray": oracle.gray_rgba, "sobel": oracle.sobel_rgba, "gray1": oracle.gray_rgba_1ch,
                           "pipeline": lambda x: oracle.pipeline_rgba(x, 5, 1.5)}[name](frames[f])
                    assert np.array_equal(got[f], ref), (name, f)
            if pinned:
                ctx.pinned_free(dst)
    finally:
        if pinned:
            ctx.pinned_free(src)


# ---- f1: the reference-shaped harness ------------------------------------------------------------------------------
REF_CSV_HEADER = ("Timestamp, Image, Resolution, Num_Iterations, avg_CPU_Time_ms, avg_OpenCL_Time_ms, "
                  "avg_OpenCL_kernel_ms, avg_OpenCL_kernel_write_ms, avg_OpenCL_kernel_read_ms, "
                  "avg_OpenCL_kernel_operation_ms, Error_MAE")   # RT/src/FileHandler.cpp:28, byte for byte


@pytest.mark.parametrize("method,mae_bound", [("GRAYSCALE", 0.0), ("EDGE", 0.0), ("GAUSSIAN", 0.01)])
def test_reference_shaped_harness(tmp_path, method, mae_bound):
    """tools/harness.py = the reference's per-image benchmark loop (src/Grayscale/grayscale.cpp:398-462): same CSV
    header, 11 columns, one row per image; a second file adds the two columns the reference's plotting script
    derives.  MAE (GPU vs the CPU restatement): 0 for gray / edge, < 0.01 grey levels for the FAST Gaussian (its
    off-by-one rate is < 1 %)."""
    out = tmp_path / "results.csv"
    run = subprocess.run([sys.executable, os.path.join(entry.ROOT, "tools", "harness.py"), "--method", method,
                          "--iterations", "5", "--cpu-iterations", "1", "--out", str(out)],
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = out.read_text().splitlines()
    assert lines[0] == REF_CSV_HEADER
    assert len(lines) >= 2
    for row in lines[1:]:
        cols = row.split(", ")
        assert len(cols) == 11
        assert cols[1] == "tulips_medium640_rgb.png" and cols[2] == "640x512" and cols[3] == "5"
        cpu_ms, gpu_ms, kern, wr, rd, op, mae = (float(c) for c in cols[4:])
        assert cpu_ms > 0 and gpu_ms > 0 and kern > 0 and wr > 0 and rd > 0
        assert abs(op - (kern + wr + rd)) < 1e-6 * max(1.0, op) and op <= gpu_ms * 1.05
        assert mae <= mae_bound
    derived = (tmp_path / "results_derived.csv").read_text().splitlines()
    assert derived[0] == REF_CSV_HEADER + ", Speedup, operation_speedup" and len(derived[1].split(", ")) == 13


# ---- f4: image2d_t-mode semantics ----------------------------------------------------------------------------------
@pytest.mark.parametrize("h,w", [(1, 1), (2, 3), (3, 3), (17, 65), (75, 75), (131, 500), (64, 256)])
def test_image2d_mode_semantics(ctx, pkg, oracle, h, w):
    """mi355_image2d_rgba8 = what the reference computes with BYPASS_IMAGE_SUPPORT = false (RT/kernel/*_images.cl,
    RT/src/Controller.cpp:246-272,374-403): bit-exact against the restated OpenCL-C semantics (oracle_image2d_*;
    parity unpinned — the reference has no CPU twin of these GPU kernels — except the weight table, which is pinned)."""
    img = rand_rgba(h, w, seed=h * 3 + w, alpha=None)
    got, prof = ctx.image2d(pkg.FILTER_GRAY, img)
    assert np.array_equal(got, oracle.image2d_gray(img))
    assert len(prof) == 6 and all(b >= a for a, b in zip(prof, prof[1:]))
    got, _ = ctx.image2d(pkg.FILTER_SOBEL, img)
    assert np.array_equal(got, oracle.image2d_sobel(img))
    for k, sigma in ((5, 1.5), (3, 0.8), (17, 6.0)):
        got, _ = ctx.image2d(pkg.FILTER_GAUSS, img, k, sigma)
        assert np.array_equal(got, oracle.image2d_gauss(img, k, sigma)), (k, sigma)
    with pytest.raises(pkg.Mi355Error):
        ctx.image2d(pkg.FILTER_PIPELINE, img)
    with pytest.raises(pkg.Mi355Error):
        ctx.image2d(pkg.FILTER_GAUSS, img, 4, 1.0)


@pytest.mark.parametrize("h,w,k,sigma", [(360, 640, 17, 6.0), (819, 1023, 5, 1.5), (70, 200, 25, 8.0), (40, 130, 27, 8.0),
                                         (33, 64, 9, 2.5), (16, 4, 7, 2.0)])
def test_image2d_mode_tiled_kernels_on_larger_frames(ctx, pkg, oracle, h, w, k, sigma):
    """The LDS-tiled image-mode Gaussian (k <= 25; k = 27 takes the per-pixel fallback), the 16 B/lane grayscale and the
    4-outputs-per-lane Sobel on frames of many tiles, ragged widths included (1023 x 819 is the reference's largest
    image, k = 17 sigma = 6 its application default): the same bytes as the restated OpenCL-C semantics."""
    img = rand_rgba(h, w, seed=h + 7 * w + k, alpha=None)
    got, _ = ctx.image2d(pkg.FILTER_GAUSS, img, k, sigma)
    assert np.array_equal(got, oracle.image2d_gauss(img, k, sigma))
    got, _ = ctx.image2d(pkg.FILTER_GRAY, img)
    assert np.array_equal(got, oracle.image2d_gray(img))
    got, _ = ctx.image2d(pkg.FILTER_SOBEL, img)
    assert np.array_equal(got, oracle.image2d_sobel(img))


def test_image2d_mode_fixture(ctx, pkg, oracle, fixture_rgba):
    got, _ = ctx.image2d(pkg.FILTER_GRAY, fixture_rgba)
    assert np.array_equal(got, oracle.image2d_gray(fixture_rgba))
    # the image path differs from the buffer path's CPU semantics by design: fp32 vs fp64 luminance, truncation of a
    # normalised product; on a photograph they stay within one grey level of each other
    assert np.abs(got.astype(int) - oracle.gray_rgba_1ch(fixture_rgba).astype(int)).max() <= 1
    got, _ = ctx.image2d(pkg.FILTER_GAUSS, fixture_rgba, 5, 1.5)
    assert np.array_equal(got, oracle.image2d_gauss(fixture_rgba, 5, 1.5))


# ---- EXACT mode at sliding-window speed (csrc/gauss_exact.hip) -----------------------------------------------------
@pytest.mark.parametrize("k,sigma", [(3, 0.8), (5, 1.5), (7, 2.0)])
@pytest.mark.parametrize("h,w", [(1, 4), (2, 8), (7, 12), (40, 252), (33, 248), (131, 500), (300, 1920), (5, 3840)])
def test_exact_mode_sliding_kernel(ctx, pkg, oracle, k, sigma, h, w):
    """MI355_GAUSS_EXACT on 4-pixel-multiple widths runs the exact-by-exception sliding kernel: bit-identical to the CPU
    path (src/GaussianBlur/GaussianBlur.cpp:234-261) and to the tiled EXACT kernel, on noise, opaque and flat frames."""
    ctx.set_gauss_mode(pkg.GAUSS_EXACT)
    try:
        for case in ("noise", "opaque", "one_hole", "flat", "alpha_0", "alpha_128", "alpha_split"):
            if case == "noise":
                img = rand_rgba(h, w, seed=h + w + k, alpha=None)
            elif case == "flat":
                img = np.full((h, w, 4), 255, np.uint8)
                img[:, w // 2:, :3] = 17
            else:
                img = oracle.synth_rgba(w, h, 1, first_frame=k, mode=1)[0].copy()   # A = 255
                if case == "one_hole":
                    img[h // 2, (w * 3) // 4, 3] = 9
                elif case == "alpha_0":          # constant alpha other than 255: the 3-channel walk with the CPU
                    img[..., 3] = 0              # chain's byte for an all-A window (round 3)
                elif case == "alpha_128":
                    img[..., 3] = 128
                elif case == "alpha_split":      # two constants: bands that meet both fall back to four channels
                    img[..., 3] = 200
                    img[h // 2:, :, 3] = 31
            ref = oracle.gauss_rgba(img, k, sigma, threads=8)
            assert np.array_equal(ctx.gauss(img, k, sigma), ref), case
            ctx.set_impl(pkg.IMPL_TILE)
            assert np.array_equal(ctx.gauss(img, k, sigma), ref), case
            ctx.set_impl(pkg.IMPL_AUTO)
    finally:
        ctx.set_impl(pkg.IMPL_AUTO)
        ctx.set_gauss_mode(pkg.GAUSS_FAST)


def test_exact_mode_batches_and_full_frames(ctx, pkg, oracle):
    ctx.set_gauss_mode(pkg.GAUSS_EXACT)
    try:
        frames = oracle.synth_rgba(1000, 300, 4, first_frame=1, mode=0).copy()
        frames[2, 150:160, 400:420, 3] = 0          # a non-opaque patch inside one frame of the batch
        got = ctx.gauss(frames, 5, 1.5)
        for f in range(4):
            assert np.array_equal(got[f], oracle.gauss_rgba(frames[f], 5, 1.5, threads=8)), f
        for mode in (0, 1):
            frame = oracle.synth_rgba(W4K, H4K, 1, first_frame=3, mode=mode)[0]
            assert np.array_equal(ctx.gauss(frame, 5, 1.5), oracle.gauss_rgba(frame, 5, 1.5, threads=_threads(oracle)))
        noisy = rand_rgba(1080, 1920, seed=21, alpha=None)
        assert np.array_equal(ctx.gauss(noisy, 3, 0.8), oracle.gauss_rgba(noisy, 3, 0.8, threads=_threads(oracle)))
    finally:
        ctx.set_gauss_mode(pkg.GAUSS_FAST)


# ---- the fused pipeline with 8 pixels per lane (csrc/pipe_slide8.hip) -----------------------------------------------
_PIPE8_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import __graft_entry__ as entry
from conftest import rand_rgba
pkg = entry.load_package(); oracle = entry.load_oracle()
bad = []
with pkg.Context(0) as ctx:
    for (n, h, w) in [(1, 2, 16), (2, 3, 24), (1, 40, 504), (2, 131, 1000), (1, 300, 1920), (1, 7, 3840), (3, 97, 496), (1, 64, 512)]:
        for k, s in ((3, 0.8), (5, 1.5)):
            for kind in ("noise", "synth", "flat", "patch64", "patch6"):
                if kind == "noise":
                    x = rand_rgba(h, w, seed=h + w + k, alpha=None, n=n)
                elif kind == "synth":
                    x = oracle.synth_rgba(w, h, n, first_frame=k, mode=1)
                elif kind == "patch64":   # flat 64 x 64 patches: the constant-window table path (exact_common.hpp)
                    x = oracle.synth_rgba(w, h, n, first_frame=k, mode=2)
                elif kind == "patch6":    # 6 x 9 patches: constant and almost-constant windows side by side
                    small = rand_rgba((h + 5) // 6, (w + 8) // 9, seed=h * w + k, alpha=255, n=n)
                    x = np.ascontiguousarray(np.repeat(np.repeat(small, 6, axis=1), 9, axis=2)[:, :h, :w])
                else:
                    x = np.full((n, h, w, 4), 200, np.uint8); x[:, :, w // 2:, :3] = 31
                got = ctx.pipeline(x, k, s)
                for f in range(n):
                    if not np.array_equal(got[f], oracle.pipeline_rgba(x[f], k, s)):
                        bad.append((n, h, w, k, kind, f))
print(bad)
"""


@pytest.mark.parametrize("h,w,n", [(70, 512, 2), (33, 1023, 1), (131, 250, 2), (5, 64, 1), (200, 1920, 1)])
def test_flat_content_takes_the_table_path_and_stays_bit_exact(ctx, pkg, oracle, h, w, n):
    """Flat content (letterbox bars, saturated regions, graphics, JPEG-blocky skies) flags every pixel of the
    exact-by-exception kernels; constant windows then read the CPU chain's value from a 256-entry table instead of
    evaluating it (exact_common.hpp: flat_windows).  Frames of 64 x 64 and of 6 x 9 flat patches, a two-level frame and a
    constant one, through the fused pipeline (4-pixel kernel here, aligned and ragged widths; the 8-pixel kernel:
    test_pipeline_eight_pixels_per_lane) and the EXACT-mode Gaussian (3- and 4-channel walks): bit-identical to the CPU
    chain."""
    small = rand_rgba((h + 5) // 6, (w + 8) // 9, seed=h * w, alpha=255, n=n)
    frames = {
        "patch64": oracle.synth_rgba(w, h, n, first_frame=3, mode=2),
        "patch6": np.ascontiguousarray(np.repeat(np.repeat(small, 6, axis=1), 9, axis=2)[:, :h, :w]),
        "const": np.full((n, h, w, 4), 255, np.uint8),
    }
    two = np.full((n, h, w, 4), 200, np.uint8)
    two[:, :, w // 2:, :3] = 31
    two[:, h // 2, :, 0] = 77
    frames["two-level"] = two
    for kind, x in frames.items():
        for k, s in ((3, 0.8), (5, 1.5), (7, 2.0)):
            got = ctx.pipeline(x, k, s)
            for f in range(n):
                assert np.array_equal(got[f], oracle.pipeline_rgba(x[f], k, s)), (kind, k, f)
        if w % 4 == 0:
            ctx.set_gauss_mode(pkg.GAUSS_EXACT)
            try:
                for k, s in ((3, 0.8), (5, 1.5)):
                    for alpha in (False, True):
                        y = x.copy()
                        if alpha:  # patchy alpha as well: the 4-channel walk, flat in all four channels
                            y[..., 3] = np.repeat(np.repeat(rand_rgba((h + 15) // 16, (w + 15) // 16, seed=k, n=n)[..., 0], 16, axis=1),
                                                  16, axis=2)[:, :h, :w]
                        got = ctx.gauss(y, k, s)
                        for f in range(n):
                            assert np.array_equal(got[f], oracle.gauss_rgba(y[f], k, s)), (kind, k, alpha, f)
            finally:
                ctx.set_gauss_mode(pkg.GAUSS_FAST)


def test_data_dependent_paths_on_seeded_random_shapes_and_content(ctx, pkg, oracle):
    """The table paths are data-dependent (constant windows, gray pixels, gray rows): 24 seeded random cases — shape, blocky
    content with random block sizes (1 x 1 = noise ... 16 x 16), a random share of gray blocks, k in {3, 5, 7} — through
    the pipeline, Sobel, gray and the EXACT Gaussian, whole frames against the CPU path."""
    rng = np.random.default_rng(20240)
    for case in range(24):
        h, w = int(rng.integers(2, 220)), int(rng.integers(4, 900))
        if case % 3 == 0:
            w = (w + 7) // 8 * 8
        by, bx = int(rng.integers(1, 17)), int(rng.integers(1, 17))
        small = rng.integers(0, 256, ((h + by - 1) // by, (w + bx - 1) // bx, 4), dtype=np.uint8)
        gray_share = float(rng.choice([0.0, 0.3, 1.0]))
        mask = rng.random(small.shape[:2]) < gray_share
        small[mask, 1] = small[mask, 0]
        small[mask, 2] = small[mask, 0]
        small[..., 3] = 255
        x = np.ascontiguousarray(np.repeat(np.repeat(small, by, axis=0), bx, axis=1)[:h, :w])
        k, sg = [(3, 0.8), (5, 1.5), (7, 2.0)][case % 3]
        tag = (case, h, w, by, bx, gray_share, k)
        assert np.array_equal(ctx.pipeline(x, k, sg), oracle.pipeline_rgba(x, k, sg)), tag
        assert np.array_equal(ctx.sobel(x), oracle.sobel_rgba(x)), tag
        assert np.array_equal(ctx.gray1(x), oracle.gray_rgba_1ch(x)), tag
        if w % 4 == 0 and k <= 5:
            ctx.set_gauss_mode(pkg.GAUSS_EXACT)
            try:
                assert np.array_equal(ctx.gauss(x, k, sg), oracle.gauss_rgba(x, k, sg)), tag
            finally:
                ctx.set_gauss_mode(pkg.GAUSS_FAST)


def test_pipeline_eight_pixels_per_lane():
    """AUTO gives k = 5 launches of >= 10^9 pixels to pipe_slide8.hip (config 5's 512-frame launch above: same checksum
    as the 4-pixel kernel's 8 x 64 frames, frames equal to the oracle).  Here the tuning build forces it
    (MI355_PIPE8=1) on small shapes: one strip / several strips / idle lanes, one band / several bands walking both
    ways, image edges, k = 3 and 5, noise, smooth and flat frames — bit-identical to the chained oracle."""
    root = entry.ROOT
    tune_lib = os.path.join(entry.ROOT, "tools", "lib", "libmi355_imgfilter_tune.so")
    assert os.path.exists(tune_lib), "run __graft_entry__.build()"
    env = dict(os.environ, MI355_IMGFILTER_LIB=tune_lib, MI355_PIPE8="1")
    out = subprocess.run([sys.executable, "-c", _PIPE8_SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "[]", out.stdout[-2000:]


_GRAPH_SCRIPT = r"""
import sys
import numpy as np
import torch                       # first: torch brings its own HIP runtime and must initialise it before the library loads
torch.cuda.init()
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import __graft_entry__ as entry
from conftest import rand_rgba
pkg = entry.load_package(); oracle = entry.load_oracle()
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(dev)
w, h = 640, 480
bad = []
with torch.cuda.stream(s):
    c = pkg.Context(0, stream=s.cuda_stream)
    frames = [rand_rgba(h, w, seed=91), rand_rgba(h, w, seed=92, alpha=None)]
    d_in = torch.from_numpy(frames[0]).to(dev)
    o_gauss = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
    o_sobel = torch.zeros((h, w), dtype=torch.uint8, device=dev)
    o_pipe = torch.zeros((h, w), dtype=torch.uint8, device=dev)
    o_g17 = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)

    def chain():
        c.filter_dev(pkg.FILTER_GAUSS, d_in.data_ptr(), o_gauss.data_ptr(), w, h, 1, 5, 1.5)
        c.filter_dev(pkg.FILTER_SOBEL, o_gauss.data_ptr(), o_sobel.data_ptr(), w, h, 1)
        c.filter_dev(pkg.FILTER_PIPELINE, d_in.data_ptr(), o_pipe.data_ptr(), w, h, 1, 5, 1.5)
        c.filter_dev(pkg.FILTER_GAUSS, d_in.data_ptr(), o_g17.data_ptr(), w, h, 1, 17, 6.0)

    c.set_gauss_mode(pkg.GAUSS_EXACT)   # bit-identical to the CPU path, so the oracle is an equality check
    chain()                             # first use: tables installed, scratch sized - not capturable, by contract
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        chain()
    for n, f in enumerate(frames):
        d_in.copy_(torch.from_numpy(f).to(dev))
        for t in (o_gauss, o_sobel, o_pipe, o_g17):
            t.zero_()
        g.replay()
        s.synchronize()
        ref_gauss = oracle.gauss_rgba(f, 5, 1.5)
        if not np.array_equal(o_gauss.cpu().numpy(), ref_gauss): bad.append((n, "gauss5"))
        if not np.array_equal(o_sobel.cpu().numpy(), oracle.sobel_rgba(ref_gauss)): bad.append((n, "sobel"))
        if not np.array_equal(o_pipe.cpu().numpy(), oracle.pipeline_rgba(f, 5, 1.5)): bad.append((n, "pipeline"))
        if not np.array_equal(o_g17.cpu().numpy(), oracle.gauss_rgba(f, 17, 6.0)): bad.append((n, "gauss17"))
    del g
    c.close()
print(bad)
"""


@pytest.mark.gpu
def test_device_resident_calls_can_be_captured_into_a_hip_graph():
    """Once the (k, sigma) tables and the scratch buffers of a frame size exist, mi355_filter_dev allocates nothing and
    synchronises nothing, so a chain of calls on the context's stream can be captured into a hipGraph and replayed
    (include/mi355_imgfilter.h says so): Gaussian 5x5 -> Sobel, the fused pipeline and a 17x17 Gaussian, replayed on the
    captured frame and on new content, equal the oracle.  Own process: torch's HIP runtime has to initialise before the
    library is loaded.  (tools/graph_probe.py times it: at 640x480 the eager chain is already bound by the kernels, 34 us
    against 38 us replayed - the property matters to callers who capture larger graphs around these calls.)"""
    out = subprocess.run([sys.executable, "-c", _GRAPH_SCRIPT, entry.ROOT], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert out.stdout.strip().splitlines()[-1] == "[]", out.stdout[-2000:]
