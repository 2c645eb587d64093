"""N>1 path on CPU: world_size 2 over gloo, running bench.py's OWN functions — shard_range, broadcast_table,
checksum_index_base, reduce_results — with the oracle standing in for the GPU kernel call only (test
infrastructure).  The frames shard by index with no data-path collective; the only collectives are the broadcast of
the coefficient table and the reduction of timing / checksum / pixel-count scalars."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, K, SIGMA = 96, 40, 5, 1.5


def _worker(rank, world, port, out_dir, frames_per_gpu, total_frames):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    import bench
    oracle = entry.load_oracle()
    pkg = entry.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, n = bench.shard_range(rank, world, frames_per_gpu, total_frames)
    frames = oracle.synth_rgba(W, H, n, first_frame=first)
    # the product library's host generator on rank 0, broadcast to everyone (bench.py's own function)
    table = bench.broadcast_table(dist, rank, K, lambda: pkg.gauss_weights(K, SIGMA), torch.device("cpu"))
    out = np.stack([oracle.pipeline_rgba(f, K, weights=table) for f in frames])   # stands in for mi355_filter_dev
    ck = oracle.checksum(out, index_base=bench.checksum_index_base(first, W, H, 1))
    red = bench.reduce_results(dist, 0.5 + rank, 0.25 + rank, ck, n * W * H, torch.device("cpu"))
    np.save(os.path.join(out_dir, "rank%d.npy" % rank),
            np.array([red["checksum"], int(red["t_max"] * 10), int(red["ms_max"] * 100), red["pixels"], ck, first, n],
                     dtype=np.uint64))
    np.save(os.path.join(out_dir, "table%d.npy" % rank), table)
    per = bench.gather_per_rank(dist, rank, world, {
        "first_frame": first, "frames": n, "avg_launch_ms": 0.25 + rank, "wall_s": 0.5 + rank,
        "probe_first_ms": 1.0 + rank, "probe_kept_ms": 0.75 + rank, "pool_candidates": 3 + rank}, torch.device("cpu"))
    import json
    json.dump(per, open(os.path.join(out_dir, "per_rank%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("frames_per_gpu,total_frames", [(3, 0), (0, 7)])
def test_two_ranks_shard_frames_and_agree(tmp_path, oracle, pkg, frames_per_gpu, total_frames):
    import bench
    world = 2
    mp.spawn(_worker, args=(world, bench.free_port(), str(tmp_path), frames_per_gpu, total_frames), nprocs=world,
             join=True)
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    nframes = total_frames or world * frames_per_gpu
    # same reduced checksum / max-over-ranks times / summed pixels on both ranks
    assert r0[0] == r1[0] and r0[1] == r1[1] == 15 and r0[2] == r1[2] == 125
    assert r0[3] == r1[3] == nframes * W * H
    # contiguous, disjoint, complete frame ranges
    assert (int(r0[5]), int(r1[5])) == (0, int(r0[6])) and int(r0[6]) + int(r1[6]) == nframes
    # the per-rank records (what makes skew visible in the line): identical on both ranks, one entry per rank
    import json
    p0, p1 = json.load(open(tmp_path / "per_rank0.json")), json.load(open(tmp_path / "per_rank1.json"))
    assert p0 == p1 and [d["rank"] for d in p0] == [0, 1]
    assert [d["avg_launch_ms"] for d in p0] == [0.25, 1.25] and [d["pool_candidates"] for d in p0] == [3, 4]
    assert [(d["first_frame"], d["frames"]) for d in p0] == [(int(r0[5]), int(r0[6])), (int(r1[5]), int(r1[6]))]
    assert bench.gather_per_rank(None, 0, 1, dict(p0[0]), None) == [p0[0]]
    t0, t1 = np.load(tmp_path / "table0.npy"), np.load(tmp_path / "table1.npy")
    assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
    assert np.array_equal(t0.view(np.uint32), oracle.gauss_weights(K, SIGMA).view(np.uint32))
    # a single-rank run over the same frames gives the same checksum: sharding changes nothing
    frames = oracle.synth_rgba(W, H, nframes, first_frame=0)
    whole = np.stack([oracle.pipeline_rgba(f, K, SIGMA) for f in frames])
    assert oracle.checksum(whole) == int(r0[0])
    assert (int(r0[4]) + int(r1[4])) % (1 << 64) == int(r0[0])
    single = bench.reduce_results(None, 1.0, 2.0, oracle.checksum(whole), nframes * W * H, None)
    assert single["checksum"] == int(r0[0]) and single["pixels"] == int(r0[3])


def test_shard_range_and_sample_ids():
    import bench
    for world in (1, 2, 4, 8):
        for total in (512, 7, 8, 9):
            spans = [bench.shard_range(r, world, 256, total) for r in range(world)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == total
            for (a, n), (b, _) in zip(spans, spans[1:]):
                assert a + n == b
        assert [bench.shard_range(r, world, 256) for r in range(world)] == [(r * 256, 256) for r in range(world)]
    assert bench.shard_range(3, 8, 256, 512) == (192, 64)      # BASELINE config 5: 64 frames per GPU
    for n in (1, 2, 3, 4, 64, 256):
        ids = bench.sample_frame_ids(n)
        assert ids[0] == 0 and ids[-1] == n - 1 and len(ids) == min(4, n) and len(set(ids)) == len(ids)
    assert bench.sample_frame_ids(256) == bench.sample_frame_ids(256)


def test_bare_multi_gpu_launch_spawns_ranks_and_fails_cleanly_without_gpus():
    """`python bench.py --gpus 2` with no torch.distributed environment spawns its two ranks itself; on a machine
    without GPUs both stop at "needs a GPU" (not at argument handling) and the exit code is non-zero."""
    if torch.cuda.is_available():
        pytest.skip("this machine has a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--frames", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode != 0
    assert run.stderr.count("bench.py needs a GPU") == 2, run.stderr[-2000:]
    assert "WORLD_SIZE" not in run.stderr
