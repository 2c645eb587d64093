"""N>1 path on CPU: world_size 2 over gloo.  The frames shard by index with no data-path collective; the
only collectives are the broadcast of the coefficient table and the reduction of timing / checksum scalars.
This test runs bench.py's host-side logic for two ranks with the oracle standing in for the GPU kernels
(test infrastructure) and checks that the sharded result equals the single-rank result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    oracle = entry.load_oracle()
    pkg = entry.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, frames_per_rank, k, sigma = 96, 40, 3, 5, 1.5
    first = rank * frames_per_rank
    frames = oracle.synth_rgba(w, h, frames_per_rank, first_frame=first)
    # coefficient table: rank 0 generates (host function of the product library), everyone receives
    table = torch.zeros(k * k, dtype=torch.float32)
    if rank == 0:
        table.copy_(torch.from_numpy(pkg.gauss_weights(k, sigma).reshape(-1)))
    dist.broadcast(table, src=0)
    weights = table.numpy().reshape(k, k)
    out = np.stack([oracle.pipeline_rgba(f, k, weights=weights) for f in frames])
    words_per_frame = w * h // 4
    ck = oracle.checksum(out, index_base=first * words_per_frame)
    c = torch.tensor([ck & 0xFFFFFFFF, ck >> 32], dtype=torch.int64)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    total = (int(c[0]) + (int(c[1]) << 32)) & 0xFFFFFFFFFFFFFFFF
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.array([total, int(float(t[0]) * 10), ck], dtype=np.uint64))
    np.save(os.path.join(out_dir, "table%d.npy" % rank), weights)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_frames_and_agree(tmp_path, oracle, pkg):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    assert r0[0] == r1[0] and r0[1] == r1[1] == 15          # same reduced checksum, max-over-ranks time
    t0, t1 = np.load(tmp_path / "table0.npy"), np.load(tmp_path / "table1.npy")
    assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
    assert np.array_equal(t0.view(np.uint32), oracle.gauss_weights(5, 1.5).view(np.uint32))
    # single-rank run over the same six frames gives the same checksum: sharding changes nothing
    frames = oracle.synth_rgba(96, 40, 6, first_frame=0)
    whole = np.stack([oracle.pipeline_rgba(f, 5, 1.5) for f in frames])
    assert oracle.checksum(whole) == int(r0[0])
    assert (int(r0[2]) + int(r1[2])) % (1 << 64) == int(r0[0])
