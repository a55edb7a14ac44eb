"""Signal sharding across ranks (dsp_amd.dist) on CPU with the gloo backend, world_size 2: the block
partition, the padded equal-size gather and the reassembly.  The per-shard compute is injected (the
CPU oracle here; on the GPU box the default is the HIP path), so what is tested is exactly the N > 1
logic that bench.py --gpus N and multi-GPU users run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dsp_amd.dist import shard_range, stft_sharded


def test_shard_range_partitions_exactly():
    for B in (0, 1, 5, 8, 1024, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_range(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d
    assert shard_range(65536, 3, 8) == (3 * 8192, 4 * 8192)


def _oracle_compute(x, win, **kw):
    from oracle import oracle as O
    ys = []
    for b in range(x.shape[0]):
        rc, y = O.stft(x[b].numpy(), win.numpy(), kw.get("stp", 160), 512, kw.get("snip_edges", False))
        assert rc == 0
        ys.append(torch.from_numpy(y))
    W = O.num_frames(x.shape[1], win.numel(), kw.get("stp", 160), kw.get("snip_edges", False))
    return torch.stack(ys) if ys else torch.empty((0, W, 257), dtype=x.dtype)


def _worker(rank, world, port, B, gather, q, chunks=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        x = torch.randn((B, 4000), generator=g, dtype=torch.float32)      # same on every rank
        win = torch.hann_window(400, periodic=False, dtype=torch.float32)
        lo, hi = shard_range(B, rank, world)
        y_local, y_full = stft_sharded(x[lo:hi], win, B, rank, world, gather=gather, compute=_oracle_compute, chunks=chunks, stp=160)
        ref = _oracle_compute(x, win, stp=160)
        ok = torch.equal(y_local, ref[lo:hi])
        if gather == "all" or (gather == "root" and rank == 0):
            ok = ok and y_full is not None and torch.equal(y_full, ref)
        else:
            ok = ok and (y_full is None)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B,gather,chunks", [(6, "all", 1), (5, "all", 1), (5, "root", 1), (1, "all", 1), (4, "none", 1),
                                             (7, "root", 3), (8, "root", 2), (1, "root", 2)])
def test_sharded_stft_world2(B, gather, chunks):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, gather, q, chunks)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _bench_worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        g = torch.Generator().manual_seed(3)
        x = torch.randn((world * B, 3000), generator=g, dtype=torch.float32)      # same on every rank
        win = torch.hann_window(400, periodic=False, dtype=torch.float32)
        res, full = bench.gather_leg(x[rank * B:(rank + 1) * B], win, B, world, rank, _oracle_compute, 2, dist.barrier, stp=160)
        ok = res["chunks"] == 2 and res["pipelined_ms"] > 0 and res["compute_then_gather_ms"] > 0
        if rank == 0:
            ok = ok and torch.equal(full, _oracle_compute(x, win, stp=160))
        else:
            ok = ok and full is None
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_bench_gather_leg_world2():
    """The benchmark's own multi-GPU leg (bench.gather_leg: equal shards per rank, gather to rank 0 alone and
    pipelined with the computation) on two gloo ranks with the CPU oracle as the per-shard compute."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
