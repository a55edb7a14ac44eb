"""Signal sharding for multi-GPU runs: one process per GPU, signals split in contiguous blocks, no
collective on the data path (each signal's STFT is independent); one optional gather of the
spectrogram shards at the end (RCCL over xGMI when the process group's backend is "nccl")."""
import torch
import torch.distributed as dist


def shard_range(B, rank, world):
    """Contiguous block partition of B signals: rank r owns [lo, hi), blocks of ceil(B / world)."""
    per = (B + world - 1) // world
    lo = min(B, rank * per)
    hi = min(B, lo + per)
    return lo, hi


def _root_of(group):
    """Global rank of the group's rank 0 (dist.gather's dst is a global rank)."""
    return dist.get_global_rank(group, 0) if group is not None else 0


def stft_sharded_pipelined(x_shard, win, B_total, rank, world, chunks, compute, group=None, **kw):
    """The shard in `chunks` pieces, the gather of piece k to the group's rank 0 running while piece k + 1 is
    computed (the gather of the spectrogram costs many times its computation: 8.4 GB per rank over xGMI against
    3 ms of kernel for 8192 utterances).  Every rank sends equal pieces (short shards padded), the root receives
    them straight into their place in Y[B_total, W, F].  Returns (Y_local, Y_full or None)."""
    lo, hi = shard_range(B_total, rank, world)
    nloc, per = hi - lo, (B_total + world - 1) // world
    root = _root_of(group)
    works, parts_local, full = [], [], None
    for c in range(chunks):
        c0, c1 = c * per // chunks, (c + 1) * per // chunks
        if c1 == c0:
            continue
        yc = compute(x_shard[min(c0, nloc):min(c1, nloc)], win, **kw)
        W, F = yc.shape[1], yc.shape[2]
        send = yc
        if yc.shape[0] < c1 - c0:                      # a short or empty shard: padded to the common piece size
            send = torch.zeros((c1 - c0, W, F), dtype=yc.dtype, device=yc.device)
            send[: yc.shape[0]] = yc
        send = send.contiguous()
        if rank == 0:
            if full is None:
                full = torch.empty((world * per, W, F), dtype=yc.dtype, device=yc.device)
            into = [full[r * per + c0: r * per + c1] for r in range(world)]
            works.append((dist.gather(send, into, dst=root, group=group, async_op=True), send))
        else:
            works.append((dist.gather(send, None, dst=root, group=group, async_op=True), send))
        parts_local.append(yc)
    for w, _ in works:
        w.wait()
    y_local = torch.cat(parts_local, dim=0) if parts_local else None
    return y_local, (full[:B_total] if rank == 0 and full is not None else None)


def stft_sharded(x_shard, win, B_total, rank=None, world=None, gather="none", compute=None, group=None, chunks=1, **kw):
    """STFT of this rank's shard x_shard[hi-lo, N] of a batch of B_total signals.

    gather: "none" -> (Y_local, None); "root" -> rank 0 also gets Y[B_total, W, F]; "all" -> every rank
    gets it.  compute(x, win, **kw) -> Y[b, W, F] defaults to the device path dsp_amd.stft_dev.
    chunks > 1 (gather "root"): the shard goes in that many pieces whose gathers overlap the computation.
    """
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if compute is None:
        from .api import stft_dev as compute
    lo, hi = shard_range(B_total, rank, world)
    if x_shard.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} owns signals [{lo}, {hi}) but was given {x_shard.shape[0]} rows")
    if gather == "root" and chunks > 1 and world > 1:
        return stft_sharded_pipelined(x_shard, win, B_total, rank, world, chunks, compute, group=group, **kw)
    y = compute(x_shard, win, **kw)
    if gather == "none" or world == 1:
        return y, (y if gather != "none" else None)
    per = (B_total + world - 1) // world
    W, F = y.shape[1], y.shape[2]
    # equal-sized messages: the last (short or empty) shards are padded to `per` signals
    send = y
    if hi - lo < per:
        send = torch.zeros((per, W, F), dtype=y.dtype, device=y.device)
        send[: hi - lo] = y
    send = send.contiguous()
    if gather == "all":
        full = torch.empty((world * per, W, F), dtype=y.dtype, device=y.device)
        dist.all_gather_into_tensor(full, send, group=group)
        return y, full[:B_total]
    if gather == "root":
        if rank == 0:
            # received straight into their place in Y[B_total, W, F] (no second copy of world * per signals)
            full = torch.empty((world * per, W, F), dtype=y.dtype, device=y.device)
            parts = [full[r * per:(r + 1) * per] for r in range(world)]
            dist.gather(send, parts, dst=_root_of(group), group=group)
            return y, full[:B_total]
        dist.gather(send, None, dst=_root_of(group), group=group)
        return y, None
    raise ValueError(f"gather must be 'none', 'root' or 'all', not {gather!r}")
