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


def stft_sharded(x_shard, win, B_total, rank=None, world=None, gather="none", compute=None, group=None, **kw):
    """STFT of this rank's shard x_shard[hi-lo, N] of a batch of B_total signals.

    gather: "none" -> (Y_local, None); "root" -> rank 0 also gets Y[B_total, W, F]; "all" -> every rank
    gets it.  compute(x, win, **kw) -> Y[b, W, F] defaults to the device path dsp_amd.stft_dev.
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
            parts = [torch.empty_like(send) for _ in range(world)]
            dist.gather(send, parts, dst=0, group=group)
            return y, torch.cat(parts, dim=0)[:B_total]
        dist.gather(send, None, dst=0, group=group)
        return y, None
    raise ValueError(f"gather must be 'none', 'root' or 'all', not {gather!r}")
