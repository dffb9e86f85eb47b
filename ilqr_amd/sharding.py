"""Sharding a batch of independent problems over the GPUs of a node (one process per GPU).

Problems never exchange data during a solve, so the only collective is the optional gather of results
(RCCL when the tensors live on GPUs, gloo on CPU tensors).
"""
import numpy as np


def shard_range(batch, rank, world):
    """Contiguous block [lo, hi) of problem indices owned by ``rank``: blocks differ in size by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("rank must be in [0, world)")
    base, extra = divmod(int(batch), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(batch, world):
    return [shard_range(batch, r, world)[1] - shard_range(batch, r, world)[0] for r in range(world)]


def gather_rows(local, batch, group=None):
    """All-gather per-problem rows ``local`` [b_rank, ...] into [batch, ...] in problem order on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = shard_sizes(batch, world)
    biggest = max(sizes)
    pad = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)


def best_start(cost, starts_per_scenario):
    """Multi-start selection: ``cost`` [scenarios * starts] -> (best cost, best start index) per scenario.
    Starts of one scenario are contiguous, so with whole scenarios per rank this needs no communication."""
    c = np.asarray(cost).reshape(-1, starts_per_scenario)
    c = np.where(np.isfinite(c), c, np.inf)
    idx = np.argmin(c, axis=1)
    return c[np.arange(c.shape[0]), idx], idx
