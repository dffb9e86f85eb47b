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
    # shard sizes as the ranks really hold them (multi-start shards are whole scenarios)
    mine = torch.tensor([int(local.shape[0])], dtype=torch.int64, device=local.device)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    sizes = [int(t.item()) for t in every]
    if sum(sizes) != int(batch):
        raise ValueError(f"ranks hold {sum(sizes)} rows, expected {batch}")
    biggest = max(max(sizes), 1)
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


def _dist_state(group=None):
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(group), dist.get_world_size(group)
    except ImportError:
        pass
    return 0, 1


def local_problem(desc, x0, goals, u_offset, lo, hi):
    """Descriptor and per-problem inputs of the problems [lo, hi) of a global batch (tables stay shared)."""
    d = type(desc).from_buffer_copy(desc)
    d.batch = hi - lo
    x0 = np.asarray(x0, dtype=np.float64).reshape(desc.batch, -1)[lo:hi]
    goals = np.asarray(goals, dtype=np.float64)
    if desc.goal_batch > 1:
        goals = goals.reshape(desc.batch, desc.horizon + 1, desc.n)[lo:hi]
        d.goal_batch = hi - lo
    if u_offset is not None:
        u_offset = np.asarray(u_offset, dtype=np.float64).reshape(desc.batch, desc.horizon, desc.m)[lo:hi]
    return d, np.ascontiguousarray(x0), np.ascontiguousarray(goals), u_offset


def solve_batch_sharded(desc, x0, goals, Q, R, cons, u_offset=None, starts_per_scenario=None, gather=True,
                        solver=None, device=None, group=None, **kw):
    """Solve a GLOBAL batch with one process per GPU: every rank solves its contiguous shard, no exchange
    during the solve; one optional gather of the results at the end (RCCL for GPU tensors, gloo for CPU ones).

    ``desc``/``x0``/``goals``/``u_offset`` describe the whole batch and are the same on every rank.
    ``starts_per_scenario`` (multi-start): the batch is scenarios x starts, whole scenarios stay on one rank, and
    the result carries the per-scenario winner (``best_cost``, ``best_start`` and its ``best_x``, ``best_u``).
    ``solver(desc, x0, goals, Q, R, cons, u_offset=..., **kw) -> dict of torch tensors`` defaults to
    ``ilqr_amd.batch.solve_batch`` on ``cuda:LOCAL_RANK`` (tests inject a stand-in to run on CPU).
    Returns a dict; with ``gather`` the per-problem tensors cover the whole batch in problem order on every rank,
    otherwise the rank's shard only (``range`` tells which).
    """
    import os
    import torch
    rank, world = _dist_state(group)
    B = int(desc.batch)
    if starts_per_scenario:
        if B % starts_per_scenario:
            raise ValueError("batch must be scenarios * starts_per_scenario")
        slo, shi = shard_range(B // starts_per_scenario, rank, world)
        lo, hi = slo * starts_per_scenario, shi * starts_per_scenario
    else:
        lo, hi = shard_range(B, rank, world)
    d, x0_l, goals_l, uo_l = local_problem(desc, x0, goals, u_offset, lo, hi)
    if solver is None:
        from . import batch as _batch
        dev = device or f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"

        def solver(dd, xx, gg, Qq, Rr, cc, u_offset=None, **k2):
            # an empty shard (more ranks than problems) goes the same way: solve_batch returns every output of the
            # call with zero rows, so all ranks hold the same keys for the gather below
            return _batch.solve_batch(dd, xx, gg, Qq, Rr, cc, device=dev, u_offset=u_offset, **k2)
    out = solver(d, x0_l, goals_l, Q, R, cons, u_offset=uo_l, **kw)
    out = {k: (v if isinstance(v, torch.Tensor) else torch.as_tensor(v)) for k, v in out.items()}
    res = {"range": (lo, hi)}
    if starts_per_scenario:
        S = starts_per_scenario
        cost = out["cost"].detach().cpu().numpy()
        bc, bi = best_start(cost, S) if cost.size else (np.zeros(0), np.zeros(0, dtype=np.int64))
        sel = torch.as_tensor(np.arange(bc.shape[0]) * S + bi, dtype=torch.long, device=out["x"].device)
        res.update(best_cost=torch.as_tensor(bc, dtype=torch.float64, device=out["x"].device),
                   best_start=torch.as_tensor(bi, dtype=torch.int64, device=out["x"].device),
                   best_x=out["x"].index_select(0, sel), best_u=out["u"].index_select(0, sel))
    if not gather or world == 1:
        res.update(out)
        return res
    if starts_per_scenario:
        n_scen = B // starts_per_scenario
        for k in ("best_cost", "best_start", "best_x", "best_u"):
            res[k] = gather_rows(res[k], n_scen, group)
    # every rank must issue the same collectives in the same order: gather over the sorted key list, after checking
    # that all ranks hold the same one (a solver that returns other outputs for an empty shard would deadlock here)
    keys = sorted(out)
    _check_same_keys(keys, out[keys[0]].device if keys else "cpu", group)
    for k in keys:
        res[k] = gather_rows(out[k], B, group)
    return res


def _check_same_keys(keys, device, group=None):
    import zlib
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    mine = torch.tensor([len(keys), zlib.crc32(",".join(keys).encode())], dtype=torch.int64, device=device)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    if any(not torch.equal(t, mine) for t in every):
        raise RuntimeError("ranks returned different sets of outputs from their shard solves; cannot gather")
