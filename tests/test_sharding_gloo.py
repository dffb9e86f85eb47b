"""Multi-rank path on CPU: two gloo ranks shard a batch and gather the per-problem rows back in order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ilqr_amd import sharding


def test_shard_ranges_partition_the_batch():
    for B in (0, 1, 7, 4096, 4097):
        for W in (1, 2, 3, 8):
            r = [sharding.shard_range(B, k, W) for k in range(W)]
            assert r[0][0] == 0 and r[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(4, 2, 2)


def test_best_start_picks_the_cheapest_finite_start():
    cost = np.array([3.0, 1.0, np.nan, 2.0, 5.0, 4.0, np.inf, 4.5])
    c, i = sharding.best_start(cost, 4)
    assert list(i) == [1, 1] and list(c) == [1.0, 4.0]


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(B, rank, world)
    # stand-in for a rank's solve output: row i of problem i is [i, 2i, 3i]
    local = torch.arange(lo, hi, dtype=torch.float64)[:, None] * torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64)
    full = sharding.gather_rows(local, B)
    cost = sharding.gather_rows(torch.arange(lo, hi, dtype=torch.float64), B)
    q.put((rank, full.numpy(), cost.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [7, 64])
def test_two_rank_gather_is_in_problem_order(B):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.arange(B, dtype=np.float64)[:, None] * np.array([1.0, 2.0, 3.0])
    for rank, full, cost in got:
        assert np.array_equal(full, want) and np.array_equal(cost, np.arange(B, dtype=np.float64))


# ---------------------------------------------------------------------------
# solve_batch_sharded on two gloo ranks with a stand-in solver (no GPU): shard, solve, gather, per-scenario arg-min
# ---------------------------------------------------------------------------
def _standin_solver(desc, x0, goals, Q, R, cons, u_offset=None, **kw):
    """Deterministic function of each problem's own inputs: what a rank would return for its shard."""
    B, H, n, m = desc.batch, desc.horizon, desc.n, desc.m
    x0 = np.asarray(x0).reshape(B, n)
    x = np.repeat(x0[:, None, :], H + 1, axis=1) + np.arange(H + 1)[None, :, None]
    u = np.zeros((B, H, m)) + x0[:, :1, None]
    if u_offset is not None:
        u = u + np.asarray(u_offset).reshape(B, H, m)
    cost = (x0 ** 2).sum(axis=1) + (0.0 if u_offset is None else np.abs(np.asarray(u_offset).reshape(B, H * m)).sum(axis=1))
    status = np.tile(np.array([[3, 1, 1, 7]], dtype=np.int32), (B, 1))
    return dict(x=torch.as_tensor(x), u=torch.as_tensor(u), cost=torch.as_tensor(cost), status=torch.as_tensor(status))


def _sharded_inputs(B, starts=None):
    from ilqr_amd import capi
    rng = np.random.default_rng(17)
    H = 5
    desc = capi.make_problem("lat", "legacy", H, B, 1.0, 0.1, v=10.0)
    x0 = rng.uniform(-1, 1, size=(B, 4))
    goals = np.zeros((H + 1, 4))
    uo = rng.uniform(-1, 1, size=(B, H, 1)) if starts else None
    return desc, x0, goals, np.eye(4), np.eye(1), np.zeros(10), uo


def _sharded_worker(rank, world, port, B, starts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    desc, x0, goals, Q, R, cons, uo = _sharded_inputs(B, starts)
    res = sharding.solve_batch_sharded(desc, x0, goals, Q, R, cons, u_offset=uo, starts_per_scenario=starts,
                                       solver=_standin_solver)
    q.put((rank, {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in res.items()}))
    dist.barrier()
    dist.destroy_process_group()


# (1, None) and (4, 4): fewer problems (scenarios) than ranks, rank 1 solves an empty shard and still takes part in
# every collective of the gather
@pytest.mark.parametrize("B,starts", [(13, None), (24, 4), (20, 4), (1, None), (4, 4)])
def test_solve_batch_sharded_two_ranks(B, starts):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, B, starts, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # what one process solving the whole batch returns
    desc, x0, goals, Q, R, cons, uo = _sharded_inputs(B, starts)
    want = {k: v.numpy() for k, v in _standin_solver(desc, x0, goals, Q, R, cons, u_offset=uo).items()}
    ranges = sorted(got[r]["range"] for r in got)
    assert ranges[0][0] == 0 and ranges[-1][1] == B and ranges[0][1] == ranges[1][0]
    if starts:
        assert all(lo % starts == 0 and hi % starts == 0 for lo, hi in ranges)      # whole scenarios per rank
        bc, bi = sharding.best_start(want["cost"], starts)
    for rank, res in got.items():
        for k in ("x", "u", "cost", "status"):
            assert np.array_equal(res[k], want[k]), (rank, k)
        if starts:
            assert np.array_equal(res["best_cost"], bc) and np.array_equal(res["best_start"], bi)
            sel = np.arange(B // starts) * starts + bi
            assert np.array_equal(res["best_x"], want["x"][sel]) and np.array_equal(res["best_u"], want["u"][sel])


def test_solve_batch_sharded_single_process_is_the_plain_solve():
    desc, x0, goals, Q, R, cons, uo = _sharded_inputs(9, 3)
    res = sharding.solve_batch_sharded(desc, x0, goals, Q, R, cons, u_offset=uo, starts_per_scenario=3, solver=_standin_solver)
    want = _standin_solver(desc, x0, goals, Q, R, cons, u_offset=uo)
    assert res["range"] == (0, 9) and np.array_equal(res["cost"].numpy(), want["cost"].numpy())
    bc, bi = sharding.best_start(want["cost"].numpy(), 3)
    assert np.array_equal(res["best_start"].numpy(), bi)


def _uneven_solver(desc, *a, **kw):
    out = _standin_solver(desc, *a, **kw)
    if desc.batch == 0:
        out.pop("status")            # a solver that returns fewer outputs for an empty shard
    return out


def _uneven_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    desc, x0, goals, Q, R, cons, uo = _sharded_inputs(1)
    try:
        sharding.solve_batch_sharded(desc, x0, goals, Q, R, cons, solver=_uneven_solver)
        q.put((rank, "no error"))
    except RuntimeError as e:
        q.put((rank, str(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_with_different_outputs_raise_instead_of_deadlocking():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_uneven_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all("different sets of outputs" in msg for msg in got.values()), got
