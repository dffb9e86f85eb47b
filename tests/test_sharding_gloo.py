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
