"""Scratch pool of the library: any number of streams, release between solves, a solver on a non-current device."""
import numpy as np
import pytest

from ilqr_amd import batch, capi, problems

pytestmark = pytest.mark.gpu


def _np(o):
    return {k: v.cpu().numpy().copy() for k, v in o.items()}


def test_more_streams_than_any_fixed_table():
    """40 streams solve concurrently (torch's pool holds 32, so handles repeat): every result equals the
    single-stream one."""
    import torch
    prob = problems.lat_legacy(64, 50, seed=51)
    ref = _np(batch.solve_batch(*prob, extras=False))
    solvers = [batch.BatchSolver(*prob) for _ in range(40)]
    streams = [torch.cuda.Stream() for _ in range(40)]
    for sv, st in zip(solvers, streams):
        with torch.cuda.stream(st):
            sv.run(st.cuda_stream)
    torch.cuda.synchronize()
    for sv in solvers:
        for k in ("x", "u", "cost", "status"):
            assert np.array_equal(sv.out[k].cpu().numpy(), ref[k], equal_nan=True), k


def test_release_between_solves():
    prob = problems.lat_legacy(48, 50, seed=52)
    a = _np(batch.solve_batch(*prob))
    batch.release()
    b = _np(batch.solve_batch(*prob))
    batch.release()
    c = _np(batch.solve_batch(*prob))
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True) and np.array_equal(a[k], c[k], equal_nan=True), k


def test_growing_and_shrinking_batches_on_one_stream():
    """The stream's slot grows with the largest batch; smaller solves afterwards reuse it."""
    outs = {}
    for B in (8, 300, 8, 64, 300):
        prob = problems.lat_legacy(B, 50, seed=53)
        o = _np(batch.solve_batch(*prob, extras=False))
        if B in outs:
            for k in o:
                assert np.array_equal(o[k], outs[B][k], equal_nan=True), (B, k)
        outs[B] = o


def test_solver_runs_on_its_own_device():
    """BatchSolver.run makes its device current for the call (with one GPU: the same device, by index and by default)."""
    import torch
    prob = problems.lat_legacy(16, 50, seed=54)
    last = torch.cuda.device_count() - 1
    ref = _np(batch.solve_batch(*prob, extras=False, device=f"cuda:{last}"))
    sv = batch.BatchSolver(*prob, device=f"cuda:{last}")
    with torch.cuda.device(0):
        sv.run()
    torch.cuda.synchronize(last)
    for k in ("x", "u", "cost"):
        assert np.array_equal(sv.out[k].cpu().numpy(), ref[k]), k
