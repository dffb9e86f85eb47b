"""The entry point that allocates nothing (alilqr_solve_batch_noalloc) against alilqr_solve_batch, through the C ABI:
bit-identical outputs on two calls in a row, the second on a smaller batch (the static scratch keeps the first
call's contents), and on a batch whose scratch exceeds the static array (fewer resident groups, same bits)."""
import ctypes as C

import numpy as np
import pytest

from ilqr_amd import batch, capi, problems

pytestmark = pytest.mark.gpu

KEYS = ("x", "u", "node_x", "node_u", "cost", "viol", "mu", "status")


def _run(solver, noalloc):
    import torch
    o = solver.run(noalloc=noalloc)
    torch.cuda.synchronize()
    return {k: o[k].cpu().numpy().copy() for k in KEYS}


def _pair(prob):
    a = batch.BatchSolver(*prob, extras=True)
    b = batch.BatchSolver(*prob, extras=True)
    return a, b


def _same(x, y, what):
    for k in KEYS:
        assert np.array_equal(x[k], y[k], equal_nan=True), (what, k)


def test_noalloc_matches_the_pooled_entry_point_twice_in_a_row():
    big = problems.lat_legacy(96, 50, seed=41)
    small = problems.lat_legacy(33, 50, seed=42)
    ref_big = _run(batch.BatchSolver(*big, extras=True), False)
    ref_small = _run(batch.BatchSolver(*small, extras=True), False)
    assert np.isfinite(ref_big["cost"]).all() and (ref_big["status"][:, 0] > 0).all()
    s_big, s_small = batch.BatchSolver(*big, extras=True), batch.BatchSolver(*small, extras=True)
    _same(_run(s_big, True), ref_big, "first call")
    _same(_run(s_small, True), ref_small, "second call, smaller batch")
    _same(_run(s_big, True), ref_big, "third call")


def test_noalloc_other_generations():
    for prob in (problems.fast_box(24), problems.new_al_box(24, H=30, seed=5), problems.full_legacy("dyn", 24, 60, 7)):
        ref = _run(batch.BatchSolver(*prob, extras=True), False)
        _same(_run(batch.BatchSolver(*prob, extras=True), True), ref, "generation")


def test_noalloc_batch_larger_than_the_static_scratch():
    """More problems than fit into the static array at once: the resident groups pull the rest."""
    lib = capi.load_library()
    prob = problems.lat_legacy(4096, 50, seed=43)
    need = lib.alilqr_workspace_bytes(C.byref(prob[0]))
    assert need > lib.alilqr_static_scratch_bytes() > 0
    ref = _run(batch.BatchSolver(*prob, extras=True), False)
    _same(_run(batch.BatchSolver(*prob, extras=True), True), ref, "large batch")


def test_noalloc_rejects_a_horizon_whose_single_wave_does_not_fit():
    desc, x0, goals, Q, R, cons = problems.lat_legacy(2, 200000, seed=1)
    s = batch.BatchSolver(desc, x0, goals, Q, R, cons)
    with pytest.raises(RuntimeError, match="static scratch"):
        s.run(noalloc=True)
