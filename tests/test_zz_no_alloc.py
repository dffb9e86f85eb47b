"""The entry point that allocates nothing (alilqr_solve_batch_noalloc) against alilqr_solve_batch, through the C ABI:
bit-identical outputs on two calls in a row, the second on a smaller batch (the static scratch keeps the first
call's contents), and on a batch whose scratch exceeds the static array (fewer resident groups, same bits).  Also that
the call really allocates nothing: the device's free memory does not move, and the call can be captured into a HIP
graph in the capture mode that refuses allocations, and replayed."""
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


def test_noalloc_leaves_the_free_device_memory_where_it_was():
    """hipMemGetInfo before and after: the call takes no device memory (the pooled entry point, on a size it has not
    seen, does)."""
    import torch
    prob = problems.lat_legacy(700, 50, seed=44)
    s = batch.BatchSolver(*prob, extras=True)          # inputs and outputs are allocated here, not by run()
    s.run(noalloc=True)                                # first call on the device: address look-up
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        s.run(noalloc=True)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    capi.release()                                     # the pooled entry point has to allocate its scratch again
    free1 = torch.cuda.mem_get_info()[0]
    s.run(noalloc=False)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] < free1        # the measure does see an allocation made by the library


def test_noalloc_call_is_captured_into_a_graph_and_replayed():
    """Stream capture in the mode that makes every allocating runtime call fail: the call is captured (one memset node,
    the row-analysis and solve kernels) and each replay of the graph solves the batch again, same bits."""
    import torch
    for prob in (problems.lat_legacy(130, 50, seed=45), problems.fast_box(24)):
        ref = _run(batch.BatchSolver(*prob, extras=True), False)
        s = batch.BatchSolver(*prob, extras=True)
        s.run(noalloc=True)                            # outside capture: the first call looks the scratch address up
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=torch.cuda.Stream(), capture_error_mode="global"):
            s.run(noalloc=True)                        # enqueued on the capturing stream (the current one in here)
        for _ in range(2):
            for k in KEYS:
                s.out[k].zero_()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            _same({k: s.out[k].cpu().numpy() for k in KEYS}, ref, "graph replay")
