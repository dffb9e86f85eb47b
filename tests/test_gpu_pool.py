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


def test_many_batches_in_one_call_give_the_bits_of_separate_calls():
    """alilqr_solve_batches: batches with buffers of their own solved as one batch (staging gather / scatter inside the
    library) - every problem gets the bits alilqr_solve_batch gives it; optional outputs may be left out per list."""
    import torch
    import ctypes as C
    seeds = (31, 32, 33, 34, 35)
    probs = [problems.lat_legacy(200, 50, seed=s) for s in seeds]
    want = [{k: v.cpu().numpy() for k, v in batch.solve_batch(*p).items()} for p in probs]
    desc, _, goals, Q, R, cons = probs[0]
    for extras in (False, True):
        mb = batch.MultiBatchSolver(desc, [p[1] for p in probs], goals, Q, R, cons, extras=extras)
        for rep in range(2):                            # the second call reuses the scratch slot and its staging
            outs = mb.run()
            torch.cuda.synchronize()
            for o, w in zip(outs, want):
                for k in o:
                    assert np.array_equal(o[k].cpu().numpy(), w[k], equal_nan=True), (extras, rep, k)
    # a 6-state generation with a table, and an empty list
    fp = [problems.fast_box_linear(40, H=50, seed=s) for s in (5, 6, 7)]
    fw = [{k: v.cpu().numpy() for k, v in batch.solve_batch(*p).items()} for p in fp]
    mb = batch.MultiBatchSolver(fp[0][0], [p[1] for p in fp], *fp[0][2:], extras=True)
    outs = mb.run()
    torch.cuda.synchronize()
    for o, w in zip(outs, fw):
        for k in o:
            assert np.array_equal(o[k].cpu().numpy(), w[k], equal_nan=True), k
    lib = capi.load_library()
    assert lib.alilqr_solve_batches(C.byref(desc), 0, *([None] * 15)) == 0
    bad = type(desc).from_buffer_copy(desc)
    bad.goal_batch = bad.batch
    assert lib.alilqr_solve_batches(C.byref(bad), 1, *([None] * 15)) == capi.E_ARG


def test_pull_order_hint_changes_nothing_but_the_schedule(tuning):
    """alilqr_extras::order / the order argument of alilqr_solve_batches: any permutation, with fewer resident groups
    than problems, gives every problem the same bits; entries outside the batch are ignored."""
    import torch
    tuning(lanes=32, max_waves=8)                      # 16 groups for 150 problems: the order really decides who runs when
    prob = problems.lat_legacy(150, 50, seed=61)
    ref = _np(batch.solve_batch(*prob))
    by_work = np.argsort(-ref["status"][:, 0], kind="stable").astype(np.int32)
    for order in (by_work, by_work[::-1].copy(), np.random.default_rng(3).permutation(150).astype(np.int32)):
        o = _np(batch.solve_batch(*prob, order=order))
        for k in ref:
            assert np.array_equal(o[k], ref[k], equal_nan=True), k
    junk = by_work.copy(); junk[7] = -5; junk[9] = 150
    bad = _np(batch.solve_batch(*prob, order=junk))    # no fault; the two positions fall back to their own index
    assert np.isfinite(bad["cost"][by_work[:5]]).all()
    tuning()
    mb = batch.MultiBatchSolver(prob[0], [prob[1], problems.lat_legacy(150, 50, seed=62)[1]], *prob[2:6])
    a = [_np(o) for o in mb.run()]
    torch.cuda.synchronize()
    b = [_np(o) for o in mb.run(order=mb.order_by_last_solve())]
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        for k in x:
            assert np.array_equal(x[k], y[k], equal_nan=True), k


def test_many_batches_with_optional_inputs_over_the_concatenated_problems():
    """alilqr_solve_batches with an alilqr_extras: multi-start control offsets (one contiguous array over all batches)
    and rectangle obstacles (shared) give the bits of alilqr_solve_batch_ex on the concatenated batch."""
    import ctypes as C
    import torch
    prob, kw = problems.new_al_lat_rect(48)
    desc, x0, goals, Q, R, cons = prob
    uo = 0.01 * np.random.default_rng(5).standard_normal((48, desc.horizon, desc.m))
    want = _np(batch.solve_batch(*prob, u_offset=uo, **kw))
    half = type(desc).from_buffer_copy(desc)
    half.batch = 24
    mb = batch.MultiBatchSolver(half, [x0[:24], x0[24:]], goals, Q, R, cons, extras=True)
    uo_d = torch.as_tensor(uo, device="cuda")
    lo = torch.as_tensor(capi.pack_rectangles(kw["left_obs"]), device="cuda")
    ro = torch.as_tensor(capi.pack_rectangles(kw["right_obs"]), device="cuda")
    ex = capi.Extras(uo_d.data_ptr(), lo.data_ptr(), ro.data_ptr(), len(kw["left_obs"]), len(kw["right_obs"]), kw["first_row"], 0)
    L = mb._lists
    P = lambda t: C.c_void_p(t.data_ptr())
    rc = mb.lib.alilqr_solve_batches(C.byref(half), 2, L[0], *[P(t) for t in mb.shared], *L[1:], C.byref(ex),
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, capi.last_error()
    torch.cuda.synchronize()
    for i, o in enumerate(mb.out):
        for k in o:
            assert np.array_equal(o[k].cpu().numpy(), want[k][24 * i:24 * (i + 1)], equal_nan=True), (i, k)
