"""What a NewALILQR object carries from one optimize() call into the next, and the constraint-object calls.

The C++ reference cannot be built here and ships no vectors, so these compare the HIP path with the CPU oracle's
restatement (parity unpinned at the solver level) and, for the table calls, with the defining formulas
(linear_constraints.h:176-265, quadratic_constraints.h:15-90).
"""
import numpy as np
import pytest

import golden_util as gu
import oracle_lib as ol
from ilqr_amd import batch, capi, problems, pybind_api as ip

pytestmark = pytest.mark.gpu


def to_np(out):
    return {k: v.cpu().numpy() for k, v in out.items()}


def test_solver_level_penalty_survives_into_the_next_solve():
    """new_al_ilqr.h:1125-1130 vs :692-695: optimize() resets the nodes' mu to 1 but not the solver's mu_, so the first
    penalty increase of a second call multiplies what the first call left."""
    desc, x0, goals, Q, R, cons = problems.new_al_box(24)
    first = to_np(batch.solve_batch(desc, x0, goals, Q, R, cons))
    ref1 = ol.solve_batch(desc, x0, goals, Q, R, cons)
    ok1 = gu.oracle_stable_mask((desc, x0, goals, Q, R, cons), ref1)        
    assert ok1.sum() >= 3
    assert np.array_equal(first["mu_solver"][ok1], ref1["mu_solver"][ok1]) and (first["mu_solver"] > 1.0).any()
    assert np.array_equal(first["mu_solver"], first["mu"])            # one call: the nodes follow the solver's penalty
    # the second call of both starts from the SAME carried value (the oracle's), so that the two stay comparable
    second = to_np(batch.solve_batch(desc, x0, goals, Q, R, cons, mu_solver=ref1["mu_solver"]))
    ref2 = ol.solve_batch(desc, x0, goals, Q, R, cons, mu_solver=ref1["mu_solver"])
    grew = second["mu_solver"] > ref1["mu_solver"]
    assert grew.any()
    # where the second call raised the penalty at all, the nodes jumped from 1 to (what the first call left) x gain
    ratio = second["mu"][grew] / ref1["mu_solver"][grew]
    assert np.all(ratio >= desc.prm.mu_gain * (1 - 1e-12))
    assert np.all(second["mu"][~grew] == 1.0)                         # no increase: the nodes stayed at their reset value
    ok = ok1 & gu.oracle_stable_mask((desc, x0, goals, Q, R, cons), ref2, mu_solver=ref1["mu_solver"])
    assert np.array_equal(second["mu_solver"][ok], ref2["mu_solver"][ok])
    assert np.array_equal(second["status"][ok, 1], ref2["status"][ok, 1])
    gu.pinned("carry_over/mu_solver second solve", ok)
    for i in np.nonzero(ok)[0]:
        assert gu.solve_err(second, ref2, i) < 1e-6
    # without the carried value the second call is the first call again
    again = to_np(batch.solve_batch(desc, x0, goals, Q, R, cons))
    assert np.array_equal(again["x"], first["x"])


def test_activated_rows_are_returned_and_start_the_next_solve():
    """UpdateConstraints mutates the nodes' tables and row indices (linear_constraints.h:360-373); a second optimize()
    starts from them: per-node tables, per-node start rows."""
    (desc, x0, goals, Q, R, cons), kw = problems.new_al_lat_rect(8)
    H, c, n, m = desc.horizon, desc.c, desc.n, desc.m
    o = to_np(batch.solve_batch(desc, x0, goals, Q, R, cons, overlay=True, **kw))
    ref = ol.solve_batch(desc, x0, goals, Q, R, cons, overlay=True, **kw)
    assert np.array_equal(o["overlay"], ref["overlay"])
    sign, idx = o["overlay"][:, :, :c], o["overlay"][:, :, 2 * c]
    assert (sign != 0).any() and set(np.unique(sign)) <= {-1.0, 0.0, 1.0}
    assert idx.min() >= kw["first_row"] and idx.max() > kw["first_row"]
    # second solve of problem 0: its nodes now own different tables and row indices
    A = np.tile(cons[: c * n].reshape(c, n), (H + 1, 1, 1))
    Bm = cons[c * n: c * (n + m)].reshape(c, m)
    Cv = np.tile(cons[c * (n + m):], (H + 1, 1))
    for t in range(H + 1):
        for i in np.nonzero(sign[0, t])[0]:
            A[t, i, :] = 0.0
            A[t, i, 1] = sign[0, t, i]
            Cv[t, i] = o["overlay"][0, t, c + i]
    tables = np.stack([capi.pack_linear(A[t], Bm, Cv[t]) for t in range(H + 1)])
    d2 = capi.make_problem("lat_k", "new_al", H, 1, desc.L, desc.dt, v=desc.v, k=desc.k, c=c, table_nodes=H + 1, params=desc.prm)
    args = (d2, x0[:1], goals, np.tile(Q, (H + 1, 1, 1)), np.tile(R, (H + 1, 1, 1)), tables)
    kw2 = dict(left_obs=kw["left_obs"], right_obs=kw["right_obs"], first_rows=idx[0].astype(np.int32), overlay=True,
               mu_solver=o["mu_solver"][:1])
    o2 = to_np(batch.solve_batch(*args, **kw2))
    r2 = ol.solve_batch(*args, **kw2)
    assert np.array_equal(o2["overlay"], r2["overlay"]) and np.array_equal(o2["status"], r2["status"])
    assert gu.solve_err(o2, r2, 0) < 1e-6
    # the rows are already there: nothing new is activated, the row indices stay
    assert not (o2["overlay"][0, :, :c] != 0).any() and np.array_equal(o2["overlay"][0, :, 2 * c], idx[0])


def test_second_optimize_of_the_pybind_shaped_solver_keeps_the_objects_state():
    v, dt, L, k, H = 10.0, 0.1, 3.0, 0.001, 50
    goals = problems.s_curve_goals(v, dt, H, False)
    Q = np.diag([1e-1, 1e-1, 1e-0, 1e-9]) * 1e3
    R = np.array([[1.0]]) * 1e2
    A = np.zeros((6, 4)); B = np.zeros((6, 1)); C = np.zeros((6, 1))
    B[0, 0], B[1, 0], C[0, 0], C[1, 0] = 1, -1, -0.6, -0.6
    lin = ip.LinearConstraints4_1_6(A, B, C)
    lin.set_current_constraints_index(1)
    nodes = [ip.NewLatBicycleNodeLinearConstraints4_1_6(L, dt, k, v, goals[i], Q, R, lin) for i in range(H + 1)]
    s = ip.NewALILQR4_1(nodes, np.zeros(4), problems.DEMO_LEFT_OBS, problems.DEMO_RIGHT_OBS)
    s.optimize(50, 100, 1e-4)
    rows = [nd.constraints_obj.get_current_constraints_index() for nd in nodes]
    assert max(rows) > 1 and min(rows) == 1                      # some nodes activated rows, others did not
    hit = [nd for nd in nodes if nd.constraints_obj.get_current_constraints_index() > 1][0].constraints_obj
    assert abs(hit.A[2, 1]) == 1.0 and hit.Cv[2] != 0.0          # the row written: +-y + C' <= 0
    assert hit.constraints(np.array([0.0, 14.0, 0.0, 0.0]), np.zeros(1))[2] == hit.A[2, 1] * 14.0 + hit.Cv[2]
    x1, mu1 = s.get_x_list(), s._mu_solver
    s.optimize(50, 100, 1e-4)                                    # starts from the mutated nodes and the kept mu_
    assert [nd.constraints_obj.get_current_constraints_index() for nd in nodes] == rows
    assert s._mu_solver >= mu1 and np.isfinite(s.get_x_list()).all() and x1.shape == s.get_x_list().shape


def test_constraint_object_calls():
    """constraints / parallel_constraints / constraints_jacobian / constraints_hessian of the table classes
    (constraints_bind.h:47-50, 88-91) run on the GPU; compared with the defining formulas."""
    rng = np.random.default_rng(3)
    A, B, C = rng.standard_normal((16, 6)), rng.standard_normal((16, 2)), rng.standard_normal(16)
    lin = ip.LinearConstraints6_2_16(A, B, C)
    x, u = rng.standard_normal(6), rng.standard_normal(2)
    assert np.allclose(lin.constraints(x, u), A @ x + B @ u + C, rtol=0, atol=1e-13)
    X, U = rng.standard_normal((6, 5)), rng.standard_normal((2, 5))
    pc = lin.parallel_constraints(X, U)
    assert pc.shape == (16, 5) and np.allclose(pc, A @ X + B @ U + C[:, None], rtol=0, atol=1e-13)
    jx, ju = lin.constraints_jacobian(x, u)
    assert np.array_equal(jx, A) and np.array_equal(ju, B)
    box = ip.BoxConstraints4_1([-1, -2, -3, -4], [1, 2, 3, 4], [-0.5], [0.5])
    xb = np.array([2.0, 0.0, -5.0, 1.0])
    assert np.array_equal(box.constraints(xb, np.array([0.7])),
                          [2.0 - 1, 0.0 - 2, -5.0 - 3, 1.0 - 4, -1 - 2.0, -2 - 0.0, -3 + 5.0, -4 - 1.0, 0.7 - 0.5, -0.5 - 0.7])
    Qc = rng.standard_normal((3, 4, 4)); Qc = Qc + Qc.transpose(0, 2, 1)
    Aq, Bq, Cq = rng.standard_normal((3, 4)), rng.standard_normal((3, 1)), rng.standard_normal(3)
    quad = ip.QuadraticConstraints4_1_3(list(Qc), Aq, Bq, Cq)
    xq, uq = rng.standard_normal(4), rng.standard_normal(1)
    want = np.array([xq @ Qc[i] @ xq for i in range(3)]) + Aq @ xq + Bq @ uq + Cq
    assert np.allclose(quad.constraints(xq, uq), want, rtol=0, atol=1e-12)
    Xq, Uq = rng.standard_normal((4, 5)), rng.standard_normal((1, 5))
    wantp = np.array([[Xq[:, j] @ Qc[i] @ Xq[:, j] for j in range(5)] for i in range(3)]) + Aq @ Xq + Bq @ Uq + Cq[:, None]
    assert np.allclose(quad.parallel_constraints(Xq, Uq), wantp, rtol=0, atol=1e-12)
    jx, ju = quad.constraints_jacobian(xq, uq)
    assert np.allclose(jx, Aq + 2 * np.einsum("ijk,k->ij", Qc, xq), rtol=0, atol=1e-12) and np.array_equal(ju, Bq)
    hxx, huu, hxu = quad.constraints_hessian(xq, uq)
    assert np.array_equal(hxx[1], 2.0 * Qc[1]) and not huu[0].any() and not hxu[0].any()
    # the oracle's restatement of the same calls
    desc = capi.make_problem("lat_k", "new_al", 1, 1, 1.0, 0.1, c=3, cons_kind=capi.CONS_QUADRATIC)
    xx = np.zeros((1, 2, 4)); xx[0, 0] = xq
    uu = np.zeros((1, 2, 1)); uu[0, 0] = uq
    lo = ol.linearize_batch(desc, xx, uu, np.zeros((1, 2, 3)), np.ones(1), np.zeros((2, 4)), np.eye(4)[None], np.eye(1)[None],
                            quad.table()[None])
    assert np.allclose(quad.constraints(xq, uq), lo["cval"][0, 0], rtol=0, atol=1e-13)
