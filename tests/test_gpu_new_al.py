"""GPU parity of the C++ generation (NEW_AL): HIP path vs the CPU oracle through the C ABI.

PARITY UNPINNED against the C++ reference itself (it cannot be built here and ships no vectors); the oracle these
tests compare with is a restatement whose node formulas are anchored on the Python goldens (tests/test_oracle_new.py).
Tolerance: 1e-6 relative on x, u and the final AL cost, on the problems the oracle itself reproduces from a start
state moved by 1e-13 (with penalties up to 1e10 some solves amplify one rounding error past any fixed tolerance).
"""
import numpy as np
import pytest

import golden_util as gu
import oracle_lib as ol
from ilqr_amd import batch, capi, problems

pytestmark = pytest.mark.gpu
TOL = 1e-6


def to_np(out):
    return {k: v.cpu().numpy() for k, v in out.items()}


def worst_err(o, ref, i):
    return gu.solve_err(o, ref, i)


def stable_mask(desc, x0, goals, Q, R, cons, ref, probes=2):
    rng = np.random.default_rng(2)
    scale = np.zeros(x0.shape[1]); scale[1:3] = 1e-13
    ok = np.ones(desc.batch, dtype=bool)
    for _ in range(probes):
        ref2 = ol.solve_batch(desc, x0 + rng.uniform(-1, 1, size=x0.shape) * scale, goals, Q, R, cons)
        ok &= np.array([worst_err(ref2, ref, i) < 1e-8 for i in range(desc.batch)])
    return ok


SETS = {
    "box6": lambda B: problems.new_al_box(B),
    "box6_h30": lambda B: problems.new_al_box(B, H=30, seed=9),
    "box6_active_bound": lambda B: problems.new_al_box(B, y_max=20.0),
    "lat4": lambda B: problems.new_al_lat(B),
    "circle6": lambda B: problems.new_al_circle(B),            # quadratic rows (quadratic_constraints.h)
    "circle4": lambda B: problems.new_al_lat_circle(B),
}
# fraction of solves the oracle must reproduce from a 1e-13-perturbed start (the rest are not comparable)
FLOOR = {"box6": 0.8, "box6_h30": 0.8, "lat4": 0.8, "box6_active_bound": 0.25, "circle6": 0.4, "circle4": 0.1}


@pytest.mark.parametrize("lanes", ["auto", "1", "4", "32"])
@pytest.mark.parametrize("name", list(SETS))
def test_new_al_matches_oracle(name, lanes, tuning):
    if lanes != "auto":
        tuning(lanes=int(lanes))
    B = 48
    desc, x0, goals, Q, R, cons = SETS[name](B)
    ref = ol.solve_batch(desc, x0, goals, Q, R, cons)
    o = to_np(batch.solve_batch(desc, x0, goals, Q, R, cons))
    ok = stable_mask(desc, x0, goals, Q, R, cons, ref)
    assert gu.pinned(f"new_al/{name}/lanes={lanes}", ok) >= FLOOR[name], name
    for i in np.nonzero(ok)[0]:
        assert worst_err(o, ref, i) < TOL, (name, i, o["status"][i], ref["status"][i])
    assert gu.rel_err(o["x_init"], ref["x_init"]) < 1e-12
    # same iteration / roll-out counts on the pinned solves, but for one in ten (at least one: a last-ulp difference of
    # the device sin/cos can add or drop an iteration that changes nothing above 1e-6)
    differ = int((o["status"][ok] != ref["status"][ok]).any(axis=1).sum())
    assert differ <= max(1, int(0.1 * ok.sum())), (differ, int(ok.sum()))
    # properties every solve must have, stable or not
    fin = np.isfinite(o["cost"])
    assert np.array_equal(o["x"][:, 0], x0)
    if not name.startswith("circle4"):      # the lateral circle demo mostly ends at the outer cap, in the oracle too
        assert ((o["status"][:, 2] & capi.FLAG_VIOL_OK) != 0)[fin].mean() > 0.5
    conv = (o["status"][:, 2] & capi.FLAG_VIOL_OK) != 0
    assert np.all(o["viol"][conv] < desc.prm.viol_stop)


WIDTH_SETS = {
    "box6": lambda: problems.new_al_box(32, H=30, seed=9),
    "lat4_ten_row_table": lambda: problems.new_al_lat(32, H=30, seed=9),       # two live rows, eight rows of zeros
    "kin_rect": lambda: problems.kin_rect(32),                                 # rows the rectangle activation fills
    "lat_rect": lambda: problems.new_al_lat_rect(32),
    "circle6": lambda: problems.new_al_circle(32, H=30),
    "fast_box_linear": lambda: problems.fast_box_linear(32, H=30),
}


@pytest.mark.parametrize("name", list(WIDTH_SETS))
def test_new_al_lane_width_does_not_change_bits(name, tuning):
    """1, 4, 8, 16 and 32 lanes per problem run different kernel forms (the 6-state kernels read the multipliers of a
    roll-out step from LDS at 8 lanes and up, the 4-state linear-table kernels at 16; the narrower ones from the
    records): same bits."""
    prob, kw = WIDTH_SETS[name](), {}
    if isinstance(prob[0], tuple):
        prob, kw = prob
    outs = []
    for g in ("1", "4", "8", "16", "32"):
        tuning(lanes=int(g))
        outs.append(to_np(batch.solve_batch(*prob, **kw)))
    for o in outs[1:]:
        for k in ("x", "u", "cost", "status", "lam", "mu"):
            assert np.array_equal(o[k], outs[0][k], equal_nan=True), k


def test_new_al_returned_pair_is_consistent():
    """x_out is the roll-out of u_out (the C++ solver returns its live arrays, new_al_ilqr.h:1177-1220)."""
    desc, x0, goals, Q, R, cons = problems.new_al_box(16, H=30, seed=9)
    o = to_np(batch.solve_batch(desc, x0, goals, Q, R, cons))
    B, H = 16, 30
    u_pad = np.concatenate([o["u"], np.zeros((B, 1, 2))], axis=1)
    lin = to_np(batch.linearize_batch(desc, o["x"], u_pad, o["lam"], o["mu"], goals, Q, R, cons))
    assert gu.rel_err(lin["f"][:, :-1], o["x"][:, 1:]) < 1e-13
    assert np.array_equal(o["node_x"], o["x"]) and np.array_equal(o["node_u"], o["u"])
    # the reported cost is the sum of the node costs at the returned pair under the returned multipliers
    assert gu.rel_err(lin["l"].sum(axis=1), o["cost"]) < 1e-12


@pytest.mark.parametrize("quad", [False, True])
@pytest.mark.parametrize("model", ["new_bicycle", "lat_k"])
def test_new_al_node_kernel_matches_oracle(model, quad):
    rng = np.random.default_rng(5)
    N = 200
    n, m = (6, 2) if model == "new_bicycle" else (4, 1)
    c = 7
    X = rng.uniform(-1, 1, size=(N, n)) * ([30, 30, 3.5, 1.2, 4, 2][:n])
    if n == 6:
        X[:, 4] += 8.0
    U = rng.uniform(-1, 1, size=(N, m)) * 0.5
    A = rng.normal(size=(c, n)); Bm = rng.normal(size=(c, m)); Cv = rng.normal(size=c)
    LAM = -np.abs(rng.normal(size=(N, c))); LAM[rng.uniform(size=(N, c)) < 0.3] = 0.0
    MU = 10.0 ** rng.uniform(0, 4, size=N)
    Q = np.diag(rng.uniform(0.1, 10, size=n)); R = np.diag(rng.uniform(0.1, 10, size=m))
    desc = capi.make_problem(model, "new_al", 1, N, 3.0, 0.1, v=10.0, k=0.001, c=c,
                             cons_kind=capi.CONS_QUADRATIC if quad else capi.CONS_LINEAR)
    goals = np.tile(rng.normal(size=n), (2, 1))
    x = np.zeros((N, 2, n)); x[:, 0] = X
    u = np.zeros((N, 2, m)); u[:, 0] = U
    lam = np.zeros((N, 2, c)); lam[:, 0] = LAM
    if quad:
        Qc = rng.normal(size=(c, n, n)) * 0.01
        Qc[2] = 0.0                                     # a row without a quadratic term
        Qc[4] = 0.5 * (Qc[4] + Qc[4].T)
        cons = capi.pack_quadratic(Qc, A, Bm, Cv)
    else:
        cons = capi.pack_linear(A, Bm, Cv)
    ref = ol.linearize_batch(desc, x, u, lam, MU, goals, Q, R, cons)
    o = to_np(batch.linearize_batch(desc, x, u, lam, MU, goals, Q, R, cons))
    for key in ("f", "fx", "fu", "l", "lx", "lu", "lxx", "luu"):
        assert gu.rel_err(o[key][:, 0], ref[key][:, 0]) < 1e-11, key
    # the remaining per-node calls: constraints, update_lambda, max_constraints_violation, parallel_cost, f_xx
    aux = to_np(batch.node_aux_batch(desc, x, u, lam, MU, goals, Q, R, cons))
    raux = ol.node_aux_batch(desc, x, u, lam, MU, goals, Q, R, cons)
    assert gu.rel_err(aux["cval"][:, 0], ref["cval"][:, 0]) < 1e-12
    assert gu.rel_err(aux["lam_new"][:, 0], ref["lam_new"][:, 0]) < 1e-12
    for key in ("viol", "par_cost", "fxx"):
        assert gu.rel_err(aux[key][:, 0], raux[key][:, 0]) < 1e-11, key
    assert np.abs(aux["fxx"][:, 0]).max() > 0


@pytest.mark.parametrize("model", ["new_bicycle", "lat_k"])
def test_new_al_equality_table_matches_oracle(model):
    """is_equality tables in the C++ generation (constraints.h:520-523, 673-676, 791-795, 880-882): node functions at
    random points and full solves with a steering-law equality row u0 = -0.5 * delta."""
    rng = np.random.default_rng(6)
    n, m = (6, 2) if model == "new_bicycle" else (4, 1)
    c = 2 if n == 6 else 1
    A = np.zeros((c, n)); Bm = np.zeros((c, m)); Cv = np.zeros(c)
    Bm[0, 0], A[0, 3] = 1.0, 0.5
    if n == 6:
        Bm[1, 1] = 1.0                                      # jerk = 0
    cons = capi.pack_linear(A, Bm, Cv)
    H, B = 30, 24
    desc = capi.make_problem(model, "new_al", H, B, 3.0, 0.1, v=10.0, k=0.001, c=c, cons_kind=capi.CONS_EQUALITY)
    desc.prm.max_outer, desc.prm.max_inner, desc.prm.viol_stop, desc.prm.viol_lambda = 30, 100, 1e-3, 5e-3
    Q = np.diag([1e-1, 1e-1, 1.0, 1e-9, 1e-6, 1e-6][:n]) * 1e3
    R = np.eye(m) * 1e2
    goals = problems.s_curve_goals(10.0, 0.1, H, n == 6)
    x0 = problems.full_start_states(B, 13) if n == 6 else problems.lat_start_states(B, 13)
    ref = ol.solve_batch(desc, x0, goals, Q, R, cons)
    o = to_np(batch.solve_batch(desc, x0, goals, Q, R, cons))
    ok = stable_mask(desc, x0, goals, Q, R, cons, ref)
    assert gu.pinned(f"new_al/equality/{model}", ok) >= 0.5
    for i in np.nonzero(ok)[0]:
        assert worst_err(o, ref, i) < TOL, (model, i, o["status"][i], ref["status"][i])
    # equality multipliers take both signs
    assert (ref["lam"] > 0).any() and (ref["lam"] < 0).any()
    # node functions
    N = 64
    d1 = capi.make_problem(model, "new_al", 1, N, 3.0, 0.1, v=10.0, k=0.001, c=c, cons_kind=capi.CONS_EQUALITY)
    X = rng.uniform(-1, 1, size=(N, n)) * ([30, 30, 2.5, 1.0, 4, 2][:n])
    x = np.zeros((N, 2, n)); x[:, 0] = X
    u = np.zeros((N, 2, m)); u[:, 0] = rng.uniform(-0.5, 0.5, size=(N, m))
    lam = np.zeros((N, 2, c)); lam[:, 0] = rng.normal(size=(N, c))
    MU = 10.0 ** rng.uniform(0, 3, size=N)
    g2 = np.tile(rng.normal(size=n), (2, 1))
    rl = ol.linearize_batch(d1, x, u, lam, MU, g2, Q, R, cons)
    gl = to_np(batch.linearize_batch(d1, x, u, lam, MU, g2, Q, R, cons))
    for key in ("l", "lx", "lu", "lxx", "luu"):
        assert gu.rel_err(gl[key][:, 0], rl[key][:, 0]) < 1e-11, key
