"""The CPU oracle against vectors produced by the reference itself (tools/make_goldens.py).

Tolerance: 1e-6 relative (the parity bar of the solve path) is asserted; the
oracle actually reproduces the reference to ~1e-11 on every solve the
reference itself reproduces from a start state moved by 1e-13 ("stable").
"""
import numpy as np
import pytest

import golden_util as gu
import oracle_lib as ol
from ilqr_amd import capi

TOL = 1e-6
SETS = ["lat_legacy", "dyn_legacy", "kin_legacy", "kin_legacy_h50", "fast_box", "fast_linear", "fast_box_linear_h100",
        "fast_equality", "lon_legacy", "interior_legacy"]


@pytest.mark.parametrize("name", SETS)
def test_solver_matches_reference_on_stable_problems(name):
    g = gu.load(name)
    desc, x0, goals, Q, R, cons = gu.problem_from_golden(g)
    out = ol.solve_batch(desc, x0, goals, Q, R, cons)
    stable = g["stable"]
    assert stable.sum() >= 3, "golden set holds too few reproducible solves"
    for i in np.nonzero(stable)[0]:
        for key in ("x", "u", "node_x", "node_u", "x_init", "u_init"):
            assert gu.rel_err(out[key][i], g[key][i]) < TOL, (name, i, key)
        assert abs(out["cost"][i] - g["cost"][i]) <= TOL * max(1.0, abs(g["cost"][i])), (name, i)
        assert gu.rel_err(out["lam"][i], g["lam"][i]) < TOL, (name, i, "lam")
        assert abs(out["mu"][i] - g["mu"][i, 0]) <= TOL * g["mu"][i, 0]
        assert out["status"][i, 1] == g["n_outer"][i], (name, i, "outer iterations")
        if name != "fast_box":
            # fast_box: the reference's BLAS rounding moves a few accept/reject ties, results agree anyway
            assert out["status"][i, 0] == g["n_inner"][i], (name, i, "inner iterations")


@pytest.mark.parametrize("name", ["fast_box", "fast_box_linear_h100"])
def test_mid_solve_iterates_match_reference(name):
    """The reference's node trajectory after chosen inner iterations (stored for the first solves of these sets) against
    the oracle stopped at the same iteration - also for solves whose LAST line searches are decided by rounding."""
    g = gu.load(name)
    prob = gu.problem_from_golden(g)
    cost, trials = ol.solve_batch_traced(*prob, capacity=1024)
    n, mean_it = gu.check_iterates(g, name, lambda stop: ol.solve_batch(*prob, stop_after=stop), trials, cost, "oracle")
    assert n >= 7 and mean_it >= 8


def test_config1_headline_numbers():
    """Single lat problem from x0 = 0: the numbers recorded from the reference run."""
    g = gu.load("lat_legacy")
    desc, x0, goals, Q, R, cons = gu.problem_from_golden(g, idx=[0])
    assert np.all(x0 == 0)
    out = ol.solve_batch(desc, x0, goals, Q, R, cons)
    assert abs(out["cost"][0] - 47.857285341711346) < 1e-9
    assert abs(out["viol"][0] - 3.2955216762931e-4) < 1e-12
    assert np.allclose(out["u"][0, :5, 0], [0.10008025, 0.10006924, 0.10005885, 0.10004914, 0.10004014], atol=1e-8)
    assert np.allclose(out["x"][0, 50], [43.2081793, 23.1946600, 0.417729303, -0.00697609280], atol=1e-7)


@pytest.mark.parametrize("tag,model,mode", [("legacy_lat", "lat", "legacy"), ("legacy_kin", "kin", "legacy"),
                                             ("legacy_dyn", "dyn", "legacy"), ("fast_box", "dyn_mid", "fast"),
                                             ("fast_linear", "dyn_mid", "fast"), ("fast_equality", "dyn_mid", "fast"),
                                             ("legacy_lon", "lon", "legacy"), ("legacy_lat_interior", "lat_interior", "legacy")])
def test_node_functions_match_reference(tag, model, mode):
    """dynamics, f_x, f_u, cost, l_x, l_u, l_xx, l_uu, constraints, multiplier update at random points."""
    k = gu.load("nodes_kat_extra" if model in ("lon", "lat_interior") else "nodes_kat")
    pre = tag + "_"
    X, U, LAM, MU = k[pre + "X"], k[pre + "U"], k[pre + "LAM"], k[pre + "MU"]
    N = X.shape[0]
    if mode == "legacy":
        cons = capi.pack_box(k[pre + "state_min"], k[pre + "state_max"], k[pre + "control_min"], k[pre + "control_max"])
        c = None
        v = float(k[pre + "v"])
    else:
        cons = capi.pack_linear(k[pre + "A"], k[pre + "B"], k[pre + "C"])
        c = k[pre + "A"].shape[0]
        v = 0.0
    # one "problem" per sample, horizon 1: node 0 carries the sample, node 1 is ignored
    kk = float(k[pre + "control_max"][0]) if model == "lat_interior" else float(k[pre + "k"])
    desc = capi.make_problem(model, mode, 1, N, float(k[pre + "L"]), float(k[pre + "dt"]), v=v, k=kk, c=c,
                             cons_kind=capi.CONS_EQUALITY if tag == "fast_equality" else capi.CONS_LINEAR)
    n, m = desc.n, desc.m
    goals = np.tile(k[pre + "goal"], (2, 1))
    x = np.zeros((N, 2, n)); x[:, 0] = X
    u = np.zeros((N, 2, m)); u[:, 0] = U
    lam = np.zeros((N, 2, desc.c)); lam[:, 0] = LAM
    out = ol.linearize_batch(desc, x, u, lam, MU, goals, k[pre + "Q"], k[pre + "R"], cons)
    for key, okey in (("f", "f"), ("fx", "fx"), ("fu", "fu"), ("l", "l"), ("lx", "lx"), ("lu", "lu"),
                      ("lxx", "lxx"), ("luu", "luu"), ("cons", "cval"), ("lam_new", "lam_new")):
        assert gu.rel_err(out[okey][:, 0], k[pre + key]) < 1e-12, (tag, key)


@pytest.mark.parametrize("tag", ["fast_box", "fast_linear"])
def test_node_violation_term_matches_reference(tag):
    """Constraints.max_violation (cilqr/constraints.py:115-119) at the golden points."""
    k = gu.load("nodes_kat")
    pre = tag + "_"
    X, U, LAM, MU = k[pre + "X"], k[pre + "U"], k[pre + "LAM"], k[pre + "MU"]
    N = X.shape[0]
    cons = capi.pack_linear(k[pre + "A"], k[pre + "B"], k[pre + "C"])
    desc = capi.make_problem("dyn_mid", "fast", 1, N, float(k[pre + "L"]), float(k[pre + "dt"]), k=float(k[pre + "k"]),
                             c=k[pre + "A"].shape[0])
    x = np.zeros((N, 2, 6)); x[:, 0] = X
    u = np.zeros((N, 2, 2)); u[:, 0] = U
    lam = np.zeros((N, 2, desc.c)); lam[:, 0] = LAM
    out = ol.node_aux_batch(desc, x, u, lam, MU, np.tile(k[pre + "goal"], (2, 1)), k[pre + "Q"], k[pre + "R"], cons)
    assert gu.rel_err(out["viol"][:, 0], k[pre + "viol"]) < 1e-12
