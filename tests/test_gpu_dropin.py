"""The reference-shaped classes on the GPU: ILQR(nodes).optimize() / FastILQR(nodes).optimize() and node methods."""
import copy

import numpy as np
import pytest

import golden_util as gu
from ilqr_amd.nodes import BoxConstraint, FastBicycleNode, FullBicycleDynamicNode, LatBicycleKinematicNode
from ilqr_amd.solvers import ILQR, FastILQR

pytestmark = pytest.mark.gpu
TOL = 1e-6


def test_ilqr_optimize_reproduces_the_reference_run_from_x0_zero():
    g = gu.load("lat_legacy")
    sb = np.array([g["state_min"], g["state_max"]]); cb = np.array([g["control_min"], g["control_max"]])
    nodes = [LatBicycleKinematicNode(L=1, dt=0.1, v=10, state_bounds=sb, control_bounds=cb, goal=goal, Q=g["Q"], R=g["R"])
             for goal in g["goals"]]
    solver = ILQR(nodes)
    nodes[0].state = np.array([0, 0, 0, 0])
    x_init, u_init, x, u = solver.optimize()
    for got, key in ((x_init, "x_init"), (u_init, "u_init"), (x, "x"), (u, "u")):
        assert gu.rel_err(got, g[key][0]) < TOL, key
    assert gu.rel_err(np.array([nd.state for nd in nodes]), g["node_x"][0]) < TOL
    assert gu.rel_err(np.array([nd.lambda_ for nd in nodes]), g["lam"][0]) < TOL
    assert nodes[0].mu == g["mu"][0, 0]
    assert abs(solver.last_status["cost"] - 47.857285341711346) < 1e-6
    assert abs(solver.compute_total_cost() - 47.857285341711346) < 1e-6     # node.cost() through the node kernel
    assert solver.last_status["outer"] == 8


def test_fast_ilqr_optimize_reproduces_the_reference_driver():
    g = gu.load("fast_box")
    box = BoxConstraint(g["state_min"], g["state_max"], g["control_min"], g["control_max"])
    nodes = [FastBicycleNode(float(g["L"]), float(g["dt"]), float(g["k"]), goal, g["Q"], g["R"], copy.deepcopy(box))
             for goal in g["goals"]]
    solver = FastILQR(nodes)
    nodes[0].state = np.array([0, 0, 0, 0, 10, 0])
    x_init, u_init, x, u = solver.optimize()
    assert gu.rel_err(x, g["x"][0]) < TOL and gu.rel_err(u, g["u"][0]) < TOL
    assert abs(solver.last_status["cost"] - 586334.1410314383) < 1e-6 * 586334
    assert gu.rel_err(np.array([nd.constraints_obj.lambda_ for nd in nodes]), g["lam"][0]) < TOL


def test_node_methods_match_reference_known_answers():
    k = gu.load("nodes_kat")
    pre = "legacy_dyn_"
    sb = np.array([k[pre + "state_min"], k[pre + "state_max"]]); cb = np.array([k[pre + "control_min"], k[pre + "control_max"]])
    nd = FullBicycleDynamicNode(float(k[pre + "L"]), float(k[pre + "dt"]), float(k[pre + "k"]), sb, cb, k[pre + "goal"],
                                k[pre + "Q"], k[pre + "R"])
    for i in range(4):
        X, U = k[pre + "X"][i], k[pre + "U"][i]
        assert gu.rel_err(nd.dynamics(X.copy(), U.copy()), k[pre + "f"][i]) < 1e-12
        A, B = nd.dynamics_jacobian(X.copy(), U.copy())
        assert gu.rel_err(A, k[pre + "fx"][i]) < 1e-12 and gu.rel_err(B, k[pre + "fu"][i]) < 1e-12
        nd.state = X.copy(); nd.control = U.copy(); nd.set_lambda(k[pre + "LAM"][i].copy()); nd.set_mu(float(k[pre + "MU"][i]))
        assert abs(nd.cost() - k[pre + "l"][i]) < 1e-12 * max(1, abs(k[pre + "l"][i]))
        jx, ju = nd.cost_jacobian(); hx, hu = nd.cost_hessian()
        assert gu.rel_err(jx, k[pre + "lx"][i]) < 1e-12 and gu.rel_err(hx, k[pre + "lxx"][i]) < 1e-12
        assert gu.rel_err(ju, k[pre + "lu"][i]) < 1e-12 and gu.rel_err(hu, k[pre + "luu"][i]) < 1e-12
        assert gu.rel_err(nd.constraints(), k[pre + "cons"][i]) < 1e-15
        nd.update_lambda()
        assert gu.rel_err(nd.lambda_, k[pre + "lam_new"][i]) < 1e-12


def test_solve_batch_extension_runs_many_start_states():
    g = gu.load("lat_legacy")
    sb = np.array([g["state_min"], g["state_max"]]); cb = np.array([g["control_min"], g["control_max"]])
    nodes = [LatBicycleKinematicNode(1, 0.1, 10, sb, cb, goal, g["Q"], g["R"]) for goal in g["goals"]]
    idx = np.nonzero(g["stable"])[0][:12]
    out = ILQR(nodes).solve_batch(g["x0"][idx])
    assert gu.rel_err(out["x"].cpu().numpy(), g["x"][idx]) < TOL
    assert gu.rel_err(out["cost"].cpu().numpy(), g["cost"][idx]) < TOL
