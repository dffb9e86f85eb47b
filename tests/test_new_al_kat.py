"""Node formulas of the C++ generation (NewALILQR) against known answers derived with SymPy from the reference's own
specification scripts (tools/make_new_al_kat.py imports cilqr/jac_full_dynamic.py, rk2.py, hessian.py and
jac_lat_dynamic.py and evaluates the symbolic Jacobians / Hessians they build at seeded points).

These pin what no Python node shares with the C++ files: the d(heading)/dv entry of NewBicycleNode's Jacobian
(new_bicycle_node.h:526-528) and dynamics_hessian_fxx of both C++ nodes (new_bicycle_node.h:605-650,
new_lat_bicycle_node.h:128-163).  Tolerance 1e-12 relative to the entry's scale.

Where the C++ file's closed form is NOT the derivative the script produces, the difference is asserted as a
documented delta instead of being masked:
  * new_lat_bicycle_node.h:148,154  x(3,3), y(3,3): the second term of the bracket is written 0.5 v (t^2+1) c / D
    where the exact second derivative has 0.25 dt v (t^2+1) c / D;
  * new_lat_bicycle_node.h:160      theta(3,3) uses tan(delta), the exact derivative tan(delta + dt u / 2): the two
    agree at u = 0, which is where the entry is checked.
"""
import numpy as np
import pytest

import golden_util as gu
import oracle_lib as ol
from ilqr_amd import capi

TOL = 1e-12


def _close(got, ref, what):
    got, ref = np.asarray(got), np.asarray(ref)
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() < TOL, f"{what}: max scaled error {err.max():.3e}"


def _full_inputs(k, L, kk):
    X, U = k["full_X"], k["full_U"]
    N = X.shape[0]
    desc = capi.make_problem("new_bicycle", "new_al", 1, N, float(L), float(k["full_dt"]), k=float(kk), c=1)
    x = np.zeros((N, 2, 6)); x[:, 0] = X
    u = np.zeros((N, 2, 2)); u[:, 0] = U
    cons = capi.pack_linear(np.zeros((1, 6)), np.zeros((1, 2)), np.array([-1.0]))
    args = (desc, x, u, np.zeros((N, 2, 1)), np.ones(N), np.zeros((2, 6)), np.eye(6), np.eye(2), cons)
    return args


def _lat_inputs(k, zero_u=False):
    X, U = k["lat_X"], k["lat_U"]
    N = X.shape[0]
    desc = capi.make_problem("lat_k", "new_al", 1, N, float(k["lat_L"]), float(k["lat_dt"]), v=float(k["lat_v"]),
                             k=float(k["lat_k"]), c=1)
    x = np.zeros((N, 2, 4)); x[:, 0] = X
    u = np.zeros((N, 2, 1))
    if not zero_u:
        u[:, 0] = U
    cons = capi.pack_linear(np.zeros((1, 4)), np.zeros((1, 1)), np.array([-1.0]))
    return (desc, x, u, np.zeros((N, 2, 1)), np.ones(N), np.zeros((2, 4)), np.eye(4), np.eye(1), cons)


def _lat_delta33(k):
    """Closed-form difference (C++ text minus exact Hessian) of the x(3,3) and y(3,3) entries."""
    X = k["lat_X"]
    v, dt, L, kk = float(k["lat_v"]), float(k["lat_dt"]), float(k["lat_L"]), float(k["lat_k"])
    D = L * (kk * v * v + 1)
    td = np.tan(X[:, 3]); t2 = td * td + 1
    thm = X[:, 2] + 0.5 * dt * v * td / D
    pref = dt * dt * v * v * (t2 / D)
    dx = -pref * (0.5 * v - 0.25 * dt * v) * t2 * np.cos(thm) / D
    dy = -pref * (0.5 * v - 0.25 * dt * v) * t2 * np.sin(thm) / D
    return dx, dy


def _check_full(lin, aux, k, which):
    fx, fu = lin["fx"][:, 0], lin["fu"][:, 0]
    _close(fx, k[f"full_{which}_fx"], f"NewBicycleNode f_x vs {which}")
    _close(fu, k[f"full_{which}_fu"], f"NewBicycleNode f_u vs {which}")
    # the entry the Python FastBicycleNode writes differently is pinned here like every other
    assert np.abs(k[f"full_{which}_fx"][:, 2, 4]).min() > 1e-4


def _check_lat(lin, aux, aux0, k):
    _close(lin["fx"][:, 0], k["lat_fx"], "NewLatBicycleNode f_x")
    _close(lin["fu"][:, 0], k["lat_fu"], "NewLatBicycleNode f_u")
    got, ref = aux["fxx"][:, 0].copy(), k["lat_fxx"].copy()
    dx, dy = _lat_delta33(k)
    # documented deltas of the (3,3) entries: asserted, then removed
    _close(got[:, 0, 3, 3] - ref[:, 0, 3, 3], dx, "x(3,3) delta")
    _close(got[:, 1, 3, 3] - ref[:, 1, 3, 3], dy, "y(3,3) delta")
    assert np.abs(dx).max() > 1e-3 and np.abs(dy).max() > 1e-3
    for r in range(3):
        got[:, r, 3, 3] = ref[:, r, 3, 3] = 0.0
    _close(got, ref, "NewLatBicycleNode f_xx (all entries but (3,3))")
    # theta(3,3) equals the exact derivative where the control is zero
    _close(aux0["fxx"][:, 0, 2, 3, 3], k["lat_fxx_u0"][:, 2, 3, 3], "theta(3,3) at u = 0")
    assert np.abs(k["lat_fxx_u0"][:, 2, 3, 3]).max() > 1e-2


# ---------------------------------------------------------------------------
# CPU: the oracle restatement
# ---------------------------------------------------------------------------
def test_oracle_new_bicycle_jacobian_matches_the_rk2_derivation():
    k = gu.load("new_al_kat")
    for which, L, kk in (("jac", k["full_jac_L"], k["full_jac_k"]), ("rk2", k["full_rk2_L"], k["full_rk2_k"])):
        args = _full_inputs(k, L, kk)
        _check_full(ol.linearize_batch(*args), None, k, which)


def test_oracle_new_bicycle_fxx_matches_the_euler_hessian():
    k = gu.load("new_al_kat")
    args = _full_inputs(k, k["full_fxx_L"], k["full_fxx_k"])
    _close(ol.node_aux_batch(*args)["fxx"][:, 0], k["full_fxx"], "NewBicycleNode f_xx")
    assert np.count_nonzero(k["full_fxx"][0]) == 10     # x and y: (2,2) (2,4) (4,2); heading: (3,3) (3,4) (4,3) (4,4)


def test_oracle_new_lat_node_matches_the_rk2_derivation():
    k = gu.load("new_al_kat")
    a, a0 = _lat_inputs(k), _lat_inputs(k, zero_u=True)
    _check_lat(ol.linearize_batch(*a), ol.node_aux_batch(*a), ol.node_aux_batch(*a0), k)


# ---------------------------------------------------------------------------
# GPU: the HIP node kernels through the C ABI
# ---------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_new_bicycle_matches_the_sympy_known_answers():
    from ilqr_amd import batch
    k = gu.load("new_al_kat")
    for which, L, kk in (("jac", k["full_jac_L"], k["full_jac_k"]), ("rk2", k["full_rk2_L"], k["full_rk2_k"])):
        args = _full_inputs(k, L, kk)
        lin = {n: t.cpu().numpy() for n, t in batch.linearize_batch(*args).items()}
        _check_full(lin, None, k, which)
    args = _full_inputs(k, k["full_fxx_L"], k["full_fxx_k"])
    aux = {n: t.cpu().numpy() for n, t in batch.node_aux_batch(*args).items()}
    _close(aux["fxx"][:, 0], k["full_fxx"], "NewBicycleNode f_xx (GPU)")


@pytest.mark.gpu
def test_gpu_new_lat_node_matches_the_sympy_known_answers():
    from ilqr_amd import batch
    k = gu.load("new_al_kat")
    a, a0 = _lat_inputs(k), _lat_inputs(k, zero_u=True)
    lin = {n: t.cpu().numpy() for n, t in batch.linearize_batch(*a).items()}
    aux = {n: t.cpu().numpy() for n, t in batch.node_aux_batch(*a).items()}
    aux0 = {n: t.cpu().numpy() for n, t in batch.node_aux_batch(*a0).items()}
    _check_lat(lin, aux, aux0, k)
