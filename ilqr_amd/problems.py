"""Synthetic problem sets of the solve path (the reference's demo configurations).

Numbers follow the reference's drivers: S-curve goals cilqr/test.py:8-23, lateral
weights and bounds cilqr/test.py:26-53, full-model weights al_ilqr_test.cc:33-35
(the shipped cilqr/test_full.py:36-37 weights make the reference itself diverge),
fast-solver setup cilqr/test_fast_full.py:31-60.
"""
import numpy as np

from . import capi


def s_curve_goals(v, dt, H, full):
    """x = v t, y = 50 sin(0.1 t), heading and steering angle of that curve (cilqr/test.py:8-23)."""
    t = np.arange(H + 1) * dt
    x = v * t
    y = 50 * np.sin(0.1 * t)
    dy = 50 * 0.1 * np.cos(0.1 * t)
    theta = np.arctan2(dy, v)
    ddy = -50 * 0.1 * 0.1 * np.sin(0.1 * t)
    curv = (v * ddy - dy * 0) / (v ** 2 + dy ** 2) ** 1.5
    delta = np.arctan(curv * 1.0)
    cols = [x, y, theta, delta]
    if full:
        cols += [np.full_like(t, v), np.zeros_like(t)]
    return np.ascontiguousarray(np.stack(cols, axis=1))


def lat_start_states(B, seed=0):
    """x0 = (0, U(-1,1), U(-0.2,0.2), U(-0.05,0.05)), drawn per problem in that order."""
    rng = np.random.default_rng(seed)
    r = rng.uniform([-1.0, -0.2, -0.05], [1.0, 0.2, 0.05], size=(B, 3))
    return np.ascontiguousarray(np.concatenate([np.zeros((B, 1)), r], axis=1))


def full_start_states(B, seed):
    """x0 = (0, U(-1,1), U(-0.1,0.1), 0, 10 + U(-1,1), 0)."""
    rng = np.random.default_rng(seed)
    r = rng.uniform([-1.0, -0.1, -1.0], [1.0, 0.1, 1.0], size=(B, 3))
    x0 = np.zeros((B, 6))
    x0[:, 1] = r[:, 0]
    x0[:, 2] = r[:, 1]
    x0[:, 4] = 10.0 + r[:, 2]
    return x0


def lat_legacy(B, H=50, seed=0, x0=None):
    """ILQR + LatBicycleKinematicNode tracking the S-curve (BASELINE configs 1 and 2)."""
    desc = capi.make_problem("lat", "legacy", H, B, L=1.0, dt=0.1, v=10.0)
    Q = np.diag([1e-3, 1e-1, 1e1, 1e-9])
    R = np.array([[50.0]])
    cons = capi.pack_box([-1000, -1000, -2 * np.pi, -10.0], [1000, 1000, 2 * np.pi, 10.0], [-0.1], [0.1])
    if x0 is None:
        x0 = lat_start_states(B, seed)
    return desc, np.ascontiguousarray(x0, dtype=np.float64), s_curve_goals(10.0, 0.1, H, False), Q, R, cons


def full_legacy(model, B, H, seed, x0=None, r_weight=10.0):
    """ILQR + FullBicycle{Kinematic,Dynamic}Node with the converging (soft) weights.  The reference reproduces its own
    solves (from a start state moved by 1e-13) for the dynamic node at H = 100 and for the kinematic node at H = 20
    (the horizon of cilqr/test_full.py:28-59) or, at H = 50, with r_weight = 100."""
    k = 0.0 if model == "kin" else 1e-3
    desc = capi.make_problem(model, "legacy", H, B, L=3.0, dt=0.1, v=10.0, k=k)
    Q = np.diag([1e-1, 1e-1, 1.0, 1e-9, 1e-6, 1e-6])
    R = np.eye(2) * float(r_weight)
    cons = capi.pack_box([-1000, -1000, -2 * np.pi, -10, -100, -10.0], [1000, 1000, 2 * np.pi, 10, 100, 10.0],
                         [-0.2, -1.0], [0.2, 1.0])
    if x0 is None:
        x0 = full_start_states(B, seed)
    return desc, np.ascontiguousarray(x0, dtype=np.float64), s_curve_goals(10.0, 0.1, H, True), Q, R, cons


def box_table(state_min, state_max, control_min, control_max):
    """BoxConstraint as a linear table (cilqr/box_constrains.py:5-32): rows [I;-I;0;0] x + [0;0;I;-I] u + C."""
    smin, smax, cmin, cmax = (np.asarray(a, dtype=np.float64) for a in (state_min, state_max, control_min, control_max))
    n, m = smin.size, cmin.size
    A = np.vstack([np.eye(n), -np.eye(n), np.zeros((2 * m, n))])
    B = np.vstack([np.zeros((2 * n, m)), np.eye(m), -np.eye(m)])
    C = np.concatenate([-smax, smin, -cmax, cmin])
    return A, B, C


def fast_box(B, H=50, seed=4, x0=None):
    """FastILQR + FastBicycleNode + BoxConstraint with active x/y bounds (cilqr/test_fast_full.py:31-60)."""
    A, Bm, C = box_table([-1000, -1000, -2 * np.pi, -10, -100, -10.0], [30, 10, 2 * np.pi, 10, 100, 10.0],
                         [-0.2, -1.0], [0.2, 1.0])
    desc = capi.make_problem("dyn_mid", "fast", H, B, L=1.0, dt=0.1, v=10.0, k=0.001, c=A.shape[0])
    Q = np.diag([1e-1, 1e-1, 1.0, 1e-9, 1e-6, 1e-6]) * 1e3
    R = np.eye(2) * 1e2
    if x0 is None:
        x0 = full_start_states(B, seed)
    return desc, np.ascontiguousarray(x0, dtype=np.float64), s_curve_goals(10.0, 0.1, H, True), Q, R, \
        capi.pack_linear(A, Bm, C)


def new_al_box(B, H=50, seed=6, x0=None, y_max=1000.0):
    """NewALILQR<6,2> + NewBicycleNode + BoxConstraints (al_ilqr_cpp/test_new_al_ilqr.cc:117-192).

    The box becomes the table the reference's BoxConstraints builds (rows [I;-I;0;0] x + [0;0;I;-I] u + C,
    al_ilqr_cpp/constraints/box_constraints.h); y_max < 1000 makes a state bound active."""
    A, Bm, C = box_table([-1000, -1000, -2 * np.pi, -10, -100, -10.0], [1000, y_max, 2 * np.pi, 10, 100, 10.0],
                         [-0.2, -1.0], [0.2, 1.0])
    desc = capi.make_problem("new_bicycle", "new_al", H, B, L=3.0, dt=0.1, v=10.0, k=0.001, c=A.shape[0])
    desc.prm.max_outer, desc.prm.max_inner = 50, 100
    desc.prm.viol_stop, desc.prm.viol_lambda = 1e-3, 5e-3
    Q = np.diag([1e-1, 1e-1, 1.0, 1e-9, 1e-6, 1e-6]) * 1e3
    R = np.eye(2) * 1e2
    if x0 is None:
        x0 = full_start_states(B, seed)
    return desc, np.ascontiguousarray(x0, dtype=np.float64), s_curve_goals(10.0, 0.1, H, True), Q, R, \
        capi.pack_linear(A, Bm, C)


def new_al_lat(B, H=50, seed=7, x0=None):
    """NewALILQR<4,1> + NewLatBicycleNode + a 10-row linear table of which two rows bound the
    steering rate (al_ilqr_cpp/test_lat_al_ilqr.cc:45-95)."""
    A = np.zeros((10, 4))
    Bm = np.zeros((10, 1))
    C = np.zeros(10)
    Bm[0, 0], Bm[1, 0] = 1.0, -1.0
    C[0] = C[1] = -0.2
    desc = capi.make_problem("lat_k", "new_al", H, B, L=3.0, dt=0.1, v=10.0, k=0.001, c=10)
    desc.prm.max_outer, desc.prm.max_inner = 50, 100
    desc.prm.viol_stop, desc.prm.viol_lambda = 1e-3, 5e-3
    Q = np.diag([1e-1, 1e-1, 1.0, 1e-9]) * 1e3
    R = np.eye(1) * 1e2
    if x0 is None:
        x0 = lat_start_states(B, seed)
    return desc, np.ascontiguousarray(x0, dtype=np.float64), s_curve_goals(10.0, 0.1, H, False), Q, R, \
        capi.pack_linear(A, Bm, C)


def multi_start(scenarios, starts, H=50, seed=3, sigma=0.02):
    """`starts` starts for each of `scenarios` lateral problems: every start of a scenario has the scenario's
    start state and the controls of its initial guess moved by sigma * N(0,1); start 0 of each scenario is the
    unperturbed one.  Problem index = scenario * starts + start.  Solved with the C++ generation's line search."""
    rng = np.random.default_rng(seed)
    B = scenarios * starts
    desc, _, goals, Q, R, cons = new_al_lat(B, H=H)
    x0 = np.repeat(lat_start_states(scenarios, seed), starts, axis=0)
    u_offset = sigma * rng.standard_normal((scenarios, starts, H, 1))
    u_offset[:, 0] = 0.0
    return desc, np.ascontiguousarray(x0), goals, Q, R, cons, np.ascontiguousarray(u_offset.reshape(B, H, 1))


def kin_linear(B, H=50, seed=1, y_max=23.5):
    """Kinematic 6/2 model (k = 0) with a static linear table: the four control-bound rows, a half-plane
    y <= y_max, a half-plane y >= -y_max and two spare all-zero rows (the row pool the reference keeps for
    rows it activates later, linear_constraints.h:360-373).  C++-generation solver, weights of
    al_ilqr_cpp/test_new_al_ilqr.cc:127-137, start states (0, U(-1,1), U(-0.1,0.1), 0, 10+U(-1,1), 0)."""
    c = 8
    A = np.zeros((c, 6)); Bm = np.zeros((c, 2)); C = np.zeros(c)
    Bm[0, 0], C[0] = 1.0, -0.2
    Bm[1, 0], C[1] = -1.0, -0.2
    Bm[2, 1], C[2] = 1.0, -1.0
    Bm[3, 1], C[3] = -1.0, -1.0
    A[4, 1], C[4] = 1.0, -y_max
    A[5, 1], C[5] = -1.0, -y_max
    desc = capi.make_problem("new_bicycle", "new_al", H, B, L=3.0, dt=0.1, v=10.0, k=0.0, c=c)
    desc.prm.max_outer, desc.prm.max_inner = 50, 100
    desc.prm.viol_stop, desc.prm.viol_lambda = 1e-3, 5e-3
    Q = np.diag([1e-1, 1e-1, 1.0, 1e-9, 1e-6, 1e-6]) * 1e3
    R = np.eye(2) * 1e2
    return desc, full_start_states(B, seed), s_curve_goals(10.0, 0.1, H, True), Q, R, capi.pack_linear(A, Bm, C)


def fast_box_linear(B, H=100, seed=2, stiff=False):
    """Benchmark configuration 4b: FastILQR + FastBicycleNode with the box of cilqr/test_fast_full.py:31-60 stacked
    with two extra half-plane rows (y <= 40, active near the crest of the S-curve, and y >= -5): a linear table of 18
    rows.  Weights and wheel base are the converging ones of al_ilqr_test.cc:33-35 (Q un-scaled, R = 10 I, L = 3), on
    which the reference reproduces its own H = 100 solves (tests/golden/fast_box_linear_h100.npz, 30 of 32).
    ``stiff``: the shipped weights of test_fast_full.py (Q * 1e3, R = 100 I, L = 1), on which neither the reference nor
    anything else reproduces a solve from a start state moved by 1e-13; kept as a named property-test workload."""
    A, Bm, C = box_table([-1000, -1000, -2 * np.pi, -10, -100, -10.0], [1000, 1000, 2 * np.pi, 10, 100, 10.0],
                         [-0.2, -1.0], [0.2, 1.0])
    A = np.vstack([A, [[0, 1.0, 0, 0, 0, 0], [0, -1.0, 0, 0, 0, 0]]])
    Bm = np.vstack([Bm, np.zeros((2, 2))])
    C = np.concatenate([C, [-40.0, -5.0]])
    desc = capi.make_problem("dyn_mid", "fast", H, B, L=1.0 if stiff else 3.0, dt=0.1, v=10.0, k=0.001, c=A.shape[0])
    Q = np.diag([1e-1, 1e-1, 1.0, 1e-9, 1e-6, 1e-6]) * (1e3 if stiff else 1.0)
    R = np.eye(2) * (1e2 if stiff else 10.0)
    return desc, full_start_states(B, seed), s_curve_goals(10.0, 0.1, H, True), Q, R, capi.pack_linear(A, Bm, C)


def circle_rows(cx, cy, r, n):
    """Circle obstacle (x-cx)^2 + (y-cy)^2 >= r^2 as one quadratic row x'Qx + Ax + C <= 0
    (al_ilqr_cpp/test_new_al_ilqr.cc:83-104)."""
    Qc = np.zeros((n, n)); Qc[0, 0] = Qc[1, 1] = -1.0
    A = np.zeros(n); A[0], A[1] = 2 * cx, 2 * cy
    return Qc, A, r * r - cx * cx - cy * cy


def new_al_circle(B, H=50, seed=8, circle=(20.0, 12.0, 4.0), x0=None):
    """NewALILQR<6,2> + NewBicycleNode + QuadraticConstraints<6,2,5>: one circle obstacle and the four
    control-bound rows (al_ilqr_cpp/test_new_al_ilqr.cc:208-266; optimize(30, 100, 1e-3))."""
    c, n, m = 5, 6, 2
    Qc = np.zeros((c, n, n)); A = np.zeros((c, n)); Bm = np.zeros((c, m)); C = np.zeros(c)
    Qc[0], A[0], C[0] = circle_rows(*circle, n)
    Bm[1:] = [[1, 0], [0, 1], [-1, 0], [0, -1]]
    C[1:] = [-0.2, -1.0, -0.2, -1.0]
    desc = capi.make_problem("new_bicycle", "new_al", H, B, L=3.0, dt=0.1, v=10.0, k=0.001, c=c,
                             cons_kind=capi.CONS_QUADRATIC)
    desc.prm.max_outer, desc.prm.max_inner = 30, 100
    desc.prm.viol_stop, desc.prm.viol_lambda = 1e-3, 5e-3
    Q = np.diag([1e-1, 1e-1, 1.0, 1e-9, 1e-6, 1e-6]) * 1e3
    R = np.eye(2) * 1e2
    if x0 is None:
        x0 = full_start_states(B, seed)
    return desc, np.ascontiguousarray(x0, dtype=np.float64), s_curve_goals(10.0, 0.1, H, True), Q, R, \
        capi.pack_quadratic(Qc, A, Bm, C)


def new_al_lat_circle(B, H=30, seed=10, circle=(30.0, 11.0, 6.0), x0=None):
    """NewALILQR<4,1> + NewLatBicycleNode + QuadraticConstraints<4,1,3>: circle + steering-rate bounds of 0.4
    (cilqr/test_lat_bicycle_pybind.py:60-104; optimize(50, 100, 1e-4))."""
    c, n, m = 3, 4, 1
    Qc = np.zeros((c, n, n)); A = np.zeros((c, n)); Bm = np.zeros((c, m)); C = np.zeros(c)
    Qc[0], A[0], C[0] = circle_rows(*circle, n)
    Bm[1, 0], Bm[2, 0] = 1.0, -1.0
    C[1] = C[2] = -0.4
    desc = capi.make_problem("lat_k", "new_al", H, B, L=3.0, dt=0.1, v=10.0, k=0.001, c=c, cons_kind=capi.CONS_QUADRATIC)
    desc.prm.max_outer, desc.prm.max_inner = 50, 100
    desc.prm.viol_stop, desc.prm.viol_lambda = 1e-4, 5e-4
    Q = np.diag([1e-1, 1e-1, 1.0, 1e-9]) * 1e6
    R = np.eye(1) * 1e2
    if x0 is None:
        x0 = lat_start_states(B, seed)
        x0[:, 2] += 1.0      # the script starts with a heading of 1 rad
    return desc, np.ascontiguousarray(x0, dtype=np.float64), s_curve_goals(10.0, 0.1, H, False), Q, R, \
        capi.pack_quadratic(Qc, A, Bm, C)


# rectangles of cilqr/test_rectangle_obs_pybind.py:74-80 (2 x 4 vertex matrices, counter-clockwise)
DEMO_LEFT_OBS = [np.array([[32.0, 32, 28, 28], [13, 15, 15, 13]])]
DEMO_RIGHT_OBS = [np.array([[18.0, 18, 14, 14], [3, 7, 7, 3]])]


def new_al_lat_rect(B, H=50, seed=12, x0=None, u_bound=0.6):
    """NewALILQR<4,1>(nodes, x0, left_obs, right_obs) + NewLatBicycleNode + LinearConstraints<4,1,6> whose rows 2..5
    are the pool the rectangle activation fills (cilqr/test_rectangle_obs_pybind.py:45-95: first row index 1,
    steering-rate bound 0.6, optimize(50, 100, 1e-4)).  Returns the problem and the keyword arguments
    (left_obs, right_obs, first_row) the solve takes."""
    A = np.zeros((6, 4)); Bm = np.zeros((6, 1)); C = np.zeros(6)
    Bm[0, 0], Bm[1, 0] = 1.0, -1.0
    C[0] = C[1] = -u_bound
    desc = capi.make_problem("lat_k", "new_al", H, B, L=3.0, dt=0.1, v=10.0, k=0.001, c=6)
    desc.prm.max_outer, desc.prm.max_inner = 50, 100
    desc.prm.viol_stop, desc.prm.viol_lambda = 1e-4, 5e-4
    Q = np.diag([1e-1, 1e-1, 1.0, 1e-9]) * 1e3
    R = np.eye(1) * 1e2
    if x0 is None:
        x0 = lat_start_states(B, seed)
    prob = (desc, np.ascontiguousarray(x0, dtype=np.float64), s_curve_goals(10.0, 0.1, H, False), Q, R,
            capi.pack_linear(A, Bm, C))
    return prob, dict(left_obs=DEMO_LEFT_OBS, right_obs=DEMO_RIGHT_OBS, first_row=1)


def kin_rect(B, H=50, seed=1):
    """Benchmark configuration 3: kinematic 6/2 model (k = 0), linear table of 8 rows (four control-bound rows +
    a pool of four rows), the two demo rectangles activating y-bounds node by node, C++-generation solver,
    weights of al_ilqr_cpp/test_new_al_ilqr.cc:127-137, start states (0, U(-1,1), U(-0.1,0.1), 0, 10+U(-1,1), 0)."""
    c = 8
    A = np.zeros((c, 6)); Bm = np.zeros((c, 2)); C = np.zeros(c)
    Bm[0, 0], C[0] = 1.0, -0.2
    Bm[1, 0], C[1] = -1.0, -0.2
    Bm[2, 1], C[2] = 1.0, -1.0
    Bm[3, 1], C[3] = -1.0, -1.0
    desc = capi.make_problem("new_bicycle", "new_al", H, B, L=3.0, dt=0.1, v=10.0, k=0.0, c=c)
    desc.prm.max_outer, desc.prm.max_inner = 50, 100
    desc.prm.viol_stop, desc.prm.viol_lambda = 1e-3, 5e-3
    Q = np.diag([1e-1, 1e-1, 1.0, 1e-9, 1e-6, 1e-6]) * 1e3
    R = np.eye(2) * 1e2
    prob = (desc, full_start_states(B, seed), s_curve_goals(10.0, 0.1, H, True), Q, R, capi.pack_linear(A, Bm, C))
    return prob, dict(left_obs=DEMO_LEFT_OBS, right_obs=DEMO_RIGHT_OBS, first_row=3)
