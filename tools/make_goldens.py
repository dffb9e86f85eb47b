#!/usr/bin/env python3
"""Generate golden vectors by *importing* the Python reference solver.

Runs only in the build container (needs /root/reference/cilqr on disk); the
GPU box never sees the reference, it only loads the .npz files written here.
Nothing from the reference is copied: its modules are imported, driven on
seeded synthetic problems, and inputs + outputs are stored as plain arrays.

Usage:  MPLBACKEND=Agg python tools/make_goldens.py [--only NAME] [--jobs 8]

Sets written to tests/golden/:
  nodes_kat.npz         per-call known answers of the node / constraint functions
  lat_legacy.npz        ILQR + LatBicycleKinematicNode (x0 = 0 case first, then rng(0) cases)
  kin_legacy.npz        ILQR + FullBicycleKinematicNode, soft weights, H=20 (the horizon of cilqr/test_full.py:28-59;
                        at H=50 the reference does not reproduce its own kinematic solves from a start state moved by 1e-13)
  kin_legacy_h50.npz    the same node at H=50 with R = 100 I, on which most solves are reproducible
  dyn_legacy.npz        ILQR + FullBicycleDynamicNode,   soft weights, H=100
  fast_box.npz          FastILQR + FastBicycleNode + BoxConstraint (test_fast_full.py setup)
  fast_linear.npz       FastILQR + FastBicycleNode(k=0) + LinearConstraints (box rows + half-planes)
  fast_box_linear_h100.npz  FastILQR + FastBicycleNode + the box stacked with two half-plane rows, H=100 (benchmark
                        configuration 4b), soft weights
  fast_equality.npz     FastILQR + FastBicycleNode(k=0) + LinearConstraints(is_equality=True)
  lon_legacy.npz        ILQR + LongitudinalNode (triple integrator, active control / acceleration bounds)
  interior_legacy.npz   ILQR + LatBicycleInteriorNode (tanh-squashed steering rate, state bounds only)
  nodes_kat_extra.npz   per-call known answers of those two nodes
fast_box and fast_box_linear_h100 also hold the node trajectory of their first SNAP_SOLVES solves after the inner
iterations SNAP_ITERS (snap_iters, snap_x, snap_u, snap_cost): a test can then compare an iterate in the middle of a
solve, before the first line search that rounding decides.
"""
import argparse
import contextlib
import copy
import io
import os
import sys
import time

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np

REF = "/root/reference/cilqr"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_ref():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import ilqr as ref_ilqr
    import fast_ilqr as ref_fast_ilqr
    import lat_bicycle_node, full_bicycle_kinematic_node, full_bicycle_dynamic_node
    import fast_bicycle_node, linear_constraints, box_constrains
    import lon_node, lat_bicycle_interior_node
    return dict(
        Lon=lon_node.LongitudinalNode, Interior=lat_bicycle_interior_node.LatBicycleInteriorNode,
        ILQR=ref_ilqr.ILQR, FastILQR=ref_fast_ilqr.FastILQR,
        Lat=lat_bicycle_node.LatBicycleKinematicNode,
        Kin=full_bicycle_kinematic_node.FullBicycleKinematicNode,
        Dyn=full_bicycle_dynamic_node.FullBicycleDynamicNode,
        Fast=fast_bicycle_node.FastBicycleNode,
        Linear=linear_constraints.LinearConstraints,
        Box=box_constrains.BoxConstraint,
    )


# ----------------------------------------------------------------------------
# problem definitions (same numbers as the reference's demo drivers)
# ----------------------------------------------------------------------------
def s_curve_goals(v, dt, H, full):
    """S-curve tracking goals: x=v t, y=50 sin(0.1 t), heading and steering from the curve."""
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
    return np.stack(cols, axis=1)


def lat_problem(H=50):
    return dict(model="lat", mode="legacy", H=H, L=1.0, dt=0.1, v=10.0, k=0.0,
                Q=np.diag([1e-3, 1e-1, 1e1, 1e-9]), R=np.array([[50.0]]),
                state_min=np.array([-1000, -1000, -2 * np.pi, -10.0]),
                state_max=np.array([1000, 1000, 2 * np.pi, 10.0]),
                control_min=np.array([-0.1]), control_max=np.array([0.1]),
                goals=s_curve_goals(10.0, 0.1, H, False))


def lat_x0(B, seed=0):
    rng = np.random.default_rng(seed)
    r = rng.uniform([-1.0, -0.2, -0.05], [1.0, 0.2, 0.05], size=(B, 3))
    return np.concatenate([np.zeros((B, 1)), r], axis=1)


def full_problem(model, H, mode="legacy"):
    return dict(model=model, mode=mode, H=H, L=3.0, dt=0.1, v=10.0, k=(1e-3 if model != "kin" else 0.0),
                Q=np.diag([1e-1, 1e-1, 1.0, 1e-9, 1e-6, 1e-6]), R=np.eye(2) * 10.0,
                state_min=np.array([-1000, -1000, -2 * np.pi, -10, -100, -10.0]),
                state_max=np.array([1000, 1000, 2 * np.pi, 10, 100, 10.0]),
                control_min=np.array([-0.2, -1.0]), control_max=np.array([0.2, 1.0]),
                goals=s_curve_goals(10.0, 0.1, H, True))


def full_x0(B, seed):
    rng = np.random.default_rng(seed)
    r = rng.uniform([-1.0, -0.1, -1.0], [1.0, 0.1, 1.0], size=(B, 3))
    x0 = np.zeros((B, 6))
    x0[:, 1] = r[:, 0]
    x0[:, 2] = r[:, 1]
    x0[:, 4] = 10.0 + r[:, 2]
    return x0


def lon_problem(H=50):
    """LongitudinalNode (cilqr/lon_node.py): follow a 10 m/s reference with bounded acceleration and jerk."""
    t = np.arange(H + 1) * 0.1
    goals = np.stack([10.0 * t, np.full_like(t, 10.0), np.zeros_like(t)], axis=1)
    return dict(model="lon", mode="legacy", H=H, L=0.0, dt=0.1, v=0.0, k=0.0,
                Q=np.diag([1.0, 1.0, 0.1]), R=np.array([[0.5]]),
                state_min=np.array([-1000, 0.0, -2.0]), state_max=np.array([1000, 14.0, 2.0]),
                control_min=np.array([-3.0]), control_max=np.array([3.0]), goals=goals)


def lon_x0(B, seed):
    rng = np.random.default_rng(seed)
    r = rng.uniform([-2.0, -3.0, -0.5], [2.0, 3.0, 0.5], size=(B, 3))
    r[:, 1] += 10.0
    return r


def interior_problem(H=50):
    """LatBicycleInteriorNode (cilqr/lat_bicycle_interior_node.py): regulate y, heading and steering to zero along
    the x axis; the steering rate is control_max * tanh(u), the table holds state bounds only."""
    t = np.arange(H + 1) * 0.1
    goals = np.stack([10.0 * t, np.zeros_like(t), np.zeros_like(t), np.zeros_like(t)], axis=1)
    return dict(model="lat_interior", mode="legacy", H=H, L=1.0, dt=0.1, v=10.0, k=0.0,
                Q=np.diag([0.0, 1.0, 1.0, 1e-3]), R=np.array([[1.0]]),
                state_min=np.array([-1000, -1000, -0.05, -0.05]),
                state_max=np.array([1000, 1000, 0.05, 0.05]),
                control_min=np.array([-0.3]), control_max=np.array([0.3]), goals=goals)


def interior_x0(B, seed):
    """(0, U(-1,1), U(-0.03,0.03), U(-0.02,0.02)): inside the heading / steering bounds of interior_problem; the
    larger lateral offsets make the heading bound active."""
    rng = np.random.default_rng(seed)
    r = rng.uniform([-1.0, -0.03, -0.02], [1.0, 0.03, 0.02], size=(B, 3))
    return np.concatenate([np.zeros((B, 1)), r], axis=1)


def fast_box_problem(H=50):
    p = full_problem("dyn_mid", H, mode="fast")
    p.update(L=1.0, k=0.001,
             Q=np.diag([1e-1, 1e-1, 1.0, 1e-9, 1e-6, 1e-6]) * 1e3, R=np.eye(2) * 1e2,
             state_max=np.array([30, 10, 2 * np.pi, 10, 100, 10.0]))
    return p


def box_to_linear(p):
    n, m = p["state_min"].size, p["control_min"].size
    A = np.vstack([np.eye(n), -np.eye(n), np.zeros((2 * m, n))])
    Bm = np.vstack([np.zeros((2 * n, m)), np.eye(m), -np.eye(m)])
    C = np.concatenate([-p["state_max"], p["state_min"], -p["control_max"], p["control_min"]])
    return A, Bm, C


def fast_linear_problem(H=50):
    """kinematic (k=0) fast node with control-box rows plus two half-planes (y <= 14, x - y <= 40)."""
    p = full_problem("dyn_mid", H, mode="fast")
    p.update(k=0.0)
    A = np.zeros((6, 6)); Bm = np.zeros((6, 2)); C = np.zeros(6)
    Bm[0, 0] = 1; C[0] = -0.2
    Bm[1, 0] = -1; C[1] = -0.2
    Bm[2, 1] = 1; C[2] = -1.0
    Bm[3, 1] = -1; C[3] = -1.0
    A[4, 1] = 1; C[4] = -14.0
    A[5, 0] = 1; A[5, 1] = -1; C[5] = -40.0
    p.update(A=A, B=Bm, C=C)
    return p


def fast_box_linear_problem(H=100):
    """Benchmark configuration 4b: the dynamic fast node (k = 1e-3) with the box of cilqr/test_fast_full.py:31-60 as a
    linear table, stacked with two half-plane rows (y <= 40, active near the crest of the S-curve, and y >= -5): 18
    rows.  Soft weights and L = 3 (al_ilqr_test.cc:33-35): with the shipped Q * 1e3, L = 1 the reference does not
    reproduce its own H = 100 solves from a start state moved by 1e-13."""
    p = full_problem("dyn_mid", H, mode="fast")
    A, Bm, C = box_to_linear(p)
    A = np.vstack([A, [[0, 1.0, 0, 0, 0, 0], [0, -1.0, 0, 0, 0, 0]]])
    Bm = np.vstack([Bm, np.zeros((2, 2))])
    C = np.concatenate([C, [-40.0, -5.0]])
    p.update(A=A, B=Bm, C=C)
    return p


def fast_equality_problem(H=50):
    """kinematic (k=0) fast node with an EQUALITY table (LinearConstraints(..., is_equality=True)):
    jerk u1 = 0 and steering rate u0 = -0.5 * steering angle.  Both rows can be met at every node."""
    p = full_problem("dyn_mid", H, mode="fast")
    p.update(k=0.0)
    A = np.zeros((2, 6)); Bm = np.zeros((2, 2)); C = np.zeros(2)
    Bm[0, 1] = 1.0
    Bm[1, 0] = 1.0; A[1, 3] = 0.5
    p.update(A=A, B=Bm, C=C, eq=True)
    return p


# ----------------------------------------------------------------------------
# traced solver runs
# ----------------------------------------------------------------------------
SNAP_ITERS = (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256)   # inner iterations whose iterate is kept
SNAP_SOLVES = 8                                                              # ... of the first solves of a set

def _make_traced(base):
    class Traced(base):
        def __init__(self, nodes):
            super().__init__(nodes)
            self.n_inner = 0
            self.n_cost_calls = 0
            self.in_forward = False
            self.trials = []
            self.iter_cost = []
            self.viol = []
            self.snaps = {}

        def compute_total_cost(self):
            if self.in_forward:
                self.n_cost_calls += 1
            return super().compute_total_cost()

        def backward(self):
            self.n_inner += 1
            return super().backward()

        def forward(self, k, K):
            self.in_forward = True
            self.n_cost_calls = 0
            out = super().forward(k, K)
            self.in_forward = False
            self.trials.append(self.n_cost_calls - 1)
            self.iter_cost.append(float(super().compute_total_cost()))
            if len(self.trials) in SNAP_ITERS:
                self.snaps[len(self.trials)] = (np.array([nd.state for nd in self.ilqr_nodes], dtype=float),
                                                np.array([nd.control for nd in self.ilqr_nodes[:-1]], dtype=float),
                                                self.iter_cost[-1])
            return out

        def compute_constrain_violation(self):
            v = super().compute_constrain_violation()
            self.viol.append(float(v))
            return v
    return Traced


def _build_nodes(ref, p):
    H = p["H"]
    nodes = []
    for t in range(H + 1):
        g = p["goals"][t].copy()
        if p["mode"] == "legacy":
            sb = np.array([p["state_min"], p["state_max"]])
            cb = np.array([p["control_min"], p["control_max"]])
            if p["model"] == "lat":
                nd = ref["Lat"](L=p["L"], dt=p["dt"], v=p["v"], state_bounds=sb, control_bounds=cb, goal=g, Q=p["Q"], R=p["R"])
            elif p["model"] == "lon":
                nd = ref["Lon"](dt=p["dt"], state_bounds=sb, control_bounds=cb, goal=g, Q=p["Q"], R=p["R"])
            elif p["model"] == "lat_interior":
                nd = ref["Interior"](L=p["L"], dt=p["dt"], v=p["v"], state_bounds=sb, control_bounds=cb, goal=g, Q=p["Q"], R=p["R"])
            elif p["model"] == "kin":
                nd = ref["Kin"](L=p["L"], dt=p["dt"], state_bounds=sb, control_bounds=cb, goal=g, Q=p["Q"], R=p["R"])
            else:
                nd = ref["Dyn"](L=p["L"], dt=p["dt"], k=p["k"], state_bounds=sb, control_bounds=cb, goal=g, Q=p["Q"], R=p["R"])
        else:
            if "A" in p:
                cons = ref["Linear"](p["A"].copy(), p["B"].copy(), p["C"].copy(), is_equality=bool(p.get("eq", False)))
            else:
                cons = ref["Box"](p["state_min"], p["state_max"], p["control_min"], p["control_max"])
            nd = ref["Fast"](p["L"], p["dt"], p["k"], g, p["Q"], p["R"], cons)
        nodes.append(nd)
    return nodes


def solve_one(args):
    p, x0 = args
    ref = _import_ref()
    nodes = _build_nodes(ref, p)
    Solver = _make_traced(ref["ILQR"] if p["mode"] == "legacy" else ref["FastILQR"])
    s = Solver(nodes)
    nodes[0].state = np.array(x0, dtype=float)
    t0 = time.time()
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            x_init, u_init, x, u = s.optimize()
            cost = float(Solver.__mro__[1].compute_total_cost(s))
    except np.linalg.LinAlgError:
        # the reference lets numpy raise once a diverging solve feeds non-finite
        # numbers to linalg.inv; such a problem is recorded as raised, not as a vector
        return None
    wall = time.time() - t0
    if p["mode"] == "legacy":
        lam = np.array([nd.lambda_ for nd in nodes])
        mu = np.array([nd.mu for nd in nodes])
    else:
        lam = np.array([nd.constraints_obj.lambda_ for nd in nodes])
        mu = np.array([nd.constraints_obj.mu for nd in nodes])
    return dict(
        x_init=np.array(x_init), u_init=np.array(u_init), x=np.array(x), u=np.array(u),
        node_x=np.array([nd.state for nd in nodes], dtype=float),
        node_u=np.array([nd.control for nd in nodes[:-1]], dtype=float),
        cost=cost, lam=lam, mu=mu, n_inner=s.n_inner, n_outer=len(s.viol),
        trials=np.array(s.trials), iter_cost=np.array(s.iter_cost), viol=np.array(s.viol), wall=wall, snaps=s.snaps)


def run_set(name, p, x0s, jobs, snaps=False):
    import multiprocessing as mp
    t0 = time.time()
    x0s = list(x0s)
    x0s_all = np.array(x0s, dtype=float)
    bump = np.zeros(len(x0s[0])); bump[1] = 1e-13; bump[2] = 1e-13
    with mp.Pool(jobs) as pool:
        both = pool.map(solve_one, [(p, x0) for x0 in x0s] + [(p, x0 + bump) for x0 in x0s], chunksize=1)
    res, res2 = both[:len(x0s)], both[len(x0s):]
    # a solve is 'stable' when the reference itself reproduces it to 1e-8 from a start state moved by 1e-13,
    # measured per component like the parity tests do (tests/golden_util.py: |a-b| / max(|b|, floor of the component))
    for extra in (os.path.dirname(os.path.dirname(OUT)), os.path.dirname(OUT)):
        if extra not in sys.path:
            sys.path.insert(0, extra)
    import golden_util as gu

    def _same(a, b):
        if a is None or b is None:
            return False
        d = 0.0
        for key in ("x", "u", "node_x", "node_u"):
            d = max(d, gu.comp_err(b[key], a[key]))
        d = max(d, abs(a["cost"] - b["cost"]) / max(1.0, abs(a["cost"])))
        return bool(d < 1e-8)
    stable_all = np.array([_same(a, b) for a, b in zip(res, res2)])
    raised = np.array([r is None for r in res])
    stable = stable_all[~raised]
    x0s = [x for x, r in zip(x0s, res) if r is not None]
    res = [r for r in res if r is not None]
    B = len(x0s)
    maxit = max(len(r["iter_cost"]) for r in res)
    maxo = max(len(r["viol"]) for r in res)

    def pad(key, n):
        out = np.full((B, n), np.nan)
        for i, r in enumerate(res):
            out[i, :len(r[key])] = r[key]
        return out
    out = {k: v for k, v in p.items() if k not in ("model", "mode")}
    out.update(model=np.array(p["model"]), mode=np.array(p["mode"]), x0=np.array(x0s, dtype=float))
    for key in ("x_init", "u_init", "x", "u", "node_x", "node_u", "lam", "mu"):
        out[key] = np.stack([r[key] for r in res])
    out["cost"] = np.array([r["cost"] for r in res])
    out["n_inner"] = np.array([r["n_inner"] for r in res])
    out["n_outer"] = np.array([r["n_outer"] for r in res])
    out["trials"] = pad("trials", maxit)
    out["iter_cost"] = pad("iter_cost", maxit)
    out["viol"] = pad("viol", maxo)
    out["ref_wall_s"] = np.array([r["wall"] for r in res])
    out["stable"] = stable
    if snaps:
        ns = min(SNAP_SOLVES, B)
        H, n, m = p["H"], res[0]["node_x"].shape[1], res[0]["node_u"].shape[1]
        sx = np.full((ns, len(SNAP_ITERS), H + 1, n), np.nan)
        su = np.full((ns, len(SNAP_ITERS), H, m), np.nan)
        sc = np.full((ns, len(SNAP_ITERS)), np.nan)
        for i in range(ns):
            for j, it in enumerate(SNAP_ITERS):
                if it in res[i]["snaps"]:
                    sx[i, j], su[i, j], sc[i, j] = res[i]["snaps"][it]
        out.update(snap_iters=np.array(SNAP_ITERS), snap_x=sx, snap_u=su, snap_cost=sc)
    out["x0_raised"] = x0s_all[raised]          # start states on which the reference raised LinAlgError
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: B={B} wall={time.time()-t0:.1f}s  ref per-solve {out['ref_wall_s'].mean():.2f}s "
          f"inner {out['n_inner'].min()}..{out['n_inner'].max()} outer {out['n_outer'].min()}..{out['n_outer'].max()} "
          f"stable {int(stable.sum())}/{B} cost[0]={out['cost'][0]!r}", flush=True)


# ----------------------------------------------------------------------------
# node-level known answers
# ----------------------------------------------------------------------------
def node_kats():
    ref = _import_ref()
    rng = np.random.default_rng(1234)
    out = {}
    N = 24
    for model, p in (("lat", lat_problem(4)), ("kin", full_problem("kin", 4)), ("dyn", full_problem("dyn", 4))):
        n = p["Q"].shape[0]; m = p["R"].shape[0]; c = 2 * (n + m)
        # tighten the bounds so that some constraints are active at the sample points
        p["state_min"] = -np.array([3, 3, 1.0, 0.3, 12, 2.0])[:n]
        p["state_max"] = np.array([3, 3, 1.0, 0.3, 12, 2.0])[:n]
        nodes = _build_nodes(ref, p)
        nd = nodes[1]
        X = rng.uniform(-1, 1, size=(N, n)) * np.array([5, 5, 4.0, 0.6, 3, 3.0])[:n]
        if n == 6:
            X[:, 4] += 10.0
        U = rng.uniform(-1, 1, size=(N, m)) * np.array([0.3, 2.0])[:m]
        LAM = np.maximum(0, rng.uniform(-1, 1, size=(N, c)))
        MU = rng.uniform(0.5, 20, size=N)
        rec = {k: [] for k in ("f", "fx", "fu", "l", "lx", "lu", "lxx", "luu", "cons", "lam_new", "xnorm")}
        for i in range(N):
            rec["f"].append(nd.dynamics(X[i].copy(), U[i].copy()))
            a, b = nd.dynamics_jacobian(X[i].copy(), U[i].copy())
            rec["fx"].append(a); rec["fu"].append(b)
            nd.state = X[i].copy(); nd.control = U[i].copy()
            nd.set_lambda(LAM[i].copy()); nd.set_mu(float(MU[i]))
            rec["l"].append(float(nd.cost()))
            jx, ju = nd.cost_jacobian(); rec["lx"].append(jx); rec["lu"].append(ju)
            hx, hu = nd.cost_hessian(); rec["lxx"].append(hx); rec["luu"].append(hu)
            rec["cons"].append(nd.constraints().copy())
            rec["xnorm"].append(nd.state.copy())
            nd.update_lambda(); rec["lam_new"].append(nd.lambda_.copy())
        pre = f"legacy_{model}_"
        for k_, v_ in rec.items():
            out[pre + k_] = np.array(v_)
        out[pre + "X"] = X; out[pre + "U"] = U; out[pre + "LAM"] = LAM; out[pre + "MU"] = MU
        for k_ in ("L", "dt", "v", "k", "Q", "R", "state_min", "state_max", "control_min", "control_max"):
            out[pre + k_] = np.array(p[k_])
        out[pre + "goal"] = np.array(nd.goal)

    # fast node + projection-form AL on box and general linear tables
    # (the equality table comes last so that the random draws of the earlier sets stay what they were)
    for tag, p in (("fast_box", fast_box_problem(4)), ("fast_linear", fast_linear_problem(4)),
                   ("fast_equality", fast_equality_problem(4))):
        n, m = 6, 2
        if tag == "fast_box":
            p["state_min"] = -np.array([3, 3, 1.0, 0.3, 12, 2.0]); p["state_max"] = np.array([3, 3, 1.0, 0.3, 12, 2.0])
        elif tag == "fast_equality":
            p["A"] = rng.uniform(-1, 1, size=(3, 6)); p["B"] = rng.uniform(-1, 1, size=(3, 2)); p["C"] = rng.uniform(-3, 3, size=3)
        else:
            p["A"] = rng.uniform(-1, 1, size=(6, 6)); p["B"] = rng.uniform(-1, 1, size=(6, 2)); p["C"] = rng.uniform(-3, 3, size=6)
        nodes = _build_nodes(ref, p)
        nd = nodes[1]
        c = nd.constraints_obj.constraint_dim
        X = rng.uniform(-1, 1, size=(N, n)) * np.array([5, 5, 4.0, 0.6, 3, 3.0]); X[:, 4] += 10.0
        U = rng.uniform(-1, 1, size=(N, m)) * np.array([0.3, 2.0])
        LAM = rng.uniform(-1, 1, size=(N, c)) if p.get("eq") else np.minimum(0, rng.uniform(-1, 1, size=(N, c)))
        MU = rng.uniform(0.5, 20, size=N)
        rec = {k: [] for k in ("f", "fx", "fu", "l", "lx", "lu", "lxx", "luu", "cons", "lam_new", "viol")}
        for i in range(N):
            rec["f"].append(nd.dynamics(X[i].copy(), U[i].copy()))
            a, b = nd.dynamics_jacobian(X[i].copy(), U[i].copy())
            rec["fx"].append(a); rec["fu"].append(b)
            nd.state = X[i].copy(); nd.control = U[i].copy()
            nd.constraints_obj.lambda_ = LAM[i].copy(); nd.constraints_obj.mu = float(MU[i])
            rec["l"].append(float(nd.cost()))
            jx, ju = nd.cost_jacobian(); rec["lx"].append(jx); rec["lu"].append(ju)
            hx, hu = nd.cost_hessian(); rec["lxx"].append(hx); rec["luu"].append(hu)
            rec["cons"].append(np.array(nd.constraints_obj.constrains(nd.state, nd.control)))
            rec["viol"].append(float(nd.constraints_obj.max_violation(nd.state, nd.control)))
            nd.constraints_obj.update_lambda(); rec["lam_new"].append(nd.constraints_obj.lambda_.copy())
        pre = tag + "_"
        for k_, v_ in rec.items():
            out[pre + k_] = np.array(v_)
        out[pre + "X"] = X; out[pre + "U"] = U; out[pre + "LAM"] = LAM; out[pre + "MU"] = MU
        A, Bm, C = (p["A"], p["B"], p["C"]) if "A" in p else box_to_linear(p)
        out[pre + "A"] = A; out[pre + "B"] = Bm; out[pre + "C"] = C
        for k_ in ("L", "dt", "k", "Q", "R"):
            out[pre + k_] = np.array(p[k_])
        out[pre + "goal"] = np.array(nd.goal)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "nodes_kat.npz"), **out)
    print("nodes_kat: written", len(out), "arrays", flush=True)


def node_kats_extra():
    """Known answers of LongitudinalNode and LatBicycleInteriorNode (own file and own random stream, so the
    vectors of nodes_kat.npz stay what they are)."""
    ref = _import_ref()
    rng = np.random.default_rng(4321)
    out = {}
    N = 24
    for model, p in (("lon", lon_problem(4)), ("lat_interior", interior_problem(4))):
        n = p["Q"].shape[0]; m = 1
        if model == "lon":
            p["state_min"] = np.array([-3.0, 8.0, -1.0]); p["state_max"] = np.array([3.0, 12.0, 1.0])
            scale = np.array([5, 3.0, 2.0]); shift = np.array([0, 10.0, 0])
        else:
            p["state_min"] = -np.array([3, 3, 1.0, 0.3]); p["state_max"] = np.array([3, 3, 1.0, 0.3])
            scale = np.array([5, 5, 4.0, 0.6]); shift = np.zeros(4)
        nodes = _build_nodes(ref, p)
        nd = nodes[1]
        c = nd.constraint_dim
        X = rng.uniform(-1, 1, size=(N, n)) * scale + shift
        U = rng.uniform(-1, 1, size=(N, m)) * (4.0 if model == "lon" else 1.5)
        LAM = np.maximum(0, rng.uniform(-1, 1, size=(N, c)))
        MU = rng.uniform(0.5, 20, size=N)
        rec = {k: [] for k in ("f", "fx", "fu", "l", "lx", "lu", "lxx", "luu", "cons", "lam_new", "xnorm")}
        for i in range(N):
            rec["f"].append(nd.dynamics(X[i].copy(), U[i].copy()))
            a, b = nd.dynamics_jacobian(X[i].copy(), U[i].copy())
            rec["fx"].append(np.array(a, dtype=float)); rec["fu"].append(np.array(b, dtype=float))
            nd.state = X[i].copy(); nd.control = U[i].copy()
            nd.set_lambda(LAM[i].copy()); nd.set_mu(float(MU[i]))
            rec["l"].append(float(nd.cost()))
            jx, ju = nd.cost_jacobian(); rec["lx"].append(jx); rec["lu"].append(ju)
            hx, hu = nd.cost_hessian(); rec["lxx"].append(hx); rec["luu"].append(hu)
            rec["cons"].append(nd.constraints().copy())
            rec["xnorm"].append(nd.state.copy())
            nd.update_lambda(); rec["lam_new"].append(nd.lambda_.copy())
        pre = f"legacy_{model}_"
        for k_, v_ in rec.items():
            out[pre + k_] = np.array(v_)
        out[pre + "X"] = X; out[pre + "U"] = U; out[pre + "LAM"] = LAM; out[pre + "MU"] = MU
        for k_ in ("L", "dt", "v", "k", "Q", "R", "state_min", "state_max", "control_min", "control_max"):
            out[pre + k_] = np.array(p[k_])
        out[pre + "goal"] = np.array(nd.goal)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "nodes_kat_extra.npz"), **out)
    print("nodes_kat_extra: written", len(out), "arrays", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--lat", type=int, default=64)
    ap.add_argument("--full", type=int, default=16)
    ap.add_argument("--kin", type=int, default=32)
    a = ap.parse_args()
    want = lambda n: a.only is None or a.only == n
    if want("nodes_kat"):
        node_kats()
    if want("nodes_kat_extra"):
        node_kats_extra()
    if want("lon_legacy"):
        run_set("lon_legacy", lon_problem(50), lon_x0(a.kin, 7), a.jobs)
    if want("interior_legacy"):
        run_set("interior_legacy", interior_problem(50), interior_x0(a.kin, 8), a.jobs)
    if want("lat_legacy"):
        x0 = np.concatenate([np.zeros((1, 4)), lat_x0(a.lat - 1, 0)])
        run_set("lat_legacy", lat_problem(50), x0, a.jobs)
    if want("kin_legacy"):
        run_set("kin_legacy", full_problem("kin", 20), full_x0(a.kin, 1), a.jobs)
    if want("kin_legacy_h50"):
        p = full_problem("kin", 50)
        p.update(R=np.eye(2) * 100.0)
        run_set("kin_legacy_h50", p, full_x0(24, 1), a.jobs)
    if want("dyn_legacy"):
        run_set("dyn_legacy", full_problem("dyn", 100), full_x0(a.full, 2), a.jobs)
    if want("fast_box"):
        x0 = full_x0(a.full, 4); x0[0] = [0, 0, 0, 0, 10, 0]
        run_set("fast_box", fast_box_problem(50), x0, a.jobs, snaps=True)
    if want("fast_box_linear_h100"):
        run_set("fast_box_linear_h100", fast_box_linear_problem(100), full_x0(32, 2), a.jobs, snaps=True)
    if want("fast_linear"):
        run_set("fast_linear", fast_linear_problem(50), full_x0(a.full, 5), a.jobs)
    if want("fast_equality"):
        run_set("fast_equality", fast_equality_problem(50), full_x0(a.full, 6), a.jobs)


if __name__ == "__main__":
    main()
