"""Load tests/golden/*.npz into (descriptor, host arrays) pairs."""
import os

import numpy as np

from ilqr_amd import capi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def problem_from_golden(g, idx=None):
    """Returns (desc, x0, goals, Q, R, cons) for the problems `idx` of a golden set."""
    model, mode = str(g["model"]), str(g["mode"])
    x0 = np.array(g["x0"], dtype=np.float64)
    if idx is not None:
        x0 = x0[idx]
    x0 = np.ascontiguousarray(x0.reshape(-1, x0.shape[-1]))
    H = int(g["H"])
    if mode == "legacy":
        cons = capi.pack_box(g["state_min"], g["state_max"], g["control_min"], g["control_max"])
        c = None
    else:
        if "A" in g.files:
            A, B, Cv = g["A"], g["B"], g["C"]
        else:
            n, m = g["state_min"].size, g["control_min"].size
            A = np.vstack([np.eye(n), -np.eye(n), np.zeros((2 * m, n))])
            B = np.vstack([np.zeros((2 * n, m)), np.eye(m), -np.eye(m)])
            Cv = np.concatenate([-g["state_max"], g["state_min"], -g["control_max"], g["control_min"]])
        cons = capi.pack_linear(A, B, Cv)
        c = A.shape[0]
    eq = ("eq" in g.files) and bool(g["eq"])
    # the interior node's steering rate is control_max * tanh(u): control_max travels in the k slot
    k = float(g["control_max"][0]) if model == "lat_interior" else float(g["k"])
    desc = capi.make_problem(model, mode, H, x0.shape[0], float(g["L"]), float(g["dt"]),
                             v=float(g["v"]), k=k, c=c,
                             cons_kind=capi.CONS_EQUALITY if eq else capi.CONS_LINEAR)
    goals = np.ascontiguousarray(g["goals"], dtype=np.float64)
    return desc, x0, goals, np.ascontiguousarray(g["Q"], dtype=np.float64), \
        np.ascontiguousarray(g["R"], dtype=np.float64), cons


def rel_err(a, b):
    """max |a-b| / max(1, max|b|) — relative to the scale of the reference array."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))


# ---------------------------------------------------------------------------
# Parity metric of the solve tests: 1e-6 RELATIVE PER COMPONENT,
#     |a - b| <= 1e-6 * max(|b|, floor_k)
# with a floor per state / control component below which the bound turns absolute (a heading of 1e-9 rad cannot be
# held to 1e-15).  Floors are small against the range the component takes on the test problems:
#   position x, y [m] 1.0 (range 0..50) | heading [rad] 0.1 (range ~0.5) | steering angle [rad] 0.01 (range ~0.1)
#   speed [m/s] 1.0 (range ~10) | acceleration [m/s^2] 0.1 | steering rate [rad/s] 0.01 (bound 0.1..0.6) | jerk 0.1
#   longitudinal node (s, v, a | jerk): 1.0, 1.0, 0.1 | 0.01        cost: max(|cost|, 1)
# ---------------------------------------------------------------------------
STATE_FLOOR = {3: (1.0, 1.0, 0.1), 4: (1.0, 1.0, 0.1, 0.01), 6: (1.0, 1.0, 0.1, 0.01, 1.0, 0.1)}
CONTROL_FLOOR = {1: (0.01,), 2: (0.01, 0.1)}


def comp_err(a, b):
    """max over entries of |a-b| / max(|b|, floor of the entry's component); the last axis is the component axis."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    k = b.shape[-1]
    floor = np.asarray(STATE_FLOOR[k] if k in STATE_FLOOR else CONTROL_FLOOR[k])
    e = np.abs(a - b) / np.maximum(np.abs(b), floor)
    e = float(np.max(e)) if e.size else 0.0
    return e if np.isfinite(e) else np.inf


def solve_err(o, ref, i, j=None, keys=("x", "u")):
    """Worst per-component relative error of solve i of `o` against solve j of `ref` (trajectories and final cost)."""
    j = i if j is None else j
    e = max(comp_err(o[k][i], ref[k][j]) for k in keys)
    c = abs(o["cost"][i] - ref["cost"][j]) / max(1.0, abs(ref["cost"][j]))
    e = max(e, c)
    return e if np.isfinite(e) else np.inf


PINNED_LOG = []


def pinned(label, ok):
    """Record and print the share of solves a parity assertion really covers (the solves the oracle reproduces from a
    start state moved by ~1e-13); the session summary lists them all."""
    ok = np.asarray(ok, dtype=bool)
    frac = float(ok.mean()) if ok.size else 0.0
    PINNED_LOG.append((label, int(ok.sum()), int(ok.size)))
    print(f"[pinned] {label}: {int(ok.sum())}/{ok.size} solves asserted to 1e-6 ({frac:.3f})")
    return frac


def oracle_stable_mask(prob, ref, probes=3, seed=1, keys=("x", "u"), tol=1e-8, **kw):
    """Solves the oracle reproduces to `tol` (per component, as above) from start states moved by ~1e-13 in `probes`
    random directions.  prob = (desc, x0, goals, Q, R, cons); kw goes to the oracle (u_offset, obstacles ...)."""
    import oracle_lib as ol
    desc, x0 = prob[0], np.asarray(prob[1], dtype=np.float64)
    rng = np.random.default_rng(seed)
    ok = np.ones(desc.batch, dtype=bool)
    for _ in range(probes):
        ref2 = ol.solve_batch(desc, x0 + rng.uniform(-1.0, 1.0, size=x0.shape) * 1e-13, *prob[2:6], **kw)
        ok &= np.array([solve_err(ref2, ref, i, keys=keys) < tol for i in range(desc.batch)], dtype=bool)
    return ok


# ---------------------------------------------------------------------------
# Iterates in the middle of a solve.  Some golden sets hold the node trajectory after chosen inner iterations
# (snap_iters / snap_x / snap_u / snap_cost of the first solves, tools/make_goldens.py).  A solve whose later line
# searches are decided by rounding can still be compared with the reference there: at the last stored iteration up
# to which both sides made the same line-search decisions.
# ---------------------------------------------------------------------------
def common_prefix(trials_a, cost_a, trials_b, cost_b, rtol=1e-9):
    """Number of leading inner iterations on which two per-iteration traces agree: the same number of line-search
    roll-outs and costs equal to rtol (relative to max(1, |cost|)).  Entries past a trace's end are NaN / 0."""
    ta, tb = np.asarray(trials_a), np.asarray(trials_b)
    ca, cb = np.asarray(cost_a, dtype=np.float64), np.asarray(cost_b, dtype=np.float64)
    L = min(len(ta), len(tb))
    with np.errstate(invalid="ignore"):
        ok = (ta[:L] == tb[:L]) & np.isfinite(ca[:L]) & np.isfinite(cb[:L]) & \
             (np.abs(ca[:L] - cb[:L]) <= rtol * np.maximum(1.0, np.abs(cb[:L])))
    bad = np.nonzero(~ok)[0]
    return int(bad[0]) if len(bad) else L


def snapshot_at(g, i, L):
    """(iteration, x, u, cost) of the last stored iterate of golden solve i at or before inner iteration L, or None."""
    if "snap_iters" not in g.files or i >= g["snap_x"].shape[0]:
        return None
    best = None
    for j, it in enumerate(g["snap_iters"]):
        if it <= L and np.isfinite(g["snap_cost"][i, j]):
            best = (int(it), g["snap_x"][i, j], g["snap_u"][i, j], float(g["snap_cost"][i, j]))
    return best


def check_iterates(g, name, solve, trials, iter_cost, who):
    """Compare the mid-solve iterates of a golden set with `solve(stop_after) -> dict(node_x, node_u, cost, status)`.
    trials / iter_cost: the per-iteration trace of the solver under test, [B][cap].  Every solve that holds
    snapshots is stopped at the last stored iteration inside the prefix it shares with the reference's trace and must
    match the reference's iterate there to 1e-6 per component.  Returns (solves compared, their mean iteration)."""
    ns = g["snap_x"].shape[0]
    stop = np.zeros(len(g["x0"]), dtype=np.int32)
    want = {}
    for i in range(ns):
        n_ref = int(g["n_inner"][i])
        L = common_prefix(trials[i][:n_ref], iter_cost[i][:n_ref], g["trials"][i][:n_ref], g["iter_cost"][i][:n_ref])
        snap = snapshot_at(g, i, L)
        if snap is not None:
            want[i] = snap
            stop[i] = snap[0]
    assert len(want) >= max(1, ns - 1), (name, who, "too few solves share a stored iterate with the reference", len(want))
    o = solve(stop)
    for i, (it, sx, su, sc) in want.items():
        assert int(o["status"][i, 0]) == it and (int(o["status"][i, 2]) & 8), (name, who, i, it, o["status"][i])
        ex, eu = comp_err(o["node_x"][i], sx), comp_err(o["node_u"][i], su)
        ec = abs(o["cost"][i] - sc) / max(1.0, abs(sc))
        assert max(ex, eu, ec) < 1e-6, (name, who, i, "iteration", it, ex, eu, ec)
    its = [w[0] for w in want.values()]
    PINNED_LOG.append((f"iterates/{name}/{who}: solves held to the reference's mid-solve iterate", len(want), ns))
    print(f"[iterates] {name}/{who}: {len(want)}/{ns} solves match the reference's iterate to 1e-6 at iterations {its}")
    return len(want), float(np.mean(its))


def sub_problem(prob, idx):
    """The problems `idx` of a batch as a batch of their own (shared goals and tables)."""
    desc = type(prob[0]).from_buffer_copy(prob[0])
    desc.batch = len(idx)
    assert desc.goal_batch == 1
    return (desc, np.ascontiguousarray(np.asarray(prob[1])[idx])) + tuple(prob[2:6])


def mid_solve_check(label, prob, idx, gpu_trials, gpu_cost, gpu_n, or_trials, or_cost, gpu_solve, oracle_solve, **kw):
    """Solves `idx` of a batch whose END cannot be pinned (their last line searches are decided by rounding, or their
    iterates amplify a rounding error): each is compared where it can be.  L = the number of leading inner iterations
    on which the GPU's trace (roll-outs per line search, cost to 1e-9) equals the oracle's.  Inside that prefix the
    comparison is made at the latest iterate - of the schedule L, 3L/4, L/2, L/4, L/8, 1 - that the oracle itself
    reproduces to 1e-8 from three start states moved by ~1e-13 (an iterate that one rounding error already moves by more
    cannot be held to 1e-6 by anything).  GPU and oracle are stopped there (stop_after) and must agree to 1e-6 per
    component.  Returns the mean share of a solve's iterations that lie before the compared iterate."""
    idx = [int(i) for i in idx]
    if not idx:
        return 1.0
    L = np.zeros(len(idx), dtype=np.int64)
    for q, i in enumerate(idx):
        L[q] = common_prefix(gpu_trials[i][:int(gpu_n[i])], gpu_cost[i][:int(gpu_n[i])], or_trials[i], or_cost[i])
        assert L[q] >= 1, (label, i, "the first inner iteration already differs from the oracle's")
    sub = sub_problem(prob, idx)
    rng = np.random.default_rng(11)
    bumps = [rng.uniform(-1.0, 1.0, size=sub[1].shape) * 1e-13 for _ in range(3)]

    def iter_err(a, b, q):
        e = max(comp_err(a["node_x"][q], b["node_x"][q]), comp_err(a["node_u"][q], b["node_u"][q]),
                abs(a["cost"][q] - b["cost"][q]) / max(1.0, abs(b["cost"][q])))
        return e if np.isfinite(e) else np.inf

    stop = np.zeros(len(idx), dtype=np.int32)
    want = {}
    for level in (1.0, 0.75, 0.5, 0.25, 0.125, 0.0):
        todo = [q for q in range(len(idx)) if stop[q] == 0]
        if not todo:
            break
        trial = stop.copy()
        for q in todo:
            trial[q] = max(1, int(level * L[q]))
        b = oracle_solve(*sub, stop_after=trial, **kw)
        ok = np.ones(len(idx), dtype=bool)
        for bump in bumps:
            b2 = oracle_solve(sub[0], sub[1] + bump, *sub[2:6], stop_after=trial, **kw)
            ok &= np.array([iter_err(b2, b, q) < 1e-8 for q in range(len(idx))])
        for q in todo:
            if ok[q]:
                stop[q] = trial[q]
                want[q] = {k: np.array(b[k][q]) for k in ("node_x", "node_u", "cost", "status")}
    assert (stop > 0).all(), (label, [idx[q] for q in np.nonzero(stop == 0)[0]],
                              "the oracle does not even reproduce the iterate after the first inner iteration")
    a = gpu_solve(sub, stop, **kw)
    share = []
    for q, i in enumerate(idx):
        assert int(a["status"][q, 0]) == int(want[q]["status"][0]) == int(stop[q]), (label, i, a["status"][q], want[q]["status"])
        ref_q = {k: v[None] for k, v in want[q].items()}
        e = iter_err({k: a[k][q:q + 1] for k in ("node_x", "node_u", "cost")}, ref_q, 0)
        assert e < 1e-6, (label, i, "iterate after inner iteration", int(stop[q]), e)
        share.append(stop[q] / max(1, len(or_trials[i])))
    PINNED_LOG.append((f"{label}: other solves, held to 1e-6 at a mid-solve iterate", len(idx), len(idx)))
    print(f"[mid-solve] {label}: {len(idx)} solves compared at iterations {stop.tolist()} (decisions agree up to {L.tolist()}), "
          f"mean share of their iterations {np.mean(share):.3f}")
    return float(np.mean(share))
