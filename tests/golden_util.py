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


NOISE_MARGIN = 1e-13


def split_pinned(o, ref, ok):
    """(strict, loose) masks of the reproducible solves `ok`.  The oracle reports every solve's decision margin: the
    smallest relative distance between the two sides of any accept / reject / stop comparison it made.  Below
    NOISE_MARGIN (exact ties, differences in the last bits: a line search on a converged trajectory) such a decision
    is taken by rounding, the reference and the oracle disagree on it themselves, and the results then differ by the
    size of the step it let through.  Those solves are held to 1e-6 where the GPU took the same decisions (equal
    iteration, outer-iteration and roll-out counts) and to a loose bound otherwise."""
    ok = np.asarray(ok, dtype=bool)
    noise = np.asarray(ref["margin"]) < NOISE_MARGIN
    same = (np.asarray(o["status"]) == np.asarray(ref["status"])).all(axis=1)
    return ok & (~noise | same), ok & noise & ~same


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
