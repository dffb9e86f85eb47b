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
