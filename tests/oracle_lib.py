"""ctypes binding of the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

from ilqr_amd import capi

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class Trace(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("length", C.c_int32),
                ("iter_cost", C.POINTER(C.c_double)), ("trials", C.POINTER(C.c_int32))]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so is not built: run `make -C oracle`")
        _LIB = C.CDLL(path)
        _LIB.oracle_solve.restype = C.c_int32
        _LIB.oracle_solve_batch.restype = C.c_int32
        _LIB.oracle_solve_batch_multistart.restype = C.c_int32
        _LIB.oracle_solve_batch_ex.restype = C.c_int32
        _LIB.oracle_linearize_batch.restype = C.c_int32
        _LIB.oracle_max_threads.restype = C.c_int32
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def solve_batch(desc, x0, goals, Q, R, cons, threads=0, extras=True, u_offset=None, left_obs=None, right_obs=None,
                first_row=0, mu_solver=None, first_rows=None, overlay=False, stop_after=None):
    """Run the oracle on a batch; arrays are host numpy float64, C-contiguous.  With ``extras`` the result also holds
    ``margin`` [B]: the smallest relative distance between the two sides of any accept / reject / stop comparison
    the solve made (a solve with a margin at rounding level is decided by rounding)."""
    B, H, n, m, c = desc.batch, desc.horizon, desc.n, desc.m, desc.c
    out = capi.alloc_outputs_numpy(B, H, n, m, c, extras)
    args = [np.ascontiguousarray(a, dtype=np.float64) for a in (x0, goals, Q, R, cons)]
    uo = None if u_offset is None else np.ascontiguousarray(u_offset, dtype=np.float64)
    assert uo is None or uo.size == B * H * m
    lo, ro = capi.pack_rectangles(left_obs), capi.pack_rectangles(right_obs)
    ex = capi.Extras(None if uo is None else uo.ctypes.data, lo.ctypes.data if len(lo) else None,
                     ro.ctypes.data if len(ro) else None, len(lo), len(ro), int(first_row), 0)
    if desc.mode == capi.MODES["new_al"]:
        out["mu_solver"] = np.ones(B)
        ex.mu_solver_out = out["mu_solver"].ctypes.data
        if mu_solver is not None:
            ms = np.ascontiguousarray(np.broadcast_to(np.asarray(mu_solver, dtype=np.float64), (B,)))
            ex.mu_solver_in = ms.ctypes.data
    if first_rows is not None:
        fr = np.ascontiguousarray(first_rows, dtype=np.int32)
        ex.first_rows = fr.ctypes.data
    if overlay:
        out["overlay"] = np.zeros((B, H + 1, 2 * c + 1))
        ex.overlay_out = out["overlay"].ctypes.data
    if stop_after is not None:
        sa = np.ascontiguousarray(np.broadcast_to(np.asarray(stop_after, dtype=np.int32), (B,)))
        ex.stop_after = sa.ctypes.data
    margin = np.ones(B) if extras else None
    lib().oracle_set_margin_out(_p(margin))
    rc = lib().oracle_solve_batch_ex(C.byref(desc), *[_p(a) for a in args], C.byref(ex),
                                     _p(out["x"]), _p(out["u"]), _p(out.get("node_x")), _p(out.get("node_u")),
                                     _p(out.get("x_init")), _p(out.get("u_init")), _p(out["cost"]), _p(out.get("viol")),
                                     _p(out.get("lam")), _p(out.get("mu")), _p(out.get("status")), C.c_int32(threads))
    lib().oracle_set_margin_out(None)
    if rc != 0:
        raise RuntimeError(f"oracle_solve_batch failed: {rc}")
    if extras:
        out["margin"] = margin
    return out


def solve_traced(desc, x0, goals, Q, R, cons, capacity=4096):
    """One problem with the per-iteration trace."""
    H, n, m, c = desc.horizon, desc.n, desc.m, desc.c
    out = capi.alloc_outputs_numpy(1, H, n, m, c, True)
    ic = np.zeros(capacity)
    tr = np.zeros(capacity, dtype=np.int32)
    trace = Trace(capacity, 0, ic.ctypes.data_as(C.POINTER(C.c_double)), tr.ctypes.data_as(C.POINTER(C.c_int32)))
    args = [np.ascontiguousarray(a, dtype=np.float64) for a in (x0, goals, Q, R, cons)]
    rc = lib().oracle_solve(C.byref(desc), *[_p(a) for a in args],
                            _p(out["x"]), _p(out["u"]), _p(out["node_x"]), _p(out["node_u"]),
                            _p(out["x_init"]), _p(out["u_init"]), _p(out["cost"]), _p(out["viol"]),
                            _p(out["lam"]), _p(out["mu"]), _p(out["status"]), C.byref(trace))
    if rc != 0:
        raise RuntimeError(f"oracle_solve failed: {rc}")
    out["iter_cost"] = ic[:trace.length].copy()
    out["trials"] = tr[:trace.length].copy()
    return out


def solve_batch_traced(desc, x0, goals, Q, R, cons, capacity=1024):
    """Per-iteration trace of every problem of a batch (one oracle_solve per problem): lists of iter_cost / trials."""
    B, H, n = desc.batch, desc.horizon, desc.n
    x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(B, n)
    goals = np.ascontiguousarray(goals, dtype=np.float64)
    d = type(desc).from_buffer_copy(desc)
    d.batch = 1
    d.goal_batch = 1
    costs, trials = [], []
    for i in range(B):
        g = goals.reshape(B, H + 1, n)[i] if desc.goal_batch > 1 else goals
        o = solve_traced(d, x0[i:i + 1], g, Q, R, cons, capacity)
        costs.append(o["iter_cost"]); trials.append(o["trials"])
    return costs, trials


def linearize_batch(desc, x, u, lam, mu, goals, Q, R, cons):
    B, H, n, m, c = desc.batch, desc.horizon, desc.n, desc.m, desc.c
    N = B * (H + 1)
    out = dict(f=np.zeros((B, H + 1, n)), fx=np.zeros((B, H + 1, n, n)), fu=np.zeros((B, H + 1, n, m)),
               l=np.zeros((B, H + 1)), lx=np.zeros((B, H + 1, n)), lu=np.zeros((B, H + 1, m)),
               lxx=np.zeros((B, H + 1, n, n)), luu=np.zeros((B, H + 1, m, m)),
               cval=np.zeros((B, H + 1, c)), lam_new=np.zeros((B, H + 1, c)), xnorm=np.zeros((B, H + 1, n)))
    args = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, u, lam, mu, goals, Q, R, cons)]
    assert args[0].size == N * n and args[1].size == N * m
    rc = lib().oracle_linearize_batch(C.byref(desc), *[_p(a) for a in args],
                                      *[_p(out[k]) for k in ("f", "fx", "fu", "l", "lx", "lu", "lxx", "luu",
                                                             "cval", "lam_new", "xnorm")])
    if rc != 0:
        raise RuntimeError(f"oracle_linearize_batch failed: {rc}")
    return out


def node_aux_batch(desc, x, u, lam, mu, goals, Q, R, cons):
    B, H, n = desc.batch, desc.horizon, desc.n
    new_al = desc.mode == capi.MODES["new_al"]
    out = dict(viol=np.zeros((B, H + 1)))
    if new_al:
        out.update(par_cost=np.zeros((B, H + 1)), fxx=np.zeros((B, H + 1, 3, n, n)))
    args = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, u, lam, mu, goals, Q, R, cons)]
    lib().oracle_node_aux_batch.restype = C.c_int32
    rc = lib().oracle_node_aux_batch(C.byref(desc), *[_p(a) for a in args], _p(out["viol"]), _p(out.get("par_cost")),
                                     _p(out.get("fxx")))
    if rc != 0:
        raise RuntimeError(f"oracle_node_aux_batch failed: {rc}")
    return out
