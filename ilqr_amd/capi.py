"""ctypes view of include/alilqr.h: descriptors, library loading, raw calls.

The HIP library is the only compute back end.  There is no CPU fallback: if
``libalilqr_hip.so`` is missing or no GPU is visible, solving raises.
"""
import ctypes as C
import os

import numpy as np

ABI_VERSION = 6
STATUS_WORDS = 4

MODELS = {"lat": 0, "kin": 1, "dyn": 2, "dyn_mid": 3, "lat_k": 4, "new_bicycle": 5, "lon": 6, "lat_interior": 7}
MODES = {"legacy": 0, "fast": 1, "new_al": 2}
MODEL_DIMS = {0: (4, 1), 1: (6, 2), 2: (6, 2), 3: (6, 2), 4: (4, 1), 5: (6, 2), 6: (3, 1), 7: (4, 1)}
DV_GATE = 0.20000000298023224   # the float literal 0.2F of new_al_ilqr.h:1039, as a double

FLAG_VIOL_OK = 1
FLAG_OUTER_CAP = 2
FLAG_NONFINITE = 4
FLAG_STOPPED = 8

E_ARG, E_HIP, E_NOMEM, E_NODEVICE = -1, -2, -3, -4


class Params(C.Structure):
    """struct alilqr_params"""
    _fields_ = [("max_outer", C.c_int32), ("max_inner", C.c_int32),
                ("inner_tol", C.c_double), ("viol_stop", C.c_double), ("viol_lambda", C.c_double),
                ("mu_gain", C.c_double), ("quu_reg", C.c_double), ("alpha_min", C.c_double),
                ("init_r", C.c_double), ("dv_gate", C.c_double)]


class Problem(C.Structure):
    """struct alilqr_problem"""
    _fields_ = [("abi_version", C.c_int32), ("model", C.c_int32), ("mode", C.c_int32),
                ("horizon", C.c_int32), ("n", C.c_int32), ("m", C.c_int32), ("c", C.c_int32),
                ("batch", C.c_int32), ("goal_batch", C.c_int32), ("table_nodes", C.c_int32),
                ("cons_kind", C.c_int32), ("reserved", C.c_int32),
                ("L", C.c_double), ("dt", C.c_double), ("v", C.c_double), ("k", C.c_double),
                ("prm", Params)]


class Extras(C.Structure):
    """struct alilqr_extras (pointers are device pointers for the HIP library, host pointers for the test oracle)"""
    _fields_ = [("u_offset", C.c_void_p), ("left_obs", C.c_void_p), ("right_obs", C.c_void_p),
                ("n_left", C.c_int32), ("n_right", C.c_int32), ("first_row", C.c_int32), ("reserved", C.c_int32),
                ("trace_cost", C.c_void_p), ("trace_trials", C.c_void_p), ("trace_cap", C.c_int32), ("reserved2", C.c_int32),
                ("mu_solver_in", C.c_void_p), ("mu_solver_out", C.c_void_p), ("first_rows", C.c_void_p),
                ("overlay_out", C.c_void_p), ("stop_after", C.c_void_p), ("order", C.c_void_p)]


def pack_rectangles(rects):
    """List of 2x4 vertex matrices (x row, y row; counter-clockwise) -> contiguous [n][2][4] array."""
    if rects is None or len(rects) == 0:
        return np.zeros((0, 2, 4))
    a = np.ascontiguousarray(np.stack([np.asarray(r, dtype=np.float64) for r in rects]))
    if a.shape[1:] != (2, 4):
        raise TypeError("every rectangle is a 2x4 matrix of vertex coordinates")
    return a


def default_params(mode):
    """Reference literals per solver generation (ilqr.py:151-174, fast_ilqr.py:136-159,
    new_al_ilqr.h:1177-1211)."""
    mode = MODES[mode] if isinstance(mode, str) else int(mode)
    if mode == MODES["legacy"]:
        return Params(20, 20, 1e-8, 1e-3, 1e-1, 8.0, 1e-9, 1e-8, 20.0, DV_GATE)
    if mode == MODES["fast"]:
        return Params(20, 20, 1e-8, 1e-3, 1e-1, 8.0, 0.0, 1e-8, 20.0, DV_GATE)
    if mode == MODES["new_al"]:
        return Params(50, 100, 1e-3, 1e-4, 5e-4, 100.0, 0.0, 1e-8, 20.0, DV_GATE)
    raise ValueError(f"unknown solver mode {mode}")


CONS_LINEAR, CONS_QUADRATIC, CONS_EQUALITY = 0, 1, 2


def make_problem(model, mode, horizon, batch, L, dt, v=0.0, k=0.0, c=None,
                 goal_batch=1, table_nodes=1, params=None, cons_kind=CONS_LINEAR):
    model_id = MODELS[model] if isinstance(model, str) else int(model)
    mode_id = MODES[mode] if isinstance(mode, str) else int(mode)
    n, m = MODEL_DIMS[model_id]
    if c is None:
        if mode_id != MODES["legacy"]:
            raise ValueError("constraint row count c is required for the fast / new_al generations")
        c = 2 * n if model_id == MODELS["lat_interior"] else 2 * (n + m)   # the interior node bounds states only
    prm = params if params is not None else default_params(mode_id)
    return Problem(ABI_VERSION, model_id, mode_id, int(horizon), n, m, int(c), int(batch),
                   int(goal_batch), int(table_nodes), int(cons_kind), 0, float(L), float(dt), float(v), float(k), prm)


def pack_box(state_min, state_max, control_min, control_max):
    """LEGACY constraint table of one node: state_min | state_max | control_min | control_max."""
    return np.concatenate([np.asarray(a, dtype=np.float64).ravel()
                           for a in (state_min, state_max, control_min, control_max)])


def pack_linear(A, B, Cv):
    """FAST / NEW_AL constraint table of one node: A (c x n) | B (c x m) | C (c)."""
    return np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in (A, B, Cv)])


def pack_quadratic(Qc, A, B, Cv):
    """NEW_AL quadratic table of one node: A (c x n) | B (c x m) | C (c) | Qc (c x n x n)."""
    return np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in (A, B, Cv, Qc)])


def cons_words(desc):
    if desc.mode == MODES["legacy"]:
        return 2 * desc.n + 2 * desc.m
    w = desc.c * (desc.n + desc.m + 1)
    if desc.cons_kind & CONS_QUADRATIC:
        w += desc.c * desc.n * desc.n
    return w


def alloc_outputs_numpy(B, H, n, m, c, extras=True):
    out = dict(x=np.zeros((B, H + 1, n)), u=np.zeros((B, H, m)), cost=np.zeros(B))
    if extras:
        out.update(node_x=np.zeros((B, H + 1, n)), node_u=np.zeros((B, H, m)),
                   x_init=np.zeros((B, H + 1, n)), u_init=np.zeros((B, H, m)),
                   viol=np.zeros(B), lam=np.zeros((B, H + 1, c)), mu=np.zeros(B),
                   status=np.zeros((B, STATUS_WORDS), dtype=np.int32))
    return out


# ---------------------------------------------------------------------------
# the HIP library
# ---------------------------------------------------------------------------
_LIB = None
LIB_PATH = os.environ.get("ALILQR_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libalilqr_hip.so")

EXPORTS = ("alilqr_abi_version", "alilqr_last_error", "alilqr_device_count", "alilqr_default_params",
           "alilqr_workspace_bytes", "alilqr_solve_batch", "alilqr_solve_batch_multistart", "alilqr_solve_batch_ex",
           "alilqr_linearize_batch", "alilqr_node_aux_batch", "alilqr_constraints_batch",
           "alilqr_solve_batches", "alilqr_release", "alilqr_solve_batch_noalloc", "alilqr_static_scratch_bytes", "alilqr_set_tuning",
           "alilqr_lanes_for")


class LibraryMissing(RuntimeError):
    pass


def load_library():
    """dlopen libalilqr_hip.so (built in-tree by __graft_entry__.build()).  No fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(f"{LIB_PATH} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(needs hipcc).  There is no CPU fallback for the solve path.")
    # torch ships its own HIP runtime (same SONAME); import it first so that the library binds to the
    # runtime that owns the tensors' device pointers instead of pulling a second copy into the process
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.alilqr_abi_version.restype = C.c_int32
    lib.alilqr_last_error.restype = C.c_char_p
    lib.alilqr_device_count.restype = C.c_int32
    lib.alilqr_default_params.restype = C.c_int32
    lib.alilqr_default_params.argtypes = [C.c_int32, C.POINTER(Params)]
    lib.alilqr_workspace_bytes.restype = C.c_int64
    lib.alilqr_workspace_bytes.argtypes = [C.POINTER(Problem)]
    lib.alilqr_lanes_for.restype = C.c_int32
    lib.alilqr_lanes_for.argtypes = [C.POINTER(Problem)]
    lib.alilqr_solve_batch.restype = C.c_int32
    lib.alilqr_solve_batch.argtypes = [C.POINTER(Problem)] + [C.c_void_p] * 17
    lib.alilqr_solve_batch_multistart.restype = C.c_int32
    lib.alilqr_solve_batch_multistart.argtypes = [C.POINTER(Problem)] + [C.c_void_p] * 18
    lib.alilqr_solve_batch_ex.restype = C.c_int32
    lib.alilqr_solve_batch_ex.argtypes = [C.POINTER(Problem)] + [C.c_void_p] * 5 + [C.POINTER(Extras)] + [C.c_void_p] * 12
    lib.alilqr_linearize_batch.restype = C.c_int32
    lib.alilqr_node_aux_batch.restype = C.c_int32
    lib.alilqr_node_aux_batch.argtypes = [C.POINTER(Problem)] + [C.c_void_p] * 14
    lib.alilqr_linearize_batch.argtypes = [C.POINTER(Problem)] + [C.c_void_p] * 17
    lib.alilqr_constraints_batch.restype = C.c_int32
    lib.alilqr_constraints_batch.argtypes = [C.POINTER(Problem)] + [C.c_void_p] * 13
    lib.alilqr_solve_batches.restype = C.c_int32
    lib.alilqr_solve_batches.argtypes = [C.POINTER(Problem), C.c_int32] + [C.c_void_p] * 13 + [C.POINTER(Extras), C.c_void_p]
    lib.alilqr_release.restype = C.c_int32
    lib.alilqr_solve_batch_noalloc.restype = C.c_int32
    lib.alilqr_solve_batch_noalloc.argtypes = [C.POINTER(Problem)] + [C.c_void_p] * 17
    lib.alilqr_static_scratch_bytes.restype = C.c_int64
    lib.alilqr_set_tuning.restype = C.c_int32
    lib.alilqr_set_tuning.argtypes = [C.c_int32] * 3
    if lib.alilqr_abi_version() != ABI_VERSION:
        raise RuntimeError("libalilqr_hip.so ABI version mismatch")
    _LIB = lib
    return lib


def set_tuning(lanes=0, max_waves=0, no_lds=0, dense_rows=0):
    """Process-wide tuning / diagnostic overrides of the library (0 = automatic); results do not depend on them.
    ``dense_rows``: evaluate the constraint rows of FAST / NEW_AL tables densely (no box-row short cut)."""
    check(load_library().alilqr_set_tuning(int(lanes), int(max_waves), int(bool(no_lds)) | (2 if dense_rows else 0)),
          "alilqr_set_tuning")


def release():
    """Free every cached device workspace of the library (waits for the solves in flight)."""
    check(load_library().alilqr_release(), "alilqr_release")


def last_error():
    return load_library().alilqr_last_error().decode()


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with status {rc}: {last_error()}")
