"""Batched solve / node-evaluation entry points over the C ABI.

torch-ROCm tensors own the device buffers and give the stream; all compute is
in libalilqr_hip.so.  There is no CPU path: without the library or a GPU these
functions raise.
"""
import ctypes as C

import numpy as np

from . import capi


def _torch():
    import torch
    return torch


def _dev_f64(a, device):
    torch = _torch()
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float64).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=device)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _require_gpu(device):
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError("ilqr_amd needs a HIP device: torch.cuda.is_available() is False "
                           "(there is no CPU fallback for the solve path)")
    return torch.device(device)


def solve_batch(desc, x0, goals, Q, R, cons, extras=True, device="cuda:0", sync=True, u_offset=None,
                left_obs=None, right_obs=None, first_row=0, trace=0, mu_solver=None, first_rows=None, overlay=False,
                stop_after=None, order=None):
    """Solve ``desc.batch`` problems on the GPU.

    Inputs may be numpy arrays or torch tensors (moved to ``device`` as fp64):
      x0 [B,n], goals [goal_batch,H+1,n], Q [T,n,n], R [T,m,m], cons [T,words].
    ``u_offset`` [B,H,m] (optional) perturbs the controls of the initial guess: B starts of one scenario.
    ``left_obs`` / ``right_obs`` (optional, NEW_AL): lists of 2x4 rectangle vertex matrices whose y-bounds are
    written into the nodes' constraint tables while the solve runs; ``first_row`` is the row index the nodes start from.
    NEW_AL state carried between optimize() calls of one solver object: ``mu_solver`` [B] is the solver-level penalty
    the previous call left (the result then holds ``mu_solver`` [B]); ``first_rows`` [H+1] gives every node its own
    start row; ``overlay`` returns ``overlay`` [B,H+1,2c+1]: the rows the rectangle activation wrote (signs, constants,
    final row index per node).
    ``trace`` > 0 also returns iter_cost / trials [B,trace]: the cost after every inner iteration and the roll-outs
    of its line search (what the reference prints while it runs); entries beyond status[:,0] are NaN / 0.
    ``stop_after`` (diagnostic; int or [B]): problem b ends right after its stop_after[b]-th inner iteration, status
    flag FLAG_STOPPED; node_x / node_u / cost are then the iterate of that iteration.
    Returns a dict of device tensors: x [B,H+1,n], u [B,H,m], cost [B], status [B,4] and, with
    ``extras``, node_x, node_u, x_init, u_init, viol, lam, mu.
    """
    torch = _torch()
    lib = capi.load_library()
    dev = _require_gpu(device)
    B, H, n, m, c = desc.batch, desc.horizon, desc.n, desc.m, desc.c
    x0_d, goals_d, Q_d, R_d, cons_d = (_dev_f64(a, dev) for a in (x0, goals, Q, R, cons))
    if x0_d.numel() != B * n:
        raise ValueError(f"x0 has {x0_d.numel()} values, expected batch*n = {B * n}")
    if goals_d.numel() != desc.goal_batch * (H + 1) * n:
        raise ValueError("goals must hold goal_batch*(H+1)*n values")
    if Q_d.numel() != desc.table_nodes * n * n or R_d.numel() != desc.table_nodes * m * m:
        raise ValueError("Q / R must hold table_nodes matrices")
    if cons_d.numel() != desc.table_nodes * capi.cons_words(desc):
        raise ValueError("constraint table has the wrong size")
    uo_d = None
    if u_offset is not None:
        uo_d = _dev_f64(u_offset, dev)
        if uo_d.numel() != B * H * m:
            raise ValueError("u_offset must hold batch*H*m values")
    lo, ro = capi.pack_rectangles(left_obs), capi.pack_rectangles(right_obs)
    lo_d = _dev_f64(lo, dev) if len(lo) else None
    ro_d = _dev_f64(ro, dev) if len(ro) else None
    ex = capi.Extras(None if uo_d is None else uo_d.data_ptr(), None if lo_d is None else lo_d.data_ptr(),
                     None if ro_d is None else ro_d.data_ptr(), len(lo), len(ro), int(first_row), 0)
    f64 = dict(dtype=torch.float64, device=dev)
    out = dict(x=torch.empty((B, H + 1, n), **f64), u=torch.empty((B, H, m), **f64), cost=torch.empty((B,), **f64),
               status=torch.zeros((B, capi.STATUS_WORDS), dtype=torch.int32, device=dev))
    keep = []
    if desc.mode == capi.MODES["new_al"]:
        out["mu_solver"] = torch.empty((B,), **f64)
        ex.mu_solver_out = out["mu_solver"].data_ptr()
        if mu_solver is not None:
            ms = _dev_f64(np.broadcast_to(np.asarray(mu_solver, dtype=np.float64), (B,)), dev)
            keep.append(ms)
            ex.mu_solver_in = ms.data_ptr()
    if first_rows is not None:
        fr_h = np.ascontiguousarray(first_rows, dtype=np.int32)
        if fr_h.size != H + 1:
            raise ValueError("first_rows must hold H+1 values")
        if fr_h.size and (fr_h.min() < 0 or fr_h.max() >= c):
            raise ValueError("every first_rows entry must be a row of the table (0 <= row < c)")
        fr = torch.as_tensor(fr_h, device=dev)
        keep.append(fr)
        ex.first_rows = fr.data_ptr()
    if overlay:
        out["overlay"] = torch.zeros((B, H + 1, 2 * c + 1), **f64)
        ex.overlay_out = out["overlay"].data_ptr()
    if order is not None:
        od = torch.as_tensor(np.ascontiguousarray(order, dtype=np.int32), device=dev) if not isinstance(order, torch.Tensor) else order.to(device=dev, dtype=torch.int32)
        if od.numel() != B:
            raise ValueError("order must be a permutation of the batch")
        keep.append(od)
        ex.order = od.data_ptr()
    if stop_after is not None:
        sa = torch.as_tensor(np.ascontiguousarray(np.broadcast_to(np.asarray(stop_after, dtype=np.int32), (B,))), device=dev)
        keep.append(sa)
        ex.stop_after = sa.data_ptr()
    if trace > 0:
        out.update(iter_cost=torch.full((B, int(trace)), float("nan"), **f64),
                   trials=torch.zeros((B, int(trace)), dtype=torch.int32, device=dev))
        ex.trace_cost, ex.trace_trials, ex.trace_cap = out["iter_cost"].data_ptr(), out["trials"].data_ptr(), int(trace)
    if extras:
        out.update(node_x=torch.empty((B, H + 1, n), **f64), node_u=torch.empty((B, H, m), **f64),
                   x_init=torch.empty((B, H + 1, n), **f64), u_init=torch.empty((B, H, m), **f64),
                   viol=torch.empty((B,), **f64), lam=torch.empty((B, H + 1, c), **f64), mu=torch.empty((B,), **f64))
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.alilqr_solve_batch_ex(C.byref(desc), _ptr(x0_d), _ptr(goals_d), _ptr(Q_d), _ptr(R_d), _ptr(cons_d),
                                    C.byref(ex), _ptr(out["x"]), _ptr(out["u"]), _ptr(out.get("node_x")), _ptr(out.get("node_u")),
                                    _ptr(out.get("x_init")), _ptr(out.get("u_init")), _ptr(out["cost"]),
                                    _ptr(out.get("viol")), _ptr(out.get("lam")), _ptr(out.get("mu")),
                                    _ptr(out.get("status")), C.c_void_p(stream))
        capi.check(rc, "alilqr_solve_batch_ex")
        if sync:
            torch.cuda.synchronize(dev)
    return out


class BatchSolver:
    """Inputs resident in HBM, outputs pre-allocated: ``run()`` only enqueues the solve."""

    def __init__(self, desc, x0, goals, Q, R, cons, extras=False, device="cuda:0", u_offset=None,
                 left_obs=None, right_obs=None, first_row=0):
        torch = _torch()
        self.lib = capi.load_library()
        self.dev = _require_gpu(device)
        self.desc = desc
        B, H, n, m, c = desc.batch, desc.horizon, desc.n, desc.m, desc.c
        self.inputs = [_dev_f64(a, self.dev) for a in (x0, goals, Q, R, cons)]
        # optional inputs of alilqr_solve_batch_ex: multi-start control offsets, rectangle obstacles
        self.ex = None
        lo, ro = capi.pack_rectangles(left_obs), capi.pack_rectangles(right_obs)
        if u_offset is not None or len(lo) or len(ro):
            self._keep = [None if u_offset is None else _dev_f64(u_offset, self.dev),
                          _dev_f64(lo, self.dev) if len(lo) else None, _dev_f64(ro, self.dev) if len(ro) else None]
            self.ex = capi.Extras(*[None if t is None else t.data_ptr() for t in self._keep], len(lo), len(ro), int(first_row), 0)
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.out = dict(x=torch.empty((B, H + 1, n), **f64), u=torch.empty((B, H, m), **f64),
                        cost=torch.empty((B,), **f64),
                        status=torch.zeros((B, capi.STATUS_WORDS), dtype=torch.int32, device=self.dev))
        if extras:
            self.out.update(node_x=torch.empty((B, H + 1, n), **f64), node_u=torch.empty((B, H, m), **f64),
                            viol=torch.empty((B,), **f64), mu=torch.empty((B,), **f64))

    def run(self, stream_ptr=None, noalloc=False):
        """Enqueue one solve.  ``noalloc`` takes the entry point that works in the library's static scratch and
        never allocates (solves that overlap on one GPU must then not use it at the same time)."""
        torch = _torch()
        o = self.out
        if noalloc and self.ex is not None:
            raise ValueError("the allocation-free entry point takes no optional inputs")
        fn = self.lib.alilqr_solve_batch_noalloc if noalloc else self.lib.alilqr_solve_batch
        outs = (_ptr(o["x"]), _ptr(o["u"]), _ptr(o.get("node_x")), _ptr(o.get("node_u")),
                None, None, _ptr(o["cost"]), _ptr(o.get("viol")), None, _ptr(o.get("mu")), _ptr(o["status"]))
        # the library launches on the device that is current on the calling thread
        with torch.cuda.device(self.dev):
            if stream_ptr is None:
                stream_ptr = torch.cuda.current_stream(self.dev).cuda_stream
            if self.ex is not None:
                rc = self.lib.alilqr_solve_batch_ex(C.byref(self.desc), *[_ptr(t) for t in self.inputs], C.byref(self.ex),
                                                    *outs, C.c_void_p(stream_ptr))
            else:
                rc = fn(C.byref(self.desc), *[_ptr(t) for t in self.inputs], *outs, C.c_void_p(stream_ptr))
        capi.check(rc, "alilqr_solve_batch_noalloc" if noalloc else "alilqr_solve_batch")
        return o


class MultiBatchSolver:
    """Several batches of one problem family, each with start states and outputs in buffers of its own, solved by ONE
    call (``alilqr_solve_batches``): the library runs them as one batch, so stragglers of one batch overlap with the
    work of the others.  ``desc.batch`` is the size of each batch; ``x0_list`` holds one [B, n] array per batch."""

    def __init__(self, desc, x0_list, goals, Q, R, cons, extras=False, device="cuda:0"):
        torch = _torch()
        self.lib = capi.load_library()
        self.dev = _require_gpu(device)
        self.desc = desc
        B, H, n, m = desc.batch, desc.horizon, desc.n, desc.m
        self.x0 = [_dev_f64(a, self.dev) for a in x0_list]
        if any(t.numel() != B * n for t in self.x0):
            raise ValueError("every start-state array must hold batch*n values")
        self.shared = [_dev_f64(a, self.dev) for a in (goals, Q, R, cons)]
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.out = []
        for _ in self.x0:
            o = dict(x=torch.empty((B, H + 1, n), **f64), u=torch.empty((B, H, m), **f64), cost=torch.empty((B,), **f64),
                     status=torch.zeros((B, capi.STATUS_WORDS), dtype=torch.int32, device=self.dev))
            if extras:
                o.update(node_x=torch.empty((B, H + 1, n), **f64), node_u=torch.empty((B, H, m), **f64),
                         viol=torch.empty((B,), **f64), mu=torch.empty((B,), **f64))
            self.out.append(o)
        n_b = len(self.x0)

        def plist(key):
            if key not in self.out[0]:
                return None
            return (C.c_void_p * n_b)(*[o[key].data_ptr() for o in self.out])
        self._lists = [(C.c_void_p * n_b)(*[t.data_ptr() for t in self.x0])] + \
            [plist(k) for k in ("x", "u", "node_x", "node_u", "cost", "viol", "mu", "status")]

    def order_by_last_solve(self, n=None):
        """Scheduling hint for the next ``run`` of the first ``n`` batches, from the status words the last run left:
        the problems that needed most inner iterations first (int32 device permutation over the n * B problems)."""
        torch = _torch()
        n = len(self.x0) if n is None else int(n)
        iters = torch.cat([o["status"][:, 0] for o in self.out[:n]])
        return torch.argsort(iters, descending=True, stable=True).to(torch.int32)

    def run(self, stream_ptr=None, n=None, order=None):
        """Enqueue the solve of all batches (or of the first ``n``); returns the list of per-batch output dicts.
        ``order``: optional int32 device permutation of the n * B problems (see ``order_by_last_solve``)."""
        torch = _torch()
        L = self._lists
        n = len(self.x0) if n is None else int(n)
        if not 0 <= n <= len(self.x0):
            raise ValueError("n must be between 0 and the number of batches")
        with torch.cuda.device(self.dev):
            if stream_ptr is None:
                stream_ptr = torch.cuda.current_stream(self.dev).cuda_stream
            if order is not None and (order.dtype != torch.int32 or order.numel() != n * self.desc.batch):
                raise ValueError("order must be an int32 permutation of the n * batch problems")
            self._order = order                                    # kept alive until the next run
            ex = None
            if order is not None:
                ex = capi.Extras()
                ex.order = order.data_ptr()
            rc = self.lib.alilqr_solve_batches(C.byref(self.desc), n, L[0], *[_ptr(t) for t in self.shared],
                                               *L[1:], None if ex is None else C.byref(ex), C.c_void_p(stream_ptr))
        capi.check(rc, "alilqr_solve_batches")
        return self.out


def release():
    """Free the library's cached device scratch (waits for the solves in flight)."""
    capi.release()


def linearize_batch(desc, x, u, lam, mu, goals, Q, R, cons, device="cuda:0"):
    """Evaluate f, f_x, f_u, l, l_x, l_u, l_xx, l_uu on B*(H+1) nodes in parallel."""
    torch = _torch()
    lib = capi.load_library()
    dev = _require_gpu(device)
    B, H, n, m = desc.batch, desc.horizon, desc.n, desc.m
    ins = [_dev_f64(a, dev) for a in (x, u, lam, mu, goals, Q, R, cons)]
    f64 = dict(dtype=torch.float64, device=dev)
    out = dict(f=torch.empty((B, H + 1, n), **f64), fx=torch.empty((B, H + 1, n, n), **f64),
               fu=torch.empty((B, H + 1, n, m), **f64), l=torch.empty((B, H + 1), **f64),
               lx=torch.empty((B, H + 1, n), **f64), lu=torch.empty((B, H + 1, m), **f64),
               lxx=torch.empty((B, H + 1, n, n), **f64), luu=torch.empty((B, H + 1, m, m), **f64))
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.alilqr_linearize_batch(C.byref(desc), *[_ptr(t) for t in ins],
                                        *[_ptr(out[k]) for k in ("f", "fx", "fu", "l", "lx", "lu", "lxx", "luu")],
                                        C.c_void_p(stream))
        capi.check(rc, "alilqr_linearize_batch")
        torch.cuda.synchronize(dev)
    return out


def node_aux_batch(desc, x, u, lam, mu, goals, Q, R, cons, device="cuda:0"):
    """Constraint values, one multiplier update, the node's violation term and (NEW_AL) the parallel cost and
    f_xx on B*(H+1) nodes in parallel."""
    torch = _torch()
    lib = capi.load_library()
    dev = _require_gpu(device)
    B, H, n, m, c = desc.batch, desc.horizon, desc.n, desc.m, desc.c
    ins = [_dev_f64(a, dev) for a in (x, u, lam, mu, goals, Q, R, cons)]
    f64 = dict(dtype=torch.float64, device=dev)
    new_al = desc.mode == capi.MODES["new_al"]
    out = dict(cval=torch.empty((B, H + 1, c), **f64), lam_new=torch.empty((B, H + 1, c), **f64),
               viol=torch.empty((B, H + 1), **f64))
    if new_al:
        out.update(par_cost=torch.empty((B, H + 1), **f64), fxx=torch.empty((B, H + 1, 3, n, n), **f64))
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.alilqr_node_aux_batch(C.byref(desc), *[_ptr(t) for t in ins],
                                       _ptr(out["cval"]), _ptr(out["lam_new"]), _ptr(out["viol"]),
                                       _ptr(out.get("par_cost")), _ptr(out.get("fxx")), C.c_void_p(stream))
        capi.check(rc, "alilqr_node_aux_batch")
        torch.cuda.synchronize(dev)
    return out


def constraints_batch(desc, x, u, cons, device="cuda:0"):
    """constraints(x,u), one column of parallel_constraints, and constraints_jacobian of a constraint table on
    B*(H+1) points in parallel (FAST / NEW_AL tables)."""
    torch = _torch()
    lib = capi.load_library()
    dev = _require_gpu(device)
    B, H, n, m, c = desc.batch, desc.horizon, desc.n, desc.m, desc.c
    zeros = (np.zeros((B, H + 1, c)), np.ones(B), np.zeros((H + 1, n)), np.eye(n), np.eye(m))
    ins = [_dev_f64(a, dev) for a in (x, u) + zeros + (cons,)]
    f64 = dict(dtype=torch.float64, device=dev)
    out = dict(cval=torch.empty((B, H + 1, c), **f64), par_cval=torch.empty((B, H + 1, c), **f64),
               jac_x=torch.empty((B, H + 1, c, n), **f64), jac_u=torch.empty((B, H + 1, c, m), **f64))
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.alilqr_constraints_batch(C.byref(desc), *[_ptr(t) for t in ins],
                                          _ptr(out["cval"]), _ptr(out["par_cval"]), _ptr(out["jac_x"]), _ptr(out["jac_u"]),
                                          C.c_void_p(stream))
        capi.check(rc, "alilqr_constraints_batch")
        torch.cuda.synchronize(dev)
    return out
