#!/usr/bin/env python3
"""Known answers for the node formulas only the C++ generation (NewALILQR) has, derived with SymPy from the
reference's own specification scripts.

The C++ reference cannot be built in this image and holds no output vectors; its closed-form Jacobians and
Hessians were, however, produced by the SymPy scripts that ship next to it:

  cilqr/jac_full_dynamic.py   RK2-midpoint step of the 6-state model, `Jx`, `Ju`       (:45-49)
  cilqr/rk2.py                the same Jacobian as I + dt J_mid (I + dt J / 2)            (:49-50)
  cilqr/hessian.py            Euler step of the 6-state model, `hessian_tensors`          (:25, 37-42)
  cilqr/jac_lat_dynamic.py    RK2 step of the 4-state lateral model, `Jx`, `Ju`,
                              `hessian_tensor`                                            (:41-54)

This script *imports* those modules (nothing is copied; their prints are swallowed), takes the symbolic objects
they leave at module level, evaluates them with 40-digit arithmetic at seeded random points and writes plain arrays
to tests/golden/new_al_kat.npz.  jac_full_dynamic.py and jac_lat_dynamic.py hard-code L = 3.0 (and k = 0.0003) in
their first RK2 stage, so their points use those values; rk2.py and hessian.py are fully symbolic and are evaluated
at other (L, k) as well.

Runs only in the build container (needs /root/reference/cilqr and sympy).
Usage:  python tools/make_new_al_kat.py
"""
import contextlib
import importlib
import io
import os
import sys

import numpy as np

REF = "/root/reference/cilqr"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "new_al_kat.npz")
NPTS = 24
DIGITS = 40


def _import_quiet(name):
    if REF not in sys.path:
        sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        return importlib.import_module(name)


def _sym(mod, name):
    import sympy as sp
    return sp.Symbol(name)


def _evalf(expr, subs):
    return float(expr.evalf(DIGITS, subs=subs))


def _matrix(M, subs):
    return np.array([[_evalf(M[i, j], subs) for j in range(M.shape[1])] for i in range(M.shape[0])])


def main():
    import sympy as sp
    rng = np.random.default_rng(2024)
    S = {n: sp.Symbol(n) for n in ("x", "y", "theta", "delta", "v", "a", "u", "u1", "u2", "L", "dt", "k")}

    out = {}
    # ---- points of the 6-state model: angles inside (-pi, pi) so that no wrap acts; speeds away from zero
    X6 = np.column_stack([rng.uniform(-50, 50, NPTS), rng.uniform(-50, 50, NPTS), rng.uniform(-3.0, 3.0, NPTS),
                          rng.uniform(-0.6, 0.6, NPTS), rng.uniform(2.0, 15.0, NPTS), rng.uniform(-2.0, 2.0, NPTS)])
    U6 = np.column_stack([rng.uniform(-0.5, 0.5, NPTS), rng.uniform(-1.0, 1.0, NPTS)])
    dt = 0.1

    def subs6(i, L, k):
        s = {S["x"]: X6[i, 0], S["y"]: X6[i, 1], S["theta"]: X6[i, 2], S["delta"]: X6[i, 3], S["v"]: X6[i, 4],
             S["a"]: X6[i, 5], S["u1"]: U6[i, 0], S["u2"]: U6[i, 1], S["L"]: L, S["k"]: k, S["dt"]: dt}
        return {key: sp.Float(float(val), DIGITS) for key, val in s.items()}

    # (1) jac_full_dynamic.py: Jx, Ju of the RK2 step; its first stage is written with L = 3.0, k = 0.0003
    jfd = _import_quiet("jac_full_dynamic")
    Lj, kj = 3.0, 0.0003
    out["full_X"], out["full_U"], out["full_dt"] = X6, U6, dt
    out["full_jac_L"], out["full_jac_k"] = Lj, kj
    out["full_jac_fx"] = np.stack([_matrix(jfd.Jx, subs6(i, Lj, kj)) for i in range(NPTS)])
    out["full_jac_fu"] = np.stack([_matrix(jfd.Ju, subs6(i, Lj, kj)) for i in range(NPTS)])

    # (2) rk2.py: the same Jacobian as a product of continuous-time Jacobians at (x, u) and at the midpoint,
    #     symbolic in L and k; the midpoint symbols are replaced by x + dt/2 f(x, u)
    rk2 = _import_quiet("rk2")
    mid_expr = rk2.state + rk2.dt / 2 * rk2.f
    mid_subs = {rk2.mid_state[i]: mid_expr[i] for i in range(6)}
    Jx_rk2 = rk2.J_x_discrete.subs(mid_subs)
    Ju_rk2 = rk2.J_u_discrete.subs(mid_subs)
    Lr, kr = 2.7, 1.0e-3
    out["full_rk2_L"], out["full_rk2_k"] = Lr, kr
    out["full_rk2_fx"] = np.stack([_matrix(Jx_rk2, subs6(i, Lr, kr)) for i in range(NPTS)])
    out["full_rk2_fu"] = np.stack([_matrix(Ju_rk2, subs6(i, Lr, kr)) for i in range(NPTS)])

    # (3) hessian.py: second derivatives of the Euler step x + dt f(x, u), first three rows
    hes = _import_quiet("hessian")
    out["full_fxx_L"], out["full_fxx_k"] = Lr, kr
    fxx = np.zeros((NPTS, 3, 6, 6))
    for i in range(NPTS):
        sb = subs6(i, Lr, kr)
        for r in range(3):
            for p in range(6):
                for q in range(6):
                    fxx[i, r, p, q] = _evalf(hes.hessian_tensors[r][p][q], sb)
    out["full_fxx"] = fxx

    # ---- lateral model (jac_lat_dynamic.py): first stage written with L = 3.0; k, v symbolic
    jl = _import_quiet("jac_lat_dynamic")
    X4 = np.column_stack([rng.uniform(-50, 50, NPTS), rng.uniform(-50, 50, NPTS), rng.uniform(-3.0, 3.0, NPTS),
                          rng.uniform(-0.6, 0.6, NPTS)])
    U4 = rng.uniform(-0.5, 0.5, (NPTS, 1))
    Ll, kl, vl = 3.0, 2.0e-3, 8.0

    def subs4(i, zero_u=False):
        s = {S["x"]: X4[i, 0], S["y"]: X4[i, 1], S["theta"]: X4[i, 2], S["delta"]: X4[i, 3], S["v"]: vl,
             S["u"]: 0.0 if zero_u else U4[i, 0], S["L"]: Ll, S["k"]: kl, S["dt"]: dt}
        return {key: sp.Float(float(val), DIGITS) for key, val in s.items()}

    out["lat_X"], out["lat_U"], out["lat_dt"] = X4, U4, dt
    out["lat_L"], out["lat_k"], out["lat_v"] = Ll, kl, vl
    out["lat_fx"] = np.stack([_matrix(jl.Jx, subs4(i)) for i in range(NPTS)])
    out["lat_fu"] = np.stack([_matrix(jl.Ju, subs4(i)) for i in range(NPTS)])
    hx = np.zeros((NPTS, 3, 4, 4))
    hx0 = np.zeros((NPTS, 3, 4, 4))
    for i in range(NPTS):
        sb, sb0 = subs4(i), subs4(i, zero_u=True)
        for r in range(3):
            for p in range(4):
                for q in range(4):
                    hx[i, r, p, q] = _evalf(jl.hessian_tensor[r][p][q], sb)
                    hx0[i, r, p, q] = _evalf(jl.hessian_tensor[r][p][q], sb0)
    out["lat_fxx"] = hx          # exact second derivatives of the RK2 step at (x, u)
    out["lat_fxx_u0"] = hx0      # the same at (x, 0)

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
