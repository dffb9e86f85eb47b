// models.hpp — bicycle models of the solve path for gfx950 (fp64, one value per lane).
//
// Each model gives the RK2-midpoint step x+ = f(x,u) and the closed-form
// Jacobian of that step.  The Jacobian is I + a few entries; only the entries
// that vary are produced ("compact" form) and the structural pattern is
// constexpr so that the Riccati code, once unrolled, touches non-zeros only.
//
// The arithmetic of step()/jac() keeps the operation order of the Python node
// files (no FMA contraction: the translation unit is built with
// -ffp-contract=off); sin/cos/tan are the range-limited kernels of trig.hpp.
//
// Reference formulas:
//   LAT      cilqr/lat_bicycle_node.py:32-91
//   KIN      cilqr/full_bicycle_kinematic_node.py:31-105
//   DYN      cilqr/full_bicycle_dynamic_node.py:32-108
//   DYN_MID  cilqr/fast_bicycle_node.py:27-139
//   LAT_K / NEW_BICYCLE  al_ilqr_cpp/model/new_lat_bicycle_node.h, new_bicycle_node.h
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/alilqr.h"
#include "trig.hpp"

namespace alilqr {

constexpr double kPi = 3.141592653589793;
constexpr double kTwoPi = 6.283185307179586;

// (a + pi) % (2 pi) - pi with Python floor-mod semantics
// (cilqr/lat_bicycle_node.py:23-25).  For a in [-pi, pi) the modulo is the
// identity, which is the case on every call but the first on a raw start state.
__device__ __forceinline__ double wrap_py(double a) {
  double t = a + kPi;
  if (t >= 0.0 && t < kTwoPi) return t - kPi;
  double r = fmod_far(t, kTwoPi);
  if (r != 0.0) {
    if (r < 0.0) r += kTwoPi;
  } else {
    r = 0.0;
  }
  return r - kPi;
}

// the argument is known to be an output of wrap_py (a fixed point of it): nothing to do
template <bool WRAPPED>
__device__ __forceinline__ double wrap_in(double a) { return WRAPPED ? a : wrap_py(a); }

struct ModelConst {
  double L, dt, v, k;
};

// ---------------------------------------------------------------------------
// Structural description shared by the Riccati code.
//   A = I + S : S(i,j) != 0 only where a_slot(i,j) >= 0 or a_const(i,j)
//   B         : b_slot / b_const likewise
// ---------------------------------------------------------------------------
template <int MODEL>
struct Model;

// ---- 4 states / 1 control ---------------------------------------------------
// Traits read by the solver: WRAPS — states 2 and 3 are angles kept in [-pi, pi); CTRL_ROWS — the LEGACY box holds
// control rows (c = 2(n+m)) or state rows only (c = 2n); GOAL_COST — the state cost is (x-g)'Q(x-g) or x'Qx.
struct Lat4Base {
  static constexpr int N = 4, M = 1;
  static constexpr int NA = 5, NB = 1;
  static constexpr bool WRAPS = true, CTRL_ROWS = true, GOAL_COST = true;
  // compact A order: (0,2) (0,3) (1,2) (1,3) (2,3)
  __host__ __device__ static constexpr int a_slot(int i, int j) {
    return (i == 0 && j == 2) ? 0 : (i == 0 && j == 3) ? 1 : (i == 1 && j == 2) ? 2 : (i == 1 && j == 3) ? 3
         : (i == 2 && j == 3) ? 4 : -1;
  }
  __host__ __device__ static constexpr bool a_is_dt(int, int) { return false; }
  // compact B order: (2,0); (3,0) is the constant dt
  __host__ __device__ static constexpr int b_slot(int i, int j) { return (i == 2 && j == 0) ? 0 : -1; }
  __host__ __device__ static constexpr int b_const(int i, int j) { return (i == 3 && j == 0) ? 1 : 0; }  // 1: dt
};

template <>
struct Model<ALILQR_MODEL_LAT> : Lat4Base {
  template <bool WRAPPED = false>
  __device__ static void step(const ModelConst& mc, const double* x, const double* u, double* xn) {
    const double v = mc.v, dt = mc.dt, L = mc.L;
    double th = wrap_in<WRAPPED>(x[2]), de = wrap_in<WRAPPED>(x[3]);
    double s1, c1;
    sincos_pi(th, &s1, &c1);
    const bool unitL = (L == 1.0);      // x / 1.0 == x: skip the division
    double k1[4] = {v * c1, v * s1, unitL ? v * tan_pi(de) : v * tan_pi(de) / L, u[0]};
    double hdt = 0.5 * dt;
    double mid[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mid[i] = x[i] + hdt * k1[i];
    double thm = wrap_py(mid[2]), dem = wrap_py(mid[3]);
    double s2, c2;
    sincos_pi(thm, &s2, &c2);
    double k2[4] = {v * c2, v * s2, unitL ? v * tan_pi(dem) : v * tan_pi(dem) / L, u[0]};
#pragma unroll
    for (int i = 0; i < 4; ++i) xn[i] = x[i] + dt * k2[i];
    xn[2] = wrap_py(xn[2]);
    xn[3] = wrap_py(xn[3]);
  }
  template <bool WRAPPED = false>
  __device__ static void jac(const ModelConst& mc, const double* x, const double* uc, double* a, double* b) {
    const double v = mc.v, dt = mc.dt, L = mc.L;
    double th = wrap_in<WRAPPED>(x[2]), de = wrap_in<WRAPPED>(x[3]);
    double u = uc[0];
    double td = tan_w(de);
    double tm = th + 0.5 * dt * v * td / L;
    double tu = tan_w(de + 0.5 * dt * u);
    double sm, cm;
    sincos_w(tm, &sm, &cm);
    double dt2 = dt * dt, v2 = v * v;
    a[0] = -dt * v * sm;
    a[1] = -0.5 * dt2 * v2 * (td * td + 1) * sm / L;
    a[2] = dt * v * cm;
    a[3] = 0.5 * dt2 * v2 * (td * td + 1) * cm / L;
    a[4] = dt * v * (tu * tu + 1) / L;
    b[0] = 0.5 * dt2 * v * (tu * tu + 1) / L;
  }
  // The Jacobian entries above are products of constants with four values: sin / cos of the mid-point heading and
  // tan^2 + 1 of the two steering angles.  A linearisation record keeps those four (NC) instead of the six entries;
  // jac_unpack() repeats the products of jac() in the same order, so the entries come out with the same bits.
  static constexpr int NC = 4;
  template <bool WRAPPED = false>
  __device__ static void jac_pack(const ModelConst& mc, const double* x, const double* uc, double* c) {
    const double v = mc.v, dt = mc.dt, L = mc.L;
    double th = wrap_in<WRAPPED>(x[2]), de = wrap_in<WRAPPED>(x[3]);
    double td = tan_w(de);
    double tm = th + 0.5 * dt * v * td / L;
    double tu = tan_w(de + 0.5 * dt * uc[0]);
    sincos_w(tm, &c[0], &c[1]);
    c[2] = td * td + 1;
    c[3] = tu * tu + 1;
  }
  __device__ static __forceinline__ void jac_unpack(const ModelConst& mc, const double* c, double* a, double* b) {
    const double v = mc.v, dt = mc.dt, L = mc.L;
    const double sm = c[0], cm = c[1];
    double dt2 = dt * dt, v2 = v * v;
    a[0] = -dt * v * sm;
    a[1] = -0.5 * dt2 * v2 * c[2] * sm / L;
    a[2] = dt * v * cm;
    a[3] = 0.5 * dt2 * v2 * c[2] * cm / L;
    a[4] = dt * v * c[3] / L;
    b[0] = 0.5 * dt2 * v * c[3] / L;
  }
};

// ---- lateral model with a squashed control (cilqr/lat_bicycle_interior_node.py) --------------------------------
// delta_dot = control_max * tanh(u) (:55); f_u(3) = dt control_max (1 - tanh(u)^2) (:88); everything else as LAT.
// control_max travels in ModelConst::k.  State bounds only, and the state cost does not subtract the goal (:96,106).
template <>
struct Model<ALILQR_MODEL_LAT_INTERIOR> : Lat4Base {
  static constexpr int NB = 2;
  static constexpr bool CTRL_ROWS = false, GOAL_COST = false;
  // compact B order: (2,0) (3,0)
  __host__ __device__ static constexpr int b_slot(int i, int j) { return (j == 0 && i == 2) ? 0 : (j == 0 && i == 3) ? 1 : -1; }
  __host__ __device__ static constexpr int b_const(int, int) { return 0; }
  template <bool WRAPPED = false>
  __device__ static void step(const ModelConst& mc, const double* x, const double* u, double* xn) {
    const double v = mc.v, dt = mc.dt, L = mc.L;
    double th = wrap_in<WRAPPED>(x[2]), de = wrap_in<WRAPPED>(x[3]);
    double s1, c1;
    sincos_pi(th, &s1, &c1);
    const double rate = mc.k * tanh(u[0]);
    double k1[4] = {v * c1, v * s1, v * tan_pi(de) / L, rate};
    double hdt = 0.5 * dt;
    double mid[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mid[i] = x[i] + hdt * k1[i];
    double thm = wrap_py(mid[2]), dem = wrap_py(mid[3]);
    double s2, c2;
    sincos_pi(thm, &s2, &c2);
    double k2[4] = {v * c2, v * s2, v * tan_pi(dem) / L, rate};
#pragma unroll
    for (int i = 0; i < 4; ++i) xn[i] = x[i] + dt * k2[i];
    xn[2] = wrap_py(xn[2]);
    xn[3] = wrap_py(xn[3]);
  }
  template <bool WRAPPED = false>
  __device__ static void jac(const ModelConst& mc, const double* x, const double* uc, double* a, double* b) {
    Model<ALILQR_MODEL_LAT>::jac<WRAPPED>(mc, x, uc, a, b);       // :75-87 are the LAT expressions
    const double th = tanh(uc[0]);
    b[1] = mc.dt * mc.k * (1 - th * th);
  }
};

// ---- longitudinal triple integrator (cilqr/lon_node.py) ----------------------------------------------------------
// states (s, v, a), control jerk; RK2 of a linear system (:21-38); constant Jacobians (:49-60); no angles.
template <>
struct Model<ALILQR_MODEL_LON> {
  static constexpr int N = 3, M = 1;
  static constexpr int NA = 1, NB = 1;                 // a[0] = dt^2/2 at (0,2); b[0] unused (B is constant)
  static constexpr bool WRAPS = false, CTRL_ROWS = true, GOAL_COST = true;
  __host__ __device__ static constexpr int a_slot(int i, int j) { return (i == 0 && j == 2) ? 0 : -1; }
  __host__ __device__ static constexpr bool a_is_dt(int i, int j) { return (i == 0 && j == 1) || (i == 1 && j == 2); }
  __host__ __device__ static constexpr int b_slot(int, int) { return -1; }
  __host__ __device__ static constexpr int b_const(int i, int j) { return (j == 0 && i == 1) ? 2 : (j == 0 && i == 2) ? 1 : 0; }
  template <bool WRAPPED = false>
  __device__ static void step(const ModelConst& mc, const double* x, const double* u, double* xn) {
    const double dt = mc.dt, hdt = 0.5 * dt;
    const double k1[3] = {x[1], x[2], u[0]};
    double mid[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) mid[i] = x[i] + hdt * k1[i];
    const double k2[3] = {mid[1], mid[2], u[0]};
#pragma unroll
    for (int i = 0; i < 3; ++i) xn[i] = x[i] + dt * k2[i];
  }
  template <bool WRAPPED = false>
  __device__ static void jac(const ModelConst& mc, const double*, const double*, double* a, double* b) {
    a[0] = 0.5 * (mc.dt * mc.dt);
    b[0] = 0.0;
  }
};

// ---- 6 states / 2 controls --------------------------------------------------
struct Full6Base {
  static constexpr int N = 6, M = 2;
  static constexpr int NA = 11, NB = 1;
  static constexpr bool WRAPS = true, CTRL_ROWS = true, GOAL_COST = true;
  // compact A order: row 0 cols 2..5, row 1 cols 2..5, row 2 cols 3..5 ; (4,5) is the constant dt
  __host__ __device__ static constexpr int a_slot(int i, int j) {
    return (i == 0 && j >= 2) ? (j - 2) : (i == 1 && j >= 2) ? (4 + j - 2) : (i == 2 && j >= 3) ? (8 + j - 3) : -1;
  }
  __host__ __device__ static constexpr bool a_is_dt(int i, int j) { return i == 4 && j == 5; }
  // compact B: (2,0) ; constants (3,0)=dt (4,1)=dt^2/2 (5,1)=dt
  __host__ __device__ static constexpr int b_slot(int i, int j) { return (i == 2 && j == 0) ? 0 : -1; }
  __host__ __device__ static constexpr int b_const(int i, int j) {
    return (i == 3 && j == 0) ? 1 : (i == 4 && j == 1) ? 2 : (i == 5 && j == 1) ? 1 : 0;  // 1: dt, 2: dt^2/2
  }
};

template <int MODEL>
struct Full6 : Full6Base {
  template <bool WRAPPED>
  __device__ static void cont(const ModelConst& mc, const double* s, const double* u, double* out) {
    double th = wrap_in<WRAPPED>(s[2]), de = wrap_in<WRAPPED>(s[3]);
    double v = s[4], acc = s[5];
    double sn, cs;
    sincos_pi(th, &sn, &cs);
    out[0] = v * cs;
    out[1] = v * sn;
    if (MODEL == ALILQR_MODEL_KIN)
      out[2] = v * tan_pi(de) / mc.L;
    else
      out[2] = v * tan_pi(de) / (mc.L * (1 + mc.k * (v * v)));
    out[3] = u[0];
    out[4] = acc;
    out[5] = u[1];
  }
  template <bool WRAPPED = false>
  __device__ static void step(const ModelConst& mc, const double* x, const double* u, double* xn) {
    double k1[6], k2[6], mid[6];
    cont<WRAPPED>(mc, x, u, k1);
    double hdt = 0.5 * mc.dt;
#pragma unroll
    for (int i = 0; i < 6; ++i) mid[i] = x[i] + hdt * k1[i];
    cont<false>(mc, mid, u, k2);
#pragma unroll
    for (int i = 0; i < 6; ++i) xn[i] = x[i] + mc.dt * k2[i];
    xn[2] = wrap_py(xn[2]);
    xn[3] = wrap_py(xn[3]);
  }
  template <bool WRAPPED = false>
  __device__ static void jac(const ModelConst& mc, const double* x, const double* uc, double* a, double* b) {
    const double dt = mc.dt, L = mc.L, k = mc.k;
    double theta = wrap_in<WRAPPED>(x[2]), delta = wrap_in<WRAPPED>(x[3]);
    double v = x[4], acc = x[5], u1 = uc[0];
    double dt2 = dt * dt;
    double tan_delta = tan_w(delta);
    double tan_u1 = tan_w(delta + 0.5 * dt * u1);
    double vt = 0.5 * acc * dt + v;
    if (MODEL == ALILQR_MODEL_KIN) {
      double theta_mid = theta + 0.5 * dt * v * tan_delta / L;
      double sn, c;
      sincos_w(theta_mid, &sn, &c);
      double tds = tan_delta * tan_delta, tus = tan_u1 * tan_u1;
      double v_tu = v * (tus + 1);
      a[0] = -dt * vt * sn;
      a[1] = -0.5 * dt2 * v * vt * (tds + 1) * sn / L;
      a[2] = dt * c - 0.5 * dt2 * vt * sn * tan_delta / L;
      a[3] = 0.5 * dt2 * c;
      a[4] = dt * vt * c;
      a[5] = 0.5 * dt2 * v * vt * (tds + 1) * c / L;
      a[6] = dt * sn + 0.5 * dt2 * vt * c * tan_delta / L;
      a[7] = 0.5 * dt2 * sn;
      a[8] = dt * v_tu / L;
      a[9] = dt * tan_u1 / L;
      a[10] = 0.5 * dt2 * tan_u1 / L;
      b[0] = 0.5 * dt2 * v * (tus + 1) / L;
    } else if (MODEL == ALILQR_MODEL_DYN) {
      double sf = 1 + k * (v * v);
      double theta_mid = theta + 0.5 * dt * v * tan_delta / (L * sf);
      double sn, c;
      sincos_w(theta_mid, &sn, &c);
      double tds = tan_delta * tan_delta, tus = tan_u1 * tan_u1;
      double v_tu = v * (tus + 1);
      double g = -dt * k * (v * v) * tan_delta / (L * (sf * sf)) + 0.5 * dt * tan_delta / (L * sf);
      a[0] = -dt * vt * sn;
      a[1] = -0.5 * dt2 * v * vt * (tds + 1) * sn / (L * sf);
      a[2] = dt * c - dt * vt * g * sn;
      a[3] = 0.5 * dt2 * c;
      a[4] = dt * vt * c;
      a[5] = 0.5 * dt2 * v * vt * (tds + 1) * c / (L * sf);
      a[6] = dt * sn + dt * vt * g * c;
      a[7] = 0.5 * dt2 * sn;
      a[8] = dt * v_tu / (L * sf);
      a[9] = dt * tan_u1 / (L * sf) - dt * k * vt * (acc * dt + 2 * v) * tan_u1 / (L * (sf * sf));
      a[10] = 0.5 * dt2 * tan_u1 / (L * sf) - 0.5 * dt2 * k * (vt * vt) * tan_u1 / (L * (sf * sf));
      b[0] = 0.5 * dt2 * v_tu / (L * sf);
    } else {  // DYN_MID
      double k_v_sq = k * (v * v);
      double k_vm_sq = k * (vt * vt);
      double denom = L * (k_v_sq + 1);
      double denom_mid = L * (k_vm_sq + 1);
      double theta_mid = theta + 0.5 * dt * v * tan_delta / denom;
      double sn, c;
      sincos_w(theta_mid, &sn, &c);
      double g = -dt * k_v_sq * tan_delta / (denom * denom) + 0.5 * dt * tan_delta / denom;
      double tms = tan_u1 * tan_u1;
      a[0] = -dt * vt * sn;
      a[1] = -0.5 * dt2 * v * vt * (tan_delta * tan_delta + 1) * sn / denom;
      a[2] = -dt * vt * g * sn + dt * c;
      a[3] = 0.5 * dt2 * c;
      a[4] = dt * vt * c;
      a[5] = 0.5 * dt2 * v * vt * (tan_delta * tan_delta + 1) * c / denom;
      a[6] = dt * vt * g * c + dt * sn;
      a[7] = 0.5 * dt2 * sn;
      a[8] = dt * vt * (tms + 1) / denom_mid;
      a[9] = -dt * k_vm_sq * (1.0 * acc * dt + 2 * v) * tan_u1 / (denom_mid * denom_mid) + dt * tan_u1 / denom_mid;
      a[10] = -dt2 * k_vm_sq * tan_u1 / (denom_mid * denom_mid) + 0.5 * dt2 * tan_u1 / denom_mid;
      b[0] = 0.5 * dt2 * vt * (tms + 1) / denom_mid;
    }
  }
};

template <> struct Model<ALILQR_MODEL_KIN> : Full6<ALILQR_MODEL_KIN> {};
template <> struct Model<ALILQR_MODEL_DYN> : Full6<ALILQR_MODEL_DYN> {};
template <> struct Model<ALILQR_MODEL_DYN_MID> : Full6<ALILQR_MODEL_DYN_MID> {};

// ---------------------------------------------------------------------------
// Models of the C++ generation (NEW_AL).  Differences to the Python nodes:
//   * angle wrap: identity inside (-pi, pi), C fmod (sign of the dividend) outside
//     (al_ilqr_cpp/model/new_ilqr_node.h:32-37)
//   * the lateral model carries the constant (1 + k v^2) factor
//   * f_xx of the first three state rows for the Riccati recursion
//   * an unwrapped "parallel" step used by the five-wide line-search ladder
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wrap_c(double a) {
  if (fabs(a) < 3.14159265358979323846) return a;
  return fmod_far(a + 3.14159265358979323846, 2 * 3.14159265358979323846) - 3.14159265358979323846;
}

template <>
struct Model<ALILQR_MODEL_LAT_K> : Lat4Base {
  static constexpr int NF = 7;   // f_xx entries: x(2,2) x(2,3) x(3,3) y(2,2) y(2,3) y(3,3) theta(3,3)
  __device__ static void cont(const ModelConst& mc, const double* s, const double* u, double* out) {
    double th = wrap_c(s[2]), de = wrap_c(s[3]);
    double sn, cs;
    sincos_w(th, &sn, &cs);
    out[0] = mc.v * cs;
    out[1] = mc.v * sn;
    out[2] = mc.v * tan_w(de) / (mc.L * (1 + mc.k * mc.v * mc.v));
    out[3] = u[0];
  }
  template <bool WRAPPED = false>
  __device__ static void step(const ModelConst& mc, const double* x, const double* u, double* xn) {
    double k1[4], k2[4], mid[4];
    cont(mc, x, u, k1);
    const double hdt = 0.5 * mc.dt;
#pragma unroll
    for (int i = 0; i < 4; ++i) mid[i] = x[i] + hdt * k1[i];
    cont(mc, mid, u, k2);
#pragma unroll
    for (int i = 0; i < 4; ++i) xn[i] = x[i] + mc.dt * k2[i];
    xn[2] = wrap_c(xn[2]);
    xn[3] = wrap_c(xn[3]);
  }
  // one column of parallel_dynamics: no wrap at all (new_lat_bicycle_node.h:60-87)
  __device__ static void par_cont(const ModelConst& mc, const double* s, const double* u, double* out) {
    const double denom = mc.L * (1.0 + mc.k * mc.v * mc.v);
    double sn, cs;
    sincos_w(s[2], &sn, &cs);
    out[0] = mc.v * cs;
    out[1] = mc.v * sn;
    out[2] = mc.v * tan_w(s[3]) / denom;
    out[3] = u[0];
  }
  __device__ static void par_step(const ModelConst& mc, const double* x, const double* u, double* xn) {
    double k1[4], k2[4], mid[4];
    par_cont(mc, x, u, k1);
#pragma unroll
    for (int i = 0; i < 4; ++i) mid[i] = x[i] + k1[i] * mc.dt * 0.5;
    par_cont(mc, mid, u, k2);
#pragma unroll
    for (int i = 0; i < 4; ++i) xn[i] = x[i] + k2[i] * mc.dt;
  }
  template <bool WRAPPED = false>
  __device__ static void jac(const ModelConst& mc, const double* x, const double* uc, double* a, double* b) {
    const double v = mc.v, dt = mc.dt, L = mc.L, k = mc.k;
    double theta = wrap_c(x[2]), delta = wrap_c(x[3]);
    double u1 = uc[0];
    double tan_delta = tan_w(delta);
    double theta_mid = theta + 0.5 * dt * v * tan_delta / (L * (k * v * v + 1));
    double tan_delta_mid = tan_w(delta + 0.5 * dt * u1);
    double denom = L * (k * v * v + 1);
    double sn, cs;
    sincos_w(theta_mid, &sn, &cs);
    double t2 = tan_delta * tan_delta, tm2 = tan_delta_mid * tan_delta_mid;
    a[0] = -dt * v * sn;
    a[1] = -0.5 * dt * dt * v * v * (t2 + 1) * sn / denom;
    a[2] = dt * v * cs;
    a[3] = 0.5 * dt * dt * v * v * (t2 + 1) * cs / denom;
    a[4] = dt * v * (tm2 + 1) / denom;
    b[0] = 0.5 * dt * dt * v * (tm2 + 1) / denom;
  }
  // new_lat_bicycle_node.h:128-163
  __device__ static void fxx(const ModelConst& mc, const double* x, const double* uc, double* f) {
    const double v = mc.v, dt = mc.dt, L = mc.L, k = mc.k;
    (void)uc;
    double theta = wrap_c(x[2]), delta = wrap_c(x[3]);
    double td = tan_w(delta), t2 = td * td;
    double theta_mid = theta + 0.5 * dt * v * td / (L * (k * v * v + 1));
    double denom = L * (k * v * v + 1);
    double sn, cs;
    sincos_w(theta_mid, &sn, &cs);
    f[0] = -dt * v * cs;
    f[1] = -0.5 * dt * dt * v * v * (t2 + 1) * cs / denom;
    f[2] = -dt * dt * v * v * ((t2 + 1) / denom) * (0.5 * v * (t2 + 1) * cs / denom + sn * td);
    f[3] = -dt * v * sn;
    f[4] = -0.5 * dt * dt * v * v * (t2 + 1) * sn / denom;
    f[5] = dt * dt * v * v * ((t2 + 1) / denom) * (cs * td - 0.5 * v * (t2 + 1) * sn / denom);
    f[6] = 2 * dt * v * (t2 + 1) * td / denom;
  }
  // (F0 Vx0 + F1 Vx1) + F2 Vx2 at entry (i,j), zero entries dropped   (new_al_ilqr.h:954-956)
  __device__ static __forceinline__ double fxx_term(const double* f, const double* Vx, int i, int j) {
    if (i == 2 && j == 2) return f[0] * Vx[0] + f[3] * Vx[1];
    if ((i == 2 && j == 3) || (i == 3 && j == 2)) return f[1] * Vx[0] + f[4] * Vx[1];
    if (i == 3 && j == 3) return (f[2] * Vx[0] + f[5] * Vx[1]) + f[6] * Vx[2];
    return 0.0;
  }
  __host__ __device__ static constexpr bool fxx_nz(int i, int j) { return (i == 2 || i == 3) && (j == 2 || j == 3); }
};

template <>
struct Model<ALILQR_MODEL_NEW_BICYCLE> : Full6Base {
  static constexpr int NF = 7;   // f_xx entries: x(2,2) x(2,4) y(2,2) y(2,4) theta(3,3) theta(3,4) theta(4,4)
  __device__ static void cont(const ModelConst& mc, const double* s, const double* u, double* out) {
    double th = wrap_c(s[2]), de = wrap_c(s[3]);
    double v = s[4], acc = s[5];
    double sn, cs;
    sincos_w(th, &sn, &cs);
    out[0] = v * cs;
    out[1] = v * sn;
    out[2] = v * tan_w(de) / (mc.L * (1 + mc.k * v * v));
    out[3] = u[0];
    out[4] = acc;
    out[5] = u[1];
  }
  template <bool WRAPPED = false>
  __device__ static void step(const ModelConst& mc, const double* x, const double* u, double* xn) {
    double k1[6], k2[6], mid[6];
    cont(mc, x, u, k1);
    const double hdt = 0.5 * mc.dt;
#pragma unroll
    for (int i = 0; i < 6; ++i) mid[i] = x[i] + hdt * k1[i];
    cont(mc, mid, u, k2);
#pragma unroll
    for (int i = 0; i < 6; ++i) xn[i] = x[i] + mc.dt * k2[i];
    xn[2] = wrap_c(xn[2]);
    xn[3] = wrap_c(xn[3]);
  }
  // new_bicycle_node.h:285-377
  __device__ static void par_cont(const ModelConst& mc, const double* s, const double* u, double* out) {
    double v = s[4];
    double sn, cs;
    sincos_w(s[2], &sn, &cs);
    out[0] = v * cs;
    out[1] = v * sn;
    out[2] = v * tan_w(s[3]) * (1.0 / (1.0 + mc.k * v * v)) / mc.L;
    out[3] = u[0];
    out[4] = s[5];
    out[5] = u[1];
  }
  __device__ static void par_step(const ModelConst& mc, const double* x, const double* u, double* xn) {
    double k1[6], k2[6], mid[6];
    par_cont(mc, x, u, k1);
#pragma unroll
    for (int i = 0; i < 6; ++i) mid[i] = x[i] + k1[i] * mc.dt * 0.5;
    par_cont(mc, mid, u, k2);
#pragma unroll
    for (int i = 0; i < 6; ++i) xn[i] = x[i] + k2[i] * mc.dt;
  }
  // step() and par_step() as one instruction stream: `par` picks, lane by lane, the arithmetic of par_step.  Every value
  // is formed by the operations of the function it stands for, in their order (a product's factors may be swapped), so
  // the bits are those of step() or par_step(); lanes of both kinds share the sin / cos / tan evaluations.
  __device__ static __forceinline__ void cont2(const ModelConst& mc, const double* s, const double* u, double* out, bool par) {
    const double th = par ? s[2] : wrap_c(s[2]), de = par ? s[3] : wrap_c(s[3]);
    const double v = s[4];
    double sn, cs;
    sincos_w(th, &sn, &cs);
    out[0] = v * cs;
    out[1] = v * sn;
    const double vt = v * tan_w(de), den = 1 + mc.k * v * v;
    if (par) out[2] = vt * (1.0 / den) / mc.L;
    else out[2] = vt / (mc.L * den);
    out[3] = u[0];
    out[4] = s[5];
    out[5] = u[1];
  }
  __device__ static void step2(const ModelConst& mc, const double* x, const double* u, double* xn, bool par) {
    double k1[6], k2[6], mid[6];
    cont2(mc, x, u, k1, par);
    const double hdt = 0.5 * mc.dt;
#pragma unroll
    for (int i = 0; i < 6; ++i) mid[i] = par ? x[i] + k1[i] * mc.dt * 0.5 : x[i] + hdt * k1[i];
    cont2(mc, mid, u, k2, par);
#pragma unroll
    for (int i = 0; i < 6; ++i) xn[i] = x[i] + mc.dt * k2[i];
    xn[2] = par ? xn[2] : wrap_c(xn[2]);
    xn[3] = par ? xn[3] : wrap_c(xn[3]);
  }
  // new_bicycle_node.h:492-545
  template <bool WRAPPED = false>
  __device__ static void jac(const ModelConst& mc, const double* x, const double* uc, double* a, double* b) {
    const double dt = mc.dt, L = mc.L, k = mc.k;
    double theta = wrap_c(x[2]), delta = wrap_c(x[3]);
    double v = x[4], acc = x[5], u1 = uc[0];
    double D = L * (k * v * v + 1);
    double td = tan_w(delta), tu = tan_w(delta + 0.5 * dt * u1);
    double tm = theta + 0.5 * dt * v * td / D;
    double vt = 0.5 * acc * dt + v;
    double Dm = L * (k * vt * vt + 1);
    double sn, cs;
    sincos_w(tm, &sn, &cs);
    double g = -1.0 * dt * k * v * v * td / (L * (k * v * v + 1) * (k * v * v + 1)) + 0.5 * dt * td / D;
    a[0] = -dt * vt * sn;
    a[1] = -0.5 * dt * dt * v * vt * (td * td + 1) * sn / D;
    a[2] = -dt * vt * g * sn + dt * cs;
    a[3] = 0.5 * dt * dt * cs;
    a[4] = dt * vt * cs;
    a[5] = 0.5 * dt * dt * v * vt * (td * td + 1) * cs / D;
    a[6] = dt * vt * g * cs + dt * sn;
    a[7] = 0.5 * dt * dt * sn;
    a[8] = dt * vt * (tu * tu + 1) / Dm;
    a[9] = -dt * k * vt * (1.0 * acc * dt + 2 * v) * tu / (L * (k * vt * vt + 1) * (k * vt * vt + 1)) + dt * tu / Dm;
    a[10] = -1.0 * dt * dt * k * vt * vt * tu / (L * (k * vt * vt + 1) * (k * vt * vt + 1)) + 0.5 * dt * dt * tu / Dm;
    b[0] = 0.5 * dt * dt * vt * (tu * tu + 1) / Dm;
  }
  // new_bicycle_node.h:605-650 (Euler form)
  __device__ static void fxx(const ModelConst& mc, const double* x, const double* uc, double* f) {
    const double dt = mc.dt, L = mc.L, k = mc.k;
    (void)uc;
    double theta = wrap_c(x[2]), delta = wrap_c(x[3]), v = x[4];
    double sn, cs;
    sincos_w(theta, &sn, &cs);
    double td = tan_w(delta);
    double tp1 = td * td + 1;
    double kv2 = k * v * v, kp1 = kv2 + 1;
    f[0] = -dt * v * cs;
    f[1] = -dt * sn;
    f[2] = -dt * v * sn;
    f[3] = dt * cs;
    f[4] = 2 * dt * v * tp1 * td / (kp1 * L);
    f[5] = dt * (1 - k * v * v) * tp1 / (kp1 * L) / (kp1);
    f[6] = dt * 2 * k * v * (kv2 - 3) * td / L / kp1 / kp1 / kp1;
  }
  __device__ static __forceinline__ double fxx_term(const double* f, const double* Vx, int i, int j) {
    if (i == 2 && j == 2) return f[0] * Vx[0] + f[2] * Vx[1];
    if ((i == 2 && j == 4) || (i == 4 && j == 2)) return f[1] * Vx[0] + f[3] * Vx[1];
    if (i == 3 && j == 3) return f[4] * Vx[2];
    if ((i == 3 && j == 4) || (i == 4 && j == 3)) return f[5] * Vx[2];
    if (i == 4 && j == 4) return f[6] * Vx[2];
    return 0.0;
  }
  __host__ __device__ static constexpr bool fxx_nz(int i, int j) { return (i >= 2 && i <= 4) && (j >= 2 && j <= 4); }
};

// What a linearisation record keeps of a node's dynamics Jacobian: the model's packed form where it has one
// (NC values, jac_pack / jac_unpack), else the compact entries themselves.
template <class MD, class = void>
struct JacPack {
  static constexpr int NC = MD::NA + MD::NB;
  template <bool WRAPPED>
  __device__ static __forceinline__ void pack(const ModelConst& mc, const double* x, const double* u, double* c) {
    MD::template jac<WRAPPED>(mc, x, u, c, c + MD::NA);
  }
  __device__ static __forceinline__ void unpack(const ModelConst&, const double* c, double* a, double* b) {
#pragma unroll
    for (int i = 0; i < MD::NA; ++i) a[i] = c[i];
#pragma unroll
    for (int i = 0; i < MD::NB; ++i) b[i] = c[MD::NA + i];
  }
};
template <class MD>
struct JacPack<MD, decltype((void)MD::NC)> {
  static constexpr int NC = MD::NC;
  template <bool WRAPPED>
  __device__ static __forceinline__ void pack(const ModelConst& mc, const double* x, const double* u, double* c) {
    MD::template jac_pack<WRAPPED>(mc, x, u, c);
  }
  __device__ static __forceinline__ void unpack(const ModelConst& mc, const double* c, double* a, double* b) {
    MD::jac_unpack(mc, c, a, b);
  }
};

}  // namespace alilqr
