/*
 * alilqr_oracle.c — CPU restatement of the reference AL-iLQR solvers.
 *
 * TEST INFRASTRUCTURE ONLY (see alilqr_oracle.h).  Plain C, dense small
 * matrices, one problem at a time, written to follow the reference statement
 * by statement so that it can be pinned against vectors produced by the
 * reference itself (tests/golden/, made by tools/make_goldens.py).
 *
 * Restated here:
 *   legacy solver   cilqr/ilqr.py:6-192
 *   legacy nodes    cilqr/lat_bicycle_node.py, full_bicycle_kinematic_node.py,
 *                   full_bicycle_dynamic_node.py
 *   fast solver     cilqr/fast_ilqr.py:5-176
 *   fast node       cilqr/fast_bicycle_node.py:7-212
 *   AL machinery    cilqr/constraints.py:4-119, linear_constraints.py:4-29,
 *                   box_constrains.py:5-32
 * The NEW_AL generation lives in alilqr_oracle_new.c.
 */
#include "alilqr_oracle.h"
#include "oracle_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* small dense helpers (row-major)                                           */
/* ------------------------------------------------------------------------ */
void om_mm(int r, int k, int c, const double* A, const double* B, double* C) {
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) {
      double s = 0.0;
      for (int q = 0; q < k; ++q) s += A[i * k + q] * B[q * c + j];
      C[i * c + j] = s;
    }
}
/* C = A^T B, A is k x r, B is k x c */
void om_mtm(int r, int k, int c, const double* A, const double* B, double* C) {
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) {
      double s = 0.0;
      for (int q = 0; q < k; ++q) s += A[q * r + i] * B[q * c + j];
      C[i * c + j] = s;
    }
}
/* inverse by LU with partial pivoting, then a solve against the identity
 * (what numpy.linalg.inv does through LAPACK gesv); m <= OR_NU */
int om_inv(int m, const double* Ain, double* Ainv) {
  double a[OR_NU * OR_NU];
  int piv[OR_NU];
  memcpy(a, Ain, sizeof(double) * m * m);
  for (int j = 0; j < m; ++j) {
    int p = j;
    for (int i = j + 1; i < m; ++i)
      if (fabs(a[i * m + j]) > fabs(a[p * m + j])) p = i;
    piv[j] = p;
    if (p != j)
      for (int q = 0; q < m; ++q) { double t = a[j * m + q]; a[j * m + q] = a[p * m + q]; a[p * m + q] = t; }
    double rcp = 1.0 / a[j * m + j];
    for (int i = j + 1; i < m; ++i) {
      a[i * m + j] *= rcp;
      for (int q = j + 1; q < m; ++q) a[i * m + q] -= a[i * m + j] * a[j * m + q];
    }
  }
  for (int col = 0; col < m; ++col) {
    double b[OR_NU];
    for (int i = 0; i < m; ++i) b[i] = (i == col) ? 1.0 : 0.0;
    for (int j = 0; j < m; ++j)
      if (piv[j] != j) { double t = b[j]; b[j] = b[piv[j]]; b[piv[j]] = t; }
    for (int i = 0; i < m; ++i)
      for (int q = 0; q < i; ++q) b[i] -= a[i * m + q] * b[q];
    for (int i = m - 1; i >= 0; --i) {
      for (int q = i + 1; q < m; ++q) b[i] -= a[i * m + q] * b[q];
      b[i] /= a[i * m + i];
    }
    for (int i = 0; i < m; ++i) Ainv[i * m + col] = b[i];
  }
  return 0;
}

/* closed-form inverse of a 1x1 / 2x2 matrix as Eigen's fixed-size inverse() computes it */
void or_inv_closed(int m, const double* M, double* Mi) {
  if (m == 1) { Mi[0] = 1.0 / M[0]; return; }
  double det = M[0] * M[3] - M[1] * M[2];
  double invdet = 1.0 / det;
  Mi[0] = M[3] * invdet;
  Mi[1] = -M[1] * invdet;
  Mi[2] = -M[2] * invdet;
  Mi[3] = M[0] * invdet;
}

/* ------------------------------------------------------------------------ */
/* angle wrap: (a + pi) % (2 pi) - pi with Python's floor-mod                 */
/* cilqr/lat_bicycle_node.py:23-25                                            */
/* ------------------------------------------------------------------------ */
static double py_mod(double a, double b) {
  double r = fmod(a, b);
  if (r != 0.0) {
    if ((b < 0.0) != (r < 0.0)) r += b;
  } else {
    r = copysign(0.0, b);
  }
  return r;
}
double or_wrap_py(double a) { return py_mod(a + OR_PI, 2.0 * OR_PI) - OR_PI; }

static void normalize_state(or_node* nd) {   /* lat_bicycle_node.py:27-30 ; lon_node.py has no angles */
  if (nd->model == ALILQR_MODEL_LON) return;
  nd->state[2] = or_wrap_py(nd->state[2]);
  nd->state[3] = or_wrap_py(nd->state[3]);
}

/* ------------------------------------------------------------------------ */
/* dynamics (RK2 midpoint) and its closed-form Jacobian                       */
/* ------------------------------------------------------------------------ */
/* lat_bicycle_node.py:48-56 ; full_bicycle_kinematic_node.py:47-58 ;
 * full_bicycle_dynamic_node.py:48-59 ; fast_bicycle_node.py:43-54 */
static void dyn_cont(const or_node* nd, const double* s, const double* u, double* out) {
  if (nd->model == ALILQR_MODEL_LON) {          /* lon_node.py:33-38 */
    out[0] = s[1];
    out[1] = s[2];
    out[2] = u[0];
    return;
  }
  double theta = or_wrap_py(s[2]);
  double delta = or_wrap_py(s[3]);
  if (nd->model == ALILQR_MODEL_LAT) {
    out[0] = nd->v * cos(theta);
    out[1] = nd->v * sin(theta);
    out[2] = nd->v * tan(delta) / nd->L;
    out[3] = u[0];
    return;
  }
  if (nd->model == ALILQR_MODEL_LAT_INTERIOR) {  /* lat_bicycle_interior_node.py:48-56 ; k holds control_max[0] */
    out[0] = nd->v * cos(theta);
    out[1] = nd->v * sin(theta);
    out[2] = nd->v * tan(delta) / nd->L;
    out[3] = nd->k * tanh(u[0]);
    return;
  }
  double v = s[4], a = s[5];
  out[0] = v * cos(theta);
  out[1] = v * sin(theta);
  if (nd->model == ALILQR_MODEL_KIN)
    out[2] = v * tan(delta) / nd->L;
  else
    out[2] = v * tan(delta) / (nd->L * (1 + nd->k * (v * v)));
  out[3] = u[0];
  out[4] = a;
  out[5] = u[1];
}

/* lat_bicycle_node.py:32-46 (same in the other node files) */
void or_py_dynamics(const or_node* nd, const double* s, const double* u, double* next) {
  double k1[OR_NX], k2[OR_NX], mid[OR_NX];
  int n = nd->n;
  dyn_cont(nd, s, u, k1);
  for (int i = 0; i < n; ++i) mid[i] = s[i] + 0.5 * nd->dt * k1[i];
  dyn_cont(nd, mid, u, k2);
  for (int i = 0; i < n; ++i) next[i] = s[i] + nd->dt * k2[i];
  if (nd->model == ALILQR_MODEL_LON) return;    /* lon_node.py:21-31 */
  next[2] = or_wrap_py(next[2]);
  next[3] = or_wrap_py(next[3]);
}

void or_py_jacobian(const or_node* nd, const double* s, const double* uc, double* A, double* B) {
  int n = nd->n, m = nd->m;
  double dt = nd->dt, L = nd->L;
  memset(A, 0, sizeof(double) * n * n);
  memset(B, 0, sizeof(double) * n * m);
  for (int i = 0; i < n; ++i) A[i * n + i] = 1.0;
  if (nd->model == ALILQR_MODEL_LON) {          /* lon_node.py:40-62 */
    A[0 * 3 + 1] = dt;
    A[0 * 3 + 2] = 0.5 * (dt * dt);
    A[1 * 3 + 2] = dt;
    B[1] = 0.5 * (dt * dt);
    B[2] = dt;
    return;
  }
  double theta = or_wrap_py(s[2]);
  double delta = or_wrap_py(s[3]);
  if (nd->model == ALILQR_MODEL_LAT || nd->model == ALILQR_MODEL_LAT_INTERIOR) {
    /* lat_bicycle_node.py:58-91 ; lat_bicycle_interior_node.py:58-91 (same f_x, f_u(3) = dt u_max (1 - tanh(u)^2)) */
    double v = nd->v, u = uc[0];
    double td = tan(delta);
    double tm = theta + 0.5 * dt * v * td / L;
    double tu = tan(delta + 0.5 * dt * u);
    A[0 * 4 + 2] = -dt * v * sin(tm);
    A[0 * 4 + 3] = -0.5 * (dt * dt) * (v * v) * (td * td + 1) * sin(tm) / L;
    A[1 * 4 + 2] = dt * v * cos(tm);
    A[1 * 4 + 3] = 0.5 * (dt * dt) * (v * v) * (td * td + 1) * cos(tm) / L;
    A[2 * 4 + 3] = dt * v * (tu * tu + 1) / L;
    B[2] = 0.5 * (dt * dt) * v * (tu * tu + 1) / L;
    if (nd->model == ALILQR_MODEL_LAT_INTERIOR) {
      double th = tanh(u);
      B[3] = dt * nd->k * (1 - th * th);
    } else {
      B[3] = dt;
    }
    return;
  }
  double v = s[4], a = s[5], u1 = uc[0];
  double dt2 = dt * dt;
  if (nd->model == ALILQR_MODEL_KIN) {          /* full_bicycle_kinematic_node.py:60-105 */
    double tan_delta = tan(delta);
    double theta_mid = theta + 0.5 * dt * v * tan_delta / L;
    double tan_delta_u1 = tan(delta + 0.5 * dt * u1);
    double c = cos(theta_mid), sn = sin(theta_mid);
    double tds = tan_delta * tan_delta, tus = tan_delta_u1 * tan_delta_u1;
    double v_tan_delta_u1 = v * (tus + 1);
    double vt = 0.5 * a * dt + v;
    A[0 * 6 + 2] = -dt * vt * sn;
    A[0 * 6 + 3] = -0.5 * dt2 * v * vt * (tds + 1) * sn / L;
    A[0 * 6 + 4] = dt * c - 0.5 * dt2 * vt * sn * tan_delta / L;
    A[0 * 6 + 5] = 0.5 * dt2 * c;
    A[1 * 6 + 2] = dt * vt * c;
    A[1 * 6 + 3] = 0.5 * dt2 * v * vt * (tds + 1) * c / L;
    A[1 * 6 + 4] = dt * sn + 0.5 * dt2 * vt * c * tan_delta / L;
    A[1 * 6 + 5] = 0.5 * dt2 * sn;
    A[2 * 6 + 3] = dt * v_tan_delta_u1 / L;
    A[2 * 6 + 4] = dt * tan_delta_u1 / L;
    A[2 * 6 + 5] = 0.5 * dt2 * tan_delta_u1 / L;
    A[4 * 6 + 5] = dt;
    B[2 * 2 + 0] = 0.5 * dt2 * v * (tus + 1) / L;
    B[3 * 2 + 0] = dt;
    B[4 * 2 + 1] = 0.5 * dt2;
    B[5 * 2 + 1] = dt;
    return;
  }
  double k = nd->k;
  if (nd->model == ALILQR_MODEL_DYN) {          /* full_bicycle_dynamic_node.py:61-108 */
    double sf = 1 + k * (v * v);
    double tan_delta = tan(delta);
    double theta_mid = theta + 0.5 * dt * v * tan_delta / (L * sf);
    double tan_delta_u1 = tan(delta + 0.5 * dt * u1);
    double c = cos(theta_mid), sn = sin(theta_mid);
    double tds = tan_delta * tan_delta, tus = tan_delta_u1 * tan_delta_u1;
    double v_tan_delta_u1 = v * (tus + 1);
    double vt = 0.5 * a * dt + v;
    double g = -dt * k * (v * v) * tan_delta / (L * (sf * sf)) + 0.5 * dt * tan_delta / (L * sf);
    A[0 * 6 + 2] = -dt * vt * sn;
    A[0 * 6 + 3] = -0.5 * dt2 * v * vt * (tds + 1) * sn / (L * sf);
    A[0 * 6 + 4] = dt * c - dt * vt * g * sn;
    A[0 * 6 + 5] = 0.5 * dt2 * c;
    A[1 * 6 + 2] = dt * vt * c;
    A[1 * 6 + 3] = 0.5 * dt2 * v * vt * (tds + 1) * c / (L * sf);
    A[1 * 6 + 4] = dt * sn + dt * vt * g * c;
    A[1 * 6 + 5] = 0.5 * dt2 * sn;
    A[2 * 6 + 3] = dt * v_tan_delta_u1 / (L * sf);
    A[2 * 6 + 4] = dt * tan_delta_u1 / (L * sf) - dt * k * vt * (a * dt + 2 * v) * tan_delta_u1 / (L * (sf * sf));
    A[2 * 6 + 5] = 0.5 * dt2 * tan_delta_u1 / (L * sf) - 0.5 * dt2 * k * (vt * vt) * tan_delta_u1 / (L * (sf * sf));
    A[4 * 6 + 5] = dt;
    B[2 * 2 + 0] = 0.5 * dt2 * v_tan_delta_u1 / (L * sf);
    B[3 * 2 + 0] = dt;
    B[4 * 2 + 1] = 0.5 * dt2;
    B[5 * 2 + 1] = dt;
    return;
  }
  /* DYN_MID: fast_bicycle_node.py:56-139 */
  {
    double tan_delta = tan(delta);
    double theta_mid = theta + 0.5 * dt * v * tan_delta / (L * (k * (v * v) + 1));
    double v_term = 0.5 * a * dt + v;
    double tan_delta_mid = tan(delta + 0.5 * dt * u1);
    double k_v_sq = k * (v * v);
    double k_v_mid_sq = k * (v_term * v_term);
    double denom = L * (k_v_sq + 1);
    double denom_mid = L * (k_v_mid_sq + 1);
    double c = cos(theta_mid), sn = sin(theta_mid);
    double g = -dt * k_v_sq * tan_delta / (denom * denom) + 0.5 * dt * tan_delta / denom;
    double tms = tan_delta_mid * tan_delta_mid;
    A[0 * 6 + 2] = -dt * v_term * sn;
    A[0 * 6 + 3] = -0.5 * dt2 * v * v_term * (tan_delta * tan_delta + 1) * sn / denom;
    A[0 * 6 + 4] = -dt * v_term * g * sn + dt * c;
    A[0 * 6 + 5] = 0.5 * dt2 * c;
    A[1 * 6 + 2] = dt * v_term * c;
    A[1 * 6 + 3] = 0.5 * dt2 * v * v_term * (tan_delta * tan_delta + 1) * c / denom;
    A[1 * 6 + 4] = dt * v_term * g * c + dt * sn;
    A[1 * 6 + 5] = 0.5 * dt2 * sn;
    A[2 * 6 + 3] = dt * v_term * (tms + 1) / denom_mid;
    A[2 * 6 + 4] = -dt * k_v_mid_sq * (1.0 * a * dt + 2 * v) * tan_delta_mid / (denom_mid * denom_mid) + dt * tan_delta_mid / denom_mid;
    A[2 * 6 + 5] = -dt2 * k_v_mid_sq * tan_delta_mid / (denom_mid * denom_mid) + 0.5 * dt2 * tan_delta_mid / denom_mid;
    A[4 * 6 + 5] = dt;
    B[2 * 2 + 0] = 0.5 * dt2 * v_term * (tms + 1) / denom_mid;
    B[3 * 2 + 0] = dt;
    B[4 * 2 + 1] = 0.5 * dt2;
    B[5 * 2 + 1] = dt;
  }
}

/* ------------------------------------------------------------------------ */
/* legacy node: box constraints folded into the cost with the I_mu mask       */
/* lat_bicycle_node.py:93-180 (kin :107-194, dyn :110-197)                    */
/* ------------------------------------------------------------------------ */
static void legacy_constraints(or_node* nd, double* c) {   /* :127-133 */
  normalize_state(nd);
  int n = nd->n, m = nd->m;
  for (int i = 0; i < n; ++i) c[i] = nd->state[i] - nd->smax[i];
  for (int i = 0; i < n; ++i) c[n + i] = nd->smin[i] - nd->state[i];
  if (nd->model == ALILQR_MODEL_LAT_INTERIOR) return;   /* state rows only: lat_bicycle_interior_node.py:125-130 */
  for (int j = 0; j < m; ++j) c[2 * n + j] = nd->control[j] - nd->cmax[j];
  for (int j = 0; j < m; ++j) c[2 * n + m + j] = nd->cmin[j] - nd->control[j];
}
static void legacy_update_Imu(or_node* nd) {               /* :174-180 */
  double c[OR_NC];
  legacy_constraints(nd, c);
  for (int i = 0; i < nd->c; ++i)
    nd->Imu[i] = (nd->lam[i] == 0.0 && c[i] <= 0.0) ? 0.0 : nd->mu;
}
static double quad_form(int n, const double* e, const double* Q) {   /* e.T @ Q @ e */
  double tmp[OR_NX], s = 0.0;
  for (int j = 0; j < n; ++j) {
    double t = 0.0;
    for (int i = 0; i < n; ++i) t += e[i] * Q[i * n + j];
    tmp[j] = t;
  }
  for (int j = 0; j < n; ++j) s += tmp[j] * e[j];
  return s;
}
static double legacy_cost(or_node* nd) {                   /* :93-102 */
  double e[OR_NX], c[OR_NC];
  normalize_state(nd);
  /* lat_bicycle_interior_node.py:96 weighs the state itself, the other nodes the distance to the goal */
  for (int i = 0; i < nd->n; ++i) e[i] = (nd->model == ALILQR_MODEL_LAT_INTERIOR) ? nd->state[i] : nd->state[i] - nd->goal[i];
  double state_cost = quad_form(nd->n, e, nd->Q);
  double control_cost = quad_form(nd->m, nd->control, nd->R);
  legacy_update_Imu(nd);
  legacy_constraints(nd, c);
  double lc = 0.0, qc = 0.0;
  for (int i = 0; i < nd->c; ++i) lc += nd->lam[i] * c[i];
  /* lat/kin: 0.5*mu*c'Imu c (lat :101, kin :115) ; dyn: 0.5*c'Imu c (dyn :118) */
  double half = (nd->model == ALILQR_MODEL_DYN) ? 0.5 : 0.5 * nd->mu;
  for (int i = 0; i < nd->c; ++i) qc += ((half * c[i]) * nd->Imu[i]) * c[i];
  return state_cost + control_cost + (lc + qc);
}
static void legacy_cost_jacobian(or_node* nd, double* Jx, double* Ju) {   /* :104-115 */
  double e[OR_NX], c[OR_NC], w[OR_NC];
  int n = nd->n, m = nd->m;
  normalize_state(nd);
  for (int i = 0; i < n; ++i) e[i] = (nd->model == ALILQR_MODEL_LAT_INTERIOR) ? nd->state[i] : nd->state[i] - nd->goal[i];
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += (2 * nd->Q[i * n + j]) * e[j];
    Jx[i] = s;
  }
  for (int i = 0; i < m; ++i) {
    double s = 0.0;
    for (int j = 0; j < m; ++j) s += (2 * nd->R[i * m + j]) * nd->control[j];
    Ju[i] = s;
  }
  legacy_update_Imu(nd);
  legacy_constraints(nd, c);
  for (int i = 0; i < nd->c; ++i) w[i] = nd->lam[i] + nd->Imu[i] * c[i];
  /* C_x = [I; -I; 0; 0], C_u = [0; 0; I; -I]: the dense products of :112-113
   * reduce to two non-zero terms each, so this is the same arithmetic */
  for (int i = 0; i < n; ++i) Jx[i] = Jx[i] + (w[i] - w[n + i]);
  if (nd->model == ALILQR_MODEL_LAT_INTERIOR) {        /* C_u = 0 (:141): the product adds zeros */
    for (int j = 0; j < m; ++j) Ju[j] = Ju[j] + 0.0;
    return;
  }
  for (int j = 0; j < m; ++j) Ju[j] = Ju[j] + (w[2 * n + j] - w[2 * n + m + j]);
}
static void legacy_cost_hessian(or_node* nd, double* Hx, double* Hu) {    /* :117-125 */
  int n = nd->n, m = nd->m;
  for (int i = 0; i < n * n; ++i) Hx[i] = 2 * nd->Q[i];
  for (int i = 0; i < m * m; ++i) Hu[i] = 2 * nd->R[i];
  legacy_update_Imu(nd);
  for (int i = 0; i < n; ++i) Hx[i * n + i] = Hx[i * n + i] + (nd->Imu[i] + nd->Imu[n + i]);
  if (nd->model == ALILQR_MODEL_LAT_INTERIOR) return;
  for (int j = 0; j < m; ++j) Hu[j * m + j] = Hu[j * m + j] + (nd->Imu[2 * n + j] + nd->Imu[2 * n + m + j]);
}
static void legacy_update_lambda(or_node* nd) {            /* :162-165 */
  double c[OR_NC];
  legacy_constraints(nd, c);
  for (int i = 0; i < nd->c; ++i) nd->lam[i] = fmax(0.0, nd->lam[i] + nd->mu * c[i]);
}

/* ------------------------------------------------------------------------ */
/* fast node: projection-form AL over a linear table                          */
/* constraints.py:47-119, linear_constraints.py:13-29, fast_bicycle_node.py:183-212 */
/* ------------------------------------------------------------------------ */
static void lin_constrains(or_node* nd, const double* x, const double* u) {  /* linear_constraints.py:13-16 */
  for (int i = 0; i < nd->c; ++i) {
    double ax = 0.0, bu = 0.0;
    for (int j = 0; j < nd->n; ++j) ax += nd->A[i * nd->n + j] * x[j];
    for (int j = 0; j < nd->m; ++j) bu += nd->B[i * nd->m + j] * u[j];
    nd->cc[i] = (ax + bu) + nd->C[i];
  }
}
static double fast_al_cost(or_node* nd) {                  /* constraints.py:59-65 */
  lin_constrains(nd, nd->state, nd->control);
  double pp = 0.0, ll = 0.0;
  for (int i = 0; i < nd->c; ++i) {
    double f = nd->lam[i] - nd->mu * nd->cc[i];
    double lp = nd->eq ? f : fmin(f, 0.0);       /* constraints.py:61-62 */
    pp += lp * lp;
  }
  for (int i = 0; i < nd->c; ++i) ll += nd->lam[i] * nd->lam[i];
  return 0.5 / nd->mu * (pp - ll);
}
static double fast_cost(or_node* nd) {                     /* fast_bicycle_node.py:183-191 */
  double e[OR_NX];
  normalize_state(nd);
  for (int i = 0; i < nd->n; ++i) e[i] = nd->state[i] - nd->goal[i];
  double state_cost = quad_form(nd->n, e, nd->Q);
  double control_cost = quad_form(nd->m, nd->control, nd->R);
  double cons_cost = fast_al_cost(nd);
  return state_cost + control_cost + cons_cost;
}
static void fast_cost_jacobian(or_node* nd, double* Jx, double* Ju) {  /* fast_bicycle_node.py:193-203 */
  double e[OR_NX];
  int n = nd->n, m = nd->m;
  normalize_state(nd);
  for (int i = 0; i < n; ++i) e[i] = nd->state[i] - nd->goal[i];
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += (2 * nd->Q[i * n + j]) * e[j];
    Jx[i] = s;
  }
  for (int i = 0; i < m; ++i) {
    double s = 0.0;
    for (int j = 0; j < m; ++j) s += (2 * nd->R[i * m + j]) * nd->control[j];
    Ju[i] = s;
  }
  /* constraints.py:67-80 */
  lin_constrains(nd, nd->state, nd->control);
  double lp[OR_NC], pj[OR_NC];
  for (int i = 0; i < nd->c; ++i) {
    double f = nd->lam[i] - nd->mu * nd->cc[i];
    lp[i] = nd->eq ? f : fmin(f, 0.0);            /* constraints.py:70-73 */
    pj[i] = nd->eq ? 1.0 : ((f > 0.0) ? 0.0 : 1.0);
  }
  for (int j = 0; j < n; ++j) {
    double s = 0.0;
    for (int i = 0; i < nd->c; ++i) s += (pj[i] * nd->A[i * n + j]) * lp[i];
    Jx[j] = Jx[j] + (-s);
  }
  for (int j = 0; j < m; ++j) {
    double s = 0.0;
    for (int i = 0; i < nd->c; ++i) s += (pj[i] * nd->B[i * m + j]) * lp[i];
    Ju[j] = Ju[j] + (-s);
  }
}
static void fast_cost_hessian(or_node* nd, double* Hx, double* Hu) {   /* fast_bicycle_node.py:205-212 */
  int n = nd->n, m = nd->m;
  for (int i = 0; i < n * n; ++i) Hx[i] = 2 * nd->Q[i];
  for (int i = 0; i < m * m; ++i) Hu[i] = 2 * nd->R[i];
  /* constraints.py:82-104 (the constraint Hessians of a linear table are zero) */
  lin_constrains(nd, nd->state, nd->control);
  double pj[OR_NC];
  for (int i = 0; i < nd->c; ++i) {
    double f = nd->lam[i] - nd->mu * nd->cc[i];
    pj[i] = nd->eq ? 1.0 : ((f > 0.0) ? 0.0 : 1.0);   /* constraints.py:91-94 */
  }
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b) {
      double s = 0.0;
      for (int i = 0; i < nd->c; ++i) s += (pj[i] * nd->A[i * n + a]) * (pj[i] * nd->A[i * n + b]);
      Hx[a * n + b] += nd->mu * s - 0.0;
    }
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b) {
      double s = 0.0;
      for (int i = 0; i < nd->c; ++i) s += (pj[i] * nd->B[i * m + a]) * (pj[i] * nd->B[i * m + b]);
      Hu[a * m + b] += nd->mu * s - 0.0;
    }
}
static double fast_max_violation(or_node* nd) {            /* constraints.py:115-119 */
  lin_constrains(nd, nd->state, nd->control);
  double v = 0.0;
  for (int i = 0; i < nd->c; ++i) {
    double dc = nd->cc[i] - fmin(nd->cc[i], 0.0);
    if (fabs(dc) > v) v = fabs(dc);
    if (isnan(dc)) v = dc;
  }
  return v;
}
static void fast_update_lambda(or_node* nd) {              /* constraints.py:106-110: uses the cached c */
  for (int i = 0; i < nd->c; ++i) {
    double f = nd->lam[i] - nd->mu * nd->cc[i];
    nd->lam[i] = nd->eq ? f : fmin(f, 0.0);
  }
}

/* ------------------------------------------------------------------------ */
/* node dispatch                                                              */
/* ------------------------------------------------------------------------ */
static double node_cost(or_node* nd) { return nd->mode == ALILQR_MODE_LEGACY ? legacy_cost(nd) : fast_cost(nd); }
static void node_cost_jacobian(or_node* nd, double* Jx, double* Ju) {
  if (nd->mode == ALILQR_MODE_LEGACY) legacy_cost_jacobian(nd, Jx, Ju); else fast_cost_jacobian(nd, Jx, Ju);
}
static void node_cost_hessian(or_node* nd, double* Hx, double* Hu) {
  if (nd->mode == ALILQR_MODE_LEGACY) legacy_cost_hessian(nd, Hx, Hu); else fast_cost_hessian(nd, Hx, Hu);
}

void or_node_init(or_node* nd, const alilqr_problem* p, int t,
                  const double* goals, const double* Q, const double* R, const double* cons) {
  memset(nd, 0, sizeof(*nd));
  int n = p->n, m = p->m, c = p->c;
  int tt = (p->table_nodes > 1) ? t : 0;
  nd->model = p->model; nd->mode = p->mode; nd->n = n; nd->m = m; nd->c = c;
  nd->eq = (p->cons_kind & ALILQR_CONS_EQUALITY) ? 1 : 0;
  nd->L = p->L; nd->dt = p->dt; nd->v = p->v; nd->k = p->k;
  nd->Q = Q + (size_t)tt * n * n;
  nd->R = R + (size_t)tt * m * m;
  memcpy(nd->goal, goals + (size_t)t * n, sizeof(double) * n);
  if (p->mode == ALILQR_MODE_LEGACY) {
    const double* b = cons + (size_t)tt * (2 * n + 2 * m);
    memcpy(nd->smin, b, sizeof(double) * n);
    memcpy(nd->smax, b + n, sizeof(double) * n);
    memcpy(nd->cmin, b + 2 * n, sizeof(double) * m);
    memcpy(nd->cmax, b + 2 * n + m, sizeof(double) * m);
  } else {
    const double* b = cons + (size_t)tt * (c * n + c * m + c);
    nd->A = b; nd->B = b + c * n; nd->C = b + c * n + c * m;
  }
  nd->mu = 1.0;
}

/* ------------------------------------------------------------------------ */
/* solver: cilqr/ilqr.py and cilqr/fast_ilqr.py                               */
/* ------------------------------------------------------------------------ */
typedef struct {
  const alilqr_problem* p;
  or_node* nodes;
  int H, n, m;
  long n_inner, n_trials;
  oracle_trace* trace;
} py_solver;

static double compute_total_cost(py_solver* s) {           /* ilqr.py:52-56 */
  double total = 0.0;
  for (int t = 0; t <= s->H; ++t) total += node_cost(&s->nodes[t]);
  return total;
}

_Thread_local const double* or_u_offset = NULL;
_Thread_local const alilqr_extras* or_extras = NULL;
_Thread_local double or_min_margin = 1.0;
_Thread_local int or_problem = 0;
static double* or_margin_out = NULL;      /* [B], set by oracle_set_margin_out: where batch solves record their margins */

/* Ask the following oracle_solve_batch* calls to record every solve's decision margin in out[B] (NULL: stop). */
void oracle_set_margin_out(double* out) { or_margin_out = out; }

/* ilqr.py:11-50 / fast_ilqr.py:10-49 */
static void linearized_initial_guess(py_solver* s, double* x, double* u) {
  int H = s->H, n = s->n, m = s->m;
  double P[OR_NX * OR_NX], A[OR_NX * OR_NX], B[OR_NX * OR_NU];
  double* Kl = (double*)malloc(sizeof(double) * H * m * n);
  double zero_u[OR_NU] = {0};
  memset(x, 0, sizeof(double) * (H + 1) * n);
  memset(u, 0, sizeof(double) * H * m);
  memcpy(x, s->nodes[0].state, sizeof(double) * n);
  memset(P, 0, sizeof(P));
  for (int i = 0; i < n; ++i) P[i * n + i] = 1.0;
  for (int t = H - 1; t >= 0; --t) {
    or_node* nd = &s->nodes[t];
    or_py_jacobian(nd, nd->goal, zero_u, A, B);
    double BtP[OR_NU * OR_NX], BtPB[OR_NU * OR_NU], BtPA[OR_NU * OR_NX], M[OR_NU * OR_NU], Mi[OR_NU * OR_NU];
    om_mtm(m, n, n, B, P, BtP);
    om_mm(m, n, m, BtP, B, BtPB);
    om_mm(m, n, n, BtP, A, BtPA);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) M[i * m + j] = ((i == j) ? 1.0 : 0.0) * s->p->prm.init_r + BtPB[i * m + j];
    om_inv(m, M, Mi);
    double* K = Kl + (size_t)t * m * n;
    om_mm(m, m, n, Mi, BtPA, K);
    double BK[OR_NX * OR_NX], AmBK[OR_NX * OR_NX], AtP[OR_NX * OR_NX], T[OR_NX * OR_NX];
    om_mm(n, m, n, B, K, BK);
    for (int i = 0; i < n * n; ++i) AmBK[i] = A[i] - BK[i];
    om_mtm(n, n, n, A, P, AtP);
    om_mm(n, n, n, AtP, AmBK, T);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) P[i * n + j] = ((i == j) ? 1.0 : 0.0) + T[i * n + j];
  }
  for (int t = 0; t < H; ++t) {
    or_node* nd = &s->nodes[t];
    const double* K = Kl + (size_t)t * m * n;
    double d[OR_NX];
    for (int i = 0; i < n; ++i) d[i] = x[t * n + i] - nd->goal[i];
    for (int j = 0; j < m; ++j) {
      double acc = 0.0;
      for (int i = 0; i < n; ++i) acc += (-K[j * n + i]) * d[i];
      if (or_u_offset) acc = acc + or_u_offset[t * m + j];
      u[t * m + j] = acc;
    }
    or_py_dynamics(nd, x + t * n, u + t * m, x + (t + 1) * n);
    memcpy(nd->control, u + t * m, sizeof(double) * m);
    memcpy(s->nodes[t + 1].state, x + (t + 1) * n, sizeof(double) * n);
  }
  for (int t = 0; t <= H; ++t) {
    memset(s->nodes[t].lam, 0, sizeof(s->nodes[t].lam));
    s->nodes[t].mu = 1.0;
  }
  free(Kl);
}

/* ilqr.py:58-100 / fast_ilqr.py:57-99 */
static void backward(py_solver* s, double* kff, double* Kfb) {
  int H = s->H, n = s->n, m = s->m;
  size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
  double* A = (double*)calloc((size_t)H * nn, sizeof(double));
  double* B = (double*)calloc((size_t)H * nm, sizeof(double));
  double* Jx = (double*)calloc((size_t)(H + 1) * n, sizeof(double));
  double* Ju = (double*)calloc((size_t)H * m, sizeof(double));
  double* Hx = (double*)calloc((size_t)(H + 1) * nn, sizeof(double));
  double* Hu = (double*)calloc((size_t)H * mm, sizeof(double));
  for (int t = 0; t < H; ++t) {
    or_node* nd = &s->nodes[t];
    or_py_jacobian(nd, nd->state, nd->control, A + t * nn, B + t * nm);
    node_cost_jacobian(nd, Jx + t * n, Ju + t * m);
    node_cost_hessian(nd, Hx + t * nn, Hu + t * mm);
  }
  {
    double dumJu[OR_NU], dumHu[OR_NU * OR_NU];
    node_cost_jacobian(&s->nodes[H], Jx + (size_t)H * n, dumJu);
    node_cost_hessian(&s->nodes[H], Hx + (size_t)H * nn, dumHu);
  }
  double Vx[OR_NX], Vxx[OR_NX * OR_NX];
  memcpy(Vx, Jx + (size_t)H * n, sizeof(double) * n);
  memcpy(Vxx, Hx + (size_t)H * nn, sizeof(double) * nn);
  double reg = s->p->prm.quu_reg;
  for (int t = H - 1; t >= 0; --t) {
    const double *At = A + t * nn, *Bt = B + t * nm;
    double Qu[OR_NU], Qx[OR_NX], Qux[OR_NU * OR_NX], Quu[OR_NU * OR_NU], Qxx[OR_NX * OR_NX];
    double BtV[OR_NU * OR_NX], AtV[OR_NX * OR_NX], tmp[OR_NX * OR_NX];
    om_mtm(m, n, 1, Bt, Vx, Qu);
    for (int j = 0; j < m; ++j) Qu[j] = Ju[t * m + j] + Qu[j];
    om_mtm(n, n, 1, At, Vx, Qx);
    for (int i = 0; i < n; ++i) Qx[i] = Jx[t * n + i] + Qx[i];
    om_mtm(m, n, n, Bt, Vxx, BtV);
    om_mm(m, n, n, BtV, At, Qux);
    om_mm(m, n, m, BtV, Bt, Quu);
    for (size_t i = 0; i < mm; ++i) Quu[i] = Hu[t * mm + i] + Quu[i];
    om_mtm(n, n, n, At, Vxx, AtV);
    om_mm(n, n, n, AtV, At, tmp);
    for (size_t i = 0; i < nn; ++i) Qxx[i] = Hx[t * nn + i] + tmp[i];

    double Qr[OR_NU * OR_NU], Qi[OR_NU * OR_NU];
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) Qr[i * m + j] = Quu[i * m + j] + ((i == j) ? 1.0 : 0.0) * reg;
    om_inv(m, Qr, Qi);
    double* K = Kfb + (size_t)t * m * n;
    double* k = kff + (size_t)t * m;
    for (int i = 0; i < m; ++i) {
      for (int j = 0; j < n; ++j) {
        double acc = 0.0;
        for (int q = 0; q < m; ++q) acc += (-Qi[i * m + q]) * Qux[q * n + j];
        K[i * n + j] = acc;
      }
      double acc = 0.0;
      for (int q = 0; q < m; ++q) acc += (-Qi[i * m + q]) * Qu[q];
      k[i] = acc;
    }
    /* Vx = Qx + K'Quu k + K'Qu + Qux'k ; Vxx = Qxx + K'Quu K + K'Qux + Qux'K */
    double KtQuu[OR_NX * OR_NU], t1[OR_NX], t2[OR_NX], t3[OR_NX];
    om_mtm(n, m, m, K, Quu, KtQuu);
    om_mm(n, m, 1, KtQuu, k, t1);
    om_mtm(n, m, 1, K, Qu, t2);
    om_mtm(n, m, 1, Qux, k, t3);
    for (int i = 0; i < n; ++i) Vx[i] = ((Qx[i] + t1[i]) + t2[i]) + t3[i];
    double m1[OR_NX * OR_NX], m2[OR_NX * OR_NX], m3[OR_NX * OR_NX];
    om_mm(n, m, n, KtQuu, K, m1);
    om_mtm(n, m, n, K, Qux, m2);
    om_mtm(n, m, n, Qux, K, m3);
    for (size_t i = 0; i < nn; ++i) Vxx[i] = ((Qxx[i] + m1[i]) + m2[i]) + m3[i];
  }
  free(A); free(B); free(Jx); free(Ju); free(Hx); free(Hu);
}

/* ilqr.py:102-149 (with its array aliasing after the first rejected step)
 * fast_ilqr.py:101-134 (clean back-tracking).
 * On return *px / *pu point at whichever buffer the reference would return. */
static void forward(py_solver* s, const double* kff, const double* Kfb,
                    double* x, double* u, double* new_x, double* new_u,
                    double** px, double** pu) {
  int H = s->H, n = s->n, m = s->m;
  int legacy = (s->p->mode == ALILQR_MODE_LEGACY);
  double alpha = 1.0;
  for (int t = 0; t <= H; ++t) memcpy(x + t * n, s->nodes[t].state, sizeof(double) * n);
  for (int t = 0; t < H; ++t) memcpy(u + t * m, s->nodes[t].control, sizeof(double) * m);
  memset(new_x, 0, sizeof(double) * (H + 1) * n);
  memset(new_u, 0, sizeof(double) * H * m);
  memcpy(new_x, x, sizeof(double) * n);
  double old_cost = compute_total_cost(s);
  double *nx = new_x, *nu = new_u;
  int trials = 0;
  while (alpha > s->p->prm.alpha_min) {
    for (int t = 0; t < H; ++t) {
      const double* K = Kfb + (size_t)t * m * n;
      double d[OR_NX], un[OR_NU], xn[OR_NX];
      for (int i = 0; i < n; ++i) d[i] = nx[t * n + i] - x[t * n + i];
      for (int j = 0; j < m; ++j) {
        double fb = 0.0;
        for (int i = 0; i < n; ++i) fb += K[j * n + i] * d[i];
        un[j] = (u[t * m + j] + alpha * kff[t * m + j]) + fb;
      }
      memcpy(nu + t * m, un, sizeof(double) * m);
      or_py_dynamics(&s->nodes[t], nx + t * n, nu + t * m, xn);
      memcpy(nx + (t + 1) * n, xn, sizeof(double) * n);
      memcpy(s->nodes[t + 1].state, nx + (t + 1) * n, sizeof(double) * n);
      memcpy(s->nodes[t].control, nu + t * m, sizeof(double) * m);
    }
    double new_cost = compute_total_cost(s);
    ++trials;
    or_decision(new_cost, old_cost);
    if (new_cost < old_cost) break;
    alpha = alpha / 2.0;
    if (legacy) { nu = u; nx = x; }          /* ilqr.py:137-138: new_u = u ; new_x = x */
  }
  if (alpha <= s->p->prm.alpha_min) {
    nx = x; nu = u;
    for (int t = 0; t < H; ++t) {
      memcpy(s->nodes[t + 1].state, nx + (t + 1) * n, sizeof(double) * n);
      memcpy(s->nodes[t].control, nu + t * m, sizeof(double) * m);
    }
  }
  s->n_trials += trials;
  if (s->trace && s->trace->length < s->trace->capacity) s->trace->trials[s->trace->length] = trials;
  *px = nx; *pu = nu;
}

static double compute_violation(py_solver* s) {
  if (s->p->mode == ALILQR_MODE_LEGACY) {                  /* ilqr.py:186-192 */
    double violation = 0.0;
    for (int t = 0; t <= s->H; ++t) {
      or_node* nd = &s->nodes[t];
      double c[OR_NC], ss = 0.0;
      legacy_constraints(nd, c);
      for (int i = 0; i < nd->c; ++i) {
        double one = fmax(-nd->lam[i] / nd->mu, c[i]);
        ss += one * one;
      }
      violation += sqrt(ss);
    }
    return violation;
  }
  double violation = -1.0;                                 /* fast_ilqr.py:171-176 */
  for (int t = 0; t <= s->H; ++t) {
    double one = fast_max_violation(&s->nodes[t]);
    violation = (isnan(one) || one > violation) ? one : violation;
  }
  return violation;
}

static int py_optimize(const alilqr_problem* p, or_node* nodes,
                       double* x_out, double* u_out, double* x_init_o, double* u_init_o,
                       double* viol_o, int32_t* status, oracle_trace* trace) {
  int H = p->horizon, n = p->n, m = p->m;
  py_solver s = {p, nodes, H, n, m, 0, 0, trace};
  size_t xs = (size_t)(H + 1) * n, us = (size_t)H * m;
  double* x_init = (double*)malloc(sizeof(double) * xs);
  double* u_init = (double*)malloc(sizeof(double) * us);
  double* kff = (double*)malloc(sizeof(double) * us);
  double* Kfb = (double*)malloc(sizeof(double) * us * n);
  /* every forward() call of the reference makes fresh arrays; the caller keeps at
   * most the pair of the previous accepted call, so two generations suffice */
  double* buf[2][4];
  for (int g = 0; g < 2; ++g) {
    buf[g][0] = (double*)malloc(sizeof(double) * xs); buf[g][1] = (double*)malloc(sizeof(double) * us);
    buf[g][2] = (double*)malloc(sizeof(double) * xs); buf[g][3] = (double*)malloc(sizeof(double) * us);
  }
  linearized_initial_guess(&s, x_init, u_init);
  double old_cost = compute_total_cost(&s);
  const double *x = x_init, *u = u_init;
  int gen = 0, n_outer = 0;
  uint32_t flags = 0;
  double violation = 0.0;
  if (trace) trace->length = 0;
  const int stop_k = (or_extras && or_extras->stop_after) ? or_extras->stop_after[or_problem] : 0;
  int stopped = 0;
  for (int j = 0; j < p->prm.max_outer; ++j) {
    old_cost = compute_total_cost(&s);
    for (int i = 0; i < p->prm.max_inner; ++i) {
      backward(&s, kff, Kfb);
      ++s.n_inner;
      /* do not overwrite the generation the caller's (x,u) may still point into */
      int g = gen;
      if (x == buf[g][0] || x == buf[g][2]) g = 1 - g;
      double *nx, *nu;
      forward(&s, kff, Kfb, buf[g][0], buf[g][1], buf[g][2], buf[g][3], &nx, &nu);
      double new_cost = compute_total_cost(&s);
      if (trace && trace->length < trace->capacity) trace->iter_cost[trace->length++] = new_cost;
      {
        /* |dJ| against the tolerance, on the scale of J (that is where the rounding of dJ lives) */
        double sc = fmax(fabs(new_cost), fabs(old_cost));
        or_decision(sc + (fabs(new_cost - old_cost) - p->prm.inner_tol), sc);
      }
      const int brk = fabs(new_cost - old_cost) < p->prm.inner_tol;
      if (!brk) {
        x = nx; u = nu; gen = g;
        old_cost = new_cost;
      }
      if (stop_k > 0 && s.n_inner >= stop_k) { stopped = 1; break; }   /* alilqr_extras::stop_after */
      if (brk) break;
    }
    if (stopped) { flags |= ALILQR_FLAG_STOPPED; break; }
    violation = compute_violation(&s);
    ++n_outer;
    or_decision(violation, p->prm.viol_stop);
    or_decision(violation, p->prm.viol_lambda);
    if (violation < p->prm.viol_stop) { flags |= ALILQR_FLAG_VIOL_OK; break; }
    else if (violation >= p->prm.viol_stop && violation < p->prm.viol_lambda) {
      for (int t = 0; t <= H; ++t) {
        if (p->mode == ALILQR_MODE_LEGACY) legacy_update_lambda(&nodes[t]); else fast_update_lambda(&nodes[t]);
      }
    } else {
      for (int t = 0; t <= H; ++t) nodes[t].mu = nodes[t].mu * p->prm.mu_gain;
    }
  }
  if (!(flags & (ALILQR_FLAG_VIOL_OK | ALILQR_FLAG_STOPPED))) flags |= ALILQR_FLAG_OUTER_CAP;
  if (x_out) memcpy(x_out, x, sizeof(double) * xs);
  if (u_out) memcpy(u_out, u, sizeof(double) * us);
  if (x_init_o) memcpy(x_init_o, x_init, sizeof(double) * xs);
  if (u_init_o) memcpy(u_init_o, u_init, sizeof(double) * us);
  if (viol_o) *viol_o = violation;
  if (status) { status[0] = (int32_t)s.n_inner; status[1] = n_outer; status[2] = (int32_t)flags; status[3] = (int32_t)s.n_trials; }
  for (int g = 0; g < 2; ++g) for (int q = 0; q < 4; ++q) free(buf[g][q]);
  free(x_init); free(u_init); free(kff); free(Kfb);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* public entry points                                                        */
/* ------------------------------------------------------------------------ */
int32_t oracle_solve(const alilqr_problem* p,
                     const double* x0, const double* goals,
                     const double* Q, const double* R, const double* cons,
                     double* x_out, double* u_out, double* node_x, double* node_u,
                     double* x_init, double* u_init, double* cost, double* viol,
                     double* lam, double* mu, int32_t* status, oracle_trace* trace) {
  if (!p || !x0 || !goals || !Q || !R || !cons) return ALILQR_E_ARG;
  if (p->n > OR_NX || p->m > OR_NU || p->c > OR_NC || p->horizon < 1) return ALILQR_E_ARG;
  if (p->mode == ALILQR_MODE_NEW_AL)
    return or_new_solve(p, x0, goals, Q, R, cons, x_out, u_out, node_x, node_u, x_init, u_init,
                        cost, viol, lam, mu, status, trace);
  int H = p->horizon, n = p->n, m = p->m;
  or_node* nodes = (or_node*)malloc(sizeof(or_node) * (H + 1));
  for (int t = 0; t <= H; ++t) or_node_init(&nodes[t], p, t, goals, Q, R, cons);
  memcpy(nodes[0].state, x0, sizeof(double) * n);
  py_optimize(p, nodes, x_out, u_out, x_init, u_init, viol, status, trace);
  {
    py_solver s = {p, nodes, H, n, m, 0, 0, NULL};
    double c = compute_total_cost(&s);
    if (cost) *cost = c;
    if (status && !isfinite(c)) status[2] |= ALILQR_FLAG_NONFINITE;
  }
  for (int t = 0; t <= H; ++t) {
    if (node_x) memcpy(node_x + (size_t)t * n, nodes[t].state, sizeof(double) * n);
    if (node_u && t < H) memcpy(node_u + (size_t)t * m, nodes[t].control, sizeof(double) * m);
    if (lam) memcpy(lam + (size_t)t * p->c, nodes[t].lam, sizeof(double) * p->c);
  }
  if (mu) *mu = nodes[0].mu;
  free(nodes);
  return ALILQR_OK;
}

int32_t oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int32_t oracle_solve_batch_ex(const alilqr_problem* p,
                           const double* x0, const double* goals,
                           const double* Q, const double* R, const double* cons, const alilqr_extras* ex,
                           double* x_out, double* u_out, double* node_x, double* node_u,
                           double* x_init, double* u_init, double* cost, double* viol,
                           double* lam, double* mu, int32_t* status, int32_t threads) {
  if (!p) return ALILQR_E_ARG;
  int B = p->batch, H = p->horizon, n = p->n, m = p->m, c = p->c;
  size_t xs = (size_t)(H + 1) * n, us = (size_t)H * m, ls = (size_t)(H + 1) * c;
  int rc = 0;
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
  for (int b = 0; b < B; ++b) {
    const double* g = goals + ((p->goal_batch > 1) ? (size_t)b * xs : 0);
    or_u_offset = (ex && ex->u_offset) ? ex->u_offset + (size_t)b * us : NULL;
    or_extras = ex;
    or_problem = b;
    or_min_margin = 1.0;
    int r = oracle_solve(p, x0 + (size_t)b * n, g, Q, R, cons,
                         x_out ? x_out + b * xs : NULL, u_out ? u_out + b * us : NULL,
                         node_x ? node_x + b * xs : NULL, node_u ? node_u + b * us : NULL,
                         x_init ? x_init + b * xs : NULL, u_init ? u_init + b * us : NULL,
                         cost ? cost + b : NULL, viol ? viol + b : NULL,
                         lam ? lam + b * ls : NULL, mu ? mu + b : NULL,
                         status ? status + (size_t)b * ALILQR_STATUS_WORDS : NULL, NULL);
    or_u_offset = NULL;
    or_extras = NULL;
    or_problem = 0;
    if (or_margin_out) or_margin_out[b] = or_min_margin;
    if (r != 0) rc = r;
  }
  (void)threads;
  return rc;
}

int32_t oracle_solve_batch_multistart(const alilqr_problem* p,
                           const double* x0, const double* goals,
                           const double* Q, const double* R, const double* cons, const double* u_offset,
                           double* x_out, double* u_out, double* node_x, double* node_u,
                           double* x_init, double* u_init, double* cost, double* viol,
                           double* lam, double* mu, int32_t* status, int32_t threads) {
  alilqr_extras ex;
  memset(&ex, 0, sizeof(ex));
  ex.u_offset = u_offset;
  return oracle_solve_batch_ex(p, x0, goals, Q, R, cons, &ex, x_out, u_out, node_x, node_u, x_init, u_init,
                               cost, viol, lam, mu, status, threads);
}

int32_t oracle_solve_batch(const alilqr_problem* p,
                           const double* x0, const double* goals,
                           const double* Q, const double* R, const double* cons,
                           double* x_out, double* u_out, double* node_x, double* node_u,
                           double* x_init, double* u_init, double* cost, double* viol,
                           double* lam, double* mu, int32_t* status, int32_t threads) {
  return oracle_solve_batch_multistart(p, x0, goals, Q, R, cons, NULL, x_out, u_out, node_x, node_u, x_init, u_init,
                                       cost, viol, lam, mu, status, threads);
}

int32_t oracle_linearize_batch(const alilqr_problem* p,
                               const double* x, const double* u,
                               const double* lam, const double* mu,
                               const double* goals, const double* Q, const double* R, const double* cons,
                               double* f, double* fx, double* fu, double* l,
                               double* lx, double* lu, double* lxx, double* luu,
                               double* cval, double* lam_new, double* xnorm) {
  if (!p || !x || !u || !goals || !Q || !R || !cons) return ALILQR_E_ARG;
  if (p->mode == ALILQR_MODE_NEW_AL)
    return or_new_linearize(p, x, u, lam, mu, goals, Q, R, cons, f, fx, fu, l, lx, lu, lxx, luu, cval, lam_new, xnorm);
  int B = p->batch, H = p->horizon, n = p->n, m = p->m, c = p->c;
  size_t xs = (size_t)(H + 1) * n;
  for (int b = 0; b < B; ++b) {
    const double* g = goals + ((p->goal_batch > 1) ? (size_t)b * xs : 0);
    for (int t = 0; t <= H; ++t) {
      size_t i = (size_t)b * (H + 1) + t;
      or_node nd;
      or_node_init(&nd, p, t, g, Q, R, cons);
      double A[OR_NX * OR_NX], Bm[OR_NX * OR_NU], nxt[OR_NX], jx[OR_NX], ju[OR_NU], hx[OR_NX * OR_NX], hu[OR_NU * OR_NU];
      or_py_dynamics(&nd, x + i * n, u + i * m, nxt);
      or_py_jacobian(&nd, x + i * n, u + i * m, A, Bm);
      memcpy(nd.state, x + i * n, sizeof(double) * n);
      memcpy(nd.control, u + i * m, sizeof(double) * m);
      if (lam) memcpy(nd.lam, lam + i * c, sizeof(double) * c);
      if (mu) nd.mu = mu[b];
      double cost = node_cost(&nd);
      node_cost_jacobian(&nd, jx, ju);
      node_cost_hessian(&nd, hx, hu);
      if (f) memcpy(f + i * n, nxt, sizeof(double) * n);
      if (fx) memcpy(fx + i * n * n, A, sizeof(double) * n * n);
      if (fu) memcpy(fu + i * n * m, Bm, sizeof(double) * n * m);
      if (l) l[i] = cost;
      if (lx) memcpy(lx + i * n, jx, sizeof(double) * n);
      if (lu) memcpy(lu + i * m, ju, sizeof(double) * m);
      if (lxx) memcpy(lxx + i * n * n, hx, sizeof(double) * n * n);
      if (luu) memcpy(luu + i * m * m, hu, sizeof(double) * m * m);
      if (xnorm) memcpy(xnorm + i * n, nd.state, sizeof(double) * n);
      if (cval) {
        if (p->mode == ALILQR_MODE_LEGACY) legacy_constraints(&nd, cval + i * c);
        else { lin_constrains(&nd, nd.state, nd.control); memcpy(cval + i * c, nd.cc, sizeof(double) * c); }
      }
      if (lam_new) {
        if (p->mode == ALILQR_MODE_LEGACY) legacy_update_lambda(&nd);
        else { (void)fast_max_violation(&nd); fast_update_lambda(&nd); }
        memcpy(lam_new + i * c, nd.lam, sizeof(double) * c);
      }
    }
  }
  return ALILQR_OK;
}

int32_t oracle_node_aux_batch(const alilqr_problem* p,
                              const double* x, const double* u,
                              const double* lam, const double* mu,
                              const double* goals, const double* Q, const double* R, const double* cons,
                              double* viol, double* par_cost, double* fxx) {
  if (!p || !x || !u || !goals || !Q || !R || !cons) return ALILQR_E_ARG;
  if (p->mode == ALILQR_MODE_NEW_AL) return or_new_node_aux(p, x, u, lam, mu, goals, Q, R, cons, viol, par_cost, fxx);
  if (par_cost || fxx) return ALILQR_E_ARG;
  int B = p->batch, H = p->horizon, n = p->n, m = p->m, c = p->c;
  size_t xs = (size_t)(H + 1) * n;
  for (int b = 0; b < B; ++b) {
    const double* g = goals + ((p->goal_batch > 1) ? (size_t)b * xs : 0);
    for (int t = 0; t <= H; ++t) {
      size_t i = (size_t)b * (H + 1) + t;
      or_node nd;
      or_node_init(&nd, p, t, g, Q, R, cons);
      memcpy(nd.state, x + i * n, sizeof(double) * n);
      memcpy(nd.control, u + i * m, sizeof(double) * m);
      if (lam) memcpy(nd.lam, lam + i * c, sizeof(double) * c);
      if (mu) nd.mu = mu[b];
      if (!viol) continue;
      if (p->mode == ALILQR_MODE_LEGACY) {                  /* one node's term of ilqr.py:186-192 */
        double cv[OR_NC], ss = 0.0;
        legacy_constraints(&nd, cv);
        for (int q = 0; q < c; ++q) {
          double one = fmax(-nd.lam[q] / nd.mu, cv[q]);
          ss += one * one;
        }
        viol[i] = sqrt(ss);
      } else {
        (void)node_cost(&nd);                               /* the node wraps its state on its first cost() call */
        viol[i] = fast_max_violation(&nd);
      }
    }
  }
  return ALILQR_OK;
}
