/*
 * alilqr_oracle_new.c — CPU restatement of the NEW_AL generation
 * (al_ilqr_cpp/new_al_ilqr.h with model/new_bicycle_node.h, model/new_lat_bicycle_node.h,
 * constraints/constraints.h, linear_constraints.h, box_constraints.h).
 * TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference holds no golden vector, assertion or fixture
 * for this generation and its C++ cannot be built in this image (Eigen and
 * Bazel-fetched dependencies are absent), so this restatement is checked only
 * by review against the cited lines and by the node / constraint formulas it
 * shares with the Python generations (tests compare those).
 *
 */
#define _GNU_SOURCE   /* M_PI under -std=c11 */
#include "alilqr_oracle.h"
#include "oracle_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PARALLEL_NUM 5          /* constraints/constraints.h:73 */

typedef struct nw_node {
  int model, n, m, c;
  double L, dt, v, k;
  const double *Q, *R, *A, *B, *C;
  double Aown[OR_NC * OR_NX], Cown[OR_NC];   /* the node's own copy of A and C (constraints are copied into the node,
                                                new_bicycle_node.h:169-170); A and C point here */
  int cur_index;                       /* current_constraints_index_ (constraints.h:1191) */
  int eq;                              /* is_equality_ */
  const double* Qc;                    /* quadratic table: c matrices n x n, or NULL (quadratic_constraints.h:94) */
  double goal[OR_NX];
  double lam[OR_NC], mu;
  /* caches of constraints.h: c_, lambda_proj_, dxdx_, dudu_ */
  double c_[OR_NC], lp_[OR_NC], dxdx[OR_NX * OR_NX], dudu[OR_NU * OR_NU];
  double cx_[OR_NC * OR_NX];           /* constraint Jacobian wrt x at the last cost_jacobian() point */
} nw_node;

/* model/new_ilqr_node.h:32-37 — identity inside (-pi, pi), C fmod (sign of the dividend) outside */
static double wrap_c(double a) {
  if (fabs(a) < M_PI) return a;
  return fmod(a + M_PI, 2 * M_PI) - M_PI;
}

/* new_lat_bicycle_node.h:41-58 ; new_bicycle_node.h:422-446 */
static void nw_cont(const nw_node* nd, const double* s, const double* u, double* out) {
  double theta = wrap_c(s[2]), delta = wrap_c(s[3]);
  if (nd->model == ALILQR_MODEL_LAT_K) {
    out[0] = nd->v * cos(theta);
    out[1] = nd->v * sin(theta);
    out[2] = nd->v * tan(delta) / (nd->L * (1 + nd->k * nd->v * nd->v));
    out[3] = u[0];
    return;
  }
  double v = s[4], a = s[5];
  out[0] = v * cos(theta);
  out[1] = v * sin(theta);
  out[2] = v * tan(delta) / (nd->L * (1 + nd->k * v * v));
  out[3] = u[0];
  out[4] = a;
  out[5] = u[1];
}
/* new_lat_bicycle_node.h:27-39 ; new_bicycle_node.h:228-244 */
static void nw_dynamics(const nw_node* nd, const double* s, const double* u, double* next) {
  double k1[OR_NX], k2[OR_NX], mid[OR_NX];
  nw_cont(nd, s, u, k1);
  for (int i = 0; i < nd->n; ++i) mid[i] = s[i] + 0.5 * nd->dt * k1[i];
  nw_cont(nd, mid, u, k2);
  for (int i = 0; i < nd->n; ++i) next[i] = s[i] + nd->dt * k2[i];
  next[2] = wrap_c(next[2]);
  next[3] = wrap_c(next[3]);
}
/* 5-wide variant used by the line-search ladder: no angle wrap anywhere
 * (new_lat_bicycle_node.h:60-87 ; new_bicycle_node.h:285-377) — one column */
static void nw_par_cont(const nw_node* nd, const double* s, const double* u, double* out) {
  if (nd->model == ALILQR_MODEL_LAT_K) {
    double denom = nd->L * (1.0 + nd->k * nd->v * nd->v);
    out[0] = nd->v * cos(s[2]);
    out[1] = nd->v * sin(s[2]);
    out[2] = nd->v * tan(s[3]) / denom;
    out[3] = u[0];
    return;
  }
  double v = s[4];
  out[0] = v * cos(s[2]);
  out[1] = v * sin(s[2]);
  out[2] = v * tan(s[3]) * (1.0 / (1.0 + nd->k * v * v)) / nd->L;
  out[3] = u[0];
  out[4] = s[5];
  out[5] = u[1];
}
static void nw_par_dynamics(const nw_node* nd, const double* s, const double* u, double* next) {
  double k1[OR_NX], k2[OR_NX], mid[OR_NX];
  nw_par_cont(nd, s, u, k1);
  for (int i = 0; i < nd->n; ++i) mid[i] = s[i] + k1[i] * nd->dt * 0.5;
  nw_par_cont(nd, mid, u, k2);
  for (int i = 0; i < nd->n; ++i) next[i] = s[i] + k2[i] * nd->dt;
}

/* new_lat_bicycle_node.h:91-126 ; new_bicycle_node.h:492-545 */
static void nw_jacobian(const nw_node* nd, const double* s, const double* uc, double* A, double* B) {
  int n = nd->n, m = nd->m;
  double dt = nd->dt, L = nd->L, k = nd->k;
  memset(A, 0, sizeof(double) * n * n);
  memset(B, 0, sizeof(double) * n * m);
  for (int i = 0; i < n; ++i) A[i * n + i] = 1.0;
  double theta = wrap_c(s[2]), delta = wrap_c(s[3]);
  double u1 = uc[0];
  if (nd->model == ALILQR_MODEL_LAT_K) {
    double v = nd->v;
    double theta_mid = theta + 0.5 * dt * v * tan(delta) / (L * (k * v * v + 1));
    double tan_delta = tan(delta);
    double tan_delta_mid = tan(delta + 0.5 * dt * u1);
    double denom = L * (k * v * v + 1);
    double sn = sin(theta_mid), cs = cos(theta_mid);
    double t2 = tan_delta * tan_delta, tm2 = tan_delta_mid * tan_delta_mid;
    A[0 * 4 + 2] = -dt * v * sn;
    A[0 * 4 + 3] = -0.5 * dt * dt * v * v * (t2 + 1) * sn / denom;
    A[1 * 4 + 2] = dt * v * cs;
    A[1 * 4 + 3] = 0.5 * dt * dt * v * v * (t2 + 1) * cs / denom;
    A[2 * 4 + 3] = dt * v * (tm2 + 1) / denom;
    B[2] = 0.5 * dt * dt * v * (tm2 + 1) / denom;
    B[3] = dt;
    return;
  }
  double v = s[4], a = s[5];
  double D = L * (k * v * v + 1);
  double tm = theta + 0.5 * dt * v * tan(delta) / D;
  double vt = 0.5 * a * dt + v;
  double Dm = L * (k * vt * vt + 1);
  double td = tan(delta), tu = tan(delta + 0.5 * dt * u1);
  double g = -1.0 * dt * k * v * v * td / (L * (k * v * v + 1) * (k * v * v + 1)) + 0.5 * dt * td / D;
  A[0 * 6 + 2] = -dt * vt * sin(tm);
  A[0 * 6 + 3] = -0.5 * dt * dt * v * vt * (td * td + 1) * sin(tm) / D;
  A[0 * 6 + 4] = -dt * vt * g * sin(tm) + dt * cos(tm);
  A[0 * 6 + 5] = 0.5 * dt * dt * cos(tm);
  A[1 * 6 + 2] = dt * vt * cos(tm);
  A[1 * 6 + 3] = 0.5 * dt * dt * v * vt * (td * td + 1) * cos(tm) / D;
  A[1 * 6 + 4] = dt * vt * g * cos(tm) + dt * sin(tm);
  A[1 * 6 + 5] = 0.5 * dt * dt * sin(tm);
  A[2 * 6 + 3] = dt * vt * (tu * tu + 1) / Dm;
  A[2 * 6 + 4] = -dt * k * vt * (1.0 * a * dt + 2 * v) * tu / (L * (k * vt * vt + 1) * (k * vt * vt + 1)) + dt * tu / Dm;
  A[2 * 6 + 5] = -1.0 * dt * dt * k * vt * vt * tu / (L * (k * vt * vt + 1) * (k * vt * vt + 1)) + 0.5 * dt * dt * tu / Dm;
  A[4 * 6 + 5] = dt;
  B[2 * 2 + 0] = 0.5 * dt * dt * vt * (tu * tu + 1) / Dm;
  B[3 * 2 + 0] = dt;
  B[4 * 2 + 1] = 0.5 * dt * dt;
  B[5 * 2 + 1] = dt;
}

/* f_xx of the first three state rows: new_lat_bicycle_node.h:128-163 (RK2 mid-point form),
 * new_bicycle_node.h:605-650 (Euler form).  H[r] is n x n. */
static void nw_fxx(const nw_node* nd, const double* s, const double* uc, double* H0, double* H1, double* H2) {
  int n = nd->n;
  (void)uc;
  memset(H0, 0, sizeof(double) * n * n);
  memset(H1, 0, sizeof(double) * n * n);
  memset(H2, 0, sizeof(double) * n * n);
  double theta = wrap_c(s[2]), delta = wrap_c(s[3]);
  double dt = nd->dt, L = nd->L, k = nd->k;
  if (nd->model == ALILQR_MODEL_LAT_K) {
    double v = nd->v;
    double theta_mid = theta + 0.5 * dt * v * tan(delta) / (L * (k * v * v + 1));
    double td = tan(delta), t2 = td * td;
    double denom = L * (k * v * v + 1);
    double cs = cos(theta_mid), sn = sin(theta_mid);
    H0[2 * 4 + 2] = -dt * v * cs;
    H0[2 * 4 + 3] = -0.5 * dt * dt * v * v * (t2 + 1) * cs / denom;
    H0[3 * 4 + 2] = -0.5 * dt * dt * v * v * (t2 + 1) * cs / denom;
    H0[3 * 4 + 3] = -dt * dt * v * v * ((t2 + 1) / denom) * (0.5 * v * (t2 + 1) * cs / denom + sn * td);
    H1[2 * 4 + 2] = -dt * v * sn;
    H1[2 * 4 + 3] = -0.5 * dt * dt * v * v * (t2 + 1) * sn / denom;
    H1[3 * 4 + 2] = -0.5 * dt * dt * v * v * (t2 + 1) * sn / denom;
    H1[3 * 4 + 3] = dt * dt * v * v * ((t2 + 1) / denom) * (cs * td - 0.5 * v * (t2 + 1) * sn / denom);
    H2[3 * 4 + 3] = 2 * dt * v * (t2 + 1) * td / denom;
    return;
  }
  double v = s[4];
  double sn = sin(theta), cs = cos(theta), td = tan(delta);
  double tp1 = td * td + 1;
  double kv2 = k * v * v, kp1 = kv2 + 1;
  H0[2 * 6 + 2] = -dt * v * cs;
  H0[2 * 6 + 4] = -dt * sn;
  H0[4 * 6 + 2] = H0[2 * 6 + 4];
  H1[2 * 6 + 2] = -dt * v * sn;
  H1[2 * 6 + 4] = dt * cs;
  H1[4 * 6 + 2] = H1[2 * 6 + 4];
  H2[3 * 6 + 3] = 2 * dt * v * tp1 * td / (kp1 * L);
  H2[3 * 6 + 4] = dt * (1 - k * v * v) * tp1 / (kp1 * L) / (kp1);
  H2[4 * 6 + 3] = H2[3 * 6 + 4];
  H2[4 * 6 + 4] = dt * 2 * k * v * (kv2 - 3) * td / L / kp1 / kp1 / kp1;
}

/* x' Q_i x as (x' Q_i) x  (quadratic_constraints.h:21-24) */
static double nw_quad_term(const nw_node* nd, int i, const double* x) {
  int n = nd->n;
  const double* Qi = nd->Qc + (size_t)i * n * n;
  double q = 0.0;
  for (int j = 0; j < n; ++j) {
    double t = 0.0;
    for (int k = 0; k < n; ++k) t += x[k] * Qi[k * n + j];
    q += t * x[j];
  }
  return q;
}
/* linear_constraints.h:176-181 ; quadratic_constraints.h:15-27 (quadratic term first in the sum) */
static void nw_constraints(const nw_node* nd, const double* x, const double* u, double* c) {
  for (int i = 0; i < nd->c; ++i) {
    double ax = 0.0, bu = 0.0;
    for (int j = 0; j < nd->n; ++j) ax += nd->A[i * nd->n + j] * x[j];
    for (int j = 0; j < nd->m; ++j) bu += nd->B[i * nd->m + j] * u[j];
    if (nd->Qc) c[i] = ((nw_quad_term(nd, i, x) + ax) + bu) + nd->C[i];
    else c[i] = ax + bu + nd->C[i];
  }
}
/* one column of parallel_constraints: linear_constraints.h ; quadratic_constraints.h:29-45 (quadratic term last) */
static void nw_par_constraints(const nw_node* nd, const double* x, const double* u, double* c) {
  for (int i = 0; i < nd->c; ++i) {
    double ax = 0.0, bu = 0.0;
    for (int j = 0; j < nd->n; ++j) ax += nd->A[i * nd->n + j] * x[j];
    for (int j = 0; j < nd->m; ++j) bu += nd->B[i * nd->m + j] * u[j];
    c[i] = ax + bu + nd->C[i];
    if (nd->Qc) c[i] = c[i] + nw_quad_term(nd, i, x);
  }
}
/* constraints.h:514-529 */
static double nw_al_cost(nw_node* nd, const double* x, const double* u) {
  nw_constraints(nd, x, u, nd->c_);
  double pp = 0.0, ll = 0.0;
  for (int i = 0; i < nd->c; ++i) {
    double f = nd->lam[i] - nd->mu * nd->c_[i];
    nd->lp_[i] = nd->eq ? f : fmin(f, 0.0);     /* constraints.h:520-528; lp_ doubles as the equality factor */
    pp += nd->lp_[i] * nd->lp_[i];
  }
  for (int i = 0; i < nd->c; ++i) ll += nd->lam[i] * nd->lam[i];
  return 0.5 / nd->mu * (pp - ll);
}
/* new_bicycle_node.h:702-724: only the diagonals of Q and R enter the cost */
static double nw_cost(nw_node* nd, const double* x, const double* u) {
  double sc = 0.0, cc = 0.0;
  for (int i = 0; i < nd->n; ++i) {
    double e = x[i] - nd->goal[i];
    sc += (e * nd->Q[i * nd->n + i]) * e;
  }
  for (int j = 0; j < nd->m; ++j) cc += (nd->R[j * nd->m + j] * u[j]) * u[j];
  double al = nw_al_cost(nd, x, u);
  return sc + cc + al;
}
/* one column of parallel_cost (new_bicycle_node.h:768-795) + parallel AL cost (constraints.h:558-580) */
static double nw_par_cost(const nw_node* nd, const double* x, const double* u) {
  double cq = 0.0, cr = 0.0, al = 0.0, c[OR_NC];
  for (int i = 0; i < nd->n; ++i) {
    double e = x[i] - nd->goal[i];
    cq += e * (e * nd->Q[i * nd->n + i]);
  }
  for (int j = 0; j < nd->m; ++j) cr += (nd->R[j * nd->m + j] * u[j]) * u[j];
  nw_par_constraints(nd, x, u, c);
  for (int i = 0; i < nd->c; ++i) {
    double f = nd->lam[i] - nd->mu * c[i];
    double p = nd->eq ? f : fmin(f, 0.0);
    al += p * p - nd->lam[i] * nd->lam[i];
  }
  al = 0.5 / nd->mu * al;
  return (cq + cr) + al;
}
/* new_bicycle_node.h:842-856 + constraints.h:658-691; needs the caches of the preceding cost() call */
static void nw_cost_jacobian(nw_node* nd, const double* x, const double* u, double* Jx, double* Ju) {
  int n = nd->n, m = nd->m;
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += (2 * nd->Q[i * n + j]) * (x[j] - nd->goal[j]);
    Jx[i] = s;
  }
  for (int i = 0; i < m; ++i) {
    double s = 0.0;
    for (int j = 0; j < m; ++j) s += (2 * nd->R[i * m + j]) * u[j];
    Ju[i] = s;
  }
  /* c_x = A (+ rows 2 Q_i x, quadratic_constraints.h:47-55) */
  for (int i = 0; i < nd->c; ++i)
    for (int a = 0; a < n; ++a) {
      double v = nd->A[i * n + a];
      if (nd->Qc) {
        const double* Qi = nd->Qc + (size_t)i * n * n;
        double qx = 0.0;
        for (int k = 0; k < n; ++k) qx += (2 * Qi[a * n + k]) * x[k];
        v = v + qx;
      }
      nd->cx_[i * n + a] = v;
    }
  /* columns of cx^T / cu^T are zeroed where lambda_proj >= 0 (projection_jacobian2, constraints.h:429-436) */
  for (int a = 0; a < n; ++a) {
    double s = 0.0;
    for (int i = 0; i < nd->c; ++i) s += ((!nd->eq && nd->lp_[i] >= 0.0) ? 0.0 : nd->cx_[i * n + a]) * nd->lp_[i];
    Jx[a] += -s;
  }
  for (int a = 0; a < m; ++a) {
    double s = 0.0;
    for (int i = 0; i < nd->c; ++i) s += ((!nd->eq && nd->lp_[i] >= 0.0) ? 0.0 : nd->B[i * m + a]) * nd->lp_[i];
    Ju[a] += -s;
  }
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b) {
      double s = 0.0;
      if (nd->eq) {                               /* mu * (cx' cx), constraints.h:791-795 */
        for (int i = 0; i < nd->c; ++i) s += nd->cx_[i * n + a] * nd->cx_[i * n + b];
        s = nd->mu * s;
      } else {
        for (int i = 0; i < nd->c; ++i) s += (nd->mu * ((nd->lp_[i] >= 0.0) ? 0.0 : nd->cx_[i * n + a])) * nd->cx_[i * n + b];
      }
      nd->dxdx[a * n + b] = s;
    }
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b) {
      double s = 0.0;
      if (nd->eq) {
        for (int i = 0; i < nd->c; ++i) s += nd->B[i * m + a] * nd->B[i * m + b];
        s = nd->mu * s;
      } else {
        for (int i = 0; i < nd->c; ++i) s += (nd->mu * ((nd->lp_[i] >= 0.0) ? 0.0 : nd->B[i * m + a])) * nd->B[i * m + b];
      }
      nd->dudu[a * m + b] = s;
    }
}
/* new_bicycle_node.h:897-910 + constraints.h:773-816: H -= sum_i lambda_proj_i * (2 Q_i)  (tensor_contract :1319-1347);
 * the constraint Hessians of a linear table are zero */
static void nw_cost_hessian(nw_node* nd, double* Hx, double* Hu) {
  int n = nd->n, m = nd->m;
  for (int i = 0; i < n * n; ++i) {
    double t = 0.0;
    if (nd->Qc)
      for (int r = 0; r < nd->c; ++r) t += (2.0 * nd->Qc[(size_t)r * n * n + i]) * nd->lp_[r];
    Hx[i] = 2 * nd->Q[i] + (nd->dxdx[i] - 1.0 * t);
  }
  for (int i = 0; i < m * m; ++i) Hu[i] = 2 * nd->R[i] + (nd->dudu[i] - 1.0 * 0.0);
}
/* constraints.h:1027-1040 */
static double nw_max_violation(const nw_node* nd, const double* x, const double* u) {
  double c[OR_NC], v = 0.0;
  nw_constraints(nd, x, u, c);
  for (int i = 0; i < nd->c; ++i) {
    double dc = fabs(c[i] - fmin(c[i], 0.0));
    if (dc > v) v = dc;
  }
  return v;
}
/* constraints.h:878-887 */
static void nw_update_lambda(nw_node* nd, const double* x, const double* u) {
  double c[OR_NC];
  nw_constraints(nd, x, u, c);
  for (int i = 0; i < nd->c; ++i) {
    double f = nd->lam[i] - nd->mu * c[i];
    nd->lam[i] = nd->eq ? f : fmin(f, 0.0);     /* constraints.h:878-887 */
  }
}

static void nw_node_init(nw_node* nd, const alilqr_problem* p, int t,
                         const double* goals, const double* Q, const double* R, const double* cons) {
  memset(nd, 0, sizeof(*nd));
  int n = p->n, m = p->m, c = p->c;
  int tt = (p->table_nodes > 1) ? t : 0;
  nd->model = p->model; nd->n = n; nd->m = m; nd->c = c;
  nd->L = p->L; nd->dt = p->dt; nd->v = p->v; nd->k = p->k;
  nd->Q = Q + (size_t)tt * n * n;
  nd->R = R + (size_t)tt * m * m;
  const int quad = (p->cons_kind & ALILQR_CONS_QUADRATIC) ? 1 : 0;
  nd->eq = (p->cons_kind & ALILQR_CONS_EQUALITY) ? 1 : 0;
  const double* b = cons + (size_t)tt * (c * n + c * m + c + (quad ? c * n * n : 0));
  memcpy(nd->Aown, b, sizeof(double) * c * n);
  memcpy(nd->Cown, b + c * n + c * m, sizeof(double) * c);
  nd->A = nd->Aown; nd->B = b + c * n; nd->C = nd->Cown;
  nd->Qc = quad ? b + c * n + c * m + c : NULL;
  nd->cur_index = or_extras ? ((or_extras->first_rows) ? or_extras->first_rows[t] : or_extras->first_row) : 0;
  memcpy(nd->goal, goals + (size_t)t * n, sizeof(double) * n);
  nd->mu = 1.0;
}

/* ------------------------------------------------------------------------ */
/* solver: new_al_ilqr.h                                                      */
/* ------------------------------------------------------------------------ */
typedef struct {
  const alilqr_problem* p;
  nw_node* nodes;
  int H, n, m;
  double *x, *u, *pre_x, *pre_u, *cost_list, *pre_cost;
  double *Jx, *Ju, *Hx, *Hu, *A, *B, *F0, *F1, *F2, *K, *kf;
  double dV_lin, dV_quad, mu;
  long n_inner, n_trials;
} nw_solver;

static double sum_cost(const nw_solver* s) {
  double t = 0.0;
  for (int i = 0; i <= s->H; ++i) t += s->cost_list[i];
  return t;
}

/* :648-696 */
static void nw_initial_guess(nw_solver* s, const double* x0) {
  int H = s->H, n = s->n, m = s->m;
  double P[OR_NX * OR_NX], A[OR_NX * OR_NX], B[OR_NX * OR_NU], zero_u[OR_NU] = {0};
  memcpy(s->x, x0, sizeof(double) * n);
  memset(P, 0, sizeof(P));
  for (int i = 0; i < n; ++i) P[i * n + i] = 1.0;
  for (int t = H - 1; t >= 0; --t) {
    nw_node* nd = &s->nodes[t];
    nw_jacobian(nd, nd->goal, zero_u, A, B);
    double BtP[OR_NU * OR_NX], BtPB[OR_NU * OR_NU], BtPA[OR_NU * OR_NX], M[OR_NU * OR_NU], Mi[OR_NU * OR_NU];
    om_mtm(m, n, n, B, P, BtP);
    om_mm(m, n, m, BtP, B, BtPB);
    om_mm(m, n, n, BtP, A, BtPA);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) M[i * m + j] = ((i == j) ? 1.0 : 0.0) * s->p->prm.init_r + BtPB[i * m + j];
    or_inv_closed(m, M, Mi);
    double* K = s->K + (size_t)t * m * n;
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
    nw_node* nd = &s->nodes[t];
    const double* K = s->K + (size_t)t * m * n;
    for (int j = 0; j < m; ++j) {
      double acc = 0.0;
      for (int i = 0; i < n; ++i) acc += (-K[j * n + i]) * (s->x[t * n + i] - nd->goal[i]);
      if (or_u_offset) acc = acc + or_u_offset[t * m + j];
      s->u[t * m + j] = acc;
    }
    nw_dynamics(nd, s->x + t * n, s->u + t * m, s->x + (t + 1) * n);
  }
  for (int t = 0; t <= H; ++t) {
    memset(s->nodes[t].lam, 0, sizeof(s->nodes[t].lam));
    s->nodes[t].mu = 1.0;
  }
}

/* :720-778 */
static void nw_calc_derivatives(nw_solver* s) {
  int H = s->H, n = s->n, m = s->m;
  size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
  double zero_u[OR_NU] = {0}, dumJu[OR_NU], dumHu[OR_NU * OR_NU];
  nw_node* ndH = &s->nodes[H];
  s->cost_list[H] = nw_cost(ndH, s->x + (size_t)H * n, zero_u);
  nw_cost_jacobian(ndH, s->x + (size_t)H * n, zero_u, s->Jx + (size_t)H * n, dumJu);
  nw_cost_hessian(ndH, s->Hx + (size_t)H * nn, dumHu);
  for (int t = 0; t < H; ++t) {
    nw_node* nd = &s->nodes[t];
    const double *x = s->x + (size_t)t * n, *u = s->u + (size_t)t * m;
    s->cost_list[t] = nw_cost(nd, x, u);
    nw_cost_jacobian(nd, x, u, s->Jx + (size_t)t * n, s->Ju + (size_t)t * m);
    nw_cost_hessian(nd, s->Hx + t * nn, s->Hu + t * mm);
    nw_jacobian(nd, x, u, s->A + t * nn, s->B + t * nm);
    nw_fxx(nd, x, u, s->F0 + t * nn, s->F1 + t * nn, s->F2 + t * nn);
  }
}

/* :929-990 */
static void nw_backward(nw_solver* s) {
  int H = s->H, n = s->n, m = s->m;
  size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
  double Vx[OR_NX], Vxx[OR_NX * OR_NX];
  memcpy(Vx, s->Jx + (size_t)H * n, sizeof(double) * n);
  memcpy(Vxx, s->Hx + (size_t)H * nn, sizeof(double) * nn);
  s->dV_lin = 0.0; s->dV_quad = 0.0;
  for (int t = H - 1; t >= 0; --t) {
    const double *At = s->A + t * nn, *Bt = s->B + t * nm;
    double Qu[OR_NU], Qx[OR_NX], Qux[OR_NU * OR_NX], Quu[OR_NU * OR_NU], Qxx[OR_NX * OR_NX];
    double BtV[OR_NU * OR_NX], AtV[OR_NX * OR_NX], tmp[OR_NX * OR_NX];
    om_mtm(m, n, 1, Bt, Vx, Qu);
    for (int j = 0; j < m; ++j) Qu[j] = s->Ju[t * m + j] + Qu[j];
    om_mtm(n, n, 1, At, Vx, Qx);
    for (int i = 0; i < n; ++i) Qx[i] = s->Jx[t * n + i] + Qx[i];
    om_mtm(m, n, n, Bt, Vxx, BtV);
    om_mm(m, n, n, BtV, At, Qux);
    om_mm(m, n, m, BtV, Bt, Quu);
    for (size_t i = 0; i < mm; ++i) Quu[i] = s->Hu[t * mm + i] + Quu[i];
    om_mtm(n, n, n, At, Vxx, AtV);
    om_mm(n, n, n, AtV, At, tmp);
    for (size_t i = 0; i < nn; ++i) Qxx[i] = s->Hx[t * nn + i] + tmp[i];
    for (size_t i = 0; i < nn; ++i)
      Qxx[i] += (s->F0[t * nn + i] * Vx[0] + s->F1[t * nn + i] * Vx[1]) + s->F2[t * nn + i] * Vx[2];
    double Qi[OR_NU * OR_NU];
    or_inv_closed(m, Quu, Qi);
    double* K = s->K + (size_t)t * m * n;
    double* k = s->kf + (size_t)t * m;
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
    /* Vx = Qx + K'(Quu k + Qu) + Qux'k ; Vxx = Qxx + K'(Quu K + Qux) + Qux'K */
    double w[OR_NU], W[OR_NU * OR_NX], t1[OR_NX], t2[OR_NX], m1[OR_NX * OR_NX], m2[OR_NX * OR_NX];
    om_mm(m, m, 1, Quu, k, w);
    for (int j = 0; j < m; ++j) w[j] = w[j] + Qu[j];
    om_mtm(n, m, 1, K, w, t1);
    om_mtm(n, m, 1, Qux, k, t2);
    double dl = 0.0, dq = 0.0;
    for (int j = 0; j < m; ++j) dl += k[j] * Qu[j];
    {
      double kq[OR_NU];
      for (int j = 0; j < m; ++j) {
        double acc = 0.0;
        for (int q = 0; q < m; ++q) acc += k[q] * Quu[q * m + j];
        kq[j] = acc;
      }
      for (int j = 0; j < m; ++j) dq += kq[j] * k[j];
    }
    for (int i = 0; i < n; ++i) Vx[i] = (Qx[i] + t1[i]) + t2[i];
    om_mm(m, m, n, Quu, K, W);
    for (int i = 0; i < m * n; ++i) W[i] = W[i] + Qux[i];
    om_mtm(n, m, n, K, W, m1);
    om_mtm(n, m, n, Qux, K, m2);
    for (size_t i = 0; i < nn; ++i) Vxx[i] = (Qxx[i] + m1[i]) + m2[i];
    s->dV_lin += dl;
    s->dV_quad += 0.5 * dq;
  }
}

/* :786-793 */
static void nw_update_trajectory(nw_solver* s, double alpha) {
  int H = s->H, n = s->n, m = s->m;
  double zero_u[OR_NU] = {0};
  for (int i = 0; i < H; ++i) {
    const double* K = s->K + (size_t)i * m * n;
    for (int j = 0; j < m; ++j) {
      double fb = 0.0;
      for (int q = 0; q < n; ++q) fb += K[j * n + q] * (s->x[i * n + q] - s->pre_x[i * n + q]);
      s->u[i * m + j] += fb + alpha * s->kf[i * m + j];
    }
    nw_dynamics(&s->nodes[i], s->x + i * n, s->u + i * m, s->x + (i + 1) * n);
    s->cost_list[i] = nw_cost(&s->nodes[i], s->x + i * n, s->u + i * m);
  }
  s->cost_list[H] = nw_cost(&s->nodes[H], s->x + (size_t)H * n, zero_u);
  s->n_trials++;
}

/* :818-898 — five step sizes alpha, alpha/3, ... rolled out side by side without angle wraps */
static void nw_parallel_search(nw_solver* s, double alpha, double* best_alpha, double* best_cost) {
  int H = s->H, n = s->n, m = s->m;
  double av[PARALLEL_NUM], cost[PARALLEL_NUM], zero_u[OR_NU] = {0};
  for (int c = 0; c < PARALLEL_NUM; ++c) { av[c] = alpha; alpha /= 3.0; }
  for (int c = 0; c < PARALLEL_NUM; ++c) {
    double xn[OR_NX], un[OR_NU], nx[OR_NX];
    memcpy(xn, s->x, sizeof(double) * n);
    double total = 0.0;
    for (int i = 0; i < H; ++i) {
      const double* K = s->K + (size_t)i * m * n;
      for (int j = 0; j < m; ++j) {
        double fb = 0.0;
        for (int q = 0; q < n; ++q) fb += K[j * n + q] * (xn[q] - s->x[i * n + q]);
        un[j] = (s->u[i * m + j] + fb) + s->kf[i * m + j] * av[c];
      }
      double one = nw_par_cost(&s->nodes[i], xn, un);
      nw_par_dynamics(&s->nodes[0], xn, un, nx);
      memcpy(xn, nx, sizeof(double) * n);
      total += one;
    }
    total += nw_par_cost(&s->nodes[H], xn, zero_u);
    cost[c] = total;
    s->n_trials++;
  }
  int best = 0;
  for (int c = 1; c < PARALLEL_NUM; ++c) {
    or_decision(cost[c], cost[best]);
    if (cost[c] < cost[best]) best = c;
  }
  *best_cost = cost[best];
  *best_alpha = av[best];
}

/* LinearConstraints / BoxConstraints / QuadraticConstraints::UpdateConstraints
 * (linear_constraints.h:360-373, box_constraints.h:80-92, quadratic_constraints.h:57-64) */
static void nw_node_update_constraints(nw_node* nd, const double* A_row, double C_row) {
  for (int i = 0; i < nd->c; ++i)
    if (nd->Cown[i] == C_row) return;
  if (!nd->Qc) {
    if (nd->cur_index + 1 >= nd->c) return;   /* the reference would write past its fixed-size table here */
    nd->cur_index += 1;
  }
  if (nd->cur_index < 0 || nd->cur_index >= nd->c) return;
  memcpy(nd->Aown + (size_t)nd->cur_index * nd->n, A_row, sizeof(double) * nd->n);
  nd->Cown[nd->cur_index] = C_row;
}

/* new_al_ilqr.h:525-625 with the edge vectors of the constructor :134-208 */
static void nw_update_constraints(nw_solver* s) {
  const alilqr_extras* ex = or_extras;
  if (!ex) return;
  int n = s->n;
  for (int side = 0; side < 2; ++side) {
    const double* obs = side == 0 ? ex->left_obs : ex->right_obs;
    int cnt = side == 0 ? ex->n_left : ex->n_right;
    if (!obs || cnt <= 0) continue;
    for (int t = 0; t <= s->H; ++t) {
      double px = s->x[(size_t)t * n + 0], py = s->x[(size_t)t * n + 1];
      for (int k = 0; k < cnt; ++k) {
        const double* vx = obs + (size_t)k * 8;       /* x of the four vertices */
        const double* vy = vx + 4;                    /* y of the four vertices */
        int inside = 1;
        double ext = vy[0];
        for (int e = 0; e < 4; ++e) {
          int e2 = (e + 1) & 3;
          double ex_ = vx[e2] - vx[e], ey_ = vy[e2] - vy[e];        /* edge vector */
          double qx = px - vx[e], qy = py - vy[e];
          double cr = qx * ey_ - qy * ex_;                            /* MultiVectorCross :215-222 */
          or_decision(qx * ey_, qy * ex_);
          if (!(cr < 0)) inside = 0;
          if (side == 0) { if (vy[e] > ext) ext = vy[e]; } else { if (vy[e] < ext) ext = vy[e]; }
        }
        if (!inside) continue;
        double A_row[OR_NX] = {0};
        A_row[1] = (side == 0) ? -1.0 : 1.0;
        nw_node_update_constraints(&s->nodes[t], A_row, (side == 0) ? ext : -ext);
      }
    }
  }
}

/* :1013-1077 */
static void nw_forward(nw_solver* s) {
  int H = s->H, n = s->n, m = s->m;
  size_t xs = (size_t)(H + 1) * n, us = (size_t)H * m;
  double old_cost = sum_cost(s), new_cost = 0.0;
  memcpy(s->pre_x, s->x, sizeof(double) * xs);
  memcpy(s->pre_u, s->u, sizeof(double) * us);
  memcpy(s->pre_cost, s->cost_list, sizeof(double) * (H + 1));
  double alpha = 1.0, best_alpha = 1.0, best_cost = 0.0;
  or_decision(fabs(s->dV_lin), s->p->prm.dv_gate);
  if (fabs(s->dV_lin) < s->p->prm.dv_gate) return;          /* the reference compares with the float literal 0.2F */
  for (int index = 0; index < 10; ++index) {
    nw_update_trajectory(s, alpha);
    new_cost = sum_cost(s);
    or_decision(new_cost, old_cost);
    if (new_cost < old_cost) break;
    alpha /= 2.0;
    memcpy(s->x, s->pre_x, sizeof(double) * xs);
    memcpy(s->u, s->pre_u, sizeof(double) * us);
  }
  if (new_cost >= old_cost) {
    nw_parallel_search(s, alpha, &best_alpha, &best_cost);
    or_decision(best_cost, old_cost);
    if (best_cost >= old_cost) {
      memcpy(s->x, s->pre_x, sizeof(double) * xs);
      memcpy(s->u, s->pre_u, sizeof(double) * us);
      memcpy(s->cost_list, s->pre_cost, sizeof(double) * (H + 1));
    } else {
      nw_update_trajectory(s, best_alpha);
      new_cost = best_cost;
    }
  }
}

int32_t or_new_solve(const alilqr_problem* p,
                     const double* x0, const double* goals,
                     const double* Q, const double* R, const double* cons,
                     double* x_out, double* u_out, double* node_x, double* node_u,
                     double* x_init, double* u_init, double* cost, double* viol,
                     double* lam, double* mu, int32_t* status, oracle_trace* trace) {
  if (p->model != ALILQR_MODEL_LAT_K && p->model != ALILQR_MODEL_NEW_BICYCLE) return ALILQR_E_ARG;
  int H = p->horizon, n = p->n, m = p->m;
  size_t xs = (size_t)(H + 1) * n, us = (size_t)H * m, nn = (size_t)n * n;
  nw_solver s;
  memset(&s, 0, sizeof(s));
  s.p = p; s.H = H; s.n = n; s.m = m;
  /* the solver's own penalty survives from one optimize() to the next (new_al_ilqr.h:1125-1130; only the nodes are reset) */
  s.mu = (or_extras && or_extras->mu_solver_in) ? or_extras->mu_solver_in[or_problem] : 1.0;
  s.nodes = (nw_node*)malloc(sizeof(nw_node) * (H + 1));
  for (int t = 0; t <= H; ++t) nw_node_init(&s.nodes[t], p, t, goals, Q, R, cons);
  size_t nm = (size_t)n * m, mm = (size_t)m * m, Hn = (size_t)H;
  size_t total_d = 3 * xs + 3 * us + 2 * (Hn + 1) + (Hn + 1) * nn + Hn * mm + 4 * Hn * nn + Hn * nm + Hn * nm + us + 64;
  double* buf = (double*)calloc(total_d, sizeof(double));
  double* q = buf;
  s.x = q; q += xs; s.pre_x = q; q += xs; s.Jx = q; q += xs;
  s.u = q; q += us; s.pre_u = q; q += us; s.Ju = q; q += us;
  s.cost_list = q; q += Hn + 1; s.pre_cost = q; q += Hn + 1;
  s.Hx = q; q += (Hn + 1) * nn; s.Hu = q; q += Hn * mm;
  s.A = q; q += Hn * nn; s.F0 = q; q += Hn * nn; s.F1 = q; q += Hn * nn; s.F2 = q; q += Hn * nn;
  s.B = q; q += Hn * nm; s.K = q; q += Hn * nm; s.kf = q; q += us;

  nw_initial_guess(&s, x0);
  if (x_init) memcpy(x_init, s.x, sizeof(double) * xs);
  if (u_init) memcpy(u_init, s.u, sizeof(double) * us);
  int n_outer = 0;
  uint32_t flags = 0;
  double violation = 0.0;
  if (trace) trace->length = 0;
  const int stop_k = (or_extras && or_extras->stop_after) ? or_extras->stop_after[or_problem] : 0;
  int stopped = 0;
  for (int outer = 0; outer < p->prm.max_outer; ++outer) {
    for (int iter = 0; iter < p->prm.max_inner; ++iter) {          /* ILQRProcess :1090-1122 */
      nw_update_constraints(&s);
      nw_calc_derivatives(&s);
      double old_cost = sum_cost(&s);
      nw_backward(&s);
      long before = s.n_trials;
      nw_forward(&s);
      ++s.n_inner;
      double new_cost = sum_cost(&s);
      if (trace && trace->length < trace->capacity) {
        trace->iter_cost[trace->length] = new_cost;
        trace->trials[trace->length++] = (int32_t)(s.n_trials - before);
      }
      {
        double sc = fmax(fabs(new_cost), fabs(old_cost));
        or_decision(sc + ((old_cost - new_cost) - p->prm.inner_tol), sc);
        or_decision(old_cost, new_cost);
      }
      if (stop_k > 0 && s.n_inner >= stop_k) { stopped = 1; break; }   /* alilqr_extras::stop_after */
      if ((old_cost - new_cost < p->prm.inner_tol) && ((old_cost - new_cost) >= 0)) break;
    }
    if (stopped) { flags |= ALILQR_FLAG_STOPPED; break; }
    double zero_u[OR_NU] = {0};
    violation = 0.0;                                                /* :1081-1087 */
    for (int t = 0; t <= H; ++t) {
      double v = nw_max_violation(&s.nodes[t], s.x + (size_t)t * n, (t < H) ? s.u + (size_t)t * m : zero_u);
      if (t == 0 || v > violation) violation = v;
    }
    ++n_outer;
    or_decision(violation, p->prm.viol_stop);
    or_decision(violation, p->prm.viol_lambda);
    if (violation < p->prm.viol_stop) { flags |= ALILQR_FLAG_VIOL_OK; break; }
    if (violation > p->prm.viol_lambda) {
      s.mu = s.mu * p->prm.mu_gain;
      for (int t = 0; t <= H; ++t) s.nodes[t].mu = s.mu;
    } else {
      for (int t = 0; t <= H; ++t) nw_update_lambda(&s.nodes[t], s.x + (size_t)t * n, (t < H) ? s.u + (size_t)t * m : zero_u);
    }
  }
  if (!(flags & (ALILQR_FLAG_VIOL_OK | ALILQR_FLAG_STOPPED))) flags |= ALILQR_FLAG_OUTER_CAP;
  /* final AL cost of the returned trajectory under the final multipliers */
  double total = 0.0, zero_u[OR_NU] = {0};
  for (int t = 0; t <= H; ++t) total += nw_cost(&s.nodes[t], s.x + (size_t)t * n, (t < H) ? s.u + (size_t)t * m : zero_u);
  if (!isfinite(total)) flags |= ALILQR_FLAG_NONFINITE;
  if (x_out) memcpy(x_out, s.x, sizeof(double) * xs);
  if (u_out) memcpy(u_out, s.u, sizeof(double) * us);
  if (node_x) memcpy(node_x, s.x, sizeof(double) * xs);
  if (node_u) memcpy(node_u, s.u, sizeof(double) * us);
  if (cost) *cost = total;
  if (viol) *viol = violation;
  if (mu) *mu = s.nodes[0].mu;
  if (lam) for (int t = 0; t <= H; ++t) memcpy(lam + (size_t)t * p->c, s.nodes[t].lam, sizeof(double) * p->c);
  if (status) { status[0] = (int32_t)s.n_inner; status[1] = n_outer; status[2] = (int32_t)flags; status[3] = (int32_t)s.n_trials; }
  if (or_extras && or_extras->mu_solver_out) or_extras->mu_solver_out[or_problem] = s.mu;
  if (or_extras && or_extras->overlay_out) {
    /* what UpdateConstraints left in the nodes' own tables, in the layout of alilqr_extras::overlay_out */
    const int c = p->c, W = 2 * c + 1;
    const int quad = (p->cons_kind & ALILQR_CONS_QUADRATIC) ? 1 : 0;
    for (int t = 0; t <= H; ++t) {
      double* o = or_extras->overlay_out + ((size_t)or_problem * (H + 1) + t) * W;
      const int tt = (p->table_nodes > 1) ? t : 0;
      const double* b = cons + (size_t)tt * (c * n + c * m + c + (quad ? c * n * n : 0));
      for (int i = 0; i < c; ++i) {
        int changed = (s.nodes[t].Cown[i] != b[c * n + c * m + i]);
        for (int j = 0; j < n; ++j) changed |= (s.nodes[t].Aown[i * n + j] != b[i * n + j]);
        o[i] = changed ? s.nodes[t].Aown[i * n + 1] : 0.0;
        o[c + i] = changed ? s.nodes[t].Cown[i] : 0.0;
      }
      o[2 * c] = (double)s.nodes[t].cur_index;
    }
  }
  free(buf);
  free(s.nodes);
  return ALILQR_OK;
}

int32_t or_new_linearize(const alilqr_problem* p,
                         const double* x, const double* u,
                         const double* lam, const double* mu,
                         const double* goals, const double* Q, const double* R, const double* cons,
                         double* f, double* fx, double* fu, double* l,
                         double* lx, double* lu, double* lxx, double* luu,
                         double* cval, double* lam_new, double* xnorm) {
  if (p->model != ALILQR_MODEL_LAT_K && p->model != ALILQR_MODEL_NEW_BICYCLE) return ALILQR_E_ARG;
  int B = p->batch, H = p->horizon, n = p->n, m = p->m, c = p->c;
  size_t xs = (size_t)(H + 1) * n;
  for (int b = 0; b < B; ++b) {
    const double* g = goals + ((p->goal_batch > 1) ? (size_t)b * xs : 0);
    for (int t = 0; t <= H; ++t) {
      size_t i = (size_t)b * (H + 1) + t;
      nw_node nd;
      nw_node_init(&nd, p, t, g, Q, R, cons);
      if (lam) memcpy(nd.lam, lam + i * c, sizeof(double) * c);
      if (mu) nd.mu = mu[b];
      double A[OR_NX * OR_NX], Bm[OR_NX * OR_NU], nxt[OR_NX], jx[OR_NX], ju[OR_NU], hx[OR_NX * OR_NX], hu[OR_NU * OR_NU];
      nw_dynamics(&nd, x + i * n, u + i * m, nxt);
      nw_jacobian(&nd, x + i * n, u + i * m, A, Bm);
      double costv = nw_cost(&nd, x + i * n, u + i * m);
      nw_cost_jacobian(&nd, x + i * n, u + i * m, jx, ju);
      nw_cost_hessian(&nd, hx, hu);
      if (f) memcpy(f + i * n, nxt, sizeof(double) * n);
      if (fx) memcpy(fx + i * n * n, A, sizeof(double) * n * n);
      if (fu) memcpy(fu + i * n * m, Bm, sizeof(double) * n * m);
      if (l) l[i] = costv;
      if (lx) memcpy(lx + i * n, jx, sizeof(double) * n);
      if (lu) memcpy(lu + i * m, ju, sizeof(double) * m);
      if (lxx) memcpy(lxx + i * n * n, hx, sizeof(double) * n * n);
      if (luu) memcpy(luu + i * m * m, hu, sizeof(double) * m * m);
      if (xnorm) memcpy(xnorm + i * n, x + i * n, sizeof(double) * n);
      if (cval) nw_constraints(&nd, x + i * n, u + i * m, cval + i * c);
      if (lam_new) { nw_update_lambda(&nd, x + i * n, u + i * m); memcpy(lam_new + i * c, nd.lam, sizeof(double) * c); }
    }
  }
  return ALILQR_OK;
}

int32_t or_new_node_aux(const alilqr_problem* p,
                        const double* x, const double* u,
                        const double* lam, const double* mu,
                        const double* goals, const double* Q, const double* R, const double* cons,
                        double* viol, double* par_cost, double* fxx) {
  int B = p->batch, H = p->horizon, n = p->n, m = p->m, c = p->c;
  size_t xs = (size_t)(H + 1) * n, nn = (size_t)n * n;
  for (int b = 0; b < B; ++b) {
    const double* g = goals + ((p->goal_batch > 1) ? (size_t)b * xs : 0);
    for (int t = 0; t <= H; ++t) {
      size_t i = (size_t)b * (H + 1) + t;
      nw_node nd;
      nw_node_init(&nd, p, t, g, Q, R, cons);
      if (lam) memcpy(nd.lam, lam + i * c, sizeof(double) * c);
      if (mu) nd.mu = mu[b];
      if (viol) viol[i] = nw_max_violation(&nd, x + i * n, u + i * m);
      if (par_cost) par_cost[i] = nw_par_cost(&nd, x + i * n, u + i * m);
      if (fxx) nw_fxx(&nd, x + i * n, u + i * m, fxx + (i * 3 + 0) * nn, fxx + (i * 3 + 1) * nn, fxx + (i * 3 + 2) * nn);
    }
  }
  return ALILQR_OK;
}
