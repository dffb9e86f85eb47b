/* oracle_internal.h — shared declarations of the CPU oracle (test infrastructure only). */
#ifndef ORACLE_INTERNAL_H
#define ORACLE_INTERNAL_H

#include <math.h>
#include "alilqr_oracle.h"

#define OR_NX 6
#define OR_NU 2
#define OR_NC 64   /* the row count the C ABI accepts */
#define OR_PI 3.141592653589793

/* One trajectory node of the Python generations: model + cost + AL state. */
typedef struct or_node {
  int model, mode, n, m, c;
  int eq;                              /* equality table (is_equality) */
  double L, dt, v, k;
  const double *Q, *R;                 /* n x n, m x m */
  double smin[OR_NX], smax[OR_NX], cmin[OR_NU], cmax[OR_NU];   /* legacy box */
  const double *A, *B, *C;             /* linear table: c x n, c x m, c */
  double goal[OR_NX], state[OR_NX], control[OR_NU];
  double lam[OR_NC], mu, Imu[OR_NC];
  double cc[OR_NC];                    /* cached constraint values (constraints.py: self.c) */
} or_node;

void om_mm(int r, int k, int c, const double* A, const double* B, double* C);
void om_mtm(int r, int k, int c, const double* A, const double* B, double* C);
int om_inv(int m, const double* Ain, double* Ainv);
void or_inv_closed(int m, const double* M, double* Mi);

double or_wrap_py(double a);
void or_node_init(or_node* nd, const alilqr_problem* p, int t,
                  const double* goals, const double* Q, const double* R, const double* cons);
void or_py_dynamics(const or_node* nd, const double* s, const double* u, double* next);
void or_py_jacobian(const or_node* nd, const double* s, const double* u, double* A, double* B);

/* multi-start: offsets [H][m] added to the initial-guess controls of the solve running on this thread, or NULL */
extern _Thread_local const double* or_u_offset;

/* optional inputs (rectangle obstacles, first row index) of the solve running on this thread, or NULL */
extern _Thread_local const alilqr_extras* or_extras;

/* index of the problem the solve running on this thread has in its batch (per-problem optional inputs / outputs) */
extern _Thread_local int or_problem;

/* Decision margin of the solve running on this thread: the smallest relative distance, over every accept / reject /
 * stop comparison the solver made, between the two sides of the comparison.  A solve whose margin is far above the
 * arithmetic noise (~1e-15) makes the same decisions in any implementation whose arithmetic differs in the last
 * bits only; a solve with a margin at noise level is decided by rounding and cannot be pinned by anything. */
extern _Thread_local double or_min_margin;
static inline void or_decision(double lhs, double rhs) {
  double sc = fabs(lhs) > fabs(rhs) ? fabs(lhs) : fabs(rhs);
  if (!(sc > 0.0) || !isfinite(sc)) return;          /* both zero, or not finite: nothing rounding could flip */
  double mg = fabs(lhs - rhs) / sc;
  if (mg < or_min_margin) or_min_margin = mg;
}

/* NEW_AL generation (alilqr_oracle_new.c) */
int32_t or_new_solve(const alilqr_problem* p,
                     const double* x0, const double* goals,
                     const double* Q, const double* R, const double* cons,
                     double* x_out, double* u_out, double* node_x, double* node_u,
                     double* x_init, double* u_init, double* cost, double* viol,
                     double* lam, double* mu, int32_t* status, oracle_trace* trace);
int32_t or_new_linearize(const alilqr_problem* p,
                         const double* x, const double* u,
                         const double* lam, const double* mu,
                         const double* goals, const double* Q, const double* R, const double* cons,
                         double* f, double* fx, double* fu, double* l,
                         double* lx, double* lu, double* lxx, double* luu,
                         double* cval, double* lam_new, double* xnorm);
int32_t or_new_node_aux(const alilqr_problem* p,
                        const double* x, const double* u,
                        const double* lam, const double* mu,
                        const double* goals, const double* Q, const double* R, const double* cons,
                        double* viol, double* par_cost, double* fxx);
#endif
