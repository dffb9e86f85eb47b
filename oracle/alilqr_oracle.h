/*
 * alilqr_oracle.h — CPU restatement of the reference AL-iLQR solvers.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP path is compared
 * against (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg); the
 * product library never links, loads or calls it.
 *
 * All pointers are HOST pointers; shapes are those of include/alilqr.h.
 */
#ifndef ALILQR_ORACLE_H
#define ALILQR_ORACLE_H

#include "../include/alilqr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Per-solve trace (optional, for debugging parity): cost after every inner
 * iteration and roll-outs used by its line search. */
typedef struct oracle_trace {
  int32_t capacity;      /* entries available in iter_cost / trials */
  int32_t length;        /* entries written */
  double* iter_cost;
  int32_t* trials;
} oracle_trace;

/* One problem (p->batch, p->goal_batch are ignored; every pointer addresses
 * exactly one problem).  Output pointers may be NULL. */
int32_t oracle_solve(const alilqr_problem* p,
                     const double* x0, const double* goals,
                     const double* Q, const double* R, const double* cons,
                     double* x_out, double* u_out, double* node_x, double* node_u,
                     double* x_init, double* u_init, double* cost, double* viol,
                     double* lam, double* mu, int32_t* status, oracle_trace* trace);

/* Same argument list as alilqr_solve_batch with host pointers; problems are
 * spread over `threads` OpenMP threads (<=0: all). */
int32_t oracle_solve_batch(const alilqr_problem* p,
                           const double* x0, const double* goals,
                           const double* Q, const double* R, const double* cons,
                           double* x_out, double* u_out, double* node_x, double* node_u,
                           double* x_init, double* u_init, double* cost, double* viol,
                           double* lam, double* mu, int32_t* status, int32_t threads);

/* Same argument list as alilqr_solve_batch_ex with host pointers (also those inside *ex). */
int32_t oracle_solve_batch_ex(const alilqr_problem* p,
                              const double* x0, const double* goals,
                              const double* Q, const double* R, const double* cons, const alilqr_extras* ex,
                              double* x_out, double* u_out, double* node_x, double* node_u,
                              double* x_init, double* u_init, double* cost, double* viol,
                              double* lam, double* mu, int32_t* status, int32_t threads);

/* Same argument list as alilqr_solve_batch_multistart with host pointers. */
int32_t oracle_solve_batch_multistart(const alilqr_problem* p,
                                      const double* x0, const double* goals,
                                      const double* Q, const double* R, const double* cons, const double* u_offset,
                                      double* x_out, double* u_out, double* node_x, double* node_u,
                                      double* x_init, double* u_init, double* cost, double* viol,
                                      double* lam, double* mu, int32_t* status, int32_t threads);

/* Same argument list as alilqr_linearize_batch with host pointers.  Extra
 * outputs (may be NULL): cval [B][H+1][c] constraint values, lam_new
 * [B][H+1][c] multipliers after one update_lambda(), xnorm [B][H+1][n] the
 * node state after the in-place angle wrap. */
int32_t oracle_linearize_batch(const alilqr_problem* p,
                               const double* x, const double* u,
                               const double* lam, const double* mu,
                               const double* goals, const double* Q, const double* R, const double* cons,
                               double* f, double* fx, double* fu, double* l,
                               double* lx, double* lu, double* lxx, double* luu,
                               double* cval, double* lam_new, double* xnorm);

/* Same argument list as alilqr_node_aux_batch with host pointers (cval and lam_new come from oracle_linearize_batch). */
int32_t oracle_node_aux_batch(const alilqr_problem* p,
                              const double* x, const double* u,
                              const double* lam, const double* mu,
                              const double* goals, const double* Q, const double* R, const double* cons,
                              double* viol, double* par_cost, double* fxx);

int32_t oracle_max_threads(void);

/* Ask the following oracle_solve_batch* calls to record, per solve, its decision margin in out[B] (NULL: stop):
 * the smallest relative distance between the two sides of any accept / reject / stop comparison the solve made. */
void oracle_set_margin_out(double* out);

#ifdef __cplusplus
}
#endif
#endif
