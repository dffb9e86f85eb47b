/*
 * alilqr.h — C ABI of the MI355X batched augmented-Lagrangian iLQR solve path.
 *
 * Everything is plain C: POD descriptors, raw pointers, int32 status codes.
 * All array arguments of the *_batch entry points are DEVICE pointers (HBM of
 * the GPU that is current on the calling thread); the descriptor itself is a
 * host struct.  Arrays at the boundary are row-major, one problem after the
 * other, exactly the shapes the reference solver returns:
 *     x (H+1, n), u (H, m)            cilqr/ilqr.py:16-17
 * The library transposes into its own lane-contiguous layout internally.
 *
 * Each entry point names the reference interface it stands in for.
 */
#ifndef ALILQR_H
#define ALILQR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALILQR_ABI_VERSION 6

/* ---- vehicle model (node type) ------------------------------------------
 * LAT      4 states / 1 control   cilqr/lat_bicycle_node.py:4-91
 * KIN      6 / 2                  cilqr/full_bicycle_kinematic_node.py:4-105
 * DYN      6 / 2, heading row of f_x evaluated at v      cilqr/full_bicycle_dynamic_node.py:61-108
 * DYN_MID  6 / 2, heading row of f_x evaluated at v_mid  cilqr/fast_bicycle_node.py:56-139
 *                 (same map as al_ilqr_cpp/model/new_bicycle_node.h:518-542)
 * LAT_K    4 / 1 with the (1+k v^2) understeer factor    al_ilqr_cpp/model/new_lat_bicycle_node.h:27-163
 * NEW_BICYCLE 6 / 2 of the C++ generation                 al_ilqr_cpp/model/new_bicycle_node.h:228-650
 *          (same map as DYN_MID; its f_x heading row d/dv term and its angle wrap differ)
 * LON      3 / 1 longitudinal triple integrator (s, v, a; jerk control), no angles   cilqr/lon_node.py:4-146
 * LAT_INTERIOR 4 / 1 lateral model whose steering rate is control_max * tanh(u); state bounds only (c = 2n), the
 *          state cost is x'Qx without the goal   cilqr/lat_bicycle_interior_node.py:4-174
 *          (the descriptor's k carries control_max[0])
 */
enum alilqr_model {
  ALILQR_MODEL_LAT = 0,
  ALILQR_MODEL_KIN = 1,
  ALILQR_MODEL_DYN = 2,
  ALILQR_MODEL_DYN_MID = 3,
  ALILQR_MODEL_LAT_K = 4,
  ALILQR_MODEL_NEW_BICYCLE = 5,
  ALILQR_MODEL_LON = 6,
  ALILQR_MODEL_LAT_INTERIOR = 7
};

/* ---- solver generation ---------------------------------------------------
 * LEGACY  ILQR      cilqr/ilqr.py:6-192       box AL built into the node (lambda>=0, I_mu mask)
 * FAST    FastILQR  cilqr/fast_ilqr.py:5-176  projection-form AL over a linear table (lambda<=0)
 * NEW_AL  NewALILQR al_ilqr_cpp/new_al_ilqr.h:52-1220
 */
enum alilqr_mode {
  ALILQR_MODE_LEGACY = 0,
  ALILQR_MODE_FAST = 1,
  ALILQR_MODE_NEW_AL = 2
};

/* ---- constraint table kind -------------------------------------------------
 * LINEAR     c(x,u) = A x + B u + C                 linear_constraints.py:13-29, al_ilqr_cpp/constraints/linear_constraints.h
 *            (LEGACY ignores the field: its table is the box  state_min | state_max | control_min | control_max)
 * QUADRATIC  c_i(x,u) = x' Qc_i x + A_i x + B_i u + C_i      al_ilqr_cpp/constraints/quadratic_constraints.h:6-98
 *            (NEW_AL only; the circle obstacles of test_new_al_ilqr.cc:208-266, test_pybind.py:284-348)
 */
enum alilqr_cons_kind {
  ALILQR_CONS_LINEAR = 0,
  ALILQR_CONS_QUADRATIC = 1,
  /* flag, OR-ed into the kind: the table holds EQUALITY rows c(x,u) = 0 (is_equality of the reference's
   * constraint classes: constraints.py:59-110, constraints.h:520-523,673-676,791-795,880-882): no projection of
   * lambda - mu c, multiplier update lambda <- lambda - mu c; FAST and NEW_AL */
  ALILQR_CONS_EQUALITY = 2
};

enum alilqr_status {
  ALILQR_OK = 0,
  ALILQR_E_ARG = -1,       /* bad descriptor / null pointer / unsupported combination */
  ALILQR_E_HIP = -2,       /* a HIP runtime call failed; see alilqr_last_error() */
  ALILQR_E_NOMEM = -3,     /* workspace allocation failed */
  ALILQR_E_NODEVICE = -4   /* no usable GPU */
};

/* bits of status word 2 (per problem) */
#define ALILQR_FLAG_VIOL_OK   1u   /* stopped because violation < viol_stop           */
#define ALILQR_FLAG_OUTER_CAP 2u   /* ran out of outer iterations                     */
#define ALILQR_FLAG_NONFINITE 4u   /* final cost is NaN/Inf (the reference is silent) */
#define ALILQR_FLAG_STOPPED   8u   /* ended by alilqr_extras::stop_after, not by a stop rule of the reference */

/* Solver constants.  The reference hard-codes these as literals; the defaults
 * returned by alilqr_default_params() are those literals, per generation:
 *   cilqr/ilqr.py:27,92,117,151-174 ; cilqr/fast_ilqr.py:26,91,116,136-159 ;
 *   al_ilqr_cpp/new_al_ilqr.h:668,1039,1045,1118,1188,1198-1211               */
typedef struct alilqr_params {
  int32_t max_outer;     /* outer (AL) iterations                         20 / 20 / arg */
  int32_t max_inner;     /* inner (iLQR) iterations per outer             20 / 20 / arg */
  double inner_tol;      /* |J_new - J_old| < tol stops the inner loop    1e-8 / 1e-8 / 1e-3 */
  double viol_stop;      /* violation below this stops the outer loop     1e-3 / 1e-3 / arg */
  double viol_lambda;    /* below: multiplier update, above: penalty x gain   1e-1 / 1e-1 / 5*viol_stop */
  double mu_gain;        /* 8 / 8 / 100 */
  double quu_reg;        /* added to diag(Quu) before inversion           1e-9 / 0 / 0 */
  double alpha_min;      /* line search gives up when alpha <= this       1e-8 / 1e-8 / (10 halvings) */
  double init_r;         /* control weight of the initial-guess LQR       20 */
  double dv_gate;        /* NEW_AL only: skip the roll-out when |dV_lin| < gate   (double)0.2F */
} alilqr_params;

/* One batch of independent problems that share model, horizon and tables. */
typedef struct alilqr_problem {
  int32_t abi_version;   /* ALILQR_ABI_VERSION */
  int32_t model;         /* enum alilqr_model */
  int32_t mode;          /* enum alilqr_mode  */
  int32_t horizon;       /* H: number of controls; H+1 nodes */
  int32_t n, m, c;       /* state, control, constraint-row counts (LEGACY: c = 2(n+m); LAT_INTERIOR: c = 2n) */
  int32_t batch;         /* B problems */
  int32_t goal_batch;    /* 1: all problems track the same goals; B: one goal list each */
  int32_t table_nodes;   /* 1: Q,R,constraint table shared by all nodes; H+1: one per node */
  int32_t cons_kind;     /* enum alilqr_cons_kind */
  int32_t reserved;      /* 0 */
  double L, dt, v, k;    /* wheel base, step, (lat) speed, understeer factor (LAT_INTERIOR: control_max) */
  alilqr_params prm;
} alilqr_problem;

/* Per-problem status: [0] inner iterations (backward passes) run, [1] outer
 * iterations run, [2] ALILQR_FLAG_* bits, [3] line-search roll-outs run. */
#define ALILQR_STATUS_WORDS 4

/* ABI version of the loaded library (== ALILQR_ABI_VERSION of its build). */
int32_t alilqr_abi_version(void);

/* Text of the last failure on the calling thread ("" if none). */
const char* alilqr_last_error(void);

/* Number of HIP devices visible, or a negative alilqr_status. */
int32_t alilqr_device_count(void);

/* Fill *out with the reference literals of the given generation
 * (cilqr/ilqr.py:151-174, cilqr/fast_ilqr.py:136-159, new_al_ilqr.h:1177-1211). */
int32_t alilqr_default_params(int32_t mode, alilqr_params* out);

/* Upper bound of the bytes of device scratch a solve of this descriptor will hold. */
int64_t alilqr_workspace_bytes(const alilqr_problem* p);

/* Lanes per problem (1, 4, 8, 16 or 32) a solve of this descriptor will run with, or a negative alilqr_status.  For
 * alilqr_solve_batches pass a descriptor whose batch is the total over the call.  Diagnostic (profiling tools run a
 * single batch at the width a many-batch call uses); no counterpart in the reference. */
int32_t alilqr_lanes_for(const alilqr_problem* p);

/*
 * Solve a batch.  Stands in for
 *   ILQR(nodes).optimize(max_iters, tol)        cilqr/ilqr.py:151-175
 *   FastILQR(nodes).optimize(max_iters, tol)    cilqr/fast_ilqr.py:136-160
 *   NewALILQR<n,m>::optimize(outer, inner, tol) al_ilqr_cpp/new_al_ilqr.h:1177-1220
 *                                               (bound at al_ilqr_cpp/ilqr_pybind.cc:25-33)
 * applied to every problem of the batch.
 *
 *  in   x0     [B][n]                    nodes[0].state
 *       goals  [goal_batch][H+1][n]      node.goal
 *       Q      [table_nodes][n][n]       node.Q
 *       R      [table_nodes][m][m]       node.R
 *       cons   [table_nodes][...]        LEGACY: state_min[n] state_max[n] control_min[m] control_max[m]
 *                                        FAST / NEW_AL: A[c][n] B[c][m] C[c]  (c(x,u) = A x + B u + C <= 0)
 *                                        NEW_AL, cons_kind QUADRATIC: A[c][n] B[c][m] C[c] Qc[c][n][n]
 *  out  x_out  [B][H+1][n], u_out [B][H][m]      the (x, u) pair optimize() returns
 *       node_x [B][H+1][n], node_u [B][H][m]     node.state / node.control after optimize() (may be NULL)
 *       x_init [B][H+1][n], u_init [B][H][m]     the initial guess optimize() also returns (may be NULL)
 *       cost   [B]                        compute_total_cost() after optimize()
 *       viol   [B]                        last constraint violation measure (may be NULL)
 *       lam    [B][H+1][c], mu [B]        final multipliers / penalty (may be NULL)
 *       status [B][ALILQR_STATUS_WORDS]   (may be NULL)
 *  stream      hipStream_t to enqueue on (NULL = default stream); the call
 *              returns after enqueueing, outputs are valid once the stream drains.
 *              Scratch comes from a pool of the library (grown with hipMalloc on first use of a size): solves
 *              on different streams get different scratch and may overlap on the GPU, solves on one stream
 *              run in order; any number of streams may be used.
 * Returns ALILQR_OK or a negative alilqr_status; never throws.
 */
int32_t alilqr_solve_batch(const alilqr_problem* p,
                           const double* x0, const double* goals,
                           const double* Q, const double* R, const double* cons,
                           double* x_out, double* u_out,
                           double* node_x, double* node_u,
                           double* x_init, double* u_init,
                           double* cost, double* viol,
                           double* lam, double* mu,
                           int32_t* status, void* stream);

/*
 * Multi-start variant of alilqr_solve_batch: u_offset [B][H][m] (device pointer, may be NULL) is added to the
 * controls of the linearised-LQR initial guess while it is rolled out,
 *     u_t = -K_t (x_t - goal_t) + u_offset_t ,  x_{t+1} = f(x_t, u_t)
 * (the roll-out of cilqr/ilqr.py:41-47, cilqr/fast_ilqr.py:40-46, al_ilqr_cpp/new_al_ilqr.h:679-690 with a
 * perturbed control), so that the B problems can be B different starts of the same scenario.  The reference has
 * no multi-start driver; everything after the initial guess is alilqr_solve_batch unchanged, and with
 * u_offset == NULL the two calls are the same call.
 */
int32_t alilqr_solve_batch_multistart(const alilqr_problem* p,
                                      const double* x0, const double* goals,
                                      const double* Q, const double* R, const double* cons,
                                      const double* u_offset,
                                      double* x_out, double* u_out,
                                      double* node_x, double* node_u,
                                      double* x_init, double* u_init,
                                      double* cost, double* viol,
                                      double* lam, double* mu,
                                      int32_t* status, void* stream);

/*
 * Optional inputs of alilqr_solve_batch_ex (a host struct; the pointers in it are device pointers).
 *
 * Rectangle obstacles (NEW_AL only) reproduce the row activation of NewALILQR::UpdateConstraints
 * (al_ilqr_cpp/new_al_ilqr.h:525-625, constructor :134-208): at the top of every inner iteration, for every node
 * whose (x, y) lies strictly inside a left rectangle the row  -y + y_max <= 0  is written into the node's own copy
 * of the constraint table, for a right rectangle the row  y - y_min <= 0  (y_max / y_min = extreme y of the four
 * vertices).  The write follows the reference's table classes: skipped when any C entry of the node's table
 * already equals the new C; linear / box tables advance the node's row index and write there
 * (linear_constraints.h:360-373, box_constraints.h:80-92); quadratic tables overwrite A and C of the row at
 * the index without advancing it (quadratic_constraints.h:57-64); the B (and Qc) part of the row is left as it is.
 * One deviation: where the reference would advance the index past the last row (undefined behaviour on its
 * fixed-size matrices) the row is dropped.
 * Rectangles are [n][2][4]: the x coordinates of the four vertices, then their y coordinates, vertices in
 * counter-clockwise order (the inside test is "all four edge cross products < 0"), shared by the whole batch.
 */
typedef struct alilqr_extras {
  const double* u_offset;    /* [B][H][m] or NULL: see alilqr_solve_batch_multistart */
  const double* left_obs;    /* [n_left][2][4] or NULL */
  const double* right_obs;   /* [n_right][2][4] or NULL */
  int32_t n_left, n_right;   /* each 0..16 */
  int32_t first_row;         /* initial row index of every node's table: set_current_constraints_index()
                                (constraints.h:189-196), 0 if never called */
  int32_t reserved;          /* 0 */
  /* Optional per-iteration trace (both NULL: none): what the reference prints per inner iteration
   * (cilqr/ilqr.py:157-167, fast_ilqr.py:142-152; new_al_ilqr.h:1100-1121): entry i of problem b holds the cost after
   * inner iteration i (counted over the whole solve) and the number of roll-outs its line search made.  Iterations
   * beyond trace_cap are not recorded; entries beyond status[0] are left untouched. */
  double* trace_cost;        /* [B][trace_cap] or NULL */
  int32_t* trace_trials;     /* [B][trace_cap] or NULL */
  int32_t trace_cap;         /* entries per problem */
  int32_t reserved2;         /* 0 */
  /* State a NewALILQR object carries from one optimize() call into the next (NEW_AL only; all optional).
   * mu_solver_in / mu_solver_out [B]: the solver's own penalty mu_ (new_al_ilqr.h:1125-1130).  optimize() resets the
   *   nodes' penalty to 1 (linearizedInitialGuess, :692-695) but not mu_, so the first penalty increase of a later
   *   call sets the nodes to (mu_ left by the earlier calls) x gain.  NULL in: 1.
   * first_rows [H+1]: the row index each node's table starts from (overrides first_row); after a solve with
   *   rectangle obstacles the nodes of the reference hold different indices.
   * overlay_out [B][H+1][2c+1]: per node, what UpdateConstraints wrote into its table (linear_constraints.h:360-373):
   *   c row signs (0: row untouched; -1 / +1: the row became  sign * y + B_i u + C'_i <= 0), c constants C'_i, and
   *   the node's final row index. */
  const double* mu_solver_in;
  double* mu_solver_out;
  const int32_t* first_rows;
  double* overlay_out;
  /* Diagnostic (no counterpart in the reference): stop_after [B] or NULL.  Problem b with stop_after[b] = k > 0 ends
   * right after its k-th inner iteration (counted over the whole solve, as the trace counts them): no violation
   * measure, multiplier or penalty update follows.  node_x / node_u / cost are then the nodes' trajectory and its cost
   * after that iteration's forward() (cilqr/ilqr.py:160, fast_ilqr.py:145, new_al_ilqr.h:1104), x_out / u_out the pair
   * optimize() would hold at that point, status[1] the outer iterations completed before, status[2] carries
   * ALILQR_FLAG_STOPPED.  A solve that ends earlier by its own rules is not affected.  Lets a test compare the
   * iterate of a chosen iteration with the reference's instead of only the end of the solve. */
  const int32_t* stop_after;
  /* Scheduling hint (no counterpart in the reference): order [B] or NULL, a permutation of 0 .. B-1.  The resident
   * lane groups pull the problems in this order instead of 0, 1, 2 ...  Results do not depend on it.  A batch ends
   * when its last solve ends; with the problems that take longest first (e.g. by the inner-iteration counts,
   * status[0], the same problems needed in the previous planning cycle) the short solves fill the end of the batch
   * instead of one long solve running alone.  An entry outside 0 .. B-1 is replaced by its own position. */
  const int32_t* order;
} alilqr_extras;

/* alilqr_solve_batch with the optional inputs above (ex may be NULL). */
int32_t alilqr_solve_batch_ex(const alilqr_problem* p,
                              const double* x0, const double* goals,
                              const double* Q, const double* R, const double* cons,
                              const alilqr_extras* ex,
                              double* x_out, double* u_out,
                              double* node_x, double* node_u,
                              double* x_init, double* u_init,
                              double* cost, double* viol,
                              double* lam, double* mu,
                              int32_t* status, void* stream);

/*
 * Parallel-in-time linearisation / quadratisation of N = B*(H+1) independent
 * (node, state, control) points: the per-node calls the reference makes at
 * the top of its backward pass (cilqr/ilqr.py:70-77, cilqr/fast_ilqr.py:69-76):
 *   node.dynamics(x,u), node.dynamics_jacobian(x,u), node.cost(),
 *   node.cost_jacobian(), node.cost_hessian().
 *  in   x [B][H+1][n], u [B][H+1][m] (terminal row = the node's control, normally 0),
 *       lam [B][H+1][c], mu [B], goals/Q/R/cons as in alilqr_solve_batch
 *  out  f [B][H+1][n], fx [B][H+1][n][n], fu [B][H+1][n][m], l [B][H+1],
 *       lx [B][H+1][n], lu [B][H+1][m], lxx [B][H+1][n][n], luu [B][H+1][m][m]
 *       (any output may be NULL)
 */
int32_t alilqr_linearize_batch(const alilqr_problem* p,
                               const double* x, const double* u,
                               const double* lam, const double* mu,
                               const double* goals, const double* Q, const double* R, const double* cons,
                               double* f, double* fx, double* fu, double* l,
                               double* lx, double* lu, double* lxx, double* luu,
                               void* stream);

/*
 * The remaining per-node calls of the reference's node / constraint objects, on B*(H+1) independent points
 * (inputs as in alilqr_linearize_batch; every output may be NULL):
 *   cval    [B][H+1][c]       node.constraints() / constraints_obj.constrains(x,u) / Constraints::constraints(x,u)
 *                             (lat_bicycle_node.py:127-133, linear_constraints.py:13-16, linear_constraints.h:176-181,
 *                              quadratic_constraints.h:15-27)
 *   lam_new [B][H+1][c]       multipliers after one update_lambda() at the point
 *                             (lat_bicycle_node.py:162-165, constraints.py:106-110, constraints.h:878-887)
 *   viol    [B][H+1]          the node's term of the violation measure: || max(-lam/mu, c) ||_2 (ilqr.py:186-192),
 *                             max_violation / max_constraints_violation = || max(c, 0) ||_inf
 *                             (constraints.py:115-119, constraints.h:1027-1040)
 *   par_cost [B][H+1]         NEW_AL only: one column of NewILQRNode::parallel_cost (new_bicycle_node.h:768-795,
 *                             constraints.h:558-580)
 *   fxx     [B][H+1][3][n][n] NEW_AL only: dynamics_hessian_fxx, the Hessians of the first three state rows
 *                             (new_bicycle_node.h:605-650, new_lat_bicycle_node.h:128-163)
 */
int32_t alilqr_node_aux_batch(const alilqr_problem* p,
                              const double* x, const double* u,
                              const double* lam, const double* mu,
                              const double* goals, const double* Q, const double* R, const double* cons,
                              double* cval, double* lam_new, double* viol, double* par_cost, double* fxx,
                              void* stream);

/*
 * The calls of the reference's constraint objects (the Linear / Box / Quadratic classes of ilqr_pybind,
 * al_ilqr_cpp/constraints/constraints_bind.h:47-50, 88-91; FAST / NEW_AL tables) on B*(H+1) independent points
 * (inputs as in alilqr_linearize_batch; every output may be NULL):
 *   cval     [B][H+1][c]      constraints(x, u)            linear_constraints.h:176-181, quadratic_constraints.h:15-27
 *   par_cval [B][H+1][c]      one column of parallel_constraints(X, U): the quadratic term is added last
 *                             (linear_constraints.h:218-224, quadratic_constraints.h:29-45)
 *   jac_x    [B][H+1][c][n]   constraints_jacobian(x, u).first  = A (+ 2 Qc_i x per row for quadratic tables)
 *   jac_u    [B][H+1][c][m]   constraints_jacobian(x, u).second = B    (linear_constraints.h:260-265, quadratic_constraints.h:47-55)
 */
int32_t alilqr_constraints_batch(const alilqr_problem* p,
                                 const double* x, const double* u,
                                 const double* lam, const double* mu,
                                 const double* goals, const double* Q, const double* R, const double* cons,
                                 double* cval, double* par_cval, double* jac_x, double* jac_u,
                                 void* stream);

/*
 * Many batches in one call.  A caller that serves independent requests holds one set of buffers per request (per
 * planning cycle, per client); solving them one alilqr_solve_batch at a time leaves the GPU idle while the longest
 * solve of each batch finishes, unless the caller keeps many streams busy.  This entry point takes n_batches batches
 * of p->batch problems each - same model, generation, horizon, goals and tables; start states and outputs in buffers
 * of their own - and solves them as ONE batch: the resident lane groups pull problems of all batches from one queue,
 * so a batch's stragglers overlap with the other batches' work.  Every problem gets the bits alilqr_solve_batch gives
 * it.  The reference has no batch driver; per problem this is optimize() as in alilqr_solve_batch.
 *   x0, x_out, u_out, cost            HOST arrays of n_batches DEVICE pointers: [B][n], [B][H+1][n], [B][H][m], [B]
 *   node_x, node_u, viol, mu, status  the same, optional: the list, or single entries of it, may be NULL
 *   goals, Q, R, cons                 device pointers shared by all batches (p->goal_batch must be 1)
 * The start states are gathered into, and the results scattered from, staging arrays inside the library's scratch by
 * small copy kernels on `stream` (one launch per array; 2.1 KB per lateral problem: microseconds per batch).  They are
 * ordinary kernels: while ANOTHER call's solve keeps every SIMD occupied they wait for its waves to finish, so results
 * of overlapping calls become visible late; a latency-sensitive caller keeps one call in flight.  Asynchronous like
 * alilqr_solve_batch: outputs are valid once the stream drains; the pointer lists are read before the call returns.
 *   ex     optional inputs as in alilqr_solve_batch_ex (may be NULL), over the problems of all batches in one index
 *          space (problem b of batch i is i * B + b): order [n_batches * B] (the pull order, see alilqr_extras::order),
 *          u_offset [n_batches * B][H][m], stop_after, trace_*, mu_solver_*: one contiguous device array each; rectangle
 *          obstacles and first_row(s) are shared by all problems as always.  overlay_out is not supported here.
 */
int32_t alilqr_solve_batches(const alilqr_problem* p, int32_t n_batches,
                             const double* const* x0, const double* goals,
                             const double* Q, const double* R, const double* cons,
                             double* const* x_out, double* const* u_out,
                             double* const* node_x, double* const* node_u,
                             double* const* cost, double* const* viol, double* const* mu,
                             int32_t* const* status, const alilqr_extras* ex, void* stream);

/* Release every cached device workspace of the calling process (waits for the solves in flight). */
int32_t alilqr_release(void);

/*
 * alilqr_solve_batch for callers that cannot allocate: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  It has no counterpart in the reference.  Arguments and results are those of
 * alilqr_solve_batch (bit-identical outputs); the scratch is a static device array of
 * ALILQR_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends, and the call makes
 * no hipMalloc / hipFree.  The solve keeps as many problems resident as fit into that array (the others are
 * pulled as resident ones finish), so any batch size is accepted; ALILQR_E_NOMEM is returned only when the scratch
 * of a single wave (a very long horizon) exceeds the array.  All calls on one GPU share the array: calls that
 * overlap on one GPU (different streams, or threads) race - enqueue them on one stream.  Every call enqueues one
 * 256-byte hipMemsetAsync (the problem counter) in front of the kernel(s).  The first call on a device looks the
 * array's address up (hipGetSymbolAddress): make that call outside graph capture.  Later calls may be captured into a
 * HIP graph, and the graph may be replayed: the enqueued work carries no state of the host.
 * The array's size is fixed when the library is built (-DALILQR_STATIC_SCRATCH_BYTES=<bytes>, default 128 MiB).
 */
#ifndef ALILQR_STATIC_SCRATCH_BYTES
#define ALILQR_STATIC_SCRATCH_BYTES (128u << 20)
#endif
int32_t alilqr_solve_batch_noalloc(const alilqr_problem* p,
                                   const double* x0, const double* goals,
                                   const double* Q, const double* R, const double* cons,
                                   double* x_out, double* u_out,
                                   double* node_x, double* node_u,
                                   double* x_init, double* u_init,
                                   double* cost, double* viol,
                                   double* lam, double* mu,
                                   int32_t* status, void* stream);

/* Size of the static scratch of alilqr_solve_batch_noalloc, in bytes (no counterpart in the reference). */
int64_t alilqr_static_scratch_bytes(void);

/*
 * Tuning / diagnostic overrides of the whole process (no counterpart in the reference); 0 = automatic.
 *   lanes      lanes per problem (1, 4, 8, 16, 32)
 *   max_waves  upper bound of resident waves per solve
 *   no_lds     bit 0: keep the linearisation records (and accumulated controls) in HBM even where they fit in LDS;
 *              bit 1: evaluate every constraint row of FAST / NEW_AL tables densely (no box-row short cut)
 * Results do not depend on any of them (tests compare the bits).
 */
int32_t alilqr_set_tuning(int32_t lanes, int32_t max_waves, int32_t no_lds);

#ifdef __cplusplus
}
#endif
#endif /* ALILQR_H */
