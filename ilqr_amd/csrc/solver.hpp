// solver.hpp — the batched AL-iLQR solve kernel for gfx950.
//
// Mapping.  One problem is solved by a *lane group* of G lanes (G = 1..64, a
// power of two, all inside one 64-lane wavefront).  A group is persistent: it
// pulls problem indices from a global counter until the batch is drained, so
// the device workspace is sized by the number of resident groups, not by the
// batch.  Inside one inner iLQR iteration the G lanes are used three ways:
//
//   P1  linearise/quadratise   parallel in time: lane g takes nodes t = g, g+G, ...
//   P2  backward Riccati       sequential in t; every lane of the group carries
//                              the same (Vx, Vxx) in registers, lane 0 stores K, k
//   P3  forward roll-outs      parallel over line-search trials: lane g rolls out
//                              trial r*G+g+1 of round r; the first (largest
//                              step) accepted trial wins, as in the sequential
//                              reference
//
// All groups of a wave run the same flat loop (one inner iteration per trip),
// so groups that are in different outer iterations, or on different problems,
// still execute the heavy phases together.
//
// Data layout in HBM (fp64).
//   blk  one block per group: H+1 step records  K_t | k_t | lambda_t
//        followed by H+1 linearisation records  f_x, f_u entries | l_x | l_u | l_xx | l_uu
//        (the linearisation records live in LDS instead where they fit at full occupancy).
//        G > 1: the block is contiguous (problem-major), so a sequential sweep reads one
//        short run of doubles per step from one base address; G = 1: element-major /
//        group-minor, so the 64 problems of a wave read one contiguous run per element.
//   trj  candidate trajectories, SS slots per problem and parity, kept only for the trials that usually
//        win (any other winner is rolled out a second time).  G > 1: [group][2][SS][T], contiguous per
//        roll-out; G = 1: [2][T][NG*SS], lane-minor.
//   ucum [group][H*M][trial - 2]   LEGACY: accumulated controls of the open-loop trials 2 .. ntrials; the
//        lanes of a roll-out pass read one contiguous run per step.
// The current trajectory of a problem is the pair (slot, parity) of the last
// accepted roll-out; nothing is copied on acceptance.  Sequential sweeps (P2, P3)
// load the record of the next step before computing the current one.
//
// Reference control flow restated here: cilqr/ilqr.py:11-192 (LEGACY, including
// the array aliasing of forward() after its first rejected step, :137-138) and
// cilqr/fast_ilqr.py:10-176 (FAST).
#pragma once
#include <cstddef>
#include <type_traits>
#include "models.hpp"

namespace alilqr {

// Debug builds (-DALILQR_DEBUG_BOUNDS): every computed workspace offset is checked before it is used.  A violation
// sets bit `code` of a word in the scratch's counter block (read back by alilqr_debug_read) and the offset is
// clamped, so a bad index is reported instead of faulting the GPU.
#ifdef ALILQR_DEBUG_BOUNDS
#define ALILQR_CHK(cond, code) ((cond) ? true : (atomicOr(ALILQR_COLD(counter) + 8, 1u << (code)), false))
#else
#define ALILQR_CHK(cond, code) (true)
#endif

#ifdef ALILQR_PROF
#define PROF_DECL unsigned long long prof_t0 = __builtin_amdgcn_s_memtime(), prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_s0 = 0, prof_roll[2] = {0, 0}, prof_p0 = 0, prof_p3[5] = {0, 0, 0, 0, 0};
#define PROF_MARK(i) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof_acc[i] += t_ - prof_t0; prof_t0 = t_; }
// a roll-out pass: PASS_START / PASS_END around it (cycles inside roll-outs, number of passes)
#define PASS_START { prof_roll[1] += 1; prof_s0 = __builtin_amdgcn_s_memtime(); }
#ifdef ALILQR_PROF_DRAIN   /* wait for every outstanding memory operation inside the measured region */
#define PASS_END { __builtin_amdgcn_s_waitcnt(0x0F70); prof_roll[0] += __builtin_amdgcn_s_memtime() - prof_s0; }
#else
#define PASS_END { prof_roll[0] += __builtin_amdgcn_s_memtime() - prof_s0; }
#endif
// -DALILQR_PROF_P3: the roll-out phase in parts.  The accumulators are updated inside the (divergent) line-search loop,
// so each lane keeps its own; lane 0's are reported: "last pass -> P3 end" is the time the wave spends on passes of
// other groups after group 0's search has ended.
#ifdef ALILQR_PROF_P3
#define P3_STAMP(i) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof_p3[i] += t_ - prof_p0; prof_p0 = t_; }
#define P3_START { __builtin_amdgcn_s_waitcnt(0); prof_p0 = __builtin_amdgcn_s_memtime(); }
#else
#define P3_STAMP(i)
#define P3_START
#endif
#define PROF_FLUSH(a) { unsigned long long* const pr_ = ALILQR_COLD(prof); if (lane == 0 && pr_) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(pr_ + i_, prof_acc[i_]); atomicAdd(pr_ + 8, 1ull); atomicAdd(pr_ + 9, prof_roll[0]); atomicAdd(pr_ + 10, prof_roll[1]); for (int i_ = 0; i_ < 5; ++i_) atomicAdd(pr_ + 11 + i_, prof_p3[i_]); } }
#else
#define PROF_DECL
#define PROF_MARK(i)
#define P3_STAMP(i)
#define P3_START
#define PASS_START
#define PASS_END
#define PROF_FLUSH(a)
#endif

// Read-only tables (Q, R, constraint tables) are addressed through the constant address space so that
// their uniform reads become scalar loads held in SGPRs instead of per-lane vector loads.
typedef const double __attribute__((address_space(4)))* ctab;
__device__ __forceinline__ ctab as_ctab(const double* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (ctab)p;
#pragma clang diagnostic pop
}
// The same tables staged in LDS (read by every lane at one address: a broadcast, no bank conflict).
typedef const double __attribute__((address_space(3)))* ltab;
__device__ __forceinline__ ltab as_ltab(const double* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (ltab)p;
#pragma clang diagnostic pop
}
// NEW_AL line search: the first columns of the ladder in the idle lanes of the last round of trials
#ifndef ALILQR_LADDER_AHEAD
#define ALILQR_LADDER_AHEAD 1
#endif
#ifndef ALILQR_TAB_LDS
#define ALILQR_TAB_LDS 1
#endif
// FAST groups of ALILQR_FAST_KEEP_G lanes and fewer keep the trajectories of the first ALILQR_FAST_KEEP trials only (0: of all);
// groups of ALILQR_FAST_TWO_G lanes and fewer form a chunk's new multipliers before its stores
#ifndef ALILQR_FAST_KEEP
#define ALILQR_FAST_KEEP 2
#endif
#ifndef ALILQR_FAST_TWO_G
#define ALILQR_FAST_TWO_G 4
#endif
#ifndef ALILQR_FAST_KEEP_G
#define ALILQR_FAST_KEEP_G 8
#endif
#ifndef ALILQR_WAVE_COPY
#define ALILQR_WAVE_COPY 1
#endif
#ifndef ALILQR_LIVE_LDS
#define ALILQR_LIVE_LDS 1
#endif
#ifndef ALILQR_MERGE_KINDS
#define ALILQR_MERGE_KINDS 1
#endif
// the value lane `l` of the wave holds (l the same in every lane).  Host builds run the groups of a wave one after the
// other: only the own group's lane 0 is ever asked for there, and the caller's own value is that value.
__device__ __forceinline__ int wave_int(int v, int l) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_readlane(v, l);
#else
  (void)l; return v;
#endif
}
__device__ __forceinline__ const double* wave_ptr(const double* p, int l) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)a, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(a >> 32), l);
  return (const double*)(((unsigned long long)hi << 32) | lo);
#else
  (void)l; return p;
#endif
}
// a value that is the same in every lane, as a scalar (host builds: the value)
__device__ __forceinline__ unsigned long long wave_uniform(unsigned long long m) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)m), hi = __builtin_amdgcn_readfirstlane((unsigned int)(m >> 32));
  return ((unsigned long long)hi << 32) | lo;
#else
  return m;
#endif
}
// orders the LDS writes of a wave's lanes before the LDS reads of its other lanes (one wave per block)
__device__ __forceinline__ void wave_lds_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// Cold kernel arguments.  solve_kernel takes one argument block (KArgs) of ~60 scalar registers' worth of pointers and
// constants.  A field that is named as a.field is loaded once at kernel entry and then held for the whole solve - or,
// with every scalar register taken, parked in lanes of a vector register and fetched back (v_readlane) inside the
// step loops.  Most fields are needed once per problem or once per inner iteration (outputs, optional inputs, the
// solver constants of the bookkeeping phase): ALILQR_COLD(field) reads such a field from the argument block in the
// constant (kernarg) segment at the place of use - one scalar load - and nothing stays live in between.  The empty asm
// makes the address opaque, so the load is not hoisted back to the kernel entry.  The argument block is the kernel's
// only parameter and therefore starts at offset 0 of the segment.  (Host builds of this source read a.field.)
template <class T>
__device__ __forceinline__ T cold_arg(size_t offset) {
#if defined(__HIP_DEVICE_COMPILE__)
  const char __attribute__((address_space(4)))* base =
      (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(base));
  return *(const T __attribute__((address_space(4)))*)(base + offset);
#else
  (void)offset;
  return T{};
#endif
}
// ALILQR_COLD: fields needed once per problem (outputs, optional inputs); ALILQR_COLD2: once per inner iteration (the
// constants of the bookkeeping phase, the trace); ALILQR_COLD3: the rectangle lists of the row activation.
#ifndef ALILQR_COLD_LEVEL
#define ALILQR_COLD_LEVEL 0
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define ALILQR_COLD_(f) (alilqr::cold_arg<decltype(alilqr::KArgs::f)>(offsetof(alilqr::KArgs, f)))
#else
#define ALILQR_COLD_(f) (a.f)
#endif
#if ALILQR_COLD_LEVEL >= 1
#define ALILQR_COLD(f) ALILQR_COLD_(f)
#else
#define ALILQR_COLD(f) (a.f)
#endif
#if ALILQR_COLD_LEVEL >= 2
#define ALILQR_COLD2(f) ALILQR_COLD_(f)
#else
#define ALILQR_COLD2(f) (a.f)
#endif
#if ALILQR_COLD_LEVEL >= 3
#define ALILQR_COLD3(f) ALILQR_COLD_(f)
#else
#define ALILQR_COLD3(f) (a.f)
#endif

#ifdef ALILQR_LIT_LAT
__constant__ const double kLitQ[16] = {1e-3, 0, 0, 0, 0, 1e-1, 0, 0, 0, 0, 1e1, 0, 0, 0, 0, 1e-9};
__constant__ const double kLitR[1] = {50.0};
__constant__ const double kLitB[10] = {-1000, -1000, -2 * 3.141592653589793, -10.0, 1000, 1000, 2 * 3.141592653589793, 10.0, -0.1, 0.1};
#endif

// ---------------------------------------------------------------------------
// Box tables (FAST / NEW_AL).  The table of BoxConstraints (box_constraints.h:35-78, box_constrains.py:5-32) is
//     [ I ; -I ; 0 ; 0 ] x + [ 0 ; 0 ; I ; -I ] u + [ -x_max ; x_min ; -u_max ; u_min ]
// and the dense row product  A_i x + B_i u  is seven multiplications by zero and one by +-1.  A one-wave kernel
// (analyze_rows_kernel, alilqr_api.hip) checks once per solve whether the first 2(n+m) rows of the table are exactly
// that; the node code then evaluates those rows as  (0.0 +- z) + C_i  with the column known at compile time.  That is
// the value of the dense sums  ax = 0.0; ax += A_i0 * x_0; ...  for finite x, u: every other term is +-0, adding +-0
// changes neither a non-zero sum nor +0, 1.0 * z is z and -1.0 * z is -z.  (A non-finite x_j makes 0 * x_j a NaN in
// the dense sum: nodes with a non-finite state or control take the dense code.)  Rows behind the box stay dense.
// ---------------------------------------------------------------------------
template <int N, int M>
struct BoxRow {
  int i;
  __device__ __forceinline__ constexpr bool is_u() const { return i >= 2 * N; }
  __device__ __forceinline__ constexpr int col() const { return is_u() ? (i - 2 * N) % M : i % N; }
  __device__ __forceinline__ constexpr bool neg() const { return is_u() ? (i - 2 * N) >= M : i >= N; }
  // A_i x + B_i u of the row
  __device__ __forceinline__ double dot(const double* x, const double* u) const {
    const double z = is_u() ? u[col()] : x[col()];
    return neg() ? (0.0 - z) : (0.0 + z);
  }
  __device__ __forceinline__ constexpr double sign() const { return neg() ? -1.0 : 1.0; }
};
// 0.0 exactly when every entry is finite, NaN otherwise
template <int N, int M>
__device__ __forceinline__ bool all_finite(const double* x, const double* u) {
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j) s += x[j] - x[j];
#pragma unroll
  for (int j = 0; j < M; ++j) s += u[j] - u[j];
  return s == 0.0;
}

// doubles of one group's share of the LDS pool: its H+1 linearisation records of rec doubles, or (later in the same
// iteration, in the same bytes) uc doubles of accumulated controls per step
__host__ __device__ inline size_t lds_group_doubles(int H, int rec, int uc) {
  const size_t lin = (size_t)(H + 1) * rec, ucd = (size_t)H * uc;
  return lin > ucd ? lin : ucd;
}

struct KArgs {
  int B, NG, H, C, goal_batch, table_nodes;
  int uc_cols;                // LEGACY with the accumulated controls in LDS: open-loop trials kept there (3 .. ntrials)
  const unsigned int* box_rows;  // FAST / NEW_AL: device word = 2(n+m) when the table starts with the box rows, else 0; or null
  int qdiag;                  // every Q table is diagonal (set in the kernel prologue)
  int eq;                     // equality table: no projection of lambda - mu c (ALILQR_CONS_EQUALITY)
  int quad;                   // constraint tables carry per-row quadratic terms (NEW_AL, ALILQR_CONS_QUADRATIC)
  unsigned long long quad_rows;   // rows whose quadratic matrix is not all zero, over all tables (kernel prologue)
  int sparse_rows;            // host: the one shared linear table may be searched for all-zero A / B rows
  unsigned long long azero, bzero;   // NEW_AL: rows whose A (B) entries are all zero (kernel prologue; 0 = none known)
  int rows_lds;               // NEW_AL: doubles of a node's multiplier block the roll-outs copy to LDS one step ahead (launch)
  int live_rows;              // host: roll-outs may skip rows that cannot contribute (all-zero table row, zero multiplier)
  unsigned long long zrows;   // NEW_AL: rows of the shared table that are all zero (kernel prologue; 0 = none known)
  ModelConst mc;
  alilqr_params prm;
  const double *x0, *goals, *Q, *R, *cons;
  const double *left_obs, *right_obs;   // rectangle obstacles [n][2][4] (NEW_AL row activation), or null
  int n_left, n_right, first_row;
  const double* u_off;        // optional [B][H][m] added to the controls of the initial guess (multi-start), or null
  double* trace_cost;         // optional [B][trace_cap]: cost after every inner iteration, or null
  int32_t* trace_trials;      // optional [B][trace_cap]: roll-outs of every inner iteration's line search, or null
  int trace_cap;
  const double* mu_s_in;      // NEW_AL, optional [B]: the solver-level penalty a previous optimize() left (null: 1)
  double* mu_s_out;           // NEW_AL, optional [B]: the solver-level penalty after this solve
  const int32_t* first_rows;  // RECT, optional [H+1]: per-node initial row index (null: first_row)
  double* overlay_out;        // RECT, optional [B][H+1][2C+1]: row signs | replacement constants | row index per node
  const int32_t* stop_after;  // optional [B]: inner iteration after which a problem ends (diagnostic), or null
  const int32_t* order;       // optional [B]: permutation in which the problems are pulled (scheduling hint), or null
  double *x_out, *u_out, *node_x, *node_u, *x_init, *u_init, *cost, *viol, *lam, *mu;
  int32_t* status;
  double *trj, *blk, *ucum;
  unsigned int* counter;      // problem counter of the scratch, never reset: this solve's problems are
  unsigned int counter_base;  // counter - counter_base = 0 .. B-1 (the host knows the value every solve leaves behind)
  unsigned long long* prof;   // optional phase cycle counters (profiling builds only)
  int lds_bytes;              // dynamic LDS of the launch (bounds-checked builds compare the pool indices with it)
};

// ---- lane-group primitives ---------------------------------------------------
template <int G>
__device__ __forceinline__ double gbcast(double v, int src) {
  if (G == 1) return v;
  return __shfl(v, src, G);
}
template <int G>
__device__ __forceinline__ int gbcast_i(int v, int src) {
  if (G == 1) return v;
  return __shfl(v, src, G);
}
// index of the first lane of the group whose predicate holds, or -1
template <int G>
__device__ __forceinline__ int gfirst(bool pred, int lane) {
  if (G == 1) return pred ? 0 : -1;
  unsigned long long b = __ballot(pred);
  if (G < 64) {
    int base = lane & ~(G - 1);
    b = (b >> base) & ((1ull << G) - 1ull);
  }
  return b ? (__ffsll((long long)b) - 1) : -1;
}

// ---- per-node record in `lin` ---------------------------------------------------
template <class MD>
struct LinLayout {
  static constexpr int N = MD::N, M = MD::M;
  static constexpr int oA = 0, oB = oA + MD::NA, oLx = oB + MD::NB, oLu = oLx + N, oLxx = oLu + M,
                       oLuu = oLxx + N * N, E = oLuu + M * M;
};

template <class MD>
__host__ __device__ constexpr bool s_nz(int i, int j) { return MD::a_slot(i, j) >= 0 || MD::a_is_dt(i, j); }
template <class MD>
__host__ __device__ constexpr bool b_nz(int i, int j) { return MD::b_slot(i, j) >= 0 || MD::b_const(i, j) != 0; }

// expand the compact Jacobian into register matrices S (= A - I) and B
template <class MD>
__device__ __forceinline__ void expand_jac(const ModelConst& mc, const double* a, const double* b,
                                           double (&S)[MD::N][MD::N], double (&Bm)[MD::N][MD::M]) {
  constexpr int N = MD::N, M = MD::M;
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (MD::a_slot(i, j) >= 0) S[i][j] = a[MD::a_slot(i, j)];
      else if (MD::a_is_dt(i, j)) S[i][j] = mc.dt;
      else S[i][j] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
      if (MD::b_slot(i, j) >= 0) Bm[i][j] = b[MD::b_slot(i, j)];
      else if (MD::b_const(i, j) == 1) Bm[i][j] = mc.dt;
      else if (MD::b_const(i, j) == 2) Bm[i][j] = 0.5 * (mc.dt * mc.dt);
      else Bm[i][j] = 0.0;
    }
  }
}

// W = V (I + S)
template <class MD>
__device__ __forceinline__ void mul_VA(const double (&V)[MD::N][MD::N], const double (&S)[MD::N][MD::N],
                                       double (&W)[MD::N][MD::N]) {
  constexpr int N = MD::N;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double acc = V[i][j];
#pragma unroll
      for (int k = 0; k < N; ++k)
        if (s_nz<MD>(k, j)) acc = fma(V[i][k], S[k][j], acc);
      W[i][j] = acc;
    }
}
// T = V B
template <class MD>
__device__ __forceinline__ void mul_VB(const double (&V)[MD::N][MD::N], const double (&Bm)[MD::N][MD::M],
                                       double (&T)[MD::N][MD::M]) {
  constexpr int N = MD::N, M = MD::M;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int b = 0; b < M; ++b) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k)
        if (b_nz<MD>(k, b)) acc = fma(V[i][k], Bm[k][b], acc);
      T[i][b] = acc;
    }
}
// inverse of an M x M matrix (M = 1 or 2), closed form
template <int M>
__device__ __forceinline__ void inv_small(const double (&Qm)[M][M], double (&Qi)[M][M]) {
  if (M == 1) {
    Qi[0][0] = 1.0 / Qm[0][0];
  } else {
    double det = Qm[0][0] * Qm[1][1] - Qm[0][1] * Qm[1][0];
    double r = 1.0 / det;
    Qi[0][0] = Qm[1][1] * r;
    Qi[0][1] = -Qm[0][1] * r;
    Qi[1][0] = -Qm[1][0] * r;
    Qi[1][1] = Qm[0][0] * r;
  }
}

// One step of the backward recursion (cilqr/ilqr.py:85-98).
template <class MD, class FLx, class FLu, class FLxx, class FLuu, class FXtra>
__device__ __forceinline__ void riccati_step(const double (&S)[MD::N][MD::N], const double (&Bm)[MD::N][MD::M],
                                             FLx lx, FLu lu, FLxx lxx, FLuu luu, FXtra xtra, double reg, double* dv_lin,
                                             double (&Vx)[MD::N], double (&Vxx)[MD::N][MD::N],
                                             double (&K)[MD::M][MD::N], double (&kf)[MD::M]) {
  constexpr int N = MD::N, M = MD::M;
  double Qx[N], Qu[M], Vo[N];
#pragma unroll
  for (int i = 0; i < N; ++i) Vo[i] = Vx[i];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double acc = lx(i) + Vx[i];
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (s_nz<MD>(k, i)) acc = fma(S[k][i], Vx[k], acc);
    Qx[i] = acc;
  }
#pragma unroll
  for (int a = 0; a < M; ++a) {
    double acc = lu(a);
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (b_nz<MD>(k, a)) acc = fma(Bm[k][a], Vx[k], acc);
    Qu[a] = acc;
  }
  double W[N][N], T[N][M];
  mul_VA<MD>(Vxx, S, W);
  mul_VB<MD>(Vxx, Bm, T);
  double Qux[M][N], Quu[M][M], Qr[M][M], Qi[M][M];
#pragma unroll
  for (int a = 0; a < M; ++a) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k)
        if (b_nz<MD>(k, a)) acc = fma(Bm[k][a], W[k][j], acc);
      Qux[a][j] = acc;
    }
#pragma unroll
    for (int b = 0; b < M; ++b) {
      double acc = luu(a, b);
#pragma unroll
      for (int k = 0; k < N; ++k)
        if (b_nz<MD>(k, a)) acc = fma(Bm[k][a], T[k][b], acc);
      Quu[a][b] = acc;
      Qr[a][b] = acc + ((a == b) ? reg : 0.0);
    }
  }
  inv_small<M>(Qr, Qi);
#pragma unroll
  for (int a = 0; a < M; ++a) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double acc = 0.0;
#pragma unroll
      for (int b = 0; b < M; ++b) acc = fma(-Qi[a][b], Qux[b][j], acc);
      K[a][j] = acc;
    }
    double acc = 0.0;
#pragma unroll
    for (int b = 0; b < M; ++b) acc = fma(-Qi[a][b], Qu[b], acc);
    kf[a] = acc;
  }
  // Quu k and Quu K
  double Qk[M], QK[M][N];
#pragma unroll
  for (int a = 0; a < M; ++a) {
    double acc = 0.0;
#pragma unroll
    for (int b = 0; b < M; ++b) acc = fma(Quu[a][b], kf[b], acc);
    Qk[a] = acc;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double acc2 = 0.0;
#pragma unroll
      for (int b = 0; b < M; ++b) acc2 = fma(Quu[a][b], K[b][j], acc2);
      QK[a][j] = acc2;
    }
  }
  // Qxx = lxx + A'W, then Vxx = Qxx + K'QuuK + K'Qux + Qux'K ; Vx likewise
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double vx = Qx[i];
#pragma unroll
    for (int a = 0; a < M; ++a) vx = fma(K[a][i], Qk[a], vx);
#pragma unroll
    for (int a = 0; a < M; ++a) vx = fma(K[a][i], Qu[a], vx);
#pragma unroll
    for (int a = 0; a < M; ++a) vx = fma(Qux[a][i], kf[a], vx);
    Vx[i] = vx;
  }
  double Vn[N][N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double acc = lxx(i, j) + W[i][j];
#pragma unroll
      for (int k = 0; k < N; ++k)
        if (s_nz<MD>(k, i)) acc = fma(S[k][i], W[k][j], acc);
      acc = xtra(acc, i, j, Vo);          // f_xx contraction of the C++ generation (identity elsewhere)
#pragma unroll
      for (int a = 0; a < M; ++a) acc = fma(K[a][i], QK[a][j], acc);
#pragma unroll
      for (int a = 0; a < M; ++a) acc = fma(K[a][i], Qux[a][j], acc);
#pragma unroll
      for (int a = 0; a < M; ++a) acc = fma(Qux[a][i], K[a][j], acc);
      Vn[i][j] = acc;
    }
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) Vxx[i][j] = Vn[i][j];
  double dl = 0.0;
#pragma unroll
  for (int a = 0; a < M; ++a) dl += kf[a] * Qu[a];
  *dv_lin = dl;
}

// ---------------------------------------------------------------------------
// Node cost / quadratisation / multiplier algebra per solver generation.
// `lam` points at this node's first multiplier in the workspace (stride NG).
// ---------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ double quad_form(const double* e, ctab Q) {  // e' Q e, reference order
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) t += e[i] * Q[i * N + j];
    s += t * e[j];
  }
  return s;
}

// e' Q e for a diagonal Q: the off-diagonal products of the dense form are exact zeros, so this is the same sum
template <int N>
__device__ __forceinline__ double quad_form_diag(const double* e, ctab Q) {
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j) s += (e[j] * Q[j * N + j]) * e[j];
  return s;
}

template <int MODEL, int MODE>
struct NodeOps;

// LEGACY: box bounds folded into the node with the I_mu mask
// (cilqr/lat_bicycle_node.py:93-180; dyn variant full_bicycle_dynamic_node.py:110-197)
template <int MODEL>
struct NodeOps<MODEL, ALILQR_MODE_LEGACY> {
  using MD = Model<MODEL>;
  static constexpr int N = MD::N, M = MD::M, C = MD::CTRL_ROWS ? 2 * (N + M) : 2 * N;

  // constraint i of (x,u): [x - smax ; smin - x ; u - cmax ; cmin - u]
  __device__ static __forceinline__ double cval(ctab box, const double* x, const double* u, int i) {
    ctab smin = box, smax = box + N, cmin = box + 2 * N, cmax = box + 2 * N + M;
    if (i < N) return x[i] - smax[i];
    if (i < 2 * N) return smin[i - N] - x[i - N];
    if (i < 2 * N + M) return u[i - 2 * N] - cmax[i - 2 * N];
    return cmin[i - 2 * N - M] - u[i - 2 * N - M];
  }
  __device__ static double cost(const KArgs& a, ctab Q, ctab R, ctab box,
                                const double* x, const double* u, const double* g,
                                const double* lam, size_t ls, double mu) {
    double e[N];
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = MD::GOAL_COST ? x[i] - g[i] : x[i];
    double sc = a.qdiag ? quad_form_diag<N>(e, Q) : quad_form<N>(e, Q);
    double cc = quad_form<M>(u, R);
    double lc = 0.0, qc = 0.0;
    const double half = (MODEL == ALILQR_MODEL_DYN) ? 0.5 : 0.5 * mu;
    // A row with lam == 0 and c <= 0 adds exact zeros to both sums.  The 2N state rows are tested with
    // compares only (x <= x_max, x >= x_min, lam == 0); when no lane of the wave has an active state row
    // the whole block is skipped — same bits, far fewer instructions on loosely bounded states.
    bool state_active = false;
#pragma unroll
    for (int i = 0; i < N; ++i)
      state_active |= !(x[i] <= box[N + i]) | !(x[i] >= box[i]) | (lam[i * ls] != 0.0) | (lam[(N + i) * ls] != 0.0);
    if (__any(state_active)) {
#pragma unroll
      for (int i = 0; i < 2 * N; ++i) {
        double c = cval(box, x, u, i);
        double l = lam[i * ls];
        double im = (l == 0.0 && c <= 0.0) ? 0.0 : mu;
        lc += l * c;
        qc += ((half * c) * im) * c;
      }
    }
#pragma unroll
    for (int i = 2 * N; i < C; ++i) {
      double c = cval(box, x, u, i);
      double l = lam[i * ls];
      double im = (l == 0.0 && c <= 0.0) ? 0.0 : mu;
      lc += l * c;
      qc += ((half * c) * im) * c;
    }
    return sc + cc + (lc + qc);
  }
  // l_x, l_u, l_xx, l_uu written straight into the node's `lin` record
  __device__ static void quadratize(const KArgs& a, ctab Q, ctab R, ctab box,
                                    const double* x, const double* u, const double* g,
                                    const double* lam, size_t ls, double mu,
                                    double* rec, size_t rs, int* code = nullptr) {
    using LL = LinLayout<MD>;
    double e[N], w[C], im[C];
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = MD::GOAL_COST ? x[i] - g[i] : x[i];
#pragma unroll
    for (int i = 0; i < C; ++i) {
      double c = cval(box, x, u, i);
      double l = lam[i * ls];
      im[i] = (l == 0.0 && c <= 0.0) ? 0.0 : mu;
      w[i] = l + im[i] * c;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) s += (2 * Q[i * N + j]) * e[j];
      rec[(LL::oLx + i) * rs] = s + (w[i] - w[N + i]);
#pragma unroll
      for (int j = 0; j < N; ++j)
        rec[(LL::oLxx + i * N + j) * rs] = (i == j) ? (2 * Q[i * N + j] + (im[i] + im[N + i])) : 2 * Q[i * N + j];
    }
#pragma unroll
    for (int i = 0; i < M; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < M; ++j) s += (2 * R[i * M + j]) * u[j];
      if constexpr (MD::CTRL_ROWS) {
        rec[(LL::oLu + i) * rs] = s + (w[2 * N + i] - w[2 * N + M + i]);
#pragma unroll
        for (int j = 0; j < M; ++j)
          rec[(LL::oLuu + i * M + j) * rs] =
              (i == j) ? (2 * R[i * M + j] + (im[2 * N + i] + im[2 * N + M + i])) : 2 * R[i * M + j];
      } else {
        // no control rows: C_u = 0 (lat_bicycle_interior_node.py:141), the sums of :111,122 add exact zeros
        rec[(LL::oLu + i) * rs] = s + 0.0;
#pragma unroll
        for (int j = 0; j < M; ++j) rec[(LL::oLuu + i * M + j) * rs] = (i == j) ? (2 * R[i * M + j] + 0.0) : 2 * R[i * M + j];
      }
    }
    // The AL part of diag(l_xx), diag(l_uu) is 0, mu or mu + mu per entry (the two bounds of a variable):
    // two bits each, states first.  hess_diag() turns them back into the same doubles.
    if (code) {
      int cd = 0;
#pragma unroll
      for (int i = 0; i < N; ++i) cd |= ((im[i] != 0.0 ? 1 : 0) + (im[N + i] != 0.0 ? 1 : 0)) << (2 * i);
      if constexpr (MD::CTRL_ROWS) {
#pragma unroll
        for (int i = 0; i < M; ++i) cd |= ((im[2 * N + i] != 0.0 ? 1 : 0) + (im[2 * N + M + i] != 0.0 ? 1 : 0)) << (2 * (N + i));
      }
      *code = cd;
    }
    (void)a;
  }
  // entry `idx` (states 0..N-1, then controls) of the AL part of the Hessian diagonal from the packed code
  __device__ static __forceinline__ double hess_diag(int code, int idx, double mu) {
    const int c = (code >> (2 * idx)) & 3;
    return c == 0 ? 0.0 : (c == 1 ? mu : mu + mu);
  }
  // per-node term of the violation measure: || max(-lam/mu, c) ||_2  (cilqr/ilqr.py:186-192)
  __device__ static double violation(const KArgs& a, ctab box, const double* x, const double* u,
                                     const double* lam, size_t ls, double mu) {
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < C; ++i) {
      double c = cval(box, x, u, i);
      double one = fmax(-lam[i * ls] / mu, c);
      ss += one * one;
    }
    (void)a;
    return sqrt(ss);
  }
  __device__ static double combine_violation(double acc, double node) { return acc + node; }
  __device__ static double violation_init() { return 0.0; }
  // lam <- max(0, lam + mu c)   (cilqr/lat_bicycle_node.py:162-165)
  __device__ static void update_lambda(const KArgs& a, ctab box, const double* x, const double* u,
                                       double* lam, size_t ls, double mu) {
#pragma unroll
    for (int i = 0; i < C; ++i) {
      double c = cval(box, x, u, i);
      lam[i * ls] = fmax(0.0, lam[i * ls] + mu * c);
    }
    (void)a;
  }
};

// The multipliers (and rectangle overlay) of a node as the row loops of the FAST and C++ generations read them: from the node's
// record in memory (a pointer and an element stride), or - the roll-outs - from LDS, where the lanes of the group have
// copied the block one step ahead (RowsLds).  The copy is a direct global -> LDS load, 16 bytes per lane: lane gl of a
// G-lane group brings doubles 2 gl, 2 gl + 1 of each chunk of 2 G doubles, and a chunk of the whole wave is 128 doubles
// of LDS (the hardware writes lane l at byte 16 l).
template <int CHD>
struct RowsLds {
  ltab p;              // the buffer + this lane's group offset (doubles)
  __device__ __forceinline__ double at(int k) const { return p[(k / CHD) * 128 + (k % CHD)]; }
};
template <class LS>
__device__ __forceinline__ double lam_at(const LS& lam, size_t ls, int k) {
  if constexpr (std::is_class<LS>::value) return lam.at(k);
  else return lam[(size_t)k * ls];
}
// Issue the copy of `nch` chunks of the block at `src` into the LDS buffer `dst` (uniform); gl = lane inside the group.
// The copies are issued from inline assembly: the compiler's wait-count pass makes every LDS read after a direct-to-LDS
// load it knows of wait for ALL outstanding memory operations, which is the wait this copy exists to avoid.  What
// orders the reads instead is the returned token: an ordinary 4-byte load issued behind the copies.  Memory operations
// return in order, so when the token has arrived the copies have landed; rows_bind() makes the LDS address of the
// reads depend on the token, and the compiler counts the operations issued since itself.  (Operations the compiler does
// not know of only make its other waits stricter: it waits for a load until at most as many operations are outstanding
// as IT issued since, and ours are among the outstanding ones.)  M0 (the LDS base of the copy) is saved and restored.
template <int G>
__device__ __forceinline__ int rows_dma(const double* src, double* dst, int nch, int gl) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  unsigned int ldsb = (unsigned int)(unsigned long long)(__attribute__((address_space(3))) void*)dst;
#pragma clang diagnostic pop
  ldsb = __builtin_amdgcn_readfirstlane(ldsb);      // uniform by construction; the constraint below needs a scalar register
  const double* s = src + 2 * gl;
  for (int ch = 0; ch < nch; ++ch) {
    unsigned int keep_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep_m0) : "v"(s), "s"(ldsb) : "memory");
    s += 2 * G;
    ldsb += 1024;
  }
  return *(const int*)src;
#else
  const int lane = threadIdx.x & 63;
  for (int ch = 0; ch < nch; ++ch)
    for (int e = 0; e < 2; ++e) dst[(size_t)ch * 128 + 2 * lane + e] = src[(size_t)ch * (2 * G) + 2 * gl + e];
  __threadfence_block();
  return 0;
#endif
}
// an LDS offset that is not formed before the token of the copy has arrived
__device__ __forceinline__ int rows_bind(int off, int token) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(off) : "v"(token));
#else
  (void)token;
#endif
  return off;
}

// FAST: projection-form AL over a linear table c(x,u) = A x + B u + C <= 0
// (cilqr/constraints.py:47-119, linear_constraints.py:13-29, fast_bicycle_node.py:183-212)
template <int MODEL>
struct NodeOps<MODEL, ALILQR_MODE_FAST> {
  using MD = Model<MODEL>;
  static constexpr int N = MD::N, M = MD::M;

  __device__ static __forceinline__ double cval(const KArgs& a, ctab tab, const double* x, const double* u, int i) {
    ctab A = tab, Bm = tab + a.C * N, Cv = tab + a.C * (N + M);
    double ax = 0.0, bu = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) ax += A[i * N + j] * x[j];
#pragma unroll
    for (int j = 0; j < M; ++j) bu += Bm[i * M + j] * u[j];
    return (ax + bu) + Cv[i];
  }
  static constexpr int NBOX = 2 * (N + M);
  // does this node evaluate the leading rows of its table as box rows?  (uniform over the wave: one lane with a
  // non-finite point sends the wave through the dense code)
  __device__ static __forceinline__ bool boxed(const KArgs& a, const double* x, const double* u) {
    if (a.box_rows == nullptr) return false;
    const unsigned int nb = *(const unsigned int __attribute__((address_space(4)))*)a.box_rows;
    return nb == (unsigned)NBOX && !__any(!all_finite<N, M>(x, u));
  }
  // Rows first .. C-1 in order, their multipliers loaded ROWS_AHEAD at a time before the first of them is used.
  static constexpr int ROWS_AHEAD = 1;   // measured: more rows ahead change nothing (the row loop is bound by its arithmetic)
  template <class LS, class F>
  __device__ static __forceinline__ void for_rows(const KArgs& a, const LS& lam, size_t ls, F&& f, int first = 0) {
    for (int i0 = first; i0 < a.C; i0 += ROWS_AHEAD) {
      double l[ROWS_AHEAD];
#pragma unroll
      for (int q = 0; q < ROWS_AHEAD; ++q) l[q] = (i0 + q < a.C) ? lam_at(lam, ls, i0 + q) : 0.0;
#pragma unroll
      for (int q = 0; q < ROWS_AHEAD; ++q)
        if (i0 + q < a.C) f(i0 + q, l[q]);
    }
  }
  template <class LS>
  __device__ static double cost(const KArgs& a, ctab Q, ctab R, ctab tab,
                                const double* x, const double* u, const double* g,
                                const LS& lam, size_t ls, double mu) {
    double e[N];
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = x[i] - g[i];
    double sc = a.qdiag ? quad_form_diag<N>(e, Q) : quad_form<N>(e, Q);
    double cc = quad_form<M>(u, R);
    double pp = 0.0, ll = 0.0;
    const bool box = boxed(a, x, u);
    if (box) {
      ctab Cv = tab + a.C * (N + M);
#pragma unroll
      for (int i = 0; i < NBOX; ++i) {
        const BoxRow<N, M> row{i};
        const double l = lam_at(lam, ls, i);
        double f = l - mu * ((row.dot(x, u) + 0.0) + Cv[i]);
        double lp = a.eq ? f : fmin(f, 0.0);
        pp += lp * lp;
        ll += l * l;
      }
    }
    for_rows(a, lam, ls, [&](int i, double l) {
      double f = l - mu * cval(a, tab, x, u, i);
      double lp = a.eq ? f : fmin(f, 0.0);          // constraints.py:61-62
      pp += lp * lp;
      ll += l * l;
    }, box ? NBOX : 0);
    return sc + cc + 0.5 / mu * (pp - ll);
  }
  __device__ static void quadratize(const KArgs& a, ctab Q, ctab R, ctab tab,
                                    const double* x, const double* u, const double* g,
                                    const double* lam, size_t ls, double mu,
                                    double* rec, size_t rs) {
    using LL = LinLayout<MD>;
    ctab A = tab, Bm = tab + a.C * N;
    double e[N], jx[N], ju[M], hx[N][N], hu[M][M];
#pragma unroll
    for (int i = 0; i < N; ++i) { e[i] = x[i] - g[i]; jx[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < M; ++i) ju[i] = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = 0; j < N; ++j) hx[i][j] = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j < M; ++j) hu[i][j] = 0.0;
    const bool box = boxed(a, x, u);
    if (box) {
      // a box row touches one gradient entry and one diagonal Hessian entry; what the dense code adds to the others
      // is +-0 (a product with a zero factor)
      ctab Cv = tab + a.C * (N + M);
#pragma unroll
      for (int i = 0; i < NBOX; ++i) {
        const BoxRow<N, M> row{i};
        double f = lam[(size_t)i * ls] - mu * ((row.dot(x, u) + 0.0) + Cv[i]);
        double lp = a.eq ? f : fmin(f, 0.0);
        double pj = a.eq ? 1.0 : ((f > 0.0) ? 0.0 : 1.0);
        const double pz = pj * row.sign();
        if (row.is_u()) { ju[row.col()] += pz * lp; hu[row.col()][row.col()] += pz * pz; }
        else { jx[row.col()] += pz * lp; hx[row.col()][row.col()] += pz * pz; }
      }
    }
    for_rows(a, lam, ls, [&](int i, double l) {
      double f = l - mu * cval(a, tab, x, u, i);
      double lp = a.eq ? f : fmin(f, 0.0);          // constraints.py:70-73, 91-94
      double pj = a.eq ? 1.0 : ((f > 0.0) ? 0.0 : 1.0);
      double pa[N], pb[M];
#pragma unroll
      for (int j = 0; j < N; ++j) { pa[j] = pj * A[i * N + j]; jx[j] += pa[j] * lp; }
#pragma unroll
      for (int j = 0; j < M; ++j) { pb[j] = pj * Bm[i * M + j]; ju[j] += pb[j] * lp; }
#pragma unroll
      for (int p = 0; p < N; ++p)
#pragma unroll
        for (int q = 0; q < N; ++q) hx[p][q] += pa[p] * pa[q];
#pragma unroll
      for (int p = 0; p < M; ++p)
#pragma unroll
        for (int q = 0; q < M; ++q) hu[p][q] += pb[p] * pb[q];
    }, box ? NBOX : 0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) s += (2 * Q[i * N + j]) * e[j];
      rec[(LL::oLx + i) * rs] = s + (-jx[i]);
#pragma unroll
      for (int j = 0; j < N; ++j) rec[(LL::oLxx + i * N + j) * rs] = 2 * Q[i * N + j] + (mu * hx[i][j] - 0.0);
    }
#pragma unroll
    for (int i = 0; i < M; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < M; ++j) s += (2 * R[i * M + j]) * u[j];
      rec[(LL::oLu + i) * rs] = s + (-ju[i]);
#pragma unroll
      for (int j = 0; j < M; ++j) rec[(LL::oLuu + i * M + j) * rs] = 2 * R[i * M + j] + (mu * hu[i][j] - 0.0);
    }
  }
  // per-node || c - min(c,0) ||_inf  (cilqr/constraints.py:115-119)
  __device__ static double violation(const KArgs& a, ctab tab, const double* x, const double* u,
                                     const double* lam, size_t ls, double mu) {
    double v = 0.0;
    for (int i = 0; i < a.C; ++i) {
      double c = cval(a, tab, x, u, i);
      double dc = fabs(c - fmin(c, 0.0));
      v = (dc > v || dc != dc) ? dc : v;
    }
    (void)lam; (void)ls; (void)mu;
    return v;
  }
  // np.maximum over nodes, seeded with -1 (cilqr/fast_ilqr.py:171-176); NaN propagates
  __device__ static double combine_violation(double acc, double node) { return (node > acc || node != node) ? node : acc; }
  __device__ static double violation_init() { return -1.0; }
  // lam <- min(lam - mu c, 0)  (cilqr/constraints.py:106-110; c is the value cached by max_violation at the same point)
  // TWO: values of a chunk before its stores (4-lane groups, where the update is a larger share: box + linear rows
  // H = 100 +2.5 ... +4.7 %; the 8- and 16-lane kernels lose 4 to 8 % with it and keep the row-by-row stores)
  template <bool TWO = false>
  __device__ static void update_lambda(const KArgs& a, ctab tab, const double* x, const double* u,
                                       double* lam, size_t ls, double mu) {
    // Eight rows at a time: their multipliers are loaded together, then the rows are updated and stored.  (A load behind
    // a store of the same array waits for the store - vector memory returns in order and the two may alias for all the
    // compiler knows - so row by row every row costs a round trip to memory.)
    constexpr int CH = 8;
    for (int i0 = 0; i0 < a.C; i0 += CH) {
      double l[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k) l[k] = lam[(size_t)((i0 + k < a.C) ? i0 + k : a.C - 1) * ls];
      // A multiplier that keeps its bits (a row that stays inactive stays at +0) is not written again: what follows waits
      // for these stores to complete, and they are single 8-byte pieces of separate lines.
      if constexpr (TWO) {
        // all eight new values first (table rows of the clamped index: the loads of a row then never follow a store),
        // then the stores
        double nl[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const double c = cval(a, tab, x, u, (i0 + k < a.C) ? i0 + k : a.C - 1);
          const double f = l[k] - mu * c;
          nl[k] = a.eq ? f : fmin(f, 0.0);
        }
#pragma unroll
        for (int k = 0; k < CH; ++k)
          if (i0 + k < a.C && __double_as_longlong(nl[k]) != __double_as_longlong(l[k])) lam[(size_t)(i0 + k) * ls] = nl[k];
      } else {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const int i = i0 + k;
          if (i < a.C) {
            double c = cval(a, tab, x, u, i);
            const double f = l[k] - mu * c;
            const double nl = a.eq ? f : fmin(f, 0.0);
            if (__double_as_longlong(nl) != __double_as_longlong(l[k])) lam[(size_t)i * ls] = nl;
          }
        }
      }
    }
  }
};


// NEW_AL: the C++ generation (al_ilqr_cpp/constraints/constraints.h:514-1040, model/new_bicycle_node.h:702-910).
// Same projection-form AL as FAST with three differences: only diag(Q), diag(R) enter the cost, the active-row
// mask is "projected multiplier < 0" (a row sitting exactly at zero drops out), and the state is not wrapped.
// Kernel-internal mode values: NEW_AL with quadratic constraint tables and / or rectangle-obstacle row
// activation.  Separate instantiations keep that code (and its registers) out of the plain NEW_AL kernels.
constexpr int kModeNewAlQuad = 3, kModeNewAlRect = 4, kModeNewAlQuadRect = 5;
__host__ __device__ constexpr bool is_new_al(int mode) { return mode >= ALILQR_MODE_NEW_AL && mode <= kModeNewAlQuadRect; }
__host__ __device__ constexpr bool is_quad(int mode) { return mode == kModeNewAlQuad || mode == kModeNewAlQuadRect; }
__host__ __device__ constexpr bool is_rect(int mode) { return mode == kModeNewAlRect || mode == kModeNewAlQuadRect; }

// With RECT a node's record carries, behind its C multipliers, an overlay of its constraint table: C row signs
// (0: row as in the shared table; +-1: row replaced by  sign * y + B_i u + C'_i <= 0), C replacement constants C'_i
// and the node's row index (LinearConstraints::UpdateConstraints, linear_constraints.h:360-373).

template <int MODEL, bool QUAD, bool RECT>
struct NewAlOps {
  using MD = Model<MODEL>;
  static constexpr int N = MD::N, M = MD::M;

  // x' Qc_i x as (x' Qc_i) x  (quadratic_constraints.h:21-24); rows with an all-zero matrix are skipped (exact zero)
  template <class TP>
  __device__ static __forceinline__ double quad_term(const KArgs& a, TP tab, const double* x, int i) {
    if (!QUAD || !((a.quad_rows >> i) & 1ull)) return 0.0;
    TP Qi = tab + a.C * (N + M + 1) + (size_t)i * N * N;
    double q = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) t += x[k] * Qi[k * N + j];
      q += t * x[j];
    }
    return q;
  }
  // entry j of row i of the node's A, and its C entry, through the overlay
  __device__ static __forceinline__ double a_ij(const KArgs& a, ctab A, const double* lam, size_t ls, int i, int j) {
    if (RECT) {
      const double sgn = lam[(size_t)(a.C + i) * ls];
      if (sgn != 0.0) return (j == 1) ? sgn : 0.0;
    }
    return A[i * N + j];
  }
  __device__ static __forceinline__ double c_i(const KArgs& a, ctab Cv, const double* lam, size_t ls, int i) {
    if (RECT) {
      if (lam[(size_t)(a.C + i) * ls] != 0.0) return lam[(size_t)(2 * a.C + i) * ls];
    }
    return Cv[i];
  }
  // One row of a node as the loops below see it: its multiplier and (RECT) its overlay entries, loaded ahead.
  struct Row { double l, sgn, cn; };
  // Rows in order, the multipliers (and overlay entries) of ROWS_AHEAD of them loaded before the first is used.
#ifdef ALILQR_EXP_ROWS_AHEAD
  static constexpr int ROWS_AHEAD = ALILQR_EXP_ROWS_AHEAD;
#else
  // measured: more rows ahead change nothing or cost registers, except with the rectangle overlay of the 6-state model
  // (three loads per row): 4 rows ahead are 7 % there, 8 spill
  static constexpr int ROWS_AHEAD = (RECT && N > 4) ? 4 : 1;
#endif
  static constexpr int NBOX = 2 * (N + M);
  // does this node evaluate the leading rows of its table as box rows?  (linear tables without rectangle overlay;
  // uniform over the wave: one lane with a non-finite point sends the wave through the dense code)
  __device__ static __forceinline__ bool boxed(const KArgs& a, const double* x, const double* u) {
    if (QUAD || RECT || a.box_rows == nullptr) return false;
    const unsigned int nb = *(const unsigned int __attribute__((address_space(4)))*)a.box_rows;
    return nb == (unsigned)NBOX && !__any(!all_finite<N, M>(x, u));
  }
  template <class LS, class F>
  __device__ static __forceinline__ void for_rows(const KArgs& a, const LS& lam, size_t ls, F&& f, int first = 0) {
    for (int i0 = first; i0 < a.C; i0 += ROWS_AHEAD) {
      Row r[ROWS_AHEAD];
#pragma unroll
      for (int q = 0; q < ROWS_AHEAD; ++q) {
        const bool in = i0 + q < a.C;
        r[q].l = in ? lam_at(lam, ls, i0 + q) : 0.0;
        r[q].sgn = (RECT && in) ? lam_at(lam, ls, a.C + i0 + q) : 0.0;
        r[q].cn = (RECT && in) ? lam_at(lam, ls, 2 * a.C + i0 + q) : 0.0;
      }
#pragma unroll
      for (int q = 0; q < ROWS_AHEAD; ++q)
        if (i0 + q < a.C) f(i0 + q, r[q]);
    }
  }
  // The rows of the bit mask `m` (uniform over the wave) in order.  The roll-outs skip rows that cannot contribute: a row
  // whose table entries are all zero, whose rectangle overlay is empty and whose multiplier is zero adds
  // (min(0 - mu * 0, 0))^2 = +0 to one sum and 0^2 = +0 to the other at a finite point - no bit of the cost changes.
  template <class LS, class F>
  __device__ static __forceinline__ void for_live(const KArgs& a, const LS& lam, size_t ls, unsigned long long m, F&& f) {
    while (m != 0ull) {
      const int i = __builtin_ctzll(m);
      m &= m - 1ull;
      Row r;
      r.l = lam_at(lam, ls, i);
      r.sgn = RECT ? lam_at(lam, ls, a.C + i) : 0.0;
      r.cn = RECT ? lam_at(lam, ls, 2 * a.C + i) : 0.0;
      f(i, r);
    }
  }
  // mask of the rows a roll-out step evaluates: `live` unless a lane of the wave holds a non-finite point (0 * inf is
  // a NaN in the dense sums of every row, so every row counts then)
  __device__ static __forceinline__ unsigned long long step_rows(const KArgs& a, unsigned long long live, bool box,
                                                                 const double* x, const double* u) {
    unsigned long long m = live;
    if (m != ~0ull && __any(!all_finite<N, M>(x, u))) m = ~0ull;
    if (a.C < 64) m &= (1ull << a.C) - 1ull;
    if (box) m &= ~((1ull << NBOX) - 1ull);
    return wave_uniform(m);
  }
  // the row-loaded forms of a_ij / c_i / cval / par_cval (same arithmetic)
  template <class TP>
  __device__ static __forceinline__ double a_ij(TP A, const Row& r, int i, int j) {
    if (RECT) {
      if (r.sgn != 0.0) return (j == 1) ? r.sgn : 0.0;
    }
    return A[i * N + j];
  }
  template <class TP>
  __device__ static __forceinline__ double c_i(TP Cv, const Row& r, int i) {
    if (RECT) {
      if (r.sgn != 0.0) return r.cn;
    }
    return Cv[i];
  }
  // Rows of zeros.  Constraint tables are mostly sparse (the reference's own: control bounds plus a pool of empty rows
  // that rectangle obstacles fill, test_lat_al_ilqr.cc:52-68).  A_i x of a row whose A entries are all zero is the sum
  // of N products with zero: +0 for a finite point, NaN otherwise, the same for every such row of a node - it is
  // formed once per node (ZeroSums) instead of once per row.  A rectangle overlay replaces such a row by  sgn * y:
  // ((0 + 0 x_0) + sgn x_1) + 0 x_2 ... = (0 + sgn x_1) + the zero products of the other columns.  Same bits as the
  // dense sums.  Which rows are zero is a property of the table: two bit masks found in the kernel prologue.
  struct ZeroSums { double xa, x1, ub; };
  __device__ static __forceinline__ ZeroSums zero_sums(const double* x, const double* u) {
    ZeroSums z;
    double sa = 0.0, s1 = 0.0, sb = 0.0;
    if (!SPARSE) { z.xa = 0.0; z.x1 = 0.0; z.ub = 0.0; return z; }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const double p = 0.0 * x[j];
      sa += p;
      if (RECT && j != 1) s1 += p;
    }
#pragma unroll
    for (int j = 0; j < M; ++j) sb += 0.0 * u[j];
    z.xa = sa; z.x1 = s1; z.ub = sb;
    return z;
  }
  // (the 4-state kernels keep the dense rows: four products per row are cheaper than the registers of the second path)
  static constexpr bool SPARSE = !QUAD && N > 4;
  __device__ static __forceinline__ bool a_zero(const KArgs& a, int i) { return SPARSE && ((a.azero >> i) & 1ull) != 0ull; }
  __device__ static __forceinline__ bool b_zero(const KArgs& a, int i) { return SPARSE && ((a.bzero >> i) & 1ull) != 0ull; }
  template <class TP>
  __device__ static __forceinline__ double ax_of(const KArgs& a, TP A, const double* x, int i, const Row& r, const ZeroSums& z) {
    if (a_zero(a, i)) return RECT ? (0.0 + r.sgn * x[1]) + z.x1 : z.xa;
    double ax = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) ax += a_ij(A, r, i, j) * x[j];
    return ax;
  }
  template <class TP>
  __device__ static __forceinline__ double bu_of(const KArgs& a, TP Bm, const double* u, int i, const ZeroSums& z) {
    if (b_zero(a, i)) return z.ub;
    double bu = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) bu += Bm[i * M + j] * u[j];
    return bu;
  }
  template <class TP>
  __device__ static __forceinline__ double cval(const KArgs& a, TP tab, const double* x, const double* u, int i, const Row& r,
                                                const ZeroSums& z) {
    TP A = tab, Bm = tab + a.C * N, Cv = tab + a.C * (N + M);
    const double ax = ax_of(a, A, x, i, r, z), bu = bu_of(a, Bm, u, i, z);
    if (QUAD) return ((quad_term(a, tab, x, i) + ax) + bu) + c_i(Cv, r, i);
    return ax + bu + c_i(Cv, r, i);
  }
  template <class TP>
  __device__ static __forceinline__ double par_cval(const KArgs& a, TP tab, const double* x, const double* u, int i, const Row& r,
                                                    const ZeroSums& z) {
    TP Cv = tab + a.C * (N + M);
    const double ax = ax_of(a, tab, x, i, r, z), bu = bu_of(a, tab + a.C * N, u, i, z);
    double c = ax + bu + c_i(Cv, r, i);
    if (QUAD) c = c + quad_term(a, tab, x, i);
    return c;
  }
  // A x + B u + C (linear_constraints.h:176-181); quadratic tables put x'Qx first in the sum (quadratic_constraints.h:26)
  __device__ static __forceinline__ double cval(const KArgs& a, ctab tab, const double* x, const double* u, int i,
                                                const double* lam, size_t ls) {
    ctab A = tab, Bm = tab + a.C * N, Cv = tab + a.C * (N + M);
    double ax = 0.0, bu = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) ax += a_ij(a, A, lam, ls, i, j) * x[j];
#pragma unroll
    for (int j = 0; j < M; ++j) bu += Bm[i * M + j] * u[j];
    if (QUAD) return ((quad_term(a, tab, x, i) + ax) + bu) + c_i(a, Cv, lam, ls, i);
    return ax + bu + c_i(a, Cv, lam, ls, i);
  }
  // one column of parallel_constraints: the quadratic term comes last (quadratic_constraints.h:44)
  __device__ static __forceinline__ double par_cval(const KArgs& a, ctab tab, const double* x, const double* u, int i,
                                                    const double* lam, size_t ls) {
    ctab A = tab, Bm = tab + a.C * N, Cv = tab + a.C * (N + M);
    double ax = 0.0, bu = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) ax += a_ij(a, A, lam, ls, i, j) * x[j];
#pragma unroll
    for (int j = 0; j < M; ++j) bu += Bm[i * M + j] * u[j];
    double c = ax + bu + c_i(a, Cv, lam, ls, i);
    if (QUAD) c = c + quad_term(a, tab, x, i);
    return c;
  }
  template <bool LIVE = false, class TP, class LS>
  __device__ static double cost(const KArgs& a, TP Q, TP R, TP tab,
                                const double* x, const double* u, const double* g,
                                const LS& lam, size_t ls, double mu, unsigned long long live = ~0ull) {
    double sc = 0.0, cc = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) { double e = x[i] - g[i]; sc += (e * Q[i * N + i]) * e; }
#pragma unroll
    for (int j = 0; j < M; ++j) cc += (R[j * M + j] * u[j]) * u[j];
    double pp = 0.0, ll = 0.0;
    const bool box = boxed(a, x, u);
    if (box) {
      TP Cv = tab + a.C * (N + M);
#pragma unroll
      for (int i = 0; i < NBOX; ++i) {
        const BoxRow<N, M> row{i};
        const double l = lam_at(lam, ls, i);
        const double f = l - mu * ((row.dot(x, u) + 0.0) + Cv[i]);
        double lp = a.eq ? f : fmin(f, 0.0);
        pp += lp * lp;
        ll += l * l;
      }
    }
    const ZeroSums zs = zero_sums(x, u);
    auto row = [&](int i, const Row& r) {
      const double l = r.l;
      const double f = l - mu * cval(a, tab, x, u, i, r, zs);
      double lp = a.eq ? f : fmin(f, 0.0);          // constraints.h:520-528
      pp += lp * lp;
      ll += l * l;
    };
    if constexpr (LIVE) for_live(a, lam, ls, step_rows(a, live, box, x, u), row);
    else for_rows(a, lam, ls, row, box ? NBOX : 0);
    return sc + cc + 0.5 / mu * (pp - ll);
  }
  // cost() and par_cost() as one instruction stream for tables without quadratic rows: `par` picks, lane by lane, which
  // of the two sums is returned.  The state and control terms of the two have the same bits (their products differ in
  // the order of the factors only), the row values are the same; the penalty is summed both ways.
  template <bool LIVE = false, class TP, class LS>
  __device__ static double cost2(const KArgs& a, TP Q, TP R, TP tab,
                                 const double* x, const double* u, const double* g,
                                 const LS& lam, size_t ls, double mu, bool par, unsigned long long live = ~0ull) {
    static_assert(!QUAD, "par_cval puts the quadratic term last");
    double sc = 0.0, cc = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) { double e = x[i] - g[i]; sc += (e * Q[i * N + i]) * e; }
#pragma unroll
    for (int j = 0; j < M; ++j) cc += (R[j * M + j] * u[j]) * u[j];
    double pp = 0.0, ll = 0.0, al = 0.0;
    const bool box = boxed(a, x, u);
    if (box) {
      TP Cv = tab + a.C * (N + M);
#pragma unroll
      for (int i = 0; i < NBOX; ++i) {
        const BoxRow<N, M> row{i};
        const double l = lam_at(lam, ls, i);
        const double f = l - mu * ((row.dot(x, u) + 0.0) + Cv[i]);
        double lp = a.eq ? f : fmin(f, 0.0);
        pp += lp * lp;
        ll += l * l;
        al += lp * lp - l * l;
      }
    }
    const ZeroSums zs = zero_sums(x, u);
    auto row = [&](int i, const Row& r) {
      const double l = r.l;
      const double f = l - mu * cval(a, tab, x, u, i, r, zs);
      double lp = a.eq ? f : fmin(f, 0.0);
      pp += lp * lp;
      ll += l * l;
      al += lp * lp - l * l;
    };
    if constexpr (LIVE) for_live(a, lam, ls, step_rows(a, live, box, x, u), row);
    else for_rows(a, lam, ls, row, box ? NBOX : 0);
    return par ? (sc + cc) + 0.5 / mu * al : sc + cc + 0.5 / mu * (pp - ll);
  }
  // one column of parallel_cost / parallel_augmented_lagrangian_cost (constraints.h:558-580)
  template <bool LIVE = false, class TP, class LS>
  __device__ static double par_cost(const KArgs& a, TP Q, TP R, TP tab,
                                    const double* x, const double* u, const double* g,
                                    const LS& lam, size_t ls, double mu, unsigned long long live = ~0ull) {
    double cq = 0.0, cr = 0.0, al = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) { double e = x[i] - g[i]; cq += e * (e * Q[i * N + i]); }
#pragma unroll
    for (int j = 0; j < M; ++j) cr += (R[j * M + j] * u[j]) * u[j];
    const bool box = boxed(a, x, u);
    if (box) {
      TP Cv = tab + a.C * (N + M);
#pragma unroll
      for (int i = 0; i < NBOX; ++i) {
        const BoxRow<N, M> row{i};
        const double l = lam_at(lam, ls, i);
        const double f = l - mu * ((row.dot(x, u) + 0.0) + Cv[i]);
        double pr = a.eq ? f : fmin(f, 0.0);
        al += pr * pr - l * l;
      }
    }
    const ZeroSums zs = zero_sums(x, u);
    auto row = [&](int i, const Row& r) {
      const double l = r.l;
      const double f = l - mu * par_cval(a, tab, x, u, i, r, zs);
      double pr = a.eq ? f : fmin(f, 0.0);
      al += pr * pr - l * l;
    };
    if constexpr (LIVE) for_live(a, lam, ls, step_rows(a, live, box, x, u), row);
    else for_rows(a, lam, ls, row, box ? NBOX : 0);
    al = 0.5 / mu * al;
    return (cq + cr) + al;
  }
  template <bool LIVE = false>
  __device__ static void quadratize(const KArgs& a, ctab Q, ctab R, ctab tab,
                                    const double* x, const double* u, const double* g,
                                    const double* lam, size_t ls, double mu,
                                    double* rec, size_t rs, unsigned long long live = ~0ull) {
    using LL = LinLayout<MD>;
    ctab A = tab, Bm = tab + a.C * N;
    double e[N], jx[N], ju[M], hx[N][N], hu[M][M], tq[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i) { e[i] = x[i] - g[i]; jx[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < M; ++i) ju[i] = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = 0; j < N; ++j) { hx[i][j] = 0.0; tq[i][j] = 0.0; }
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j < M; ++j) hu[i][j] = 0.0;
    const bool box = boxed(a, x, u);
    if (box) {
      // a box row touches one gradient entry and one diagonal Hessian entry; what the dense code adds to the others
      // is +-0 (a product with a zero factor)
      ctab Cv = tab + a.C * (N + M);
#pragma unroll
      for (int i = 0; i < NBOX; ++i) {
        const BoxRow<N, M> row{i};
        const double fi = lam[(size_t)i * ls] - mu * ((row.dot(x, u) + 0.0) + Cv[i]);
        double lp = a.eq ? fi : fmin(fi, 0.0);
        double on = a.eq ? 1.0 : ((lp >= 0.0) ? 0.0 : 1.0);
        const double s = row.sign();
        const double pz = on * s;
        if (row.is_u()) {
          ju[row.col()] += pz * lp;
          hu[row.col()][row.col()] += a.eq ? (s * s) : ((mu * pz) * s);
        } else {
          jx[row.col()] += pz * lp;
          hx[row.col()][row.col()] += a.eq ? (s * s) : ((mu * pz) * s);
        }
      }
    }
    const ZeroSums zs = zero_sums(x, u);
    // (a skipped row - all-zero table row, zero multiplier, finite point - has lp = +-0 and coefficients 0: every term it
    //  adds below is +-0, and a sum that started at +0 keeps its bits)
    auto row = [&](int i, const Row& r) {
      const double fi = r.l - mu * cval(a, tab, x, u, i, r, zs);
      double lp = a.eq ? fi : fmin(fi, 0.0);        // the equality branch works on the factor itself (constraints.h:673-676)
      double on = a.eq ? 1.0 : ((lp >= 0.0) ? 0.0 : 1.0);
      const bool az = a_zero(a, i), bz = b_zero(a, i);
      if (az || bz) {
        // a zero row (or its overlay  sgn * y) touches at most one gradient and one diagonal Hessian entry; what the
        // dense code below adds to the others is +-0 at a finite point
        if (az) {
          if (RECT) {
            const double p1 = on * r.sgn;
            jx[1] += p1 * lp;
            hx[1][1] += a.eq ? (r.sgn * r.sgn) : ((mu * p1) * r.sgn);
          }
        } else {
          double ca[N], qa[N];
#pragma unroll
          for (int j = 0; j < N; ++j) { ca[j] = a_ij(A, r, i, j); qa[j] = on * ca[j]; jx[j] += qa[j] * lp; }
#pragma unroll
          for (int p = 0; p < N; ++p)
#pragma unroll
            for (int q = 0; q < N; ++q) hx[p][q] += a.eq ? (ca[p] * ca[q]) : ((mu * qa[p]) * ca[q]);
        }
        if (!bz) {
          double qb[M];
#pragma unroll
          for (int j = 0; j < M; ++j) { qb[j] = on * Bm[i * M + j]; ju[j] += qb[j] * lp; }
#pragma unroll
          for (int p = 0; p < M; ++p)
#pragma unroll
            for (int q = 0; q < M; ++q) hu[p][q] += a.eq ? (Bm[i * M + p] * Bm[i * M + q]) : ((mu * qb[p]) * Bm[i * M + q]);
        }
        return;
      }
      double cx[N], pa[N], pb[M];
#pragma unroll
      for (int j = 0; j < N; ++j) cx[j] = a_ij(A, r, i, j);
      if (QUAD && ((a.quad_rows >> i) & 1ull)) {
        // c_x row: A_i + 2 Qc_i x ; constraint Hessian 2 Qc_i weighted by the projected multiplier
        // (quadratic_constraints.h:47-55, 86-88; constraints.h:797-802, 1341-1345)
        ctab Qi = tab + a.C * (N + M + 1) + (size_t)i * N * N;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          double qx = 0.0;
#pragma unroll
          for (int k = 0; k < N; ++k) {
            qx += (2 * Qi[j * N + k]) * x[k];
            tq[j][k] += (2.0 * Qi[j * N + k]) * lp;
          }
          cx[j] = cx[j] + qx;
        }
      }
#pragma unroll
      for (int j = 0; j < N; ++j) { pa[j] = on * cx[j]; jx[j] += pa[j] * lp; }
#pragma unroll
      for (int j = 0; j < M; ++j) { pb[j] = on * Bm[i * M + j]; ju[j] += pb[j] * lp; }
#pragma unroll
      for (int p = 0; p < N; ++p)
#pragma unroll
        for (int q = 0; q < N; ++q) hx[p][q] += a.eq ? (cx[p] * cx[q]) : ((mu * pa[p]) * cx[q]);   // mu (cx'cx) vs (mu P cx)'cx, constraints.h:791-802
#pragma unroll
      for (int p = 0; p < M; ++p)
#pragma unroll
        for (int q = 0; q < M; ++q) hu[p][q] += a.eq ? (Bm[i * M + p] * Bm[i * M + q]) : ((mu * pb[p]) * Bm[i * M + q]);
    };
    if constexpr (LIVE) for_live(a, lam, ls, step_rows(a, live, box, x, u), row);
    else for_rows(a, lam, ls, row, box ? NBOX : 0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) s += (2 * Q[i * N + j]) * e[j];
      rec[(LL::oLx + i) * rs] = s + (-jx[i]);
#pragma unroll
      for (int j = 0; j < N; ++j) rec[(LL::oLxx + i * N + j) * rs] = 2 * Q[i * N + j] + ((a.eq ? mu * hx[i][j] : hx[i][j]) - 1.0 * tq[i][j]);
    }
#pragma unroll
    for (int i = 0; i < M; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < M; ++j) s += (2 * R[i * M + j]) * u[j];
      rec[(LL::oLu + i) * rs] = s + (-ju[i]);
#pragma unroll
      for (int j = 0; j < M; ++j) rec[(LL::oLuu + i * M + j) * rs] = 2 * R[i * M + j] + (a.eq ? mu * hu[i][j] : hu[i][j]);
    }
  }
  // (a skipped row has c = +0: it neither raises the violation nor moves its multiplier off zero)
  template <bool LIVE = false>
  __device__ static double violation(const KArgs& a, ctab tab, const double* x, const double* u,
                                     const double* lam, size_t ls, double mu, unsigned long long live = ~0ull) {
    double v = 0.0;
    auto row = [&](int i) {
      double c = cval(a, tab, x, u, i, lam, ls);
      double dc = fabs(c - fmin(c, 0.0));
      v = (dc > v) ? dc : v;
    };
    if constexpr (LIVE) {
      for (unsigned long long m = step_rows(a, live, false, x, u); m != 0ull; m &= m - 1ull) row(__builtin_ctzll(m));
    } else {
      for (int i = 0; i < a.C; ++i) row(i);
    }
    (void)mu;
    return v;
  }
  __device__ static double combine_violation(double acc, double node) { return (node > acc) ? node : acc; }
  __device__ static double violation_init() { return -1.0; }
  template <bool LIVE = false>
  __device__ static void update_lambda(const KArgs& a, ctab tab, const double* x, const double* u,
                                       double* lam, size_t ls, double mu, unsigned long long live = ~0ull) {
    auto row = [&](int i) {
      double c = cval(a, tab, x, u, i, lam, ls);
      const double l = lam[i * ls];
      const double f = l - mu * c;
      const double nl = a.eq ? f : fmin(f, 0.0);    // constraints.h:878-887
      if (__double_as_longlong(nl) != __double_as_longlong(l)) lam[i * ls] = nl;   // (unchanged bits are not written again)
    };
    if constexpr (LIVE) {
      for (unsigned long long m = step_rows(a, live, false, x, u); m != 0ull; m &= m - 1ull) row(__builtin_ctzll(m));
    } else {
      for (int i = 0; i < a.C; ++i) row(i);
    }
  }
};
template <int MODEL>
struct NodeOps<MODEL, ALILQR_MODE_NEW_AL> : NewAlOps<MODEL, false, false> {};
template <int MODEL>
struct NodeOps<MODEL, kModeNewAlQuad> : NewAlOps<MODEL, true, false> {};
template <int MODEL>
struct NodeOps<MODEL, kModeNewAlRect> : NewAlOps<MODEL, false, true> {};
template <int MODEL>
struct NodeOps<MODEL, kModeNewAlQuadRect> : NewAlOps<MODEL, true, true> {};


// ---------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------
// Per-step record of a problem, contiguous in the group's block:
//   K_t | k_t | lambda_t        (RecLayout; x_t, u_t are read from the current trajectory, goals from the input)
// followed (after all H+1 records) by the per-node linearisation records:
//   LEGACY  f_x entries | f_u entries | l_x | l_u | diagonal AL terms of l_xx | of l_uu
//   FAST    f_x entries | f_u entries | l_x | l_u | l_xx (N x N) | l_uu (M x M)
template <class MD>
struct RecLayout {
  static constexpr int N = MD::N, M = MD::M;
  static constexpr int rK = 0, rk = rK + M * N, rl = rk + M;
};
// PACK (LEGACY records that have to share a small LDS budget among many groups of a wave): the Jacobian in the
// model's packed form (JacPack) and the AL part of the Hessian diagonal as one code word; the Riccati sweep expands
// both with the arithmetic that produced the unpacked entries, so the bits are the same.
template <class MD, int MODE, bool PACK = false>
struct NodeRec {
  static constexpr int N = MD::N, M = MD::M;
  static constexpr bool LEGACY = (MODE == ALILQR_MODE_LEGACY);
  static constexpr bool NEWAL = is_new_al(MODE);
  static_assert(!PACK || LEGACY, "packed records exist for the LEGACY generation");
  static constexpr int HX = PACK ? 1 : (LEGACY ? N : N * N), HU = PACK ? 0 : (LEGACY ? M : M * M), NF = NEWAL ? 7 : 0;
  static constexpr int NJA = PACK ? JacPack<MD>::NC : MD::NA, NJB = PACK ? 0 : MD::NB;
  static constexpr int oA = 0, oB = oA + NJA, oLx = oB + NJB, oLu = oLx + N, oHx = oLu + M, oHu = oHx + HX,
                       oF = oHu + HU, E = (oF + NF + 1) & ~1;
};

// doubles of one group's block (host and device must agree)
// doubles of one step record: K | k | lambda[C] (| overlay: sign[C] | C'[C] | row index)
template <class MD, int MODE>
__host__ __device__ constexpr int record_doubles(int C) {
  return (RecLayout<MD>::rl + C + (is_rect(MODE) ? 2 * C + 1 : 0) + 1) & ~1;
}
template <class MD, int MODE>
__host__ __device__ inline size_t block_doubles(int H, int C) {
  const size_t RS = (size_t)record_doubles<MD, MODE>(C);
  return (size_t)(H + 1) * (RS + NodeRec<MD, MODE>::E);
}

#ifndef ALILQR_LEAN_LOADS
#define ALILQR_LEAN_LOADS 0   // 1: a roll-out lane loads the reference point / gains / multipliers of a node only when its trial needs them (measured slower: divergent loads, more spills)
#endif
#ifndef ALILQR_TRIP_N6
#define ALILQR_TRIP_N6 0   // per-trip passes for the 6-state kernels too (measured slower: their register allocation suffers)
#endif
#ifndef ALILQR_LEGACY_ROUNDS
#define ALILQR_LEGACY_ROUNDS 0   // 1: LEGACY groups narrower than 32 lanes take their trials in rounds on the group's own lanes, one round per trip (measured: same bits, same throughput as the wave-wide passes - lateral 218 k against 217 k at 16 lanes, dynamic H = 100 50.5 k against 52.2 k)
#endif
#ifndef ALILQR_MINW_N4
#define ALILQR_MINW_N4 2   // waves per SIMD the 4-state kernels are compiled for (256 registers per lane)
#endif
#ifndef ALILQR_MINW_N6
#define ALILQR_MINW_N6 1   // waves per SIMD the 6-state kernels are compiled for (512 registers per lane)
#endif
// trajectory slots per problem and parity: the usual winners + one spare for a repeated roll-out.  The 6-state kernels
// of the C++ generation keep all ten trials (a winner among trials 8-10 - one iteration in ten - costs them a whole
// further pass otherwise: +3 to +9 % measured); the 4-state ones keep seven: at large batches the trajectory writes of
// the three extra lanes cost more than the repeats (65536 lateral problems: 253 k against 235 k solves/s).
template <int MODE, int N>
struct TrjSlots {
  static constexpr int SS = (MODE == ALILQR_MODE_LEGACY) ? 4 : (MODE == ALILQR_MODE_FAST || N <= 4) ? 8 : 11;
};

// f_xx of a node, only the C++ generation has it
template <class MD, bool ON>
struct FxxOf {
  __device__ static __forceinline__ void eval(const ModelConst&, const double*, const double*, double*) {}
  __device__ static __forceinline__ double add(double acc, const double*, const double*, int, int) { return acc; }
};
template <class MD>
struct FxxOf<MD, true> {
  __device__ static __forceinline__ void eval(const ModelConst& mc, const double* x, const double* u, double* f) { MD::fxx(mc, x, u, f); }
  __device__ static __forceinline__ double add(double acc, const double* f, const double* Vx, int i, int j) {
    return MD::fxx_nz(i, j) ? acc + MD::fxx_term(f, Vx, i, j) : acc;
  }
};

// MINW: waves per SIMD the instantiation is compiled for (0: 2 for the 4-state models, 1 for the 6-state ones).
// One wave per SIMD gives a wave 512 registers and a 40 KB share of the LDS.
// LDSLIN: 0 the linearisation records live in the groups' blocks in HBM; 1 in LDS; 2 (LEGACY) in LDS, and the same
// LDS bytes hold the accumulated controls of the open-loop line-search trials afterwards: the records are dead once
// the Riccati sweep has read them, the controls are written after it and dead when the line search ends; 3 (LEGACY,
// 32 lanes) in LDS in their packed form.
// ROWS (FAST and C++ generations): 1 the roll-outs read a node's multiplier block from LDS buffers filled one step ahead (the
// host launches this form where the buffers fit), 0 from the node's record.
template <int MODEL, int MODE, int G, int LDSLIN = 0, int MINW = 0, int ROWS = 0>
__global__ __launch_bounds__(64, (MINW ? MINW : (Model<MODEL>::N <= 4 ? ALILQR_MINW_N4 : ALILQR_MINW_N6))) void solve_kernel(KArgs a) {
  using MD = Model<MODEL>;
  using OPS = NodeOps<MODEL, MODE>;
  using LL = LinLayout<MD>;
  using RL = RecLayout<MD>;
  constexpr int N = MD::N, M = MD::M;
  constexpr bool LEGACY = (MODE == ALILQR_MODE_LEGACY);
  // packed records: where many groups share the wave's LDS, and (LDSLIN == 3) for 32-lane groups whose unpacked
  // records do not fit (long horizons)
  constexpr bool PACK = LEGACY && ((LDSLIN && (G < 32)) || LDSLIN == 3);
  constexpr bool LDSUC = LEGACY && (LDSLIN == 2);      // accumulated controls of trials 3 .. ntrials in LDS too
  using NR = NodeRec<MD, MODE, PACK>;
  constexpr bool NEWAL = is_new_al(MODE);
  constexpr bool QUAD = is_quad(MODE);
  constexpr bool RECT = is_rect(MODE);
  // One roll-out pass per trip of the wave loop where a wave carries more than two groups (their line searches need
  // different numbers of passes, and with the passes of a search back to back the wave runs as many as its slowest
  // group needs while the lanes of the others idle: half of the roll-out phase of 65536 kinematic + rectangle problems
  // at 8 lanes).  The 4-state kernels and the 6-state FAST kernel (box + linear H = 100, 16384 problems: 74 k -> 83 k
  // solves/s) take it; the 6-state NEW_AL kernels lose 20-30 % with it (register allocation) and keep the passes of
  // a search back to back, as do two groups per wave.  (LEGACY searches are wave-wide, below.)
  constexpr bool TRIP = (G < 32) && (N <= 4 || MODE == ALILQR_MODE_FAST || ALILQR_TRIP_N6);
  // LEGACY line search: wave-wide passes (below), or - a development switch, ALILQR_LEGACY_ROUNDS - for groups narrower
  // than 32 lanes the trials in rounds of G on the group's OWN lanes, one round per roll-out pass of the wave, through
  // the state machine of the other generations.
  constexpr bool LWIDE = LEGACY && !(G < 32 && ALILQR_LEGACY_ROUNDS);
  constexpr bool PM = true;                            // block layout: problem-major (one contiguous block per group)
  constexpr int CL = LEGACY ? (MD::CTRL_ROWS ? 2 * (N + M) : 2 * N) : 1;         // multipliers held in registers (LEGACY only)
  constexpr int SS = TrjSlots<MODE, N>::SS;               // trajectory slots kept per problem and parity

  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);                       // lane inside the group
  const size_t lg = (size_t)blockIdx.x * 64 + lane;    // global lane
  const size_t gid = lg / G;                           // global group
  if (gid >= (size_t)a.NG) return;                     // whole groups only: NG*G is a multiple of 64 by construction

  const int H = a.H, C = a.C;
  const size_t NG = (size_t)a.NG, LT = NG * G;
  const int XS = (H + 1) * N, US = H * M, T = XS + US;
  const ModelConst mc = a.mc;
  const int RS = record_doubles<MD, MODE>(C);
  const size_t ES = PM ? (size_t)1 : NG;               // element stride inside the block
  double* const blk = a.blk + (PM ? gid * block_doubles<MD, MODE>(H, C) : gid);
  // Linearisation records of the group: in LDS when the host found room for them at full occupancy (the
  // sequential Riccati sweep then reads them with LDS latency and they cause no memory traffic), else in the
  // group's block in HBM.
  extern __shared__ double lds_pool[];
  // doubles of a group's share of the pool (the host sizes the launch with the same expression: lds_group_doubles)
  const size_t PGS = lds_group_doubles(H, NR::E, LDSUC ? a.uc_cols * M : 0);
  double* const linb = LDSLIN ? lds_pool + (size_t)(lane / G) * PGS : blk + (size_t)(H + 1) * RS * ES;
  // bounds-checked builds: the wave's groups fit the launch's LDS; a group's share holds its records and (later in the
  // iteration) its accumulated controls
  (void)ALILQR_CHK(!LDSLIN || (size_t)(64 / G) * PGS * sizeof(double) <= (size_t)a.lds_bytes, 16);
  (void)ALILQR_CHK(!LDSLIN || ((size_t)(H + 1) * NR::E <= PGS && (!LDSUC || (size_t)H * M * (size_t)(a.uc_cols > 0 ? a.uc_cols : 1) <= PGS)), 17);

  double* const trj = a.trj;
  const int cw = LEGACY ? (2 * N + 2 * M) : C * (N + M + 1) + (QUAD ? C * N * N : 0);   // doubles per constraint table

  // is every state-weight table diagonal?  (uniform scalar reads, once per kernel)
  {
    bool diag = true;
    ctab Qt = as_ctab(a.Q);
    for (int t = 0; t < a.table_nodes; ++t)
      for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
          if (i != j && Qt[(size_t)t * N * N + i * N + j] != 0.0) diag = false;
    a.qdiag = diag ? 1 : 0;
    // which constraint rows carry a quadratic term in any table
    a.quad_rows = 0ull;
    if (QUAD) {
      ctab ct = as_ctab(a.cons);
      for (int t = 0; t < a.table_nodes; ++t)
        for (int i = 0; i < C; ++i) {
          bool nz = false;
          for (int e = 0; e < N * N; ++e) nz |= (ct[(size_t)t * cw + C * (N + M + 1) + i * N * N + e] != 0.0);
          if (nz) a.quad_rows |= 1ull << i;
        }
    }
    // which rows of the shared table are all zero (A, B, C and the quadratic term)
    a.zrows = 0ull;
    if (is_new_al(MODE) && a.live_rows && a.table_nodes == 1) {
      ctab ct = as_ctab(a.cons);
      for (int i = 0; i < C && i < 64; ++i) {
        bool z = (ct[C * (N + M) + i] == 0.0) && !(QUAD && ((a.quad_rows >> i) & 1ull));
        for (int j = 0; j < N; ++j) z &= (ct[i * N + j] == 0.0);
        for (int j = 0; j < M; ++j) z &= (ct[C * N + i * M + j] == 0.0);
        if (z) a.zrows |= 1ull << i;
      }
    }
    // which rows of the shared linear table have no A (no B) entry
    a.azero = 0ull; a.bzero = 0ull;
    if (is_new_al(MODE) && !QUAD && a.sparse_rows && a.table_nodes == 1 && a.C <= 64) {
      ctab ct = as_ctab(a.cons);
      for (int i = 0; i < C; ++i) {
        bool za = true, zb = true;
        for (int j = 0; j < N; ++j) za &= (ct[i * N + j] == 0.0);
        for (int j = 0; j < M; ++j) zb &= (ct[C * N + i * M + j] == 0.0);
        if (za) a.azero |= 1ull << i;
        if (zb) a.bzero |= 1ull << i;
      }
    }
  }

  // number of line-search trials the reference can make: alpha = 1, 1/2, ... while alpha > alpha_min
  int ntrials = 0;
  {
    const double alpha_min = ALILQR_COLD(prm.alpha_min);
    for (double al = 1.0; al > alpha_min && ntrials < 64; al = al / 2.0) ++ntrials;
  }
  if (NEWAL) ntrials = 10;                             // new_al_ilqr.h:1045

  // ---- per-group state (identical in every lane of a group) ----
  int p = -1;                 // problem being solved, -1: none
  bool drained = false;       // the batch counter ran out
  int cur_slot = 0, cur_par = 0;
  int ref_slot = 0;           // slot of the trajectory the running line search started from (cur_slot moves to the winner before a repeat)
  double mu = 1.0, old_cost = 0.0, viol = 0.0, dv_lin = 0.0;
  double mu_s = 1.0;          // NEW_AL: the solver's own penalty; the nodes' `mu` follows it from the first increase on
  int outer = 0, inner = 0, n_inner = 0, n_trials = 0;
  bool outer_start = false;
  unsigned long long live_g = 0ull;   // NEW_AL: all-zero table rows with a live multiplier or overlay on this lane's nodes (see live_rows)
  // Each group is a small state machine that advances by at most one linearise + Riccati pass and one roll-out
  // pass per trip of the wave loop: a group whose line search needs further passes (more rounds of trials, the
  // ladder, a repeat) takes them in the following trips while the other groups of the wave, already on their
  // next iteration, run their own linearisation and first round in the same instructions.
  enum { PH_LIN = 0, PH_ROLL = 1 };
  enum { ST_TRIALS = 0, ST_LADDER = 1, ST_REPLAY = 2, ST_DONE = 3 };
  int phase = PH_LIN, stage = ST_DONE;
  int np_ = 0;                // parity the candidates of the running line search are written to
  double new_cost = 0.0, J_last = 0.0, replay_alpha = 0.0, bestJ = 0.0, bestA = 0.0;
  bool keep_old = false, gate_skip = false, have = false;
  int win_trial = -1, win_lane = -1, r0 = 0, replay_kind = 0;
  int nt0 = 0;                // roll-out count when the running line search started (trace output)
  bool xo_pend = false;       // the (x, u) pair to return is a trajectory in the slots, not yet copied to x_out / u_out
  int xo_par = 0, xo_slot = 0;
  // LEGACY: bit t set = some multiplier of node t is not zero (nodes 0 .. 127; longer horizons load them always).
  // A roll-out step reads the node's multipliers only then: they are all zero during the first outer iteration
  // and on most nodes later.
  unsigned long long lmask0 = 0ull, lmask1 = 0ull;
  const bool lam_all = (H + 1 > 128);

  // candidate trajectories.  G > 1: [group][parity][slot][element], a lane writes its roll-out as one contiguous
  // run and a node's (x, u) is one short run; G = 1: [parity][element][group * SS + slot], lane-minor, so the
  // 64 problems of a wave store one contiguous run per element.
  auto trj_at = [&](int par, int off, int slot) -> double* {
    if (!ALILQR_CHK(par >= 0 && par < 2, 0)) par = 0;
    if (!ALILQR_CHK(off >= 0 && off < T, 1)) off = 0;
    if (!ALILQR_CHK(slot >= 0 && slot < SS, 2)) slot = 0;
    if (PM) return trj + ((gid * 2 + par) * SS + slot) * (size_t)T + off;
    return trj + ((size_t)par * T + off) * (NG * SS) + gid * SS + slot;
  };
  auto rec = [&](int t) -> double* {
    if (!ALILQR_CHK(t >= 0 && t <= H, 3)) t = 0;
    return blk + (size_t)t * RS * ES;
  };
  auto lrec = [&](int t) -> double* {
    if (!ALILQR_CHK(t >= 0 && t <= H, 4)) t = 0;
    return linb + (size_t)t * NR::E * ES;
  };
#ifdef ALILQR_LIT_LAT   /* timing experiment only: the tables of the headline problem as literals */
  auto tabQ = [&](int) { return (ctab)kLitQ; };
  auto tabR = [&](int) { return (ctab)kLitR; };
  auto tabC = [&](int) { return (ctab)kLitB; };
#else
  // (bounds-checked builds: the node index of a per-node table, and the problem index of a per-problem goal list, are
  //  caller-input indices like the workspace offsets above)
  auto tnode = [&](int t) -> int {
    if (!ALILQR_CHK(t >= 0 && t <= H, 14)) t = 0;
    return (a.table_nodes > 1) ? t : 0;
  };
  auto tabQ = [&](int t) { return as_ctab(a.Q) + (size_t)tnode(t) * N * N; };
  auto tabR = [&](int t) { return as_ctab(a.R) + (size_t)tnode(t) * M * M; };
  auto tabC = [&](int t) { return as_ctab(a.cons) + (size_t)tnode(t) * cw; };
#endif
  // NEW_AL: the tables the roll-outs read (constraint table, Q, R) are staged in LDS.  A roll-out step evaluates the
  // rows of a node one after the other; with the table behind scalar loads every row waits for two round trips to the
  // scalar cache, with the table in LDS the reads are issued rows ahead and return in order.  A table shared by all
  // nodes is staged once per kernel; per-node tables are staged node by node inside the roll-out.
  constexpr bool TLDS = QUAD && (ALILQR_TAB_LDS != 0);
  // the two kinds of NEW_AL roll-out (trial, ladder column) as one instruction stream: 6-state model, no quadratic rows
  // (the 4-state kernels spill twice as much with it at 4 and 8 lanes and lose 40 % there)
  // FAST: the returned pair (x, u of the last iteration that did not end an inner loop) is copied by the whole wave
  constexpr bool WCOPY = (MODE == ALILQR_MODE_FAST) && (G < 32) && (ALILQR_WAVE_COPY != 0);
  constexpr bool MERGE = NEWAL && !QUAD && (N > 4) && (ALILQR_MERGE_KINDS != 0);
  // the first ladder columns of a NEW_AL line search in the idle lanes of its last round of trials.  For groups of 16 and
  // 32 lanes the first round is the last one and every search carries the columns that three in ten need: only where
  // the two kinds share one instruction stream (the columns then cost no pass time: kinematic + rectangles at 16 lanes
  // 64 k -> 73 k solves/s, 387 -> 317 us per iteration at light load; divergent kinds, 4-state model: 136 -> 171 us)
  constexpr bool AHEAD = NEWAL && (G > 1) && (G <= 8 || MERGE) && (ALILQR_LADDER_AHEAD != 0);
  // Roll-outs skip rows that cannot contribute (below).  Measured on 4096 problems: lateral model with the reference's
  // ten-row table (two live rows) 73 -> 101 k solves/s, with rectangle rows 41 -> 51 k; 6-state tables with two to four
  // spare rows unchanged; quadratic tables (every row live) lose a third with the masked loop and keep the plain one.
  constexpr bool LIVE = NEWAL && !QUAD;
  const ltab ltC = as_ltab(lds_pool), ltQ = ltC + cw, ltR = ltQ + N * N;
  auto stage_tables = [&](int t) {           // every active lane writes the same values
    if constexpr (TLDS) {
      (void)ALILQR_CHK((size_t)(cw + N * N + M * M) * sizeof(double) <= (size_t)a.lds_bytes, 16);
      ctab sc = tabC(t), sq = tabQ(t), sr = tabR(t);
      for (int k = 0; k < cw; ++k) lds_pool[k] = sc[k];
      for (int k = 0; k < N * N; ++k) lds_pool[cw + k] = sq[k];
      for (int k = 0; k < M * M; ++k) lds_pool[cw + N * N + k] = sr[k];
      wave_lds_fence();
    }
  };
  if (TLDS && a.table_nodes <= 1) stage_tables(0);
  // NEW_AL roll-outs read a node's multipliers (and rectangle overlay) from LDS: the lanes of the group copy the block of
  // node t + 1 while step t computes (RowsLds / rows_dma above), two buffers by node parity, behind the staged tables.
  // With the block behind ordinary loads, every step waited for them - and, memory operations returning in order, for
  // the trajectory stores of the step before; the copy is one step old when it is needed and its wait leaves the newer
  // stores in flight.
  constexpr int CHD = 2 * G;                                        // doubles of a group's chunk (16 bytes per lane)
  constexpr bool RLDS = !LEGACY && (ROWS != 0);
  const int rows_lim = RLDS ? a.rows_lds : 0;                      // doubles of the block
  const int rows_nch = (rows_lim + CHD - 1) / CHD;                 // chunks per node
  const int rows_off = TLDS ? ((cw + N * N + M * M + 1) & ~1) : 0;  // first double of the buffers (16-byte aligned)
  (void)ALILQR_CHK(!RLDS || (size_t)(rows_off + 2 * rows_nch * 128 + 64) * sizeof(double) <= (size_t)a.lds_bytes, 17);

  // goal list of problem q (clamped: idle groups still form addresses)
  auto goal_of = [&](int q) -> const double* {
    if (!ALILQR_CHK(q < ALILQR_COLD_(B), 15)) q = 0;
    return a.goals + ((a.goal_batch > 1 && q > 0) ? (size_t)q * XS : 0);
  };
  // node t of the current trajectory (slot of the last accepted roll-out)
  auto load_node = [&](int t, double* x, double* u) {
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = *trj_at(cur_par, t * N + i, cur_slot);
#pragma unroll
    for (int j = 0; j < M; ++j) u[j] = (t < H) ? *trj_at(cur_par, XS + t * M + j, cur_slot) : 0.0;
  };
  // cost of node t at (x,u) with the node's goal and multipliers read from its record
  unsigned long long live_now = ~0ull;       // NEW_AL: the rows node_cost evaluates (set per trip of the wave loop, see live_rows)
  auto node_cost = [&](int t, const double* x, const double* u) -> double {
    const double* r = rec(t);
    double g[N];
#pragma unroll
    for (int i = 0; i < N; ++i) g[i] = goal_of(p)[(size_t)t * N + i];
    if (LEGACY) {
      double lam[CL];
#pragma unroll
      for (int i = 0; i < CL; ++i) lam[i] = r[(size_t)(RL::rl + i) * ES];
      return OPS::cost(a, tabQ(t), tabR(t), tabC(t), x, u, g, lam, 1, mu);
    }
    if constexpr (LIVE) return OPS::template cost<true>(a, tabQ(t), tabR(t), tabC(t), x, u, g, r + (size_t)RL::rl * ES, ES, mu, live_now);
    else return OPS::cost(a, tabQ(t), tabR(t), tabC(t), x, u, g, r + (size_t)RL::rl * ES, ES, mu);
  };

  // total cost of the current trajectory: node costs in parallel, summed in node order (ilqr.py:52-56)
  auto total_cost = [&]() -> double {
    double total = 0.0;
    for (int t0 = 0; t0 <= H; t0 += G) {
      int t = t0 + gl;
      double ct = 0.0;
      if (t <= H) {
        double x[N], u[M];
        load_node(t, x, u);
        ct = node_cost(t, x, u);
      }
      for (int g = 0; g < G && t0 + g <= H; ++g) total += gbcast<G>(ct, g);
    }
    return total;
  };

  PROF_DECL
  for (;;) {
    PROF_MARK(7)
    // ================= problem switch: pull + initial guess =================
    if (p < 0 && !drained) {
      int np = 0;
      if (gl == 0) np = (int)(atomicAdd(ALILQR_COLD(counter), 1u) - ALILQR_COLD(counter_base));
      np = gbcast_i<G>(np, 0);
      if (np >= ALILQR_COLD(B) || !ALILQR_CHK(np >= 0, 11)) {
        drained = true;
      } else {
        p = np;
        {
          // scheduling hint: the np-th pull takes problem order[np] (read from the argument block here, see ALILQR_COLD)
          const int32_t* const ord = ALILQR_COLD_(order);
          if (ord) { const int q = ord[np]; if ((unsigned)q < (unsigned)ALILQR_COLD(B)) p = q; }
        }
        const double* goals = a.goals + ((a.goal_batch > 1) ? (size_t)p * XS : 0);
        // outputs / optional inputs of the initial guess, read from the argument block here (see ALILQR_COLD)
        double* const o_xi = ALILQR_COLD(x_init);
        double* const o_ui = ALILQR_COLD(u_init);
        double* const o_xo = ALILQR_COLD(x_out);
        double* const o_uo = ALILQR_COLD(u_out);
        const double* const i_uoff = ALILQR_COLD(u_off);
        const double init_r = ALILQR_COLD(prm.init_r);
        // zero multipliers into the records, lanes in parallel
        lmask0 = 0ull; lmask1 = 0ull;
        for (int t = gl; t <= H; t += G) {
          double* r = rec(t);
          for (int i = 0; i < C; ++i) r[(size_t)(RL::rl + i) * ES] = 0.0;
          if (RECT) {
            for (int i = 0; i < 2 * C; ++i) r[(size_t)(RL::rl + C + i) * ES] = 0.0;
            // the caller's per-node start rows are device data the host cannot check: keep them inside the table
            const int32_t* const frs = ALILQR_COLD(first_rows);
            int fr = frs ? frs[t] : ALILQR_COLD(first_row);
            fr = fr < 0 ? 0 : (fr >= C ? C - 1 : fr);
            r[(size_t)(RL::rl + 3 * C) * ES] = (double)fr;
          }
        }
        // ---- linearised-LQR initial guess (ilqr.py:11-50): every lane of the group
        //      carries the same recursion; lane 0 stores.
        double P[N][N];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
          for (int j = 0; j < N; ++j) P[i][j] = (i == j) ? 1.0 : 0.0;
        for (int t = H - 1; t >= 0; --t) {
          double ac[MD::NA], bc[MD::NB], S[N][N], Bm[N][M], zu[M];
#pragma unroll
          for (int j = 0; j < M; ++j) zu[j] = 0.0;
          double gt[N];
#pragma unroll
          for (int i = 0; i < N; ++i) gt[i] = goals[(size_t)t * N + i];
          MD::template jac<false>(mc, gt, zu, ac, bc);
          expand_jac<MD>(mc, ac, bc, S, Bm);
          double W[N][N], Tm[N][M], Mx[M][M], Mi[M][M], BtPA[M][N], Kt[M][N];
          mul_VA<MD>(P, S, W);
          mul_VB<MD>(P, Bm, Tm);
#pragma unroll
          for (int r = 0; r < M; ++r) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
              double acc = 0.0;
#pragma unroll
              for (int k = 0; k < N; ++k)
                if (b_nz<MD>(k, r)) acc = fma(Bm[k][r], W[k][j], acc);
              BtPA[r][j] = acc;
            }
#pragma unroll
            for (int b = 0; b < M; ++b) {
              double acc = (r == b) ? init_r : 0.0;
#pragma unroll
              for (int k = 0; k < N; ++k)
                if (b_nz<MD>(k, r)) acc = fma(Bm[k][r], Tm[k][b], acc);
              Mx[r][b] = acc;
            }
          }
          inv_small<M>(Mx, Mi);
          double* rt = rec(t);
#pragma unroll
          for (int r = 0; r < M; ++r)
#pragma unroll
            for (int j = 0; j < N; ++j) {
              double acc = 0.0;
#pragma unroll
              for (int b = 0; b < M; ++b) acc = fma(Mi[r][b], BtPA[b][j], acc);
              Kt[r][j] = acc;
              if (gl == 0) rt[(size_t)(RL::rK + r * N + j) * ES] = acc;
            }
          // P = I + A'(W - T K)
          double D[N][N];
#pragma unroll
          for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
              double acc = W[i][j];
#pragma unroll
              for (int b = 0; b < M; ++b) acc = fma(-Tm[i][b], Kt[b][j], acc);
              D[i][j] = acc;
            }
#pragma unroll
          for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
              double acc = ((i == j) ? 1.0 : 0.0) + D[i][j];
#pragma unroll
              for (int k = 0; k < N; ++k)
                if (s_nz<MD>(k, i)) acc = fma(S[k][i], D[k][j], acc);
              P[i][j] = acc;
            }
        }
        __threadfence_block();
        // roll-out u = -K (x - goal) from the raw start state; trajectory goes to slot (lane 0, parity 0)
        {
          const int cs = 0;
          double x[N];
#pragma unroll
          for (int i = 0; i < N; ++i) x[i] = ALILQR_COLD(x0)[(size_t)p * N + i];
          if (gl == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
              if (o_xi) o_xi[(size_t)p * XS + i] = x[i];
              o_xo[(size_t)p * XS + i] = x[i];
              // node 0 keeps the wrapped start state from its first cost() call on (lat_bicycle_node.py:27-30,94)
              *trj_at(0, i, cs) = (!NEWAL && MD::WRAPS && (i == 2 || i == 3)) ? wrap_py(x[i]) : x[i];
            }
          }
          for (int t = 0; t < H; ++t) {
            const double* rt = rec(t);
            double u[M], xn[N];
#pragma unroll
            for (int r = 0; r < M; ++r) {
              double acc = 0.0;
#pragma unroll
              for (int i = 0; i < N; ++i) acc += (-rt[(size_t)(RL::rK + r * N + i) * ES]) * (x[i] - goals[(size_t)t * N + i]);
              if (i_uoff) acc = acc + i_uoff[(size_t)p * US + t * M + r];
              u[r] = acc;
            }
            MD::template step<false>(mc, x, u, xn);
            if (gl == 0) {
#pragma unroll
              for (int r = 0; r < M; ++r) {
                *trj_at(0, XS + t * M + r, cs) = u[r];
                if (o_ui) o_ui[(size_t)p * US + t * M + r] = u[r];
                o_uo[(size_t)p * US + t * M + r] = u[r];
              }
#pragma unroll
              for (int i = 0; i < N; ++i) {
                *trj_at(0, (t + 1) * N + i, cs) = xn[i];
                if (o_xi) o_xi[(size_t)p * XS + (t + 1) * N + i] = xn[i];
                o_xo[(size_t)p * XS + (t + 1) * N + i] = xn[i];
              }
            }
#pragma unroll
            for (int i = 0; i < N; ++i) x[i] = xn[i];
          }
        }
        mu = 1.0;
        {
          const double* const msi = NEWAL ? ALILQR_COLD(mu_s_in) : nullptr;
          mu_s = msi ? msi[p] : 1.0;
        }
        xo_pend = false;
        cur_slot = 0; cur_par = 0;
        outer = 0; inner = 0; n_inner = 0; n_trials = 0;
        outer_start = true;
        phase = PH_LIN;
        __threadfence_block();
      }
    }
    PROF_MARK(0)
    // every group of the wave idle and nothing left to pull: done
    if (__ballot(p >= 0) == 0ull) break;

    bool tables_changed = false;
    if (RECT && p >= 0 && phase == PH_LIN) {
      {
        // ---- row activation (new_al_ilqr.h:525-625): nodes in parallel; a node inside a rectangle gets the
        //      rectangle's y-bound written into the next row of its own table
        bool mine = false;
        for (int t = gl; t <= H; t += G) {
          double* r = rec(t);
          const double px = *trj_at(cur_par, t * N + 0, cur_slot), py = *trj_at(cur_par, t * N + 1, cur_slot);
          ctab Cv = tabC(t) + C * (N + M);
          for (int side = 0; side < 2; ++side) {
            ctab obs = as_ctab(side == 0 ? ALILQR_COLD3(left_obs) : ALILQR_COLD3(right_obs));
            const int cnt = side == 0 ? ALILQR_COLD3(n_left) : ALILQR_COLD3(n_right);
            for (int k = 0; k < cnt; ++k) {
              ctab vx = obs + (size_t)k * 8, vy = vx + 4;
              bool inside = true;
              double ext = vy[0];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int e2 = (e + 1) & 3;
                const double ex = vx[e2] - vx[e], ey = vy[e2] - vy[e];
                const double qx = px - vx[e], qy = py - vy[e];
                const double cr = qx * ey - qy * ex;
                if (!(cr < 0)) inside = false;
                if (side == 0) { if (vy[e] > ext) ext = vy[e]; } else { if (vy[e] < ext) ext = vy[e]; }
              }
              if (!inside) continue;
              const double sgn = (side == 0) ? -1.0 : 1.0, cnew = (side == 0) ? ext : -ext;
              bool exist = false;
              for (int i = 0; i < C; ++i) {
                const double ci = (r[(size_t)(RL::rl + C + i) * ES] != 0.0) ? r[(size_t)(RL::rl + 2 * C + i) * ES] : Cv[i];
                exist |= (ci == cnew);
              }
              if (exist) continue;
              int idx = (int)r[(size_t)(RL::rl + 3 * C) * ES];
              if (!QUAD) {
                if (idx + 1 >= C) continue;          // the reference would write past its fixed-size table here
                idx += 1;
                r[(size_t)(RL::rl + 3 * C) * ES] = (double)idx;
              }
              if (idx < 0 || idx >= C) continue;
              r[(size_t)(RL::rl + C + idx) * ES] = sgn;
              r[(size_t)(RL::rl + 2 * C + idx) * ES] = cnew;
              mine = true;
            }
          }
        }
        tables_changed = gfirst<G>(mine, lane) >= 0;
        __threadfence_block();
      }
    }
    // NEW_AL: the rows this trip of the loop evaluates.  A row of the shared table that is all zero counts only at nodes
    // where its multiplier or its rectangle overlay is not zero (NodeOps::for_live): the lanes of each group look through
    // the group's nodes, and a row that is live in any group of the wave is evaluated by the whole wave (evaluating a
    // dead row changes nothing).  Dead rows stay dead until the next row activation: their c is +0, so the multiplier
    // update leaves a zero multiplier at zero.
    // The look happens where the multipliers or the overlay can have changed - a group's first trip with a problem, after
    // a multiplier update (outer_start) and after a row activation - and its result (live_g) is kept in between.
    unsigned long long live_rows = ~0ull;
    if constexpr (LIVE) {
      const unsigned long long zr = a.zrows;
      if (zr != 0ull) {
        // (4-state kernels; the 6-state ones look on every trip: the two more registers of the kept mask cost them
        //  7 to 12 %)
        constexpr bool KEEP = N <= 4;
        // 6-state kernels with the multiplier buffers in LDS keep the look there, one word per lane behind the buffers
        constexpr bool KEEPL = !KEEP && RLDS && (ALILQR_LIVE_LDS != 0);
        unsigned long long* const live_slot = reinterpret_cast<unsigned long long*>(lds_pool + rows_off + (size_t)2 * rows_nch * 128) + lane;
        if (p >= 0 && ((!KEEP && !KEEPL) || (phase == PH_LIN && (outer_start || tables_changed)))) {
          unsigned long long lm = 0ull;
          for (int t = gl; t <= H; t += G) {
            const double* lp = rec(t) + (size_t)RL::rl * ES;
            for (unsigned long long z = zr; z != 0ull; z &= z - 1ull) {
              const int i = __builtin_ctzll(z);
              bool nz = lp[(size_t)i * ES] != 0.0;
              if (RECT) nz = nz || (lp[(size_t)(C + i) * ES] != 0.0);
              if (nz) lm |= 1ull << i;
            }
          }
          live_g = lm;                                  // this lane's nodes; the union over the wave is formed below
          if constexpr (KEEPL) *live_slot = lm;
        } else if (KEEPL) {
          live_g = (p >= 0) ? *live_slot : 0ull;
        } else if (!KEEP) {
          live_g = 0ull;
        }
        const unsigned long long lm = (p >= 0) ? live_g : 0ull;
        live_rows = ~zr;
        for (unsigned long long z = zr; z != 0ull; z &= z - 1ull) {
          const int i = __builtin_ctzll(z);
          if (__ballot(((lm >> i) & 1ull) != 0ull) != 0ull) live_rows |= 1ull << i;
        }
      }
    }
    live_now = live_rows;

    if (p >= 0 && phase == PH_LIN) {
     {
      // ================= outer-iteration start: J under the current multipliers =================
      if (outer_start || tables_changed) {
        // J of the current trajectory under the current multipliers and tables (what CalcDerivatives leaves in cost_list_)
        old_cost = total_cost();
        if (outer_start) inner = 0;
        outer_start = false;
      }
      np_ = 1 - cur_par;
      ref_slot = cur_slot;

      PROF_MARK(1)
      // ================= P1: linearise + quadratise, parallel in time =================
      for (int t = gl; t <= H; t += G) {
        double x[N], u[M], g[N];
        load_node(t, x, u);
        double* r = rec(t);
        double* lr = lrec(t);
#pragma unroll
        for (int i = 0; i < N; ++i) g[i] = goal_of(p)[(size_t)t * N + i];
        if (t < H) {
          if constexpr (PACK) {
            double jc[NR::NJA];
            JacPack<MD>::template pack<true>(mc, x, u, jc);
#pragma unroll
            for (int i = 0; i < NR::NJA; ++i) lr[(size_t)(NR::oA + i) * ES] = jc[i];
          } else {
            double ac[MD::NA], bc[MD::NB];
            MD::template jac<!NEWAL>(mc, x, u, ac, bc);
#pragma unroll
            for (int i = 0; i < MD::NA; ++i) lr[(size_t)(NR::oA + i) * ES] = ac[i];
#pragma unroll
            for (int i = 0; i < MD::NB; ++i) lr[(size_t)(NR::oB + i) * ES] = bc[i];
          }
          if (NEWAL) {
            double ff[8];
            FxxOf<MD, NEWAL>::eval(mc, x, u, ff);
#pragma unroll
            for (int i = 0; i < NR::NF; ++i) lr[(size_t)(NR::oF + i) * ES] = ff[i];
          }
        }
        double q[LL::E];
        if constexpr (LEGACY) {
          double lam[CL];
#pragma unroll
          for (int i = 0; i < CL; ++i) lam[i] = r[(size_t)(RL::rl + i) * ES];
          // l_xx = 2Q + diag, l_uu = 2R + diag: the record keeps the diagonal only (2Q, 2R are uniform tables), or,
          // packed, the code of its AL part
          if constexpr (PACK) {
            int hcode = 0;
            OPS::quadratize(a, tabQ(t), tabR(t), tabC(t), x, u, g, lam, 1, mu, q, 1, &hcode);
            lr[(size_t)NR::oHx * ES] = (double)hcode;
          } else {
            OPS::quadratize(a, tabQ(t), tabR(t), tabC(t), x, u, g, lam, 1, mu, q, 1);
#pragma unroll
            for (int i = 0; i < N; ++i) lr[(size_t)(NR::oHx + i) * ES] = q[LL::oLxx + i * N + i];
#pragma unroll
            for (int i = 0; i < M; ++i) lr[(size_t)(NR::oHu + i) * ES] = q[LL::oLuu + i * M + i];
          }
        } else {
          if constexpr (LIVE) OPS::template quadratize<true>(a, tabQ(t), tabR(t), tabC(t), x, u, g, r + (size_t)RL::rl * ES, ES, mu, q, 1, live_rows);
          else OPS::quadratize(a, tabQ(t), tabR(t), tabC(t), x, u, g, r + (size_t)RL::rl * ES, ES, mu, q, 1);
#pragma unroll
          for (int i = 0; i < N * N; ++i) lr[(size_t)(NR::oHx + i) * ES] = q[LL::oLxx + i];
#pragma unroll
          for (int i = 0; i < M * M; ++i) lr[(size_t)(NR::oHu + i) * ES] = q[LL::oLuu + i];
        }
#pragma unroll
        for (int i = 0; i < N; ++i) lr[(size_t)(NR::oLx + i) * ES] = q[LL::oLx + i];
#pragma unroll
        for (int i = 0; i < M; ++i) lr[(size_t)(NR::oLu + i) * ES] = q[LL::oLu + i];
      }
      __threadfence_block();

      PROF_MARK(2)
      // ================= P2: backward Riccati sweep =================
      {
        // Registers of one linearisation record.  The part needed first (f_x, f_u, l_x, l_u) is loaded one step
        // ahead into a second buffer; the second-order part (l_xx, l_uu, f_xx), needed late in a step, is loaded
        // at the top of its own step and lands while the first products are computed.  LEGACY records are
        // short (diagonal second-order part) and are loaded ahead whole.
        constexpr int EA = LEGACY ? NR::E : NR::oHx;
        struct LinRegs { double v[EA]; };
        struct LinLate { double v[NR::E - EA + 1]; };
        auto load_lin = [&](int t, LinRegs& L) {
          const double* lr = lrec(t);
#pragma unroll
          for (int i = 0; i < EA; ++i) L.v[i] = lr[(size_t)i * ES];
        };
        auto load_late = [&](int t, LinLate& Lt) {
          const double* lr = lrec(t);
#pragma unroll
          for (int i = EA; i < NR::E; ++i) Lt.v[i - EA] = lr[(size_t)i * ES];
        };
        double Vx[N], Vxx[N][N];
        {
          LinRegs LH;
          LinLate LHt;
          load_lin(H, LH);
          load_late(H, LHt);
          ctab Q = tabQ(H);
#pragma unroll
          for (int i = 0; i < N; ++i) {
            Vx[i] = LH.v[NR::oLx + i];
#pragma unroll
            for (int j = 0; j < N; ++j) {
              if constexpr (PACK) Vxx[i][j] = (i == j) ? 2 * Q[i * N + j] + OPS::hess_diag((int)LH.v[NR::oHx], i, mu) : 2 * Q[i * N + j];
              else if constexpr (LEGACY) Vxx[i][j] = (i == j) ? LH.v[NR::oHx + i] : 2 * Q[i * N + j];
              else Vxx[i][j] = LHt.v[NR::oHx - EA + i * N + j];
            }
          }
        }
        auto sweep = [&](int t, const LinRegs& L, LinRegs& Lnext) {
          LinLate Lt;
          if (!LEGACY) load_late(t, Lt);
          if (t > 0) load_lin(t - 1, Lnext);            // in flight while this step computes
          double S[N][N], Bm[N][M], K[M][N], kf[M], dl;
          int hcode = 0;
          if constexpr (PACK) {
            double ac[MD::NA], bc[MD::NB];
            JacPack<MD>::unpack(mc, &L.v[NR::oA], ac, bc);
            expand_jac<MD>(mc, ac, bc, S, Bm);
            hcode = (int)L.v[NR::oHx];
          } else {
            expand_jac<MD>(mc, &L.v[NR::oA], &L.v[NR::oB], S, Bm);
          }
          ctab Q = tabQ(t);
          ctab R = tabR(t);
          riccati_step<MD>(S, Bm,
                           [&](int i) { return L.v[NR::oLx + i]; },
                           [&](int r) { return L.v[NR::oLu + r]; },
                           [&](int i, int j) {
                             if constexpr (PACK) return (i == j) ? 2 * Q[i * N + j] + OPS::hess_diag(hcode, i, mu) : 2 * Q[i * N + j];
                             else if constexpr (LEGACY) return (i == j) ? L.v[NR::oHx + i] : 2 * Q[i * N + j];
                             else return Lt.v[NR::oHx - EA + i * N + j];
                           },
                           [&](int r, int b) {
                             if constexpr (PACK) return (r == b) ? 2 * R[r * M + b] + OPS::hess_diag(hcode, N + r, mu) : 2 * R[r * M + b];
                             else if constexpr (LEGACY) return (r == b) ? L.v[NR::oHu + r] : 2 * R[r * M + b];
                             else return Lt.v[NR::oHu - EA + r * M + b];
                           },
                           [&](double acc, int i, int j, const double* Vo) {
                             return FxxOf<MD, NEWAL>::add(acc, &Lt.v[NEWAL ? NR::oF - EA : 0], Vo, i, j);
                           },
                           a.prm.quu_reg, &dl, Vx, Vxx, K, kf);
          dv_lin += dl;
          if (gl == 0) {
            double* rt = rec(t);
#pragma unroll
            for (int r = 0; r < M; ++r) {
#pragma unroll
              for (int j = 0; j < N; ++j) rt[(size_t)(RL::rK + r * N + j) * ES] = K[r][j];
              rt[(size_t)(RL::rk + r) * ES] = kf[r];
            }
          }
        };
        dv_lin = 0.0;
        LinRegs LA, LB;
        load_lin(H - 1, LA);
        int t = H - 1;
        for (; t >= 1; t -= 2) {
          sweep(t, LA, LB);
          sweep(t - 1, LB, LA);
        }
        if (t == 0) sweep(0, LA, LB);
      }
      __threadfence_block();
      // ---- a new line search starts
      nt0 = n_trials;
      new_cost = old_cost;
      keep_old = false;                          // NEW_AL: the trial costs were not finite, the trajectory stays
      win_trial = -1; win_lane = -1;
      J_last = 0.0;
      gate_skip = NEWAL && (fabs(dv_lin) < ALILQR_COLD2(prm.dv_gate));      // new_al_ilqr.h:1039
      stage = gate_skip ? ST_DONE : ST_TRIALS;
      r0 = 0;                                    // first trial of the current round, or first ladder column
      replay_kind = 0; replay_alpha = 0.0; bestJ = 0.0; bestA = 0.0; have = false;
      phase = PH_ROLL;
     }

    }
      PROF_MARK(3)
      P3_START
      // ================= P3: roll-outs, one line-search trial per lane =================
      // What a roll-out reads and writes of a problem: the group's block (step records), the trajectory slots
      // of the parity being written, the penalty.  A lane normally works for its own group; in the compacted
      // passes of the LEGACY line search it works for whichever group of the wave still has trials to run.
      //   blk   the group's block (K, k, multipliers per node)        cur   its current trajectory (x | u)
      //   trjp  the trajectory slots of the parity being written       goal  its goal list
      //   lm0/1 which nodes carry a multiplier that is not zero (LEGACY)
      struct Tgt { double* blk; double* trjp; const double* cur; const double* goal; double mu; unsigned long long lm0, lm1; };
      struct RecRegs { double x[N], u[M], K[M][N], k[M], g[N], lam[CL]; };
      // A lane loads what its trial needs of node t: the goal (through scalar loads when all problems share one
      // goal list), the reference point and the gains only for a closed-loop trial, the multipliers only where
      // the node has any.  The open-loop trials of the LEGACY line search (all but the first) need neither.
      auto load_rec = [&](const Tgt& tg, int t, RecRegs& R, bool closed) {
        const double* r = tg.blk + (size_t)t * RS * ES;
#pragma unroll
        for (int i = 0; i < N; ++i) R.g[i] = tg.goal[(size_t)t * N + i];
        if (!LEGACY || closed || !ALILQR_LEAN_LOADS) {
#pragma unroll
          for (int i = 0; i < N; ++i) R.x[i] = tg.cur[(size_t)t * N + i];
#pragma unroll
          for (int j = 0; j < M; ++j) {
            R.u[j] = (t < H) ? tg.cur[(size_t)XS + (size_t)t * M + j] : 0.0;
            R.k[j] = r[(size_t)(RL::rk + j) * ES];
#pragma unroll
            for (int i = 0; i < N; ++i) R.K[j][i] = r[(size_t)(RL::rK + j * N + i) * ES];
          }
        }
        if (LEGACY) {
          const bool lam_on = !ALILQR_LEAN_LOADS || lam_all || (((t < 64 ? tg.lm0 >> t : tg.lm1 >> (t - 64)) & 1ull) != 0ull);
          if (lam_on) {
#pragma unroll
            for (int i = 0; i < CL; ++i) R.lam[i] = r[(size_t)(RL::rl + i) * ES];
          } else {
#pragma unroll
            for (int i = 0; i < CL; ++i) R.lam[i] = 0.0;
          }
        }
      };
      // Only the trials that usually win keep their trajectory (LEGACY: 1, 2 and the last one, which the
      // reference is left with when every step is rejected; FAST: the first SS-1).  Any other winner is
      // rolled out once more into the spare slot SS-1: same code, same inputs, same bits.
      auto store_slot = [&](int j) -> int {
        if (LEGACY) return (j == 1) ? 0 : (j == 2) ? 1 : (j == ntrials) ? 2 : -1;
        // FAST at 4 and 8 lanes: the first trial wins four searches in five and the phase behind the roll-outs waits for
        // their stores; trials 3 and up keep nothing and are rolled out again when they win (box + linear rows H = 100 +3
        // to +7 %, box rows -2 to +1 %; keeping trial 1 only: +3 % / -4 %; 16 lanes: box rows -9 to -14 %, not there)
        if constexpr (MODE == ALILQR_MODE_FAST && G <= ALILQR_FAST_KEEP_G && ALILQR_FAST_KEEP > 0) return (j <= ALILQR_FAST_KEEP) ? j - 1 : -1;
        return (j <= SS - 1) ? j - 1 : -1;
      };
      // roll out one trial.  kind 0: closed loop, step alpha (Python generations: (u + alpha k) + K dx;
      // C++ generation: u + (K dx + alpha k)); kind 1: LEGACY open loop on the accumulated controls of
      // lane `ul`; kind 2: one column of the C++ generation's five-wide ladder (no angle wrap, (u + K dx) + k alpha)
      // fly2 (accumulated controls in LDS): the lane runs trial 2, whose controls u + k / 2 are formed from the record
      auto rollout = [&](const Tgt& tg, bool valid, int kind, double alpha, const double* ucol, size_t ucs, int sslot, bool fly2 = false) -> double {
        double J = 0.0;
        double x[N];
        const double mu = tg.mu;                       // shadows the group's own penalty
        (void)ALILQR_CHK(sslot < SS, 5);
        const size_t bdoubles = block_doubles<MD, MODE>(H, C);
        (void)ALILQR_CHK(tg.blk >= a.blk && tg.blk + bdoubles <= a.blk + NG * bdoubles, 6);
        (void)bdoubles;
        (void)ALILQR_CHK(tg.trjp >= trj && tg.trjp + (size_t)SS * T <= trj + NG * 2 * SS * (size_t)T, 7);
        (void)ALILQR_CHK(tg.cur >= trj && tg.cur + T <= trj + NG * 2 * SS * (size_t)T, 8);
        (void)ALILQR_CHK(!(LEGACY && !LDSUC && kind == 1 && valid) || (ucol >= a.ucum && ucol + (size_t)(US - 1) * ucs < a.ucum + NG * (size_t)US * ucs), 9);
        double* const tout = tg.trjp + (size_t)(sslot >= 0 && sslot < SS ? sslot : 0) * T;
        auto rec = [&](int t) -> double* {
          if (!ALILQR_CHK(t >= 0 && t <= H, 10)) t = 0;
          return tg.blk + (size_t)t * RS * ES;
        };
        const bool closed = !(LEGACY && kind == 1);
        PASS_START
        RecRegs RA, RB;
        load_rec(tg, 0, RA, closed);
        int rows_tok = 0;                              // NEW_AL: token of the multiplier copy in flight (rows_dma)
        if constexpr (RLDS) rows_tok = rows_dma<G>(rec(0) + (size_t)RL::rl * ES, lds_pool + rows_off, rows_nch, gl);
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = tg.cur[i];
        double uown[M];
#pragma unroll
        for (int r = 0; r < M; ++r) uown[r] = (LEGACY && kind == 1 && valid && !(LDSUC && fly2)) ? ucol[(size_t)r * ucs] : 0.0;
        // tables of node t as the roll-out reads them
        auto rQ = [&](int t) { if constexpr (TLDS) { (void)t; return ltQ; } else return tabQ(t); };
        auto rR = [&](int t) { if constexpr (TLDS) { (void)t; return ltR; } else return tabR(t); };
        auto rC = [&](int t) { if constexpr (TLDS) { (void)t; return ltC; } else return tabC(t); };
        auto roll = [&](int t, const RecRegs& R, RecRegs& Rn) {
          load_rec(tg, t + 1, Rn, closed);             // next record in flight while this step computes
          if (TLDS && a.table_nodes > 1) stage_tables(t);
          double unext[M];
#pragma unroll
          for (int r = 0; r < M; ++r)
            unext[r] = (LEGACY && kind == 1 && valid && !(LDSUC && fly2) && t + 1 < H) ? ucol[(size_t)((t + 1) * M + r) * ucs] : 0.0;
          double u[M];
          if (LEGACY && kind == 1) {
#pragma unroll
            for (int r = 0; r < M; ++r) u[r] = uown[r];
            if constexpr (LDSUC) {
              if (fly2) {
                const double al2 = 1.0 / 2.0;                                   // the step of trial 2
#pragma unroll
                for (int r = 0; r < M; ++r) u[r] = R.u[r] + al2 * R.k[r];      // the first accumulation (ilqr.py:119 after one rejected step)
              }
            }
          } else {
#pragma unroll
            for (int r = 0; r < M; ++r) {
              double fb = 0.0;
#pragma unroll
              for (int i = 0; i < N; ++i) fb += R.K[r][i] * (x[i] - R.x[i]);
              if (NEWAL) u[r] = (kind == 2) ? ((R.u[r] + fb) + R.k[r] * alpha) : (R.u[r] + (fb + alpha * R.k[r]));
              else u[r] = (R.u[r] + alpha * R.k[r]) + fb;
            }
          }
          // The node's (x_t, u_t) is stored between its cost and its dynamics: the stores then complete under the
          // dynamics' arithmetic instead of ahead of the next node's loads (loads and stores share one in-order
          // counter, so a wait for a load also waits for every store issued before it).
          auto keep = [&]() {
            if (sslot >= 0) {
#pragma unroll
              for (int r = 0; r < M; ++r) tout[XS + t * M + r] = u[r];
#pragma unroll
              for (int i = 0; i < N; ++i) tout[t * N + i] = x[i];
            }
          };
          if constexpr (!LEGACY) {
            if (valid) {
              auto node_J = [&](const auto& lp) -> double {
                if constexpr (MERGE) {
                  return OPS::template cost2<LIVE>(a, rQ(t), rR(t), rC(t), x, u, R.g, lp, ES, mu, kind == 2, live_rows);
                } else if constexpr (NEWAL) {
                  if (kind == 2) return OPS::template par_cost<LIVE>(a, rQ(t), rR(t), rC(t), x, u, R.g, lp, ES, mu, live_rows);
                  return OPS::template cost<LIVE>(a, rQ(t), rR(t), rC(t), x, u, R.g, lp, ES, mu, live_rows);
                } else {
                  return OPS::cost(a, tabQ(t), tabR(t), tabC(t), x, u, R.g, lp, ES, mu);
                }
              };
              if constexpr (RLDS) {
                const RowsLds<CHD> lp{as_ltab(lds_pool + rows_off + (size_t)(t & 1) * rows_nch * 128) + rows_bind((lane / G) * CHD, rows_tok)};
                J += node_J(lp);
              } else {
                const double* lp = rec(t) + (size_t)RL::rl * ES;
                J += node_J(lp);
              }
            }
            // the next node's multiplier block: every lane of the group brings its 16 bytes, issued behind this node's
            // LDS reads and ahead of its stores (the wait for the copy then leaves the stores in flight)
            if constexpr (RLDS) rows_tok = rows_dma<G>(rec(t + 1) + (size_t)RL::rl * ES, lds_pool + rows_off + (size_t)((t + 1) & 1) * rows_nch * 128, rows_nch, gl);
            if (valid) {
              double xn[N];
              keep();
              if constexpr (MERGE) {
                MD::step2(mc, x, u, xn, kind == 2);
              } else if constexpr (NEWAL) {
                if (kind == 2) MD::par_step(mc, x, u, xn);
                else MD::template step<false>(mc, x, u, xn);
              } else {
                MD::template step<true>(mc, x, u, xn);
              }
#pragma unroll
              for (int i = 0; i < N; ++i) x[i] = xn[i];
            }
          } else {
            if (valid) {
              double xn[N];
              J += OPS::cost(a, tabQ(t), tabR(t), tabC(t), x, u, R.g, R.lam, 1, mu);
              keep();
              MD::template step<true>(mc, x, u, xn);
#pragma unroll
              for (int i = 0; i < N; ++i) x[i] = xn[i];
            }
          }
#pragma unroll
          for (int r = 0; r < M; ++r) uown[r] = unext[r];
        };
        int t = 0;
        for (; t + 1 < H; t += 2) {
          roll(t, RA, RB);
          roll(t + 1, RB, RA);
        }
        if (t < H) {
          roll(t, RA, RB);
#pragma unroll
          for (int i = 0; i < N; ++i) RA.g[i] = RB.g[i];
          if (LEGACY) {
#pragma unroll
            for (int i = 0; i < CL; ++i) RA.lam[i] = RB.lam[i];
          }
        }
        if (valid) {
          if (sslot >= 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) tout[H * N + i] = x[i];
          }
          double zu[M];
#pragma unroll
          for (int r = 0; r < M; ++r) zu[r] = 0.0;
          if constexpr (!LEGACY) {
            if (TLDS && a.table_nodes > 1) stage_tables(H);
            auto node_J = [&](const auto& lp) -> double {
              if constexpr (MERGE) {
                return OPS::template cost2<LIVE>(a, rQ(H), rR(H), rC(H), x, zu, RA.g, lp, ES, mu, kind == 2, live_rows);
              } else if constexpr (NEWAL) {
                if (kind == 2) return OPS::template par_cost<LIVE>(a, rQ(H), rR(H), rC(H), x, zu, RA.g, lp, ES, mu, live_rows);
                return OPS::template cost<LIVE>(a, rQ(H), rR(H), rC(H), x, zu, RA.g, lp, ES, mu, live_rows);
              } else {
                return OPS::cost(a, tabQ(H), tabR(H), tabC(H), x, zu, RA.g, lp, ES, mu);
              }
            };
            if constexpr (RLDS) {
              const RowsLds<CHD> lp{as_ltab(lds_pool + rows_off + (size_t)(H & 1) * rows_nch * 128) + rows_bind((lane / G) * CHD, rows_tok)};
              J += node_J(lp);
            } else {
              const double* lp = rec(H) + (size_t)RL::rl * ES;
              J += node_J(lp);
            }
          } else {
            J += OPS::cost(a, tabQ(H), tabR(H), tabC(H), x, zu, RA.g, RA.lam, 1, mu);
          }
        }
        PASS_END
        return J;
      };
      auto alpha_of = [&](int j) -> double {     // 2^-(j-1) by exact halvings (ilqr.py:136)
        double al = 1.0;
        for (int q = 1; q < j && q < 64; ++q) al = al / 2.0;
        return al;
      };
      if constexpr (LWIDE) {
        // ---- LEGACY line search, wave-wide.  Trial 1 is the closed-loop step; trials >= 2 run open loop on
        //      accumulated controls (ilqr.py:119 with new_u aliasing u after the first rejected step).
        //      Pass A: every group rolls out its first G trials on its own lanes.  Passes B: the trials that are
        //      left of the groups whose pass A found no acceptable step are spread over ALL 64 lanes of the
        //      wave, so a narrow group borrows the lanes of the groups that are already done.  The first
        //      (largest step) accepted trial wins, exactly the sequential rule.  Pass C (rare): a winner whose
        //      trajectory was not kept is rolled out once more, lane 0 keeps it in the spare slot.
        //      All passes go through ONE roll-out call site (the code exists once in the instruction cache).
        const bool act = (p >= 0);
        const size_t gid0 = gid - (size_t)(lane / G);                    // first group of this wave
        // open-loop trials uc0 .. ntrials keep their accumulated controls: all of them (2 ..) in HBM, or 3 .. in the
        // group's share of the LDS pool (trial 2 is formed on the fly)
        const int uc0 = LDSUC ? 3 : 2;
        const size_t UCS = LDSUC ? (size_t)(a.uc_cols > 0 ? a.uc_cols : 1) : (size_t)(ntrials > 1 ? ntrials - 1 : 1);
        const size_t UC = LDSUC ? PGS : UCS * US;                        // accumulated controls of one group (LDS: its pool share)
        double* const ucbase = LDSUC ? lds_pool : a.ucum;
        double* const ucg = LDSUC ? linb : a.ucum + gid * UC;            // [t * M + r][trial - uc0]: a step's controls of all trials are contiguous
        if (act) {
          // all accumulated controls, nodes in parallel:  u_j = u_{j-1} + 2^-(j-1) k
          for (int t = gl; t < H; t += G) {
            const double* rt = rec(t);
#pragma unroll
            for (int r = 0; r < M; ++r) {
              const double kf = rt[(size_t)(RL::rk + r) * ES];
              double sacc = *trj_at(cur_par, XS + t * M + r, cur_slot);
              double al = alpha_of(2);
              for (int i = 2; i <= ntrials; ++i) {
                sacc = sacc + al * kf;
                al = al / 2.0;
                if (i >= uc0) ucg[(size_t)(t * M + r) * UCS + (i - uc0)] = sacc;
              }
            }
          }
        }
        __threadfence_block();
        const Tgt own{blk, trj + ((gid * 2 + (size_t)np_) * SS) * (size_t)T, trj_at(cur_par, 0, ref_slot), goal_of(p), mu, lmask0, lmask1};
        const int R = ntrials - G;                                       // trials left per group after pass A
        const int nA = (G < ntrials) ? G : ntrials;
        enum { PASS_A = 0, PASS_B = 1, PASS_C = 2 };
        int pass = PASS_A, base = 0, total = 0, my_rank = 0, found = -1;
        unsigned long long nb = 0ull;
        double foundJ = 0.0, JlastL = 0.0;
        bool need = false;
        for (;;) {
          Tgt tg = own;
          bool v;
          int kind, sslot, trial;
          double oc_t = old_cost;
          size_t gid_t = gid;
          if (pass == PASS_A) {
            trial = gl + 1;
            v = act && trial <= ntrials;
          } else if (pass == PASS_B) {
            const int item = base + lane;
            v = item < total;
            const int fr = v ? item / R : 0;
            trial = G + 1 + (v ? item % R : 0);
            int ol = 0;                                                  // first lane of the fr-th searching group
            {
              unsigned long long m = nb;
              for (int q = 0; q < fr; ++q) m &= m - 1ull;
              ol = m ? (__ffsll((long long)m) - 1) : 0;
            }
            const double mu_t = __shfl(mu, ol);
            oc_t = __shfl(old_cost, ol);
            const int np_t = __shfl(np_, ol), cs_t = __shfl(ref_slot, ol), p_t = __shfl(p, ol);
            const unsigned long long lm0_t = __shfl(lmask0, ol), lm1_t = __shfl(lmask1, ol);
            gid_t = gid0 + (size_t)(ol / G);
            tg = Tgt{a.blk + gid_t * block_doubles<MD, MODE>(H, C), trj + ((gid_t * 2 + (size_t)np_t) * SS) * (size_t)T,
                     trj + ((gid_t * 2 + (size_t)(1 - np_t)) * SS + (size_t)cs_t) * (size_t)T, goal_of(p_t), mu_t, lm0_t, lm1_t};
          } else {
            trial = win_trial;
            v = act && store_slot(win_trial) < 0;
          }
          (void)ALILQR_CHK(!v || (trial >= 1 && trial <= ntrials), 12);
          (void)ALILQR_CHK(gid_t < NG, 13);
          kind = (trial != 1) ? 1 : 0;
          sslot = !v ? -1 : (pass == PASS_C) ? ((gl == 0) ? SS - 1 : -1) : store_slot(trial);
          const size_t gsel = LDSUC ? (size_t)(gid_t - gid0) : gid_t;    // the target group's controls: its pool share, or its HBM block
          const double J = rollout(tg, v, kind, 1.0, ucbase + gsel * UC + (size_t)(trial >= uc0 ? trial - uc0 : 0), UCS, sslot,
                                   LDSUC && trial == 2);
          if (pass == PASS_C) break;
          bool search_over = false;
          if (pass == PASS_A) {
            const int w = gfirst<G>(v && (J < old_cost), lane);
            if (w >= 0) found = w + 1;
            foundJ = gbcast<G>(J, w >= 0 ? w : 0);
            JlastL = gbcast<G>(J, nA - 1);                               // the last trial's cost once all have run
            need = act && found < 0 && R > 0;
            nb = __ballot(need && gl == 0);                              // first lanes of the groups still searching
            total = (R > 0) ? __popcll(nb) * R : 0;
            my_rank = __popcll(nb & ((1ull << (lane & ~(G - 1))) - 1ull));
            if (total > 0) pass = PASS_B; else search_over = true;
          } else {
            const unsigned long long okb = __ballot(v && (J < oc_t));
            // lanes [lo, hi) of this pass ran my group's trials G+1 .. ntrials
            const int lo = my_rank * R - base, hi = lo + R;
            int src_w = lane, src_l = lane, hit_trial = -1;
            bool has_last = false;
            if (need && found < 0) {
              const int l0 = lo < 0 ? 0 : lo, h0 = hi > 64 ? 64 : hi;
              if (l0 < h0) {
                const unsigned long long msk = ((h0 - l0 >= 64) ? ~0ull : ((1ull << (h0 - l0)) - 1ull)) << l0;
                const unsigned long long mm = okb & msk;
                if (mm) {
                  const int wl = __ffsll((long long)mm) - 1;
                  src_w = wl;
                  hit_trial = G + 1 + (base + wl - my_rank * R);
                }
                if (hi >= 1 && hi <= 64) { has_last = true; src_l = hi - 1; }
              }
            }
            const double Jw = __shfl(J, src_w), Jl = __shfl(J, src_l);
            if (hit_trial >= 0) { found = hit_trial; foundJ = Jw; }
            if (has_last) JlastL = Jl;
            base += 64;
            if (base >= total || __ballot(need && found < 0) == 0ull) search_over = true;
          }
          if (search_over) {
            if (act) {
              if (found >= 0) {
                win_trial = found; new_cost = foundJ; n_trials += found;
              } else {
                // every step rejected: the reference is left holding the last trial's arrays (ilqr.py:137-147)
                win_trial = ntrials; new_cost = JlastL; n_trials += ntrials;
              }
            }
            if (__ballot(act && store_slot(win_trial) < 0) == 0ull) break;
            pass = PASS_C;
          }
        }
        if (act) {
          const int ws = store_slot(win_trial);
          cur_slot = (ws < 0) ? SS - 1 : ws;
          stage = ST_DONE;
        }
      } else if (p >= 0) {
        const Tgt own{blk, trj + ((gid * 2 + (size_t)np_) * SS) * (size_t)T, trj_at(cur_par, 0, ref_slot), goal_of(p), mu,
                      LEGACY ? lmask0 : 0ull, LEGACY ? lmask1 : 0ull};
        // LEGACY: trial 1 is the closed-loop step, trials >= 2 run open loop on accumulated controls (ilqr.py:119 with
        // new_u aliasing u after the first rejected step); the controls of trials 2 .. ntrials are formed, nodes in
        // parallel, when the search starts:  u_j = u_{j-1} + 2^-(j-1) k
        static_assert(!(LEGACY && LDSUC), "accumulated controls in LDS belong to the wave-wide search of 32-lane groups");
        const size_t UCSr = LEGACY ? (size_t)(ntrials > 1 ? ntrials - 1 : 1) : 0;
        double* const ucgr = LEGACY ? a.ucum + gid * (UCSr * US) : nullptr;      // [t * M + r][trial - 2]
        if (LEGACY && stage == ST_TRIALS && r0 == 0) {
          for (int t = gl; t < H; t += G) {
            const double* rt = rec(t);
#pragma unroll
            for (int r = 0; r < M; ++r) {
              const double kf = rt[(size_t)(RL::rk + r) * ES];
              double sacc = *trj_at(cur_par, XS + t * M + r, cur_slot);
              double al = alpha_of(2);
              for (int i = 2; i <= ntrials; ++i) {
                sacc = sacc + al * kf;
                al = al / 2.0;
                ucgr[(size_t)(t * M + r) * UCSr + (i - 2)] = sacc;
              }
            }
          }
        }
      // The passes of the line search (rounds of trials, the C++ generation's ladder, the repeat of a winner
      // whose trajectory was not kept) go through ONE roll-out call: the groups of a wave that are in
      // different passes still execute it together, and the roll-out code exists once.
      P3_STAMP(0)                                   /* MARK(3) -> first loop top */
      while (stage != ST_DONE) {
        bool valid = true;
        int kind = 0, sslot = -1;
        double alpha = 1.0;
        int uct = 2;                             // LEGACY: the trial whose accumulated controls this lane reads
        bool is_trial = false;
        int spec = 0;                            // NEW_AL: ladder columns run ahead in the spare lanes of the last round
        if (stage == ST_TRIALS) {
          const int j = r0 + gl + 1;             // trial number, 1-based
          valid = j <= ntrials;
          is_trial = valid;
          if (LEGACY) { kind = (j != 1) ? 1 : 0; alpha = 1.0; uct = (valid && j >= 2) ? j : 2; }
          else alpha = alpha_of(j);
          sslot = valid ? store_slot(j) : -1;
          if constexpr (AHEAD) {
            // The ladder follows when every trial is rejected (new_al_ilqr.h:1045-1062) and reads nothing the trials
            // write: its first columns are rolled out by the lanes the last round of trials leaves idle, and are
            // used only if the search gets there.
            const int nv = ntrials - r0;
            if (nv <= G) {
              spec = (G - nv < 5) ? G - nv : 5;
              const int i = gl - nv;
              if (i >= 0 && i < spec) {
                valid = true; kind = 2;
                alpha = alpha_of(ntrials + 1);
                for (int q = 0; q < i; ++q) alpha /= 3.0;
              }
            }
          }
        } else if (stage == ST_LADDER) {
          // five-wide ladder alpha, alpha/3, ... from alpha = 2^-10 (new_al_ilqr.h:818-898)
          const int i = r0 + gl;
          valid = i < 5;
          kind = 2;
          alpha = alpha_of(ntrials + 1);
          for (int q = 0; q < i && q < 5; ++q) alpha /= 3.0;
        } else {
          // every lane of the group repeats the winner's roll-out, lane 0 keeps it in the spare slot
          kind = replay_kind; alpha = replay_alpha;
          sslot = (gl == 0) ? SS - 1 : -1;
          if (LEGACY) uct = (win_trial >= 2) ? win_trial : 2;
        }
        __threadfence_block();
        P3_STAMP(1)                                 /* loop top -> roll-out call */
        const double J = rollout(own, valid, kind, alpha, LEGACY ? ucgr + (uct - 2) : nullptr, UCSr, sslot);
        P3_STAMP(2)                                 /* the roll-out */

        if (stage == ST_TRIALS) {
          const bool ok = (AHEAD ? is_trial : valid) && (J < old_cost);
          const int w = gfirst<G>(ok, lane);
          int nv = ntrials - r0; if (nv > G) nv = G;      // trials evaluated this round
          if (w >= 0) {
            win_lane = w; win_trial = r0 + w + 1;
            new_cost = gbcast<G>(J, w);
            n_trials += w + 1;
          } else {
            n_trials += nv;
            if (r0 + G >= ntrials) {
              J_last = gbcast<G>(J, nv - 1);
            }
          }
          if (win_trial >= 0) {
            const int ws = store_slot(win_trial);
            if (ws < 0) {
              replay_kind = (LEGACY && win_trial != 1) ? 1 : 0;
              replay_alpha = LEGACY ? 1.0 : alpha_of(win_trial);
              cur_slot = SS - 1;
              stage = ST_REPLAY;
            } else {
              cur_slot = ws;
              stage = ST_DONE;
            }
          } else if (r0 + G < ntrials) {
            r0 += G;                             // next round of trials
          } else if (NEWAL) {
            if (J_last >= old_cost) {
              stage = ST_LADDER; r0 = 0; have = false;
              if constexpr (AHEAD) {
                for (int g = 0; g < spec; ++g) {
                  const double Jg = gbcast<G>(J, nv + g), ag = gbcast<G>(alpha, nv + g);
                  if (!have || Jg < bestJ) { bestJ = Jg; bestA = ag; have = true; }   // first smallest wins
                }
                r0 = spec;
                if (r0 >= 5) {
                  n_trials += 5;
                  if (!(bestJ >= old_cost)) {    // new_al_ilqr.h:1062
                    replay_kind = 0; replay_alpha = bestA;
                    cur_slot = SS - 1;
                    win_trial = 0;
                    stage = ST_REPLAY;
                  } else {
                    stage = ST_DONE;
                  }
                }
              }
            } else {
              new_cost = J_last;                 // not comparable (NaN): nothing is accepted, nothing converged
              keep_old = true;
              stage = ST_DONE;
            }
          } else {
            if (LEGACY) {
              // every step rejected: the reference is left holding the last trial's arrays (ilqr.py:137-147)
              win_trial = ntrials; new_cost = J_last;
              cur_slot = store_slot(ntrials);
            }
            stage = ST_DONE;
          }
        } else if (stage == ST_LADDER) {
          for (int g = 0; g < G && r0 + g < 5; ++g) {
            const double Jg = gbcast<G>(J, g), ag = gbcast<G>(alpha, g);
            if (!have || Jg < bestJ) { bestJ = Jg; bestA = ag; have = true; }   // first smallest wins
          }
          r0 += G;
          if (r0 >= 5) {
            n_trials += 5;
            if (!(bestJ >= old_cost)) {          // new_al_ilqr.h:1062
              replay_kind = 0; replay_alpha = bestA;
              cur_slot = SS - 1;
              win_trial = 0;
              stage = ST_REPLAY;
            } else {
              stage = ST_DONE;
            }
          }
        } else {
          if (NEWAL && win_trial == 0) { new_cost = J; ++n_trials; }   // the ladder winner is re-run with the sequential arithmetic
          stage = ST_DONE;
        }
        P3_STAMP(3)                                 /* bookkeeping after the roll-out */
        if (TRIP) break;                         // further passes in the following trips
      }
      }
      P3_STAMP(4)                                   /* loop end -> MARK(4) */
      PROF_MARK(4)
     bool wc_need = false;                       // FAST: this group's returned pair is to be copied behind the phase (WCOPY)
     const double* wc_src = nullptr;
     int wc_p = 0;
     if (p >= 0 && stage == ST_DONE) {
      if (win_trial >= 0) cur_par = np_;
      __threadfence_block();

      // ================= P4: inner / outer bookkeeping (ilqr.py:158-174) =================
      ++n_inner; ++inner;
      {
        const int trace_cap = ALILQR_COLD2(trace_cap);
        if (gl == 0 && n_inner <= trace_cap) {
          double* const tc = ALILQR_COLD2(trace_cost);
          int32_t* const tt = ALILQR_COLD2(trace_trials);
          if (tc) tc[(size_t)p * trace_cap + (n_inner - 1)] = new_cost;
          if (tt) tt[(size_t)p * trace_cap + (n_inner - 1)] = n_trials - nt0;
        }
      }
      // inner stop rule: |dJ| < tol (ilqr.py:162) ; 0 <= J_old - J_new < tol (new_al_ilqr.h:1118)
      const double inner_tol = ALILQR_COLD2(prm.inner_tol);
      const int max_inner = ALILQR_COLD2(prm.max_inner);
      const bool brk = NEWAL ? ((old_cost - new_cost < inner_tol) && (old_cost - new_cost >= 0))
                             : (fabs(new_cost - old_cost) < inner_tol);
      if (!brk && !keep_old) {
        if (!NEWAL) {
          // x, u = new_x, new_u : the pair optimize() will return (the C++ generation returns its live arrays).
          // Only noted here; the copy is made when the inner loop ends (below).
          xo_pend = true; xo_par = cur_par; xo_slot = cur_slot;
        }
        old_cost = new_cost;
      }
      bool finished = false;
      unsigned flags = 0;
      // diagnostic stop (alilqr_extras::stop_after): the pointer is read from the argument block here, once per inner
      // iteration, so that nothing of it stays live in the step loops
      bool stop_now = false;
      {
        const int32_t* const sa = ALILQR_COLD_(stop_after);
        if (sa) { const int stop_k = sa[p]; stop_now = stop_k > 0 && n_inner >= stop_k; }
      }
      if (brk || inner >= max_inner || stop_now) {
        if (!NEWAL && xo_pend) {
          // The pair to return is the trajectory of the last iteration that did not end the inner loop.  It sits in
          // the parity this iteration's candidates were NOT written to (or is the one just accepted), so it is
          // still whole; the next outer iteration's first line search will overwrite it.
          if constexpr (WCOPY) {
            // noted for the copy by the whole wave behind this phase
            wc_need = true; wc_src = trj_at(xo_par, 0, xo_slot); wc_p = p;
          } else {
            double* const o_xo = ALILQR_COLD(x_out);
            double* const o_uo = ALILQR_COLD(u_out);
            for (int e = gl; e < XS; e += G) o_xo[(size_t)p * XS + e] = *trj_at(xo_par, e, xo_slot);
            for (int e = gl; e < US; e += G) o_uo[(size_t)p * US + e] = *trj_at(xo_par, XS + e, xo_slot);
          }
          xo_pend = false;
        }
      }
      if (stop_now) {
        finished = true; flags = ALILQR_FLAG_STOPPED;
      } else if (brk || inner >= max_inner) {
        // constraint violation of the node trajectory, nodes in parallel, combined in node order
        double vtot = OPS::violation_init();
        for (int t0 = 0; t0 <= H; t0 += G) {
          int t = t0 + gl;
          double vt = 0.0;
          if (t <= H) {
            double x[N], u[M];
            load_node(t, x, u);
            if (LEGACY) {
              double lam[CL];
#pragma unroll
              for (int i = 0; i < CL; ++i) lam[i] = rec(t)[(size_t)(RL::rl + i) * ES];
              vt = OPS::violation(a, tabC(t), x, u, lam, 1, mu);
            } else {
              if constexpr (LIVE) vt = OPS::template violation<true>(a, tabC(t), x, u, rec(t) + (size_t)RL::rl * ES, ES, mu, live_rows);
              else vt = OPS::violation(a, tabC(t), x, u, rec(t) + (size_t)RL::rl * ES, ES, mu);
            }
          }
          for (int g = 0; g < G && t0 + g <= H; ++g) vtot = OPS::combine_violation(vtot, gbcast<G>(vt, g));
        }
        viol = vtot;
        ++outer;
        const double viol_stop = ALILQR_COLD2(prm.viol_stop), viol_lambda = ALILQR_COLD2(prm.viol_lambda);
        if (viol < viol_stop) {
          finished = true; flags = ALILQR_FLAG_VIOL_OK;
        } else {
          // multiplier update vs penalty increase: ilqr.py:169-174 ; new_al_ilqr.h:1198-1211
          if (NEWAL ? !(viol > viol_lambda) : (viol >= viol_stop && viol < viol_lambda)) {
            lmask0 = 0ull; lmask1 = 0ull;
            for (int t0 = 0; t0 <= H; t0 += G) {
              const int t = t0 + gl;
              bool nz = false;
              if (t <= H) {
                double x[N], u[M];
                load_node(t, x, u);
                double* lp = rec(t) + (size_t)RL::rl * ES;
                if constexpr (LIVE) OPS::template update_lambda<true>(a, tabC(t), x, u, lp, ES, mu, live_rows);
                else if constexpr (MODE == ALILQR_MODE_FAST) OPS::template update_lambda<(G <= ALILQR_FAST_TWO_G)>(a, tabC(t), x, u, lp, ES, mu);
                else OPS::update_lambda(a, tabC(t), x, u, lp, ES, mu);
                if (LEGACY) {
#pragma unroll
                  for (int i = 0; i < CL; ++i) nz |= (lp[(size_t)i * ES] != 0.0);
                }
              }
              if (LEGACY) {
                // the group's lanes hold nodes t0 .. t0+G-1: their flags, taken from the wave's ballot
                const unsigned long long bits = (__ballot(nz) >> (lane & ~(G - 1))) & ((G >= 64) ? ~0ull : ((1ull << G) - 1ull));
                if (t0 < 64) lmask0 |= bits << t0;
                else if (t0 < 128) lmask1 |= bits << (t0 - 64);
              }
            }
            __threadfence_block();
          } else {
            const double mu_gain = ALILQR_COLD2(prm.mu_gain);
            if constexpr (NEWAL) { mu_s = mu_s * mu_gain; mu = mu_s; }   // UpdateMu, new_al_ilqr.h:1125-1130
            else mu = mu * mu_gain;
          }
          if (outer >= ALILQR_COLD2(prm.max_outer)) { finished = true; flags = ALILQR_FLAG_OUTER_CAP; }
          else outer_start = true;
        }
      }
      if (finished) {
        double fc = total_cost();
        if (NEWAL) {
          double* const o_xo = ALILQR_COLD(x_out);
          double* const o_uo = ALILQR_COLD(u_out);
          for (int e = gl; e < XS; e += G) o_xo[(size_t)p * XS + e] = *trj_at(cur_par, e, cur_slot);
          for (int e = gl; e < US; e += G) o_uo[(size_t)p * US + e] = *trj_at(cur_par, XS + e, cur_slot);
        }
        double* const o_nx = ALILQR_COLD(node_x);
        double* const o_nu = ALILQR_COLD(node_u);
        double* const o_lam = ALILQR_COLD(lam);
        if (o_nx) for (int e = gl; e < XS; e += G) o_nx[(size_t)p * XS + e] = *trj_at(cur_par, e, cur_slot);
        if (o_nu) for (int e = gl; e < US; e += G) o_nu[(size_t)p * US + e] = *trj_at(cur_par, XS + e, cur_slot);
        if (o_lam)
          for (int e = gl; e < (H + 1) * C; e += G)
            o_lam[(size_t)p * (H + 1) * C + e] = rec(e / C)[(size_t)(RL::rl + e % C) * ES];
        if constexpr (RECT) {
          double* const o_ov = ALILQR_COLD(overlay_out);
          if (o_ov)
            for (int e = gl; e < (H + 1) * (2 * C + 1); e += G)
              o_ov[(size_t)p * (H + 1) * (2 * C + 1) + e] = rec(e / (2 * C + 1))[(size_t)(RL::rl + C + e % (2 * C + 1)) * ES];
        }
        if (gl == 0) {
          double* const o_ms = NEWAL ? ALILQR_COLD(mu_s_out) : nullptr;
          double* const o_viol = ALILQR_COLD(viol);
          double* const o_mu = ALILQR_COLD(mu);
          int32_t* const o_st = ALILQR_COLD(status);
          if (o_ms) o_ms[p] = mu_s;
          ALILQR_COLD(cost)[p] = fc;
          if (o_viol) o_viol[p] = viol;
          if (o_mu) o_mu[p] = mu;
          if (o_st) {
            if (!(fc - fc == 0.0)) flags |= ALILQR_FLAG_NONFINITE;
            o_st[(size_t)p * ALILQR_STATUS_WORDS + 0] = n_inner;
            o_st[(size_t)p * ALILQR_STATUS_WORDS + 1] = outer;
            o_st[(size_t)p * ALILQR_STATUS_WORDS + 2] = (int32_t)flags;
            o_st[(size_t)p * ALILQR_STATUS_WORDS + 3] = n_trials;
          }
        }
        p = -1;
      }
      phase = PH_LIN;
     }
     if constexpr (WCOPY) {
       // The returned pair of the groups whose inner loop ended in this trip, copied by all 64 lanes, one group after the
       // other: whole 512-byte pieces per store instruction instead of G x 8 bytes per group (a narrow group would issue
       // (XS + US) / G loads and stores of its own, and the next phase waits for the stores).
       unsigned long long m = wave_uniform(__ballot(wc_need && gl == 0));
       if (m != 0ull) {
         double* const o_xo = ALILQR_COLD(x_out);
         double* const o_uo = ALILQR_COLD(u_out);
         while (m != 0ull) {
           const int l = __builtin_ctzll(m);
           m &= m - 1ull;
           const double* const src = wave_ptr(wc_src, l);
           const int pp = wave_int(wc_p, l);
#if defined(__HIP_DEVICE_COMPILE__)
           const int e0 = lane, es = 64;
#else
           const int e0 = gl, es = G;            // host builds: a group's lanes are all there is of the wave
#endif
           for (int e = e0; e < XS; e += es) o_xo[(size_t)pp * XS + e] = src[e];
           for (int e = e0; e < US; e += es) o_uo[(size_t)pp * US + e] = src[XS + e];
         }
       }
     }
      PROF_MARK(5)
  }
  PROF_FLUSH(a)
  // Every group leaves the loop after exactly one failing pull, so the counter ends at base + B + NG.
}

}  // namespace alilqr
