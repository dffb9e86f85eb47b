// alilqr_api.hip — C ABI of the batched AL-iLQR solve path (see include/alilqr.h).
//
// Host side: argument checks, workspace, lane-group width selection, launch.
// No CPU compute path exists here: without a GPU every compute entry point
// returns ALILQR_E_NODEVICE / ALILQR_E_HIP.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/alilqr.h"
#include "launch.hpp"

// the kernel families: defined here in single-translation-unit builds, in their own objects otherwise
namespace alilqr {
#ifdef ALILQR_SINGLE_TU
#define X(m, k) ALILQR_DEFINE_FAMILY(m, k)
#else
#define X(m, k) ALILQR_DECLARE_FAMILY(m, k)
#endif
#include "families.inc"
#undef X
}  // namespace alilqr

namespace {

thread_local char g_err[512] = "";

void set_err(const char* fmt, const char* a = "", const char* b = "") {
  std::snprintf(g_err, sizeof(g_err), fmt, a, b);
}

#define HIP_TRY(call)                                                        \
  do {                                                                       \
    hipError_t e_ = (call);                                                  \
    if (e_ != hipSuccess) {                                                  \
      set_err("%s: %s", #call, hipGetErrorString(e_));                       \
      return ALILQR_E_HIP;                                                   \
    }                                                                        \
  } while (0)

// Device scratch.  A solve needs scratch for its resident lane groups only (they pull problems from a counter),
// so a slot is a few hundred MB at most.  Slots form a pool per device: a solve takes the slot its stream used
// last (stream order already serialises the two kernels), else a slot whose last kernel has finished (its event
// says so), else a new slot, else - the pool is full - the least recently used slot, after which the stream
// waits for that slot's event on the GPU.  Every reuse is preceded by hipStreamWaitEvent on the slot's last event,
// so a recycled stream handle or a slot handed to another stream can never overlap its previous kernel.  The pool
// lock is held until the kernel is enqueued and the slot's event recorded: a grow or a release on another thread
// cannot free scratch that a launch is about to use.
struct Slot {
  void* ptr = nullptr;
  size_t bytes = 0;
  hipStream_t stream = nullptr;   // stream of the last solve in this slot
  hipEvent_t done = nullptr;      // recorded behind the last solve in this slot
  bool used = false;              // a solve has been enqueued in this slot
  unsigned counter_next = 0;      // value the slot's problem counter holds once its last solve has finished
  unsigned long long tick = 0;    // for least-recently-used
};
constexpr int kMaxDev = 16, kMaxSlots = 32;
Slot g_slots[kMaxDev][kMaxSlots];
unsigned long long g_tick = 0;
void* g_last_ws[kMaxDev] = {};   // scratch of the most recent solve on the device (profiling builds read it)
std::mutex g_mu;

// Tuning overrides (alilqr_set_tuning): 0 = automatic.
std::atomic<int> g_opt_lanes{0}, g_opt_max_waves{0}, g_opt_no_lds{0}, g_opt_dense_rows{0};

// ---- scratch of the entry point that allocates nothing ----------------------------------------------------------
// A file-scope device array: every GPU the library runs on holds its own copy until the process ends.  Host code
// never uses the array's name as an address; the per-device address comes from hipGetSymbolAddress (static_scratch()).
constexpr size_t kStaticScratchBytes = (size_t)ALILQR_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
struct StaticScratch {
  void* ptr = nullptr;            // device address of `scratch` on this device
  unsigned counter_next = 0;
};
StaticScratch g_static[kMaxDev];

struct Dims {
  int N, M, NA, NB, E;
};

bool model_dims(int model, Dims* d) {
  using namespace alilqr;
  switch (model) {
    case ALILQR_MODEL_LON: *d = {3, 1, Model<ALILQR_MODEL_LON>::NA, Model<ALILQR_MODEL_LON>::NB, LinLayout<Model<ALILQR_MODEL_LON>>::E}; return true;
    case ALILQR_MODEL_LAT:
    case ALILQR_MODEL_LAT_INTERIOR:
    case ALILQR_MODEL_LAT_K: *d = {4, 1, Model<ALILQR_MODEL_LAT>::NA, Model<ALILQR_MODEL_LAT>::NB, LinLayout<Model<ALILQR_MODEL_LAT>>::E}; return true;
    case ALILQR_MODEL_KIN:
    case ALILQR_MODEL_DYN:
    case ALILQR_MODEL_DYN_MID:
    case ALILQR_MODEL_NEW_BICYCLE: *d = {6, 2, Model<ALILQR_MODEL_KIN>::NA, Model<ALILQR_MODEL_KIN>::NB, LinLayout<Model<ALILQR_MODEL_KIN>>::E}; return true;
    default: return false;
  }
}

int32_t check_problem(const alilqr_problem* p, Dims* d) {
  if (!p) { set_err("null problem descriptor"); return ALILQR_E_ARG; }
  if (p->abi_version != ALILQR_ABI_VERSION) { set_err("descriptor abi_version mismatch"); return ALILQR_E_ARG; }
  if (!model_dims(p->model, d)) { set_err("unsupported model"); return ALILQR_E_ARG; }
  if (p->n != d->N || p->m != d->M) { set_err("n/m do not match the model"); return ALILQR_E_ARG; }
  if (p->horizon < 1 || p->batch < 0) { set_err("horizon must be >= 1 and batch >= 0"); return ALILQR_E_ARG; }
  if (p->goal_batch != 1 && p->goal_batch != p->batch) { set_err("goal_batch must be 1 or batch"); return ALILQR_E_ARG; }
  if (p->table_nodes != 1 && p->table_nodes != p->horizon + 1) { set_err("table_nodes must be 1 or horizon+1"); return ALILQR_E_ARG; }
  if (p->cons_kind < 0 || p->cons_kind > (ALILQR_CONS_QUADRATIC | ALILQR_CONS_EQUALITY)) { set_err("unknown cons_kind"); return ALILQR_E_ARG; }
  if ((p->cons_kind & ALILQR_CONS_EQUALITY) && p->mode == ALILQR_MODE_LEGACY) {
    set_err("equality tables are defined for FAST and NEW_AL"); return ALILQR_E_ARG;
  }
  if ((p->cons_kind & ALILQR_CONS_QUADRATIC) && p->mode != ALILQR_MODE_NEW_AL) {
    set_err("quadratic constraint tables are defined for NEW_AL"); return ALILQR_E_ARG;
  }
  if (p->mode == ALILQR_MODE_LEGACY) {
    if (p->model != ALILQR_MODEL_LAT && p->model != ALILQR_MODEL_KIN && p->model != ALILQR_MODEL_DYN &&
        p->model != ALILQR_MODEL_LON && p->model != ALILQR_MODEL_LAT_INTERIOR) {
      set_err("LEGACY is defined for LAT, KIN, DYN, LON, LAT_INTERIOR"); return ALILQR_E_ARG;
    }
    if (p->model == ALILQR_MODEL_LAT_INTERIOR) {
      if (p->c != 2 * p->n) { set_err("LAT_INTERIOR bounds states only: c = 2n"); return ALILQR_E_ARG; }
    } else if (p->c != 2 * (p->n + p->m)) { set_err("LEGACY needs c = 2(n+m)"); return ALILQR_E_ARG; }
  } else if (p->mode == ALILQR_MODE_FAST) {
    if (p->c < 1 || p->c > 64) { set_err("FAST needs 1 <= c <= 64"); return ALILQR_E_ARG; }
    if (p->model != ALILQR_MODEL_DYN_MID) { set_err("FAST is defined for DYN_MID"); return ALILQR_E_ARG; }
  } else if (p->mode == ALILQR_MODE_NEW_AL) {
    if (p->c < 1 || p->c > 64) { set_err("NEW_AL needs 1 <= c <= 64"); return ALILQR_E_ARG; }
    if (p->model != ALILQR_MODEL_LAT_K && p->model != ALILQR_MODEL_NEW_BICYCLE) {
      set_err("NEW_AL is defined for LAT_K, NEW_BICYCLE"); return ALILQR_E_ARG;
    }
  } else {
    set_err("unsupported solver mode"); return ALILQR_E_ARG;
  }
  if (p->prm.max_outer < 1 || p->prm.max_inner < 1 || !(p->prm.alpha_min > 0.0)) {
    set_err("max_outer, max_inner must be >= 1 and alpha_min > 0"); return ALILQR_E_ARG;
  }
  return ALILQR_OK;
}

// Lanes per problem.  Wide groups put every line-search trial of a round in flight and shorten the solve of
// one problem; narrow groups put more problems into a wave and raise throughput once the batch alone fills
// the machine.  Measured on MI355X (batch 4096 and 65536, every generation): the wide width wins while the
// batch fits in 2 waves per SIMD at that width, the next narrower one beyond; a single lane per problem never
// wins (its roll-outs are sequential and the kernel spills).
int pick_group(const alilqr_problem* p) {
  {
    const int g = g_opt_lanes.load(std::memory_order_relaxed);
    if (g == 1 || g == 4 || g == 8 || g == 16 || g == 32) return g;
  }
  const long target_waves = 2048;   // 2 waves on each of the 1024 SIMDs
  // LEGACY needs up to 27 trials per round; the other generations accept early (FAST) or stop at 10 (NEW_AL),
  // and their 6-state kernels carry twice the registers per lane
  const bool wide = (p->mode == ALILQR_MODE_LEGACY) || (p->n == 4);
  const int g0 = wide ? 32 : 16, g1 = wide ? 16 : 8;
  // The LEGACY line search of the small models runs all of its up to 27 trials in one pass on 32 lanes; on 16 it needs a
  // second pass whenever one group of the wave finds no step, and the last solves of a batch - each alone in its wave -
  // take twice as long.  Measured at 40 960 and 81 920 lateral problems per launch: 32 lanes 228 k solves/s, 16 lanes 217 k.
  if (p->mode == ALILQR_MODE_LEGACY && p->n <= 4) return g0;
  // FAST (6-state, one wave per SIMD = 1024 resident waves): the widest group that still gives every problem a resident
  // group.  Its line search accepts the first trial four times out of five, and since the groups of a wave take their
  // passes trip by trip, narrow groups lose little.  Measured (box + linear H = 100, one launch): 4096 problems 16 lanes
  // 56 k solves/s (8: 50 k), 8192 8 lanes 79 k (16: 61 k, 4: 64 k), 16384 4 lanes 95 k (8: 82 k, 16: 71 k).
  if (p->mode == ALILQR_MODE_FAST) {
    const long one_per_simd = 1024;
    return ((long)p->batch * 16 / 64 <= one_per_simd) ? 16 : ((long)p->batch * 8 / 64 <= one_per_simd) ? 8 : 4;
  }
  // The 6-state LEGACY models at large batches: 8 lanes from 8192 problems, 4 from 16384.  Dynamic model H = 100: 16384
  // problems in one launch 34.5 k solves/s at 8 lanes, 33.0 k at 4 (16 lanes 32.0 k), six such batches in one call
  // 66.7 k at 4 lanes, 60.8 k at 8, 50 k at 16; kinematic model H = 50: 8192 problems 40.6 k at 8 lanes (16 lanes
  // 27.4 k, 4 lanes 29.6 k), 16384 problems 49.4 k at 4 lanes (8 lanes 40.9 k).
  // The dynamic model, whose 6 x 6 sweep is the largest share of an iteration, one lane per problem from 65536 problems
  // (the wave-wide line search spreads the trials of a lane's problem over the wave, so a single lane does not roll out
  // its trials one after the other).  H = 100, one launch: 65536 problems 61.5 k solves/s at 1 lane against 59.4 k at 4,
  // 131072 73.0 k / 68.7 k, 262144 81.6 k / 74.4 k; 32768 43.4 k / 48.9 k.  The kinematic model (H = 50) keeps 4 lanes at
  // every size (262144 problems: 58.3 k at 1 lane, 61.7 k at 4).
  if (p->mode == ALILQR_MODE_LEGACY && p->model == ALILQR_MODEL_DYN && p->batch >= 65536) return 1;
  if (p->mode == ALILQR_MODE_LEGACY && p->batch >= 16384) return 4;
  if (p->mode == ALILQR_MODE_LEGACY && p->batch >= 8192) return 8;
  // The 4-state NEW_AL model from 262144 problems: 4 lanes.  One launch of that size runs as fast at 4 lanes as at 16
  // (lateral box set H = 50: 291 k against 295 k solves/s); with several launches in flight the narrow groups win
  // (256 starts x 1024 scenarios, 12 calls in flight: 337 k at 4 lanes, 296 k at 8, 313 k at 16).  Smaller single
  // launches lose at 4 lanes (131072 problems: 273 k against 288 k at 16).
  if (p->mode == ALILQR_MODE_NEW_AL && p->n == 4 && p->batch >= 262144) return 4;
  return ((long)p->batch * g0 / 64 <= target_waves) ? g0 : g1;
}

struct Plan {
  int G, NG;
  size_t trj, blk, ucum, counter, total;   // byte offsets / total
};

// kernel instantiation of a NEW_AL descriptor: constraint kind x rectangle activation
int new_al_variant(const alilqr_problem* p, bool rect) {
  const bool quad = ((p->cons_kind & ALILQR_CONS_QUADRATIC) != 0);
  if (quad) return rect ? alilqr::kModeNewAlQuadRect : alilqr::kModeNewAlQuad;
  return rect ? alilqr::kModeNewAlRect : ALILQR_MODE_NEW_AL;
}

size_t block_doubles_for(const alilqr_problem* p, bool rect) {
  using namespace alilqr;
  const int H = p->horizon, C = p->c;
  if (p->mode == ALILQR_MODE_LEGACY) {
    if (p->model == ALILQR_MODEL_LAT) return block_doubles<Model<ALILQR_MODEL_LAT>, ALILQR_MODE_LEGACY>(H, C);
    if (p->model == ALILQR_MODEL_LON) return block_doubles<Model<ALILQR_MODEL_LON>, ALILQR_MODE_LEGACY>(H, C);
    if (p->model == ALILQR_MODEL_LAT_INTERIOR) return block_doubles<Model<ALILQR_MODEL_LAT_INTERIOR>, ALILQR_MODE_LEGACY>(H, C);
    return block_doubles<Model<ALILQR_MODEL_KIN>, ALILQR_MODE_LEGACY>(H, C);
  }
  if (p->mode == ALILQR_MODE_NEW_AL) {
    // the record layout depends on the rectangle overlay only
    if (p->model == ALILQR_MODEL_LAT_K)
      return rect ? block_doubles<Model<ALILQR_MODEL_LAT_K>, kModeNewAlRect>(H, C) : block_doubles<Model<ALILQR_MODEL_LAT_K>, ALILQR_MODE_NEW_AL>(H, C);
    return rect ? block_doubles<Model<ALILQR_MODEL_NEW_BICYCLE>, kModeNewAlRect>(H, C) : block_doubles<Model<ALILQR_MODEL_NEW_BICYCLE>, ALILQR_MODE_NEW_AL>(H, C);
  }
  return block_doubles<Model<ALILQR_MODEL_DYN_MID>, ALILQR_MODE_FAST>(H, C);
}

// doubles of one linearisation record of a LEGACY solve
int legacy_lin_doubles(const alilqr_problem* p, bool packed) {
  using namespace alilqr;
#define E_OF(m) (packed ? NodeRec<Model<m>, ALILQR_MODE_LEGACY, true>::E : NodeRec<Model<m>, ALILQR_MODE_LEGACY>::E)
  switch (p->model) {
    case ALILQR_MODEL_LAT: return E_OF(ALILQR_MODEL_LAT);
    case ALILQR_MODEL_LON: return E_OF(ALILQR_MODEL_LON);
    case ALILQR_MODEL_LAT_INTERIOR: return E_OF(ALILQR_MODEL_LAT_INTERIOR);
    case ALILQR_MODEL_DYN: return E_OF(ALILQR_MODEL_DYN);
    default: return E_OF(ALILQR_MODEL_KIN);
  }
#undef E_OF
}

// Does the constraint table start with the rows of a box,  [I; -I; 0; 0] x + [0; 0; I; -I] u + C ?  One wave, lane =
// row; *out = 2(n+m) if every one of the first 2(n+m) rows is exactly that, else 0 (see BoxRow in solver.hpp).
__global__ void analyze_rows_kernel(const double* tab, int C, int N, int M, unsigned int* out) {
  const int i = threadIdx.x, nbox = 2 * (N + M);
  bool ok = true;
  if (i < nbox && i < C) {
    for (int j = 0; j < N; ++j) {
      const double want = (i < N && j == i) ? 1.0 : ((i >= N && i < 2 * N && j == i - N) ? -1.0 : 0.0);
      ok &= (tab[(size_t)i * N + j] == want);
    }
    for (int j = 0; j < M; ++j) {
      const int k = i - 2 * N;
      const double want = (k >= 0 && k < M && j == k) ? 1.0 : ((k >= M && j == k - M) ? -1.0 : 0.0);
      ok &= (tab[(size_t)C * N + (size_t)i * M + j] == want);
    }
  }
  const bool all_ok = (C >= nbox) && (__ballot(!ok) == 0ull);
  if (i == 0) *out = all_ok ? (unsigned)nbox : 0u;
}

// Gather / scatter between the per-batch buffers of alilqr_solve_batches and the contiguous staging arrays the solve
// works on: one launch moves one array of up to 64 batches (8-byte words; blockIdx.y = batch).
struct SegPtrs { void* p[64]; };
__global__ __launch_bounds__(256) void seg_copy_kernel(SegPtrs seg, unsigned long long* staging, size_t words_per_seg, int to_segments) {
  unsigned long long* const own = (unsigned long long*)seg.p[blockIdx.y];
  if (!own) return;
  unsigned long long* const st = staging + (size_t)blockIdx.y * words_per_seg;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words_per_seg; i += (size_t)gridDim.x * blockDim.x) {
    if (to_segments) own[i] = st[i]; else st[i] = own[i];
  }
}
// move one array between `n_seg` per-batch buffers and its staging array (bytes_per_seg is a multiple of 8)
hipError_t seg_copy(void* const* ptrs, void* staging, size_t bytes_per_seg, int n_seg, bool to_segments, hipStream_t st) {
  const size_t words = bytes_per_seg / 8;
  for (int s0 = 0; s0 < n_seg; s0 += 64) {
    SegPtrs sp{};
    const int n = (n_seg - s0 < 64) ? n_seg - s0 : 64;
    for (int i = 0; i < n; ++i) sp.p[i] = ptrs[s0 + i];
    unsigned bx = (unsigned)((words + 255) / 256);
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(seg_copy_kernel, dim3(bx, (unsigned)n), dim3(256), 0, st, sp,
                       (unsigned long long*)((char*)staging + (size_t)s0 * bytes_per_seg), words, to_segments ? 1 : 0);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// Layout of a solve's scratch.  `cap` > 0 bounds the total: the number of resident waves is reduced until the
// scratch fits (the groups then pull more problems each); total == 0 says that not even one wave fits.
Plan make_plan(const alilqr_problem* p, const Dims& d, bool rect = false, size_t cap = 0) {
  Plan pl{};
  pl.G = pick_group(p);
  long max_waves = 4096;
  { const long w = g_opt_max_waves.load(std::memory_order_relaxed); if (w > 0) max_waves = w; }
  // a kernel built for one wave per SIMD (8-lane LEGACY groups with their records in LDS) has 1024 resident waves
  if (p->mode == ALILQR_MODE_LEGACY && !g_opt_no_lds.load(std::memory_order_relaxed) &&
      d.N <= 4 && alilqr::legacy_lds_mode(d.N, pl.G, p->horizon, legacy_lin_doubles(p, pl.G < 32)) == 1 && max_waves > 1024)
    max_waves = 1024;
  long gpw = 64 / pl.G;                             // groups per wave
  long need_waves = ((long)p->batch + gpw - 1) / gpw;
  long waves = need_waves < max_waves ? need_waves : max_waves;
  if (waves < 1) waves = 1;
  const size_t H = p->horizon, N = d.N, M = d.M;
  const size_t T = (H + 1) * N + H * M;
  const size_t SS = (p->mode == ALILQR_MODE_LEGACY) ? alilqr::TrjSlots<ALILQR_MODE_LEGACY, 4>::SS
                  : (p->mode == ALILQR_MODE_FAST) ? alilqr::TrjSlots<ALILQR_MODE_FAST, 6>::SS
                  : (d.N <= 4) ? alilqr::TrjSlots<ALILQR_MODE_NEW_AL, 4>::SS : alilqr::TrjSlots<ALILQR_MODE_NEW_AL, 6>::SS;
  // LEGACY: one column of accumulated controls per open-loop line-search trial (2 .. ntrials; the kernel counts
  // its trials the same way: alpha = 1, 1/2, ... while alpha > alpha_min, at most 64)
  size_t ucols = 0;
  if (p->mode == ALILQR_MODE_LEGACY) {
    int ntrials = 0;
    for (double al = 1.0; al > p->prm.alpha_min && ntrials < 64; al = al / 2.0) ++ntrials;
    ucols = ntrials > 1 ? (size_t)(ntrials - 1) : 1;
  }
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  auto lay = [&](long w) {
    pl.NG = (int)(w * gpw);
    size_t off = 0;
    pl.counter = off; off = up(off + 256);
    pl.trj = off; off = up(off + sizeof(double) * 2 * T * (size_t)pl.NG * SS);
    pl.ucum = off; off = up(off + sizeof(double) * H * M * (size_t)pl.NG * ucols);
    pl.blk = off; off = up(off + sizeof(double) * block_doubles_for(p, rect) * (size_t)pl.NG);
    pl.total = off;
  };
  lay(waves);
  if (cap > 0 && pl.total > cap) {
    lay(1);
    const size_t per_wave = pl.total;               // upper bound of what one more wave adds (alignment included)
    if (per_wave > cap) { pl.total = 0; return pl; }
    long fit = (long)(cap / per_wave);
    if (fit > waves) fit = waves;
    lay(fit);
    while (pl.total > cap && fit > 1) lay(--fit);
  }
  return pl;
}

// ---------------------------------------------------------------------------
// node evaluation kernel (alilqr_linearize_batch): one thread per node
// ---------------------------------------------------------------------------
struct LArgs {
  int B, H, C, goal_batch, table_nodes, quad, eq;
  alilqr::ModelConst mc;
  const double *x, *u, *lam, *mu, *goals, *Q, *R, *cons;
  double *f, *fx, *fu, *l, *lx, *lu, *lxx, *luu;
  double *cval, *lam_new, *viol, *pcost, *fxx;   // alilqr_node_aux_batch
  double *pcval, *cjx, *cju;                     // alilqr_constraints_batch
};

template <int MODEL, int MODE>
__global__ __launch_bounds__(256) void linearize_kernel(LArgs a) {
  using namespace alilqr;
  using MD = Model<MODEL>;
  using OPS = NodeOps<MODEL, MODE>;
  using LL = LinLayout<MD>;
  constexpr int N = MD::N, M = MD::M;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)a.B * (a.H + 1);
  if (i >= total) return;
  const int b = (int)(i / (a.H + 1)), t = (int)(i % (a.H + 1));
  const int tt = (a.table_nodes > 1) ? t : 0;
  const int cw = (MODE == ALILQR_MODE_LEGACY) ? (2 * N + 2 * M) : a.C * (N + M + 1) + (a.quad ? a.C * N * N : 0);
  KArgs ka{};
  ka.C = a.C;
  ka.qdiag = 0;
  ka.quad = a.quad;
  ka.eq = a.eq;
  ka.quad_rows = ~0ull;   // no row is skipped here: an all-zero matrix contributes exact zeros
  double x[N], u[M], xn[N], ac[MD::NA], bc[MD::NB], S[N][N], Bm[N][M];
#pragma unroll
  for (int q = 0; q < N; ++q) x[q] = a.x[i * N + q];
#pragma unroll
  for (int q = 0; q < M; ++q) u[q] = a.u[i * M + q];
  MD::step(a.mc, x, u, xn);
  MD::jac(a.mc, x, u, ac, bc);
  expand_jac<MD>(a.mc, ac, bc, S, Bm);
  if constexpr (!is_new_al(MODE) && MD::WRAPS) {
    // the Python node keeps its state angle-wrapped from its first cost call on
    x[2] = wrap_py(x[2]);
    x[3] = wrap_py(x[3]);
  }
  const double* g = a.goals + ((a.goal_batch > 1) ? (size_t)b * (a.H + 1) * N : 0) + (size_t)t * N;
  ctab Q = as_ctab(a.Q) + (size_t)tt * N * N;
  ctab R = as_ctab(a.R) + (size_t)tt * M * M;
  ctab tab = as_ctab(a.cons) + (size_t)tt * cw;
  const double* lam = a.lam + i * a.C;
  const double mu = a.mu[b];
  double rec[LL::E];
  double cost = OPS::cost(ka, Q, R, tab, x, u, g, lam, 1, mu);
  OPS::quadratize(ka, Q, R, tab, x, u, g, lam, 1, mu, rec, 1);
  if (a.f) for (int q = 0; q < N; ++q) a.f[i * N + q] = xn[q];
  if (a.fx)
    for (int r = 0; r < N; ++r)
      for (int q = 0; q < N; ++q) a.fx[(i * N + r) * N + q] = ((r == q) ? 1.0 : 0.0) + S[r][q];
  if (a.fu)
    for (int r = 0; r < N; ++r)
      for (int q = 0; q < M; ++q) a.fu[(i * N + r) * M + q] = Bm[r][q];
  if (a.l) a.l[i] = cost;
  if (a.lx) for (int q = 0; q < N; ++q) a.lx[i * N + q] = rec[LL::oLx + q];
  if (a.lu) for (int q = 0; q < M; ++q) a.lu[i * M + q] = rec[LL::oLu + q];
  if (a.lxx) for (int q = 0; q < N * N; ++q) a.lxx[i * N * N + q] = rec[LL::oLxx + q];
  if (a.luu) for (int q = 0; q < M * M; ++q) a.luu[i * M * M + q] = rec[LL::oLuu + q];
  // ---- the remaining per-node calls (alilqr_node_aux_batch)
  if (a.cval || a.lam_new) {
    double ln[64];
    for (int q = 0; q < a.C; ++q) ln[q] = lam[q];
    if constexpr (MODE == ALILQR_MODE_LEGACY) {
      if (a.cval) for (int q = 0; q < a.C; ++q) a.cval[i * a.C + q] = OPS::cval(tab, x, u, q);
    } else if constexpr (is_new_al(MODE)) {
      if (a.cval) for (int q = 0; q < a.C; ++q) a.cval[i * a.C + q] = OPS::cval(ka, tab, x, u, q, lam, 1);
    } else {
      if (a.cval) for (int q = 0; q < a.C; ++q) a.cval[i * a.C + q] = OPS::cval(ka, tab, x, u, q);
    }
    if (a.lam_new) {
      OPS::update_lambda(ka, tab, x, u, ln, 1, mu);
      for (int q = 0; q < a.C; ++q) a.lam_new[i * a.C + q] = ln[q];
    }
  }
  if (a.viol) a.viol[i] = OPS::violation(ka, tab, x, u, lam, 1, mu);
  // ---- constraint object calls (alilqr_constraints_batch): table generations only
  if constexpr (MODE != ALILQR_MODE_LEGACY) {
    const int Cn = a.C;
    if (a.pcval)
      for (int q = 0; q < Cn; ++q) {
        if constexpr (is_new_al(MODE)) a.pcval[i * Cn + q] = OPS::par_cval(ka, tab, x, u, q, lam, 1);
        else a.pcval[i * Cn + q] = OPS::cval(ka, tab, x, u, q);
      }
    if (a.cjx)
      for (int q = 0; q < Cn; ++q)
        for (int j = 0; j < N; ++j) {
          double v = tab[q * N + j];
          if (a.quad) {
            // row q of c_x: A_q + 2 Qc_q x   (quadratic_constraints.h:47-55)
            ctab Qi = tab + Cn * (N + M + 1) + (size_t)q * N * N;
            double qx = 0.0;
            for (int k = 0; k < N; ++k) qx += (2 * Qi[j * N + k]) * x[k];
            v = v + qx;
          }
          a.cjx[(i * Cn + q) * N + j] = v;
        }
    if (a.cju)
      for (int q = 0; q < Cn; ++q)
        for (int j = 0; j < M; ++j) a.cju[(i * Cn + q) * M + j] = tab[Cn * N + q * M + j];
  }
  if constexpr (is_new_al(MODE)) {
    if (a.pcost) a.pcost[i] = OPS::par_cost(ka, Q, R, tab, x, u, g, lam, 1, mu);
    if (a.fxx) {
      double ff[8];
      MD::fxx(a.mc, x, u, ff);
      for (int k = 0; k < 3; ++k) {
        double e[3] = {0.0, 0.0, 0.0};
        e[k] = 1.0;
        for (int r = 0; r < N; ++r)
          for (int q = 0; q < N; ++q)
            a.fxx[((i * 3 + k) * N + r) * N + q] = MD::fxx_nz(r, q) ? MD::fxx_term(ff, e, r, q) : 0.0;
      }
    }
  }
}

}  // namespace

extern "C" {

int32_t alilqr_abi_version(void) { return ALILQR_ABI_VERSION; }

const char* alilqr_last_error(void) { return g_err; }

int32_t alilqr_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { set_err("hipGetDeviceCount: %s", hipGetErrorString(e)); return ALILQR_E_NODEVICE; }
  return n;
}

int32_t alilqr_default_params(int32_t mode, alilqr_params* out) {
  if (!out) { set_err("null params"); return ALILQR_E_ARG; }
  alilqr_params p{};
  p.init_r = 20.0;
  p.dv_gate = (double)0.2F;   /* new_al_ilqr.h:1039 compares with a float literal */
  p.alpha_min = 1e-8;
  switch (mode) {
    case ALILQR_MODE_LEGACY:
      p.max_outer = 20; p.max_inner = 20; p.inner_tol = 1e-8; p.viol_stop = 1e-3; p.viol_lambda = 1e-1;
      p.mu_gain = 8.0; p.quu_reg = 1e-9; break;
    case ALILQR_MODE_FAST:
      p.max_outer = 20; p.max_inner = 20; p.inner_tol = 1e-8; p.viol_stop = 1e-3; p.viol_lambda = 1e-1;
      p.mu_gain = 8.0; p.quu_reg = 0.0; break;
    case ALILQR_MODE_NEW_AL:
      p.max_outer = 50; p.max_inner = 100; p.inner_tol = 1e-3; p.viol_stop = 1e-4; p.viol_lambda = 5e-4;
      p.mu_gain = 100.0; p.quu_reg = 0.0; break;
    default: set_err("unknown solver mode"); return ALILQR_E_ARG;
  }
  *out = p;
  return ALILQR_OK;
}

int64_t alilqr_workspace_bytes(const alilqr_problem* p) {
  Dims d;
  int32_t rc = check_problem(p, &d);
  if (rc != ALILQR_OK) return rc;
  /* NEW_AL: sized for a solve with rectangle obstacles (the larger record) */
  return (int64_t)make_plan(p, d, p->mode == ALILQR_MODE_NEW_AL).total;
}

int32_t alilqr_lanes_for(const alilqr_problem* p) {
  Dims d;
  int32_t rc = check_problem(p, &d);
  if (rc != ALILQR_OK) return rc;
  return pick_group(p);
}

int32_t alilqr_solve_batch(const alilqr_problem* p,
                           const double* x0, const double* goals,
                           const double* Q, const double* R, const double* cons,
                           double* x_out, double* u_out,
                           double* node_x, double* node_u,
                           double* x_init, double* u_init,
                           double* cost, double* viol,
                           double* lam, double* mu,
                           int32_t* status, void* stream) {
  return alilqr_solve_batch_multistart(p, x0, goals, Q, R, cons, nullptr, x_out, u_out, node_x, node_u, x_init, u_init,
                                       cost, viol, lam, mu, status, stream);
}

int32_t alilqr_solve_batch_multistart(const alilqr_problem* p,
                                      const double* x0, const double* goals,
                                      const double* Q, const double* R, const double* cons,
                                      const double* u_offset,
                                      double* x_out, double* u_out,
                                      double* node_x, double* node_u,
                                      double* x_init, double* u_init,
                                      double* cost, double* viol,
                                      double* lam, double* mu,
                                      int32_t* status, void* stream) {
  alilqr_extras ex{};
  ex.u_offset = u_offset;
  return alilqr_solve_batch_ex(p, x0, goals, Q, R, cons, &ex, x_out, u_out, node_x, node_u, x_init, u_init,
                               cost, viol, lam, mu, status, stream);
}

// Address of the static scratch on the current device (once per device, with that device current).
static int32_t static_scratch(int dev, StaticScratch** out) {
  StaticScratch& ss = g_static[dev];
  if (!ss.ptr) {
    void* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) { set_err("hipGetSymbolAddress(scratch): %s", hipGetErrorString(e)); return ALILQR_E_HIP; }
    ss.ptr = ptr;
  }
  *out = &ss;
  return ALILQR_OK;
}

// Take a scratch slot of the pool for a solve of `bytes` on stream `st` (pool lock held by the caller).
static int32_t take_slot(int dev, hipStream_t st, size_t bytes, Slot** out) {
  Slot* pool = g_slots[dev];
  Slot* slot = nullptr;
  for (int i = 0; i < kMaxSlots && !slot; ++i)             // the slot this stream used last
    if (pool[i].used && pool[i].stream == st) slot = &pool[i];
  for (int i = 0; i < kMaxSlots && !slot; ++i)             // a slot whose last kernel has finished
    if (pool[i].used && hipEventQuery(pool[i].done) == hipSuccess) slot = &pool[i];
  for (int i = 0; i < kMaxSlots && !slot; ++i)             // a slot never used
    if (!pool[i].used) slot = &pool[i];
  if (!slot) {                                             // pool full and busy: queue behind the least recently used
    slot = &pool[0];
    for (int i = 1; i < kMaxSlots; ++i)
      if (pool[i].tick < slot->tick) slot = &pool[i];
  }
  if (!slot->done) HIP_TRY(hipEventCreateWithFlags(&slot->done, hipEventDisableTiming));
  if (slot->used) HIP_TRY(hipStreamWaitEvent(st, slot->done, 0));   // orders this solve behind the slot's last one
  if (slot->bytes < bytes) {
    if (slot->ptr) {
      // the old block may still be read by the slot's last kernel
      if (slot->used) HIP_TRY(hipEventSynchronize(slot->done));
      HIP_TRY(hipFree(slot->ptr));
      slot->ptr = nullptr; slot->bytes = 0;
    }
    hipError_t e = hipMalloc(&slot->ptr, bytes);
    if (e != hipSuccess) { set_err("hipMalloc(workspace): %s", hipGetErrorString(e)); slot->ptr = nullptr; return ALILQR_E_NOMEM; }
    slot->bytes = bytes;
    HIP_TRY(hipMemsetAsync(slot->ptr, 0, 256, st));         // counter block of the new scratch
    slot->counter_next = 0;
  }
  *out = slot;
  return ALILQR_OK;
}

// Segments of alilqr_solve_batches: n_seg batches of seg_batch problems each, every one with buffers of its own.  The
// solve runs once over all of them on contiguous staging arrays behind the scratch (same slot, same event ordering);
// start states are gathered in front of the kernel and results scattered behind it with device-to-device copies.
struct Segments {
  int n_seg, seg_batch;
  const double* const* x0;
  double* const* x_out; double* const* u_out; double* const* node_x; double* const* node_u;
  double* const* cost; double* const* viol; double* const* mu; int32_t* const* status;
};

static int32_t solve_impl(const alilqr_problem* p,
                          const double* x0, const double* goals,
                          const double* Q, const double* R, const double* cons,
                          const alilqr_extras* ex,
                          double* x_out, double* u_out,
                          double* node_x, double* node_u,
                          double* x_init, double* u_init,
                          double* cost, double* viol,
                          double* lam, double* mu,
                          int32_t* status, void* stream, bool use_static, const Segments* seg = nullptr) {
  const double* u_offset = ex ? ex->u_offset : nullptr;
  const int n_left = ex ? ex->n_left : 0, n_right = ex ? ex->n_right : 0;
  const bool rect = (n_left > 0) || (n_right > 0);
  if (ex) {
    if (n_left < 0 || n_left > 16 || n_right < 0 || n_right > 16) { set_err("n_left, n_right must be 0..16"); return ALILQR_E_ARG; }
    if ((n_left > 0 && !ex->left_obs) || (n_right > 0 && !ex->right_obs)) { set_err("null rectangle list"); return ALILQR_E_ARG; }
    if (rect && (!p || p->mode != ALILQR_MODE_NEW_AL)) { set_err("rectangle obstacles are defined for NEW_AL"); return ALILQR_E_ARG; }
    if (rect && (ex->first_row < 0 || ex->first_row >= p->c)) { set_err("first_row must be a row of the table"); return ALILQR_E_ARG; }
  }
  Dims d;
  int32_t rc = check_problem(p, &d);
  if (rc != ALILQR_OK) return rc;
  if (p->batch == 0) return ALILQR_OK;   /* an empty batch is valid and touches nothing */
  if (!seg && (!x0 || !x_out || !u_out || !cost)) { set_err("x0, x_out, u_out, cost must not be null"); return ALILQR_E_ARG; }
  if (!goals || !Q || !R || !cons) { set_err("goals, Q, R, cons must not be null"); return ALILQR_E_ARG; }
  int dev = 0;
  {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { set_err("no HIP device is visible"); return ALILQR_E_NODEVICE; }
  }
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDev) { set_err("device index out of range"); return ALILQR_E_ARG; }
  hipStream_t st = (hipStream_t)stream;
  Plan pl = make_plan(p, d, rect, use_static ? kStaticScratchBytes : 0);
  if (pl.total == 0) {
    set_err("the scratch of one wave of this problem exceeds the static scratch (ALILQR_STATIC_SCRATCH_BYTES)");
    return ALILQR_E_NOMEM;
  }
  // staging arrays of a segmented solve, behind the scratch: x0 | x | u | node_x | node_u | cost | viol | mu | status
  const size_t Bt = (size_t)p->batch, XSb = (size_t)(p->horizon + 1) * d.N * sizeof(double), USb = (size_t)p->horizon * d.M * sizeof(double);
  size_t so[10] = {0};
  if (seg) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t part[9] = {Bt * d.N * sizeof(double), Bt * XSb, Bt * USb, seg->node_x ? Bt * XSb : 0, seg->node_u ? Bt * USb : 0,
                            Bt * sizeof(double), seg->viol ? Bt * sizeof(double) : 0, seg->mu ? Bt * sizeof(double) : 0,
                            seg->status ? Bt * ALILQR_STATUS_WORDS * sizeof(int32_t) : 0};
    so[0] = up(pl.total);
    for (int i = 0; i < 9; ++i) so[i + 1] = up(so[i] + part[i]);
    pl.total = so[9];
  }

  // The pool lock is held until the kernel is enqueued: nothing can free or re-assign the scratch in between.
  std::lock_guard<std::mutex> lk(g_mu);
  char* base = nullptr;
  unsigned* counter_next = nullptr;
  Slot* slot = nullptr;
  if (use_static) {
    StaticScratch* ss = nullptr;
    rc = static_scratch(dev, &ss);
    if (rc != ALILQR_OK) return rc;
    // The counter block is cleared in front of every solve (a memset node when the stream is being captured), and the
    // solve counts from zero: a captured launch carries nothing of the host's state and can be replayed.
    HIP_TRY(hipMemsetAsync(ss->ptr, 0, 256, st));
    ss->counter_next = 0;
    base = (char*)ss->ptr;
    counter_next = &ss->counter_next;
    g_last_ws[dev] = ss->ptr;
  } else {
    rc = take_slot(dev, st, pl.total, &slot);
    if (rc != ALILQR_OK) return rc;
    base = (char*)slot->ptr;
    counter_next = &slot->counter_next;
    g_last_ws[dev] = slot->ptr;
  }
#ifdef ALILQR_PROF
  HIP_TRY(hipMemsetAsync(base + pl.counter + 128, 0, 128, st));   /* phase counters of profiling builds */
#endif
#ifdef ALILQR_DEBUG_BOUNDS
  HIP_TRY(hipMemsetAsync(base + pl.counter + 32, 0, 4, st));      /* violation bits of debug builds */
#endif

  if (seg) {
    // the kernel works on the staging arrays; start states are gathered now, in stream order in front of it
    x0 = (const double*)(base + so[0]); x_out = (double*)(base + so[1]); u_out = (double*)(base + so[2]);
    node_x = seg->node_x ? (double*)(base + so[3]) : nullptr; node_u = seg->node_u ? (double*)(base + so[4]) : nullptr;
    cost = (double*)(base + so[5]); viol = seg->viol ? (double*)(base + so[6]) : nullptr; mu = seg->mu ? (double*)(base + so[7]) : nullptr;
    status = seg->status ? (int32_t*)(base + so[8]) : nullptr;
    HIP_TRY(seg_copy((void* const*)seg->x0, base + so[0], (size_t)seg->seg_batch * d.N * sizeof(double), seg->n_seg, false, st));
  }
  alilqr::KArgs ka{};
  ka.B = p->batch; ka.NG = pl.NG; ka.H = p->horizon; ka.C = p->c;
  ka.goal_batch = p->goal_batch; ka.table_nodes = p->table_nodes;
  ka.mc = {p->L, p->dt, p->v, p->k};
  ka.quad = ((p->cons_kind & ALILQR_CONS_QUADRATIC) != 0) ? 1 : 0;
  ka.eq = (p->cons_kind & ALILQR_CONS_EQUALITY) ? 1 : 0;
  ka.prm = p->prm;
  ka.x0 = x0; ka.goals = goals; ka.Q = Q; ka.R = R; ka.cons = cons; ka.u_off = u_offset;
  if (ex && (ex->trace_cost || ex->trace_trials)) {
    if (ex->trace_cap < 1) { set_err("trace_cap must be positive"); return ALILQR_E_ARG; }
    ka.trace_cost = ex->trace_cost; ka.trace_trials = ex->trace_trials; ka.trace_cap = ex->trace_cap;
  }
  if (ex && (ex->mu_solver_in || ex->mu_solver_out || ex->first_rows || ex->overlay_out)) {
    if (p->mode != ALILQR_MODE_NEW_AL) { set_err("mu_solver / first_rows / overlay_out are defined for NEW_AL"); return ALILQR_E_ARG; }
    if ((ex->first_rows || ex->overlay_out) && !rect) { set_err("first_rows / overlay_out need rectangle obstacles"); return ALILQR_E_ARG; }
    ka.mu_s_in = ex->mu_solver_in; ka.mu_s_out = ex->mu_solver_out; ka.first_rows = ex->first_rows; ka.overlay_out = ex->overlay_out;
  }
  if (ex) { ka.stop_after = ex->stop_after; ka.order = ex->order; }
  if (rect) { ka.left_obs = ex->left_obs; ka.right_obs = ex->right_obs; ka.n_left = n_left; ka.n_right = n_right; ka.first_row = ex->first_row; }
  ka.x_out = x_out; ka.u_out = u_out; ka.node_x = node_x; ka.node_u = node_u;
  ka.x_init = x_init; ka.u_init = u_init; ka.cost = cost; ka.viol = viol; ka.lam = lam; ka.mu = mu;
  ka.status = status;
  ka.trj = (double*)(base + pl.trj); ka.blk = (double*)(base + pl.blk); ka.ucum = (double*)(base + pl.ucum);
  ka.counter = (unsigned int*)(base + pl.counter);
  // Linear tables shared by all nodes: is it a box table?  (one word of the counter block, written before the solve)
  if (p->mode != ALILQR_MODE_LEGACY && p->table_nodes == 1 && !(p->cons_kind & ALILQR_CONS_QUADRATIC) && !rect &&
      2 * (d.N + d.M) <= 64 && !g_opt_dense_rows.load(std::memory_order_relaxed)) {
    unsigned int* word = (unsigned int*)(base + pl.counter + 48);
    hipLaunchKernelGGL(analyze_rows_kernel, dim3(1), dim3(64), 0, st, cons, (int)p->c, (int)d.N, (int)d.M, word);
    if (hipGetLastError() != hipSuccess) { set_err("row analysis launch failed"); return ALILQR_E_HIP; }
    ka.box_rows = word;
  }
  // NEW_AL, one linear table for all nodes: the kernel prologue looks for rows without A (B) entries
  ka.sparse_rows = (p->mode == ALILQR_MODE_NEW_AL && p->table_nodes == 1 && !(p->cons_kind & ALILQR_CONS_QUADRATIC) &&
                    p->c <= 64 && !g_opt_dense_rows.load(std::memory_order_relaxed)) ? 1 : 0;
  // NEW_AL, one table for all nodes: roll-outs skip all-zero rows whose multipliers (and rectangle overlay) are zero
  ka.live_rows = (p->mode == ALILQR_MODE_NEW_AL && p->table_nodes == 1 && p->c <= 64 &&
                  !g_opt_dense_rows.load(std::memory_order_relaxed)) ? 1 : 0;
  // The problem counter is never reset: a solve starts from the value the scratch's previous solve left, which
  // the host knows exactly - every problem is pulled once and every group makes one failing pull at the end.
  ka.counter_base = *counter_next;
  ka.prof = (unsigned long long*)(base + pl.counter + 128);   // read by profiling builds
  const int waves = pl.NG * pl.G / 64;

  // kernel family = model x kernel mode (families.inc)
  const int kmode = (p->mode == ALILQR_MODE_NEW_AL) ? new_al_variant(p, rect) : p->mode;
  const int no_lds = g_opt_no_lds.load(std::memory_order_relaxed) ? 1 : 0;
  int lr = -1;
#define X(m, k) if (p->model == (m) && kmode == (k)) lr = alilqr::ALILQR_FAMILY_NAME(m, k)(ka, pl.G, waves, st, no_lds);
#include "families.inc"
#undef X
  if (lr == alilqr::LAUNCH_OK) {
    *counter_next += (unsigned)p->batch + (unsigned)pl.NG;
    if (seg) {
      // results back to the segments' own buffers, behind the kernel in stream order
      const size_t sb = (size_t)seg->seg_batch;
      hipError_t e = hipSuccess;
      auto scatter = [&](void* const* dst, void* src, size_t per_problem) {
        if (!dst || e != hipSuccess) return;
        e = seg_copy(dst, src, sb * per_problem, seg->n_seg, true, st);
      };
      scatter((void* const*)seg->x_out, x_out, XSb);
      scatter((void* const*)seg->u_out, u_out, USb);
      scatter((void* const*)seg->node_x, node_x, XSb);
      scatter((void* const*)seg->node_u, node_u, USb);
      scatter((void* const*)seg->cost, cost, sizeof(double));
      scatter((void* const*)seg->viol, viol, sizeof(double));
      scatter((void* const*)seg->mu, mu, sizeof(double));
      scatter((void* const*)seg->status, status, ALILQR_STATUS_WORDS * sizeof(int32_t));
      if (e != hipSuccess) {
        (void)hipStreamSynchronize(st);
        if (slot) { slot->used = true; slot->stream = st; slot->tick = ++g_tick; }
        set_err("%s: %s", "scatter launch", hipGetErrorString(e));
        return ALILQR_E_HIP;
      }
    }
    if (slot) {
      // the slot is in use from here on, whatever happens to the event: a later solve must wait for this kernel
      slot->used = true; slot->stream = st; slot->tick = ++g_tick;
      const hipError_t e = hipEventRecord(slot->done, st);
      if (e != hipSuccess) {
        // without the event nothing orders later users of the slot behind this kernel: let it finish now
        (void)hipStreamSynchronize(st);
        set_err("%s: %s", "hipEventRecord(slot->done)", hipGetErrorString(e));
        return ALILQR_E_HIP;
      }
    }
    return ALILQR_OK;
  }
  if (lr == alilqr::LAUNCH_BAD_WIDTH) { set_err("bad lane-group width"); return ALILQR_E_ARG; }
  if (lr == alilqr::LAUNCH_FAILED) { set_err("kernel launch failed"); return ALILQR_E_HIP; }
  set_err("unsupported model / mode combination (or a development build without this kernel family)");
  return ALILQR_E_ARG;
}

int32_t alilqr_solve_batch_ex(const alilqr_problem* p,
                              const double* x0, const double* goals,
                              const double* Q, const double* R, const double* cons,
                              const alilqr_extras* ex,
                              double* x_out, double* u_out,
                              double* node_x, double* node_u,
                              double* x_init, double* u_init,
                              double* cost, double* viol,
                              double* lam, double* mu,
                              int32_t* status, void* stream) {
  return solve_impl(p, x0, goals, Q, R, cons, ex, x_out, u_out, node_x, node_u, x_init, u_init,
                    cost, viol, lam, mu, status, stream, false);
}

int32_t alilqr_solve_batch_noalloc(const alilqr_problem* p,
                                   const double* x0, const double* goals,
                                   const double* Q, const double* R, const double* cons,
                                   double* x_out, double* u_out,
                                   double* node_x, double* node_u,
                                   double* x_init, double* u_init,
                                   double* cost, double* viol,
                                   double* lam, double* mu,
                                   int32_t* status, void* stream) {
  return solve_impl(p, x0, goals, Q, R, cons, nullptr, x_out, u_out, node_x, node_u, x_init, u_init,
                    cost, viol, lam, mu, status, stream, true);
}

int32_t alilqr_solve_batches(const alilqr_problem* p, int32_t n_batches,
                             const double* const* x0, const double* goals,
                             const double* Q, const double* R, const double* cons,
                             double* const* x_out, double* const* u_out,
                             double* const* node_x, double* const* node_u,
                             double* const* cost, double* const* viol, double* const* mu,
                             int32_t* const* status, const alilqr_extras* ex, void* stream) {
  if (!p) { set_err("null problem descriptor"); return ALILQR_E_ARG; }
  if (n_batches < 0 || n_batches > 4096) { set_err("n_batches must be 0..4096"); return ALILQR_E_ARG; }
  if (p->goal_batch != 1) { set_err("alilqr_solve_batches takes batches that share one goal list (goal_batch = 1)"); return ALILQR_E_ARG; }
  if (n_batches == 0 || p->batch == 0) {
    Dims d;
    return check_problem(p, &d);
  }
  if (!x0 || !x_out || !u_out || !cost) { set_err("x0, x_out, u_out, cost pointer lists must not be null"); return ALILQR_E_ARG; }
  for (int i = 0; i < n_batches; ++i)
    if (!x0[i] || !x_out[i] || !u_out[i] || !cost[i]) { set_err("x0, x_out, u_out, cost of every batch must not be null"); return ALILQR_E_ARG; }
  if ((long long)p->batch * n_batches > 0x7fffffffLL / 4) { set_err("too many problems in one call"); return ALILQR_E_ARG; }
  alilqr_problem all = *p;
  all.batch = p->batch * n_batches;
  Segments seg{n_batches, p->batch, x0, x_out, u_out, node_x, node_u, cost, viol, mu, status};
  if (ex && ex->overlay_out) { set_err("overlay_out is not supported by alilqr_solve_batches"); return ALILQR_E_ARG; }
  return solve_impl(&all, nullptr, goals, Q, R, cons, ex, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                    nullptr, nullptr, nullptr, nullptr, nullptr, stream, false, &seg);
}

int64_t alilqr_static_scratch_bytes(void) { return (int64_t)kStaticScratchBytes; }

int32_t alilqr_set_tuning(int32_t lanes, int32_t max_waves, int32_t no_lds) {
  if (!(lanes == 0 || lanes == 1 || lanes == 4 || lanes == 8 || lanes == 16 || lanes == 32)) {
    set_err("lanes must be 0 (automatic), 1, 4, 8, 16 or 32"); return ALILQR_E_ARG;
  }
  if (max_waves < 0) { set_err("max_waves must be >= 0"); return ALILQR_E_ARG; }
  g_opt_lanes.store(lanes); g_opt_max_waves.store(max_waves);
  g_opt_no_lds.store((no_lds & 1) ? 1 : 0); g_opt_dense_rows.store((no_lds & 2) ? 1 : 0);
  return ALILQR_OK;
}

static int32_t node_eval(const alilqr_problem* p,
                         const double* x, const double* u,
                         const double* lam, const double* mu,
                         const double* goals, const double* Q, const double* R, const double* cons,
                         double* f, double* fx, double* fu, double* l,
                         double* lx, double* lu, double* lxx, double* luu,
                         double* cval, double* lam_new, double* viol, double* pcost, double* fxx,
                         void* stream, double* pcval = nullptr, double* cjx = nullptr, double* cju = nullptr) {
  Dims d;
  int32_t rc = check_problem(p, &d);
  if (rc != ALILQR_OK) return rc;
  if (p->batch == 0) return ALILQR_OK;
  if ((pcost || fxx) && p->mode != ALILQR_MODE_NEW_AL) { set_err("par_cost and fxx exist for NEW_AL only"); return ALILQR_E_ARG; }
  if (!x || !u || !lam || !mu || !goals || !Q || !R || !cons) { set_err("null input"); return ALILQR_E_ARG; }
  {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { set_err("no HIP device is visible"); return ALILQR_E_NODEVICE; }
  }
  LArgs a{};
  a.B = p->batch; a.H = p->horizon; a.C = p->c; a.goal_batch = p->goal_batch; a.table_nodes = p->table_nodes;
  a.quad = ((p->cons_kind & ALILQR_CONS_QUADRATIC) != 0) ? 1 : 0;
  a.eq = (p->cons_kind & ALILQR_CONS_EQUALITY) ? 1 : 0;
  a.mc = {p->L, p->dt, p->v, p->k};
  a.x = x; a.u = u; a.lam = lam; a.mu = mu; a.goals = goals; a.Q = Q; a.R = R; a.cons = cons;
  a.cval = cval; a.lam_new = lam_new; a.viol = viol; a.pcost = pcost; a.fxx = fxx;
  a.pcval = pcval; a.cjx = cjx; a.cju = cju;
  if ((pcval || cjx || cju) && p->mode == ALILQR_MODE_LEGACY) { set_err("constraint tables exist for FAST and NEW_AL"); return ALILQR_E_ARG; }
  a.f = f; a.fx = fx; a.fu = fu; a.l = l; a.lx = lx; a.lu = lu; a.lxx = lxx; a.luu = luu;
  const size_t total = (size_t)p->batch * (p->horizon + 1);
  dim3 block(256), grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (p->mode == ALILQR_MODE_LEGACY) {
    switch (p->model) {
      case ALILQR_MODEL_LAT: hipLaunchKernelGGL((linearize_kernel<ALILQR_MODEL_LAT, ALILQR_MODE_LEGACY>), grid, block, 0, st, a); break;
      case ALILQR_MODEL_KIN: hipLaunchKernelGGL((linearize_kernel<ALILQR_MODEL_KIN, ALILQR_MODE_LEGACY>), grid, block, 0, st, a); break;
      case ALILQR_MODEL_DYN: hipLaunchKernelGGL((linearize_kernel<ALILQR_MODEL_DYN, ALILQR_MODE_LEGACY>), grid, block, 0, st, a); break;
      case ALILQR_MODEL_LON: hipLaunchKernelGGL((linearize_kernel<ALILQR_MODEL_LON, ALILQR_MODE_LEGACY>), grid, block, 0, st, a); break;
      case ALILQR_MODEL_LAT_INTERIOR: hipLaunchKernelGGL((linearize_kernel<ALILQR_MODEL_LAT_INTERIOR, ALILQR_MODE_LEGACY>), grid, block, 0, st, a); break;
      default: set_err("unsupported model / mode combination"); return ALILQR_E_ARG;
    }
  } else if (p->mode == ALILQR_MODE_NEW_AL) {
    if (a.quad) {
      if (p->model == ALILQR_MODEL_LAT_K) hipLaunchKernelGGL((linearize_kernel<ALILQR_MODEL_LAT_K, alilqr::kModeNewAlQuad>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((linearize_kernel<ALILQR_MODEL_NEW_BICYCLE, alilqr::kModeNewAlQuad>), grid, block, 0, st, a);
    } else {
      if (p->model == ALILQR_MODEL_LAT_K) hipLaunchKernelGGL((linearize_kernel<ALILQR_MODEL_LAT_K, ALILQR_MODE_NEW_AL>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((linearize_kernel<ALILQR_MODEL_NEW_BICYCLE, ALILQR_MODE_NEW_AL>), grid, block, 0, st, a);
    }
  } else {
    hipLaunchKernelGGL((linearize_kernel<ALILQR_MODEL_DYN_MID, ALILQR_MODE_FAST>), grid, block, 0, st, a);
  }
  HIP_TRY(hipGetLastError());
  return ALILQR_OK;
}

int32_t alilqr_linearize_batch(const alilqr_problem* p,
                               const double* x, const double* u,
                               const double* lam, const double* mu,
                               const double* goals, const double* Q, const double* R, const double* cons,
                               double* f, double* fx, double* fu, double* l,
                               double* lx, double* lu, double* lxx, double* luu,
                               void* stream) {
  return node_eval(p, x, u, lam, mu, goals, Q, R, cons, f, fx, fu, l, lx, lu, lxx, luu,
                   nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

int32_t alilqr_node_aux_batch(const alilqr_problem* p,
                              const double* x, const double* u,
                              const double* lam, const double* mu,
                              const double* goals, const double* Q, const double* R, const double* cons,
                              double* cval, double* lam_new, double* viol, double* par_cost, double* fxx,
                              void* stream) {
  return node_eval(p, x, u, lam, mu, goals, Q, R, cons, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                   cval, lam_new, viol, par_cost, fxx, stream);
}

int32_t alilqr_constraints_batch(const alilqr_problem* p,
                                 const double* x, const double* u,
                                 const double* lam, const double* mu,
                                 const double* goals, const double* Q, const double* R, const double* cons,
                                 double* cval, double* par_cval, double* jac_x, double* jac_u,
                                 void* stream) {
  return node_eval(p, x, u, lam, mu, goals, Q, R, cons, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                   cval, nullptr, nullptr, nullptr, nullptr, stream, par_cval, jac_x, jac_u);
}

#ifdef ALILQR_DEBUG_BOUNDS
/* debug builds only: the violation bits of the last solve on this device (0 = every offset was in range) */
int32_t alilqr_debug_read(unsigned* out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || !g_last_ws[dev]) return ALILQR_E_HIP;
  if (hipDeviceSynchronize() != hipSuccess) return ALILQR_E_HIP;
  if (hipMemcpy(out, (char*)g_last_ws[dev] + 32, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return ALILQR_E_HIP;
  return ALILQR_OK;
}
#endif

#ifdef ALILQR_PROF
/* profiling builds only: copy the 9 phase counters and the 2 roll-out counters of the last solve on this device to the host */
int32_t alilqr_prof_read(unsigned long long* out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || !g_last_ws[dev]) return ALILQR_E_HIP;
  if (hipDeviceSynchronize() != hipSuccess) return ALILQR_E_HIP;
  if (hipMemcpy(out, (char*)g_last_ws[dev] + 128, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return ALILQR_E_HIP;
  return ALILQR_OK;
}
#endif

int32_t alilqr_release(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < kMaxDev; ++i) {
    bool any = false;
    for (int k = 0; k < kMaxSlots; ++k) any |= (g_slots[i][k].ptr != nullptr) || (g_slots[i][k].done != nullptr);
    int cur = 0;
    if (any && hipGetDevice(&cur) == hipSuccess) {
      (void)hipSetDevice(i);
      (void)hipDeviceSynchronize();
      for (int k = 0; k < kMaxSlots; ++k) {
        if (g_slots[i][k].ptr) (void)hipFree(g_slots[i][k].ptr);
        if (g_slots[i][k].done) (void)hipEventDestroy(g_slots[i][k].done);
      }
      (void)hipSetDevice(cur);
    }
    for (int k = 0; k < kMaxSlots; ++k) g_slots[i][k] = Slot{};
    g_last_ws[i] = nullptr;
  }
  return ALILQR_OK;
}

}  // extern "C"
