// launch.hpp — launching one kernel family (model x kernel mode) of the solve path.
//
// Every family lives in its own translation unit so that the library builds in parallel; the C ABI file only
// sees plain functions  int launch_m<model>_k<mode>(args, lanes, waves, stream, no_lds)  returning
//   0 launched, 1 unsupported lane-group width, 2 the launch failed (hipGetLastError has the reason).
#pragma once
#include "solver.hpp"

#ifndef ALILQR_NO_LDS_UCUM
#define ALILQR_NO_LDS_UCUM 0   // development builds: 1 keeps the accumulated controls in HBM
#endif

namespace alilqr {

enum { LAUNCH_OK = 0, LAUNCH_BAD_WIDTH = 1, LAUNCH_FAILED = 2 };

// Where the linearisation records of a LEGACY solve live (lds_mode): 0 in the groups' blocks in HBM; 2 in LDS with
// two waves per SIMD (a wave's share of the 160 KB of a CU is 20 KB); 1 in LDS with one wave per SIMD (40 KB).
// The sequential Riccati sweep then reads them with LDS latency and they cause no memory traffic.
__host__ inline int legacy_lds_mode(int n_states, int G, int H, int rec_doubles) {
  const size_t lds = (size_t)(64 / G) * (H + 1) * rec_doubles * sizeof(double);
  const size_t share2 = 160 * 1024 / 8, share1 = 160 * 1024 / 4;
  if (n_states <= 4) {
    if ((G == 32 || G == 16) && lds <= share2) return 2;
    if (G == 8 && lds <= share1) return 1;
    return 0;
  }
  return ((G == 32 || G == 16) && lds <= share1) ? 1 : 0;
}

template <int MODEL, int MODE>
inline int launch_family(const KArgs& ka, int G, int waves, hipStream_t st, int no_lds) {
  dim3 grid(waves), block(64);
  if constexpr (MODE == ALILQR_MODE_LEGACY) {
    using MDx = Model<MODEL>;
    int E = (G == 32) ? NodeRec<MDx, MODE>::E : NodeRec<MDx, MODE, true>::E;     // narrower groups pack their records
    int mode = no_lds ? 0 : legacy_lds_mode(MDx::N, G, ka.H, E);
    bool packed32 = false;
    if (G == 32 && mode == 0 && !no_lds && legacy_lds_mode(MDx::N, G, ka.H, NodeRec<MDx, MODE, true>::E) != 0) {
      // long horizons: the unpacked records of a wave's two groups do not fit its LDS share, the packed ones do
      E = NodeRec<MDx, MODE, true>::E;
      mode = legacy_lds_mode(MDx::N, G, ka.H, E);
      packed32 = true;
    }
    const size_t lds = (size_t)(64 / G) * (ka.H + 1) * E * sizeof(double);
    if (mode != 0) {
      if (G == 32) {
        // the accumulated controls of the open-loop trials 3 .. ntrials share the records' LDS bytes when a wave's two
        // groups fit their share of the CU's LDS with them
        int ntrials = 0;
        for (double al = 1.0; al > ka.prm.alpha_min && ntrials < 64; al = al / 2.0) ++ntrials;
        const int cols = ntrials > 2 ? ntrials - 2 : 0;
        const size_t both = 2 * lds_group_doubles(ka.H, E, cols * MDx::M) * sizeof(double);
        const size_t share = (MDx::N <= 4) ? 160 * 1024 / 8 : 160 * 1024 / 4;
        KArgs k2 = ka;
        k2.lds_bytes = (int)lds;
        if (packed32) {
          hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 32, 3>), grid, block, lds, st, k2);
        } else if (both <= share && !ALILQR_NO_LDS_UCUM) {
          k2.uc_cols = cols;
          k2.lds_bytes = (int)both;
          hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 32, 2>), grid, block, both, st, k2);
        } else {
          hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 32, 1>), grid, block, lds, st, k2);
        }
      } else {
        KArgs k2 = ka;
        k2.lds_bytes = (int)lds;
        if (G == 16) hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 16, 1>), grid, block, lds, st, k2);
        else {
          if constexpr (MDx::N <= 4) hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 8, 1, 1>), grid, block, lds, st, k2);
          else return LAUNCH_BAD_WIDTH;
        }
      }
      return hipGetLastError() == hipSuccess ? LAUNCH_OK : LAUNCH_FAILED;
    }
  }
  // quadratic tables: the wave stages the tables its roll-outs read (one constraint table, Q, R) in LDS; at most
  // 64 rows (checked by the C ABI), 24 KB
  size_t tl = 0;
  KArgs kt = ka;
  if constexpr (is_quad(MODE) && ALILQR_TAB_LDS) {
    using MDx = Model<MODEL>;
    const size_t cw = (size_t)ka.C * (MDx::N + MDx::M + 1) + (size_t)ka.C * MDx::N * MDx::N;
    tl = (((cw + MDx::N * MDx::N + MDx::M * MDx::M + 1) & ~(size_t)1)) * sizeof(double);
  }
  // FAST / NEW_AL: two LDS buffers for the multiplier block (C doubles; 3 C + 1 with the rectangle overlay) of the node a
  // roll-out evaluates next, in chunks of 2 G doubles per group = 1 KB per wave.  Blocks of more than 8 chunks (16 KB of
  // buffers; a wave's share of the LDS at two waves per SIMD is 20 KB) are read from the records instead.
  if constexpr (MODE != ALILQR_MODE_LEGACY) {
    const int block = is_rect(MODE) ? 3 * ka.C + 1 : ka.C;
    const int chd = 2 * G;
    const int nch = (block + chd - 1) / chd;
    // Measured (solves/s against the form that reads the records, 4096 / 32768-65536 problems): 6-state tables +6 to +14 %
    // at 8 and 16 lanes, +5 % at 32; 4-state linear tables +3 to +9 % at 16 lanes, but -45 % at 8 lanes and -13 to -40 %
    // at 32 lanes with rectangle or quadratic rows (their register allocation suffers).  The form is therefore built
    // and used for the 6-state kernels (FAST included) at 8 lanes and up and for the 4-state linear-table kernels at 16 lanes.
    constexpr bool six = Model<MODEL>::N > 4;
    const bool wanted = six ? (G >= 8) : (G == 16 && !is_quad(MODE));
    // (+ 512 bytes behind the buffers: one 8-byte word per lane, where the 6-state NEW_AL kernels keep their live-row look)
    const bool fits = wanted && nch <= 8 && tl + (size_t)2 * nch * 1024 + 512 <= 40 * 1024;
    kt.rows_lds = fits ? block : 0;
    if (fits) tl += (size_t)2 * nch * 1024 + 512;
  }
  kt.lds_bytes = (int)tl;
  if constexpr (MODE != ALILQR_MODE_LEGACY) {
    if (kt.rows_lds > 0) {
#ifdef ALILQR_ONLY_G
      if (G != ALILQR_ONLY_G) return LAUNCH_BAD_WIDTH;
      hipLaunchKernelGGL((solve_kernel<MODEL, MODE, ALILQR_ONLY_G, 0, 0, 1>), grid, block, tl, st, kt);
#else
      if constexpr (Model<MODEL>::N > 4) {
        switch (G) {
          case 8: hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 8, 0, 0, 1>), grid, block, tl, st, kt); break;
          case 16: hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 16, 0, 0, 1>), grid, block, tl, st, kt); break;
          case 32: hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 32, 0, 0, 1>), grid, block, tl, st, kt); break;
          default: return LAUNCH_BAD_WIDTH;
        }
      } else if constexpr (!is_quad(MODE)) {
        hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 16, 0, 0, 1>), grid, block, tl, st, kt);
      }
#endif
      return hipGetLastError() == hipSuccess ? LAUNCH_OK : LAUNCH_FAILED;
    }
  }
#ifdef ALILQR_ONLY_G   /* development builds: one lane-group width only (seconds to compile) */
  if (G != ALILQR_ONLY_G) return LAUNCH_BAD_WIDTH;
  hipLaunchKernelGGL((solve_kernel<MODEL, MODE, ALILQR_ONLY_G>), grid, block, tl, st, kt);
  return hipGetLastError() == hipSuccess ? LAUNCH_OK : LAUNCH_FAILED;
#else
  switch (G) {
    case 1: hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 1>), grid, block, tl, st, kt); break;
    case 4: hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 4>), grid, block, tl, st, kt); break;
    case 8: hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 8>), grid, block, tl, st, kt); break;
    case 16: hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 16>), grid, block, tl, st, kt); break;
    case 32: hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 32>), grid, block, tl, st, kt); break;
    default: return LAUNCH_BAD_WIDTH;
  }
  return hipGetLastError() == hipSuccess ? LAUNCH_OK : LAUNCH_FAILED;
#endif
}

#define ALILQR_FAMILY_NAME_(m, k) launch_m##m##_k##k
#define ALILQR_FAMILY_NAME(m, k) ALILQR_FAMILY_NAME_(m, k)
#define ALILQR_DECLARE_FAMILY(m, k) int ALILQR_FAMILY_NAME(m, k)(const KArgs& ka, int G, int waves, hipStream_t st, int no_lds);
#define ALILQR_DEFINE_FAMILY(m, k)                                                                   \
  int ALILQR_FAMILY_NAME(m, k)(const KArgs& ka, int G, int waves, hipStream_t st, int no_lds) {      \
    return launch_family<m, k>(ka, G, waves, st, no_lds);                                            \
  }

}  // namespace alilqr
