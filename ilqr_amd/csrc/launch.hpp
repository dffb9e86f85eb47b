// launch.hpp — launching one kernel family (model x kernel mode) of the solve path.
//
// Every family lives in its own translation unit so that the library builds in parallel; the C ABI file only
// sees plain functions  int launch_m<model>_k<mode>(args, lanes, waves, stream, no_lds)  returning
//   0 launched, 1 unsupported lane-group width, 2 the launch failed (hipGetLastError has the reason).
#pragma once
#include "solver.hpp"

namespace alilqr {

enum { LAUNCH_OK = 0, LAUNCH_BAD_WIDTH = 1, LAUNCH_FAILED = 2 };

template <int MODEL, int MODE>
inline int launch_family(const KArgs& ka, int G, int waves, hipStream_t st, int no_lds) {
  dim3 grid(waves), block(64);
  if constexpr (MODE == ALILQR_MODE_LEGACY) {
    // 32-lane LEGACY kernels keep their linearisation records in LDS when two groups' worth of them fit eight
    // (4-state, 2 waves per SIMD) or four (6-state, 1 wave per SIMD) times into the 160 KB of a CU
    using MDx = Model<MODEL>;
    const size_t lds = (size_t)2 * (ka.H + 1) * NodeRec<MDx, MODE>::E * sizeof(double);
    const size_t room = (MDx::N <= 4) ? 160 * 1024 / 8 : 160 * 1024 / 4;
    if (G == 32 && lds <= room && !no_lds) {
      hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 32, true>), grid, block, lds, st, ka);
      return hipGetLastError() == hipSuccess ? LAUNCH_OK : LAUNCH_FAILED;
    }
  }
  switch (G) {
    case 1: hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 1>), grid, block, 0, st, ka); break;
    case 4: hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 4>), grid, block, 0, st, ka); break;
    case 8: hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 8>), grid, block, 0, st, ka); break;
    case 16: hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 16>), grid, block, 0, st, ka); break;
    case 32: hipLaunchKernelGGL((solve_kernel<MODEL, MODE, 32>), grid, block, 0, st, ka); break;
    default: return LAUNCH_BAD_WIDTH;
  }
  return hipGetLastError() == hipSuccess ? LAUNCH_OK : LAUNCH_FAILED;
}

#define ALILQR_FAMILY_NAME_(m, k) launch_m##m##_k##k
#define ALILQR_FAMILY_NAME(m, k) ALILQR_FAMILY_NAME_(m, k)
#define ALILQR_DECLARE_FAMILY(m, k) int ALILQR_FAMILY_NAME(m, k)(const KArgs& ka, int G, int waves, hipStream_t st, int no_lds);
#define ALILQR_DEFINE_FAMILY(m, k)                                                                   \
  int ALILQR_FAMILY_NAME(m, k)(const KArgs& ka, int G, int waves, hipStream_t st, int no_lds) {      \
    return launch_family<m, k>(ka, G, waves, st, no_lds);                                            \
  }

}  // namespace alilqr
