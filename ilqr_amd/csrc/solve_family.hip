// solve_family.hip — one kernel family of the solve path (see families.inc), selected at compile time:
//   hipcc -c -DALILQR_FAMILY_MODEL=<model id> -DALILQR_FAMILY_MODE=<kernel mode> solve_family.hip
// How the family's kernels read the argument block (ALILQR_COLD_LEVEL, solver.hpp): level 3 reads the fields that are
// needed once per problem or once per inner iteration at their place of use, level 0 holds every field in scalar
// registers.  Which one is faster is a matter of the register allocation it leaves the step loops with; measured per
// family on MI355X (one launch of 16 384 problems, level 0 -> level 3, solves/s):
//   LEGACY lateral 162.8 k -> 151.9 k | LEGACY dynamic H = 100 31.4 k -> 30.9 k | LEGACY kinematic 34.0 k -> 33.9 k
//   FAST box 50.4 k -> 51.9 k, box + linear H = 100 70.2 k -> 70.5 k
//   NEW_AL 4-state: linear 120.8 k -> 143.6 k | circle 95.8 k -> 85.9 k | rectangles 80.2 k -> 91.1 k
//   NEW_AL 6-state: box 127.8 k -> 128.6 k, kinematic + linear 63.9 k -> 65.7 k | circle 52.3 k -> 52.9 k | rectangles 42.0 k -> 35.5 k
#if !defined(ALILQR_COLD_LEVEL) && defined(ALILQR_FAMILY_MODEL) && defined(ALILQR_FAMILY_MODE)
#if (ALILQR_FAMILY_MODEL == 4 && (ALILQR_FAMILY_MODE == 2 || ALILQR_FAMILY_MODE == 4)) || \
    (ALILQR_FAMILY_MODEL == 3 && ALILQR_FAMILY_MODE == 1) ||                             \
    (ALILQR_FAMILY_MODEL == 5 && (ALILQR_FAMILY_MODE == 2 || ALILQR_FAMILY_MODE == 3))
#define ALILQR_COLD_LEVEL 3
#else
#define ALILQR_COLD_LEVEL 0
#endif
#endif
#include "launch.hpp"

#if !defined(ALILQR_FAMILY_MODEL) || !defined(ALILQR_FAMILY_MODE)
#error "compile with -DALILQR_FAMILY_MODEL=<model id> -DALILQR_FAMILY_MODE=<kernel mode>"
#endif

namespace alilqr {
ALILQR_DEFINE_FAMILY(ALILQR_FAMILY_MODEL, ALILQR_FAMILY_MODE)
}
