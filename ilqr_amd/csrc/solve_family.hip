// solve_family.hip — one kernel family of the solve path (see families.inc), selected at compile time:
//   hipcc -c -DALILQR_FAMILY_MODEL=<model id> -DALILQR_FAMILY_MODE=<kernel mode> solve_family.hip
#include "launch.hpp"

#if !defined(ALILQR_FAMILY_MODEL) || !defined(ALILQR_FAMILY_MODE)
#error "compile with -DALILQR_FAMILY_MODEL=<model id> -DALILQR_FAMILY_MODE=<kernel mode>"
#endif

namespace alilqr {
ALILQR_DEFINE_FAMILY(ALILQR_FAMILY_MODEL, ALILQR_FAMILY_MODE)
}
