// gprf_solve_wide32.hip — k_solve_panel's largest instantiation: units of up to 32 tiles per edge (512 points), one workgroup per CU.
#include "gprf_solve_panel.h"

namespace gprf {

void launch_solve_wide32(const UnitTab &utp, const Pools &p, int dy, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((k_solve_panel<SOLVE_PANEL_MAXT, 1, true>), grid, dim3(256), 0, s, utp, p, dy);
}

}  // namespace gprf
