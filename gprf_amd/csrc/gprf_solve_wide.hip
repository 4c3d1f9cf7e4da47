// gprf_solve_wide.hip — k_solve_panel's instantiations for launches whose largest unit has 21 .. 28 tiles per edge.
#include "gprf_solve_panel.h"

namespace gprf {

void launch_solve_wide(const UnitTab &utp, const Pools &p, int dy, dim3 grid, hipStream_t s) {
    if (utp.max_T <= 26)           // (one panel buffer, two workgroups per CU: the paper-scale catalogue's pairs of 390 points)
        hipLaunchKernelGGL((k_solve_panel<26, 2, true, 1>), grid, dim3(256), 0, s, utp, p, dy);
    else                           // (448 points: the seismic configuration's pairs at every block size below 210; 11 % faster
                                   // there than the 32-tile instantiation)
        hipLaunchKernelGGL((k_solve_panel<28, 1, true>), grid, dim3(256), 0, s, utp, p, dy);
}

}  // namespace gprf
