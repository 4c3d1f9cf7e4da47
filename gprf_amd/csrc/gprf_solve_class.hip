// gprf_solve_class.hip — the forward substitution of ONE of the Cholesky's two size classes (k_solve_panel's class
// instantiations; a translation unit of their own for the build's sake), for launch_potrf's by-class pipelines.
#include "gprf_solve_panel.h"

namespace gprf {

// The substitution by size class (round 6).  In-kernel records of where the stage's time goes (-DGPRF_WGTRACE=1): the launch is
// ~3 rounds of workgroups of which the FIRST — every unit's Y workgroup and first identity workgroup, all full T-step chains
// starting together — lasts 50 of the 76 us at three workgroups per CU, their waves waiting for panels and updating in step with
// each other; the 16-tile instantiation's 128 accumulator registers are what holds the CU at three.  352 of the north-star's
// 442 units have at most 13 tiles: k_solve_panel<13, 4> fits 128 registers and 30 KB of LDS — FOUR workgroups per CU — and
// the Cholesky already runs as two kernels over exactly these two classes on two queues: each class's substitution goes
// behind its own Cholesky kernel, on that queue, and the queues join behind the substitutions instead of in front.
constexpr int SOLVE_CLASS_MAXT = 20;      // the large class's largest instantiation = the generating Cholesky kernels' limit (launches
                                          // with larger units: every stage one launch)
bool solve_by_class(const UnitTab &ut) {
    return diag("solve_class", 1) != 0 && potrf_small_maxT() == 13 && ut.max_T > potrf_small_maxT() && ut.max_T <= SOLVE_CLASS_MAXT;
}
void launch_solve_class(const UnitTab &ut, const Pools &p, int dy, int which, hipStream_t s) {
    UnitTab utp = ut;
    utp.pm_group = 0;
    if (which == 1) {
        if (ut.grid_big <= 0) return;
        // (the large list + at most grid_big - |large list| surplus units: never more than grid_big; the instantiation by the
        // launch's largest unit, as launch_solve picks it: 16 tiles at three workgroups per CU, 20 — a pair growing past 256
        // points during an optimisation — at two)
        utp.max_T = ut.max_T <= 16 ? 16 : 20;
        const int nparts = (utp.max_T + 3) / 4 + 1;
        dim3 grid(xcd_grid(ut.grid_big, nparts));
        if (ut.max_T <= 16) hipLaunchKernelGGL((k_solve_panel<16, 3, true, 1, 1>), grid, dim3(256), 0, s, utp, p, dy);
        else hipLaunchKernelGGL((k_solve_panel<20, 2, true, 2, 1>), grid, dim3(256), 0, s, utp, p, dy);
    } else {
        if (ut.grid_small <= 0) return;
        utp.max_T = 13;
        const int nparts = (utp.max_T + 3) / 4 + 1;
        hipLaunchKernelGGL((k_solve_panel<13, 4, true, 1, 2>), dim3(xcd_grid(ut.grid_small, nparts)), dim3(256), 0, s, utp, p, dy);
    }
}

}  // namespace gprf
