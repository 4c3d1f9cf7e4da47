// gprf_solve.hip — forward substitution U^T [W | Z] = [I | Y[unit rows]] (k_solve_panel; replaces dtrtri / dpotri / dpotrs,
// gpy_linalg.py:139-171,219-253) and At = Z^T W = (K^-1 Y)^T (k_at, k_at_wide) of the units of up to 512 / 1024 points.
#include "gprf_solve_panel.h"

namespace gprf {

// k_at_wide: the throughput form of k_at (many units per CU): one workgroup per 16 column tiles of the unit; wave w owns the column tiles
// I = I0 + w, 7-w, 8+w, 15-w (W is lower triangular: column tile I has T - I row tiles, and wave w always lands on SIMD w —
// dealt w, w+4, w+8, w+12, wave 0 of every workgroup on a CU carried 40 tile steps of a 16-tile unit against wave 3's 28;
// the snake gives 34 each) and all four 16-row blocks of At for each (16 accumulators).  The k-loop runs
// DOWN from the last row tile so the four waves need the same Z chunk at the same time (shared through L1):
// per k-tile 16 Z operands are loaded once and reused for up to four column tiles.
template <int CLS = 0>
__global__ __launch_bounds__(256, 2) void k_at_wide(UnitTab ut, Pools pl, int first_round, int skip_T) {
    int slot_, part_;
    WgTrace trace(ut, pl, 2);
    const ClassList cl = class_list<CLS>(ut);
    if (cl.n <= 0) return;
    if (!xcd_map(blockIdx.x, cl.n, (ut.max_T + 15) >> 4, &slot_, &part_)) return;
    if (CLS != 0 && part_ >= ((ut.max_T + 15) >> 4)) return;
    // A launch of at most two workgroups per CU is resident all at once: workgroup first_round + j (first_round = the CUs)
    // becomes the second resident of the CU that took workgroup j.  The launch order is largest unit first, so the CU of the
    // largest unit also got the largest of the rest, and the launch lasted as long as those two sharing four SIMDs; with the
    // second round in ASCENDING size the largest unit is paired with the smallest.
    if (first_round > 0 && slot_ >= first_round) slot_ = cl.n - 1 - (slot_ - first_round);
    const UnitRef ur = class_unit<CLS>(ut, cl, slot_);
    int m = ur.m;
    int mp = pad16(m), T = mp >> 4;
    int I0 = 16 * part_;
    if (I0 >= T || T > skip_T) return;      // (skip_T: the units launch_big_at takes)
    int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar wave index
    int lr = lane & 15, lg = lane >> 4;
    size_t roff = ur.row_off;
    const double *__restrict__ W = pl.W + ur.mat_off;
    const double *__restrict__ Z = pl.Z + roff * YPAD;
    double *__restrict__ At = pl.At + roff * YPAD;
    d4 acc[4][4];   // [owned column tile][16-row block of At]
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[o][c] = d4{0.0, 0.0, 0.0, 0.0};
    int Imin = I0 + wave;
    for (int kt = T - 1; kt >= I0; --kt) {
        if (kt < Imin) continue;   // nothing of this wave's tiles reaches up here (keeps the waves in step)
        const double *zp = Z + (size_t)(16 * kt + lg) * YPAD + lr;
        double a[4][4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) a[s][c] = zp[(size_t)(4 * s) * YPAD + 16 * c];
        const double *wrow = W + (size_t)(16 * kt + lg) * mp + lr;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            int I = I0 + 4 * o + ((o & 1) ? 3 - wave : wave);
            if (I <= kt && I < T) {
                double b[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) b[s] = wrow[(size_t)(4 * s) * mp + 16 * I];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[o][c] = mfma(a[s][c], b[s], acc[o][c]);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        int I = I0 + 4 * o + ((o & 1) ? 3 - wave : wave);
        if (I < T) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) At[(size_t)(16 * c + lg + 4 * q) * mp + 16 * I + lr] = acc[o][c][q];
        }
    }
    trace.done(T * 8 + part_);
}

// k_at: At = Z^T W with one workgroup per AT_TILES column tiles of the unit; wave w owns the column tiles
// I = I0 + w, w+4 and all four 16-row blocks of At for each (8 accumulators).  The k-loop runs DOWN from the
// last row tile so the four waves need the same Z chunk at the same time (shared through L1): per k-tile 16 Z
// operands are loaded once and reused for both column tiles; two register sets alternate so that the next
// step's operands are already in flight.
constexpr int AT_TILES = 8;    // column tiles of At per workgroup (two per wave)

__global__ __launch_bounds__(256, 2) void k_at(UnitTab ut, Pools pl, int skip_T) {
    int slot_, part_;
    WgTrace trace(ut, pl, 2);
    if (!xcd_map(blockIdx.x, ut.n_ids, (ut.max_T + AT_TILES - 1) / AT_TILES, &slot_, &part_)) return;
    const UnitRef ur = unit_ref(ut.srec, slot_);
    int m = ur.m;
    int mp = pad16(m), T = mp >> 4;
    int I0 = AT_TILES * part_;
    if (I0 >= T || T > skip_T) return;      // (skip_T: the units launch_big_at takes)
    int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar wave index
    int lr = lane & 15, lg = lane >> 4;
    size_t roff = ur.row_off;
    const double *__restrict__ W = pl.W + ur.mat_off;
    const double *__restrict__ Z = pl.Z + roff * YPAD;
    double *__restrict__ At = pl.At + roff * YPAD;
    constexpr int NO = AT_TILES / 4;
    d4 acc[NO][4];   // [owned column tile][16-row block of At]
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[o][c] = d4{0.0, 0.0, 0.0, 0.0};
    int Imin = I0 + wave;
    if (Imin >= T) return;
    // operands of one k step: the Z chunk (A, shared by the wave's tiles) and the W tiles (B); the next step's
    // are in flight while this step's MFMAs run — a unit's chain of T steps is otherwise a chain of T memory
    // round trips
    auto fetch = [&](int kt, double (&a)[4][4], double (&b)[NO][4]) {
        const double *zp = Z + (size_t)(16 * kt + lg) * YPAD + lr;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) a[s][c] = zp[(size_t)(4 * s) * YPAD + 16 * c];
        const double *wrow = W + (size_t)(16 * kt + lg) * mp + lr;
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            int I = I0 + wave + 4 * o;
            bool on = I <= kt && I < T;
#pragma unroll
            for (int s = 0; s < 4; ++s) b[o][s] = on ? wrow[(size_t)(4 * s) * mp + 16 * I] : 0.0;
        }
    };
    auto mma = [&](int kt, const double (&a)[4][4], const double (&b)[NO][4]) {
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            int I = I0 + wave + 4 * o;
            if (I <= kt && I < T) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[o][c] = mfma(a[s][c], b[o][s], acc[o][c]);
            }
        }
    };
    double a0[4][4], b0[NO][4], a1[4][4], b1[NO][4];
    fetch(T - 1, a0, b0);
    for (int kt = T - 1; kt >= Imin; kt -= 2) {      // steps below Imin hold none of this wave's tiles (W is lower)
        if (kt - 1 >= Imin) fetch(kt - 1, a1, b1);
        mma(kt, a0, b0);
        if (kt - 1 >= Imin) {
            if (kt - 2 >= Imin) fetch(kt - 2, a0, b0);
            mma(kt - 1, a1, b1);
        }
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        int I = I0 + wave + 4 * o;
        if (I < T) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) At[(size_t)(16 * c + lg + 4 * q) * mp + 16 * I + lr] = acc[o][c][q];
        }
    }
    trace.done(T * 8 + part_);
}


// ... and At of one class behind its substitution (k_at_wide over the class's list; the large class is fewer workgroups than
// CUs, the small one a single round of two per CU)
void launch_at_class(const UnitTab &ut, const Pools &p, int which, hipStream_t s) {
    const int parts = (ut.max_T + 15) / 16;
    if (which == 1) {
        if (ut.grid_big > 0) hipLaunchKernelGGL(k_at_wide<1>, dim3(xcd_grid(ut.grid_big, parts)), dim3(256), 0, s, ut, p, 0, MAX_T);
    } else {
        if (ut.grid_small > 0) hipLaunchKernelGGL(k_at_wide<2>, dim3(xcd_grid(ut.grid_small, parts)), dim3(256), 0, s, ut, p, 0, MAX_T);
    }
}

void launch_solve(const UnitTab &ut_all, const Pools &p, const KParams &kp, hipStream_t s) {
    if (ut_all.n_ids == 0) return;
    // PM: the grid walked part by part (part_major_map) — launches at most two rounds of CUs wide; diag part_major=0 / 1 forces
    const int pm_d = diag("part_major", -1);
    // (n_launch, not n_ids: one half of a split launch takes the form the whole launch would take — UnitTab::n_launch)
    const bool pm = pm_d >= 0 ? pm_d == 1 : ut_all.n_launch <= 2 * device_cus();
    static_assert(BIG_LA_T <= SOLVE_PANEL_MAXT, "every unit the blocked path leaves alone fits a k_solve_panel instantiation");
    {
        // (the launch's units of more than BIG_LA_T tiles go through launch_big_solve; the instantiation follows the others)
        UnitTab ut = ut_all;
        if (ut.max_T > BIG_LA_T) ut.max_T = BIG_LA_T;
        const int nparts = (ut.max_T + 3) / 4 + 1;
        dim3 grid(xcd_grid(ut.n_ids, nparts));
        UnitTab utp = ut;
        utp.pm_group = 0;
        if (ut.max_T <= 12) {
            if (pm) hipLaunchKernelGGL((k_solve_panel<12, 3, true>), grid, dim3(256), 0, s, utp, p, kp.dy);
            else hipLaunchKernelGGL((k_solve_panel<12, 3, false>), grid, dim3(256), 0, s, ut, p, kp.dy);
        } else if (ut.max_T <= 13 && pm) {
            // (round 6: 13 tiles fit 128 registers and 30 KB of LDS — FOUR workgroups per CU; the by-class pipelines' small-class
            // instantiation, here for a launch whose every unit is that small)
            hipLaunchKernelGGL((k_solve_panel<13, 4, true, 1>), grid, dim3(256), 0, s, utp, p, kp.dy);
        } else if (ut.max_T <= 16) {
            // units of 13 .. 16 tiles: ONE panel buffer at THREE workgroups per CU (35 KB of LDS, 157 VGPRs) against the
            // double-buffered 18-tile instantiation's two — round 4, measured: C3 87 -> 79 us, C4 633 -> 596
            if (pm) hipLaunchKernelGGL((k_solve_panel<16, 3, true, 1>), grid, dim3(256), 0, s, utp, p, kp.dy);
            else hipLaunchKernelGGL((k_solve_panel<16, 3, false, 1>), grid, dim3(256), 0, s, ut, p, kp.dy);
        } else if (ut.max_T <= 18) {
            if (pm) hipLaunchKernelGGL((k_solve_panel<18, 2, true>), grid, dim3(256), 0, s, utp, p, kp.dy);
            else hipLaunchKernelGGL((k_solve_panel<18, 2, false>), grid, dim3(256), 0, s, ut, p, kp.dy);
        } else if (ut.max_T <= 20) {
            // (two panels of 19 tile columns + V_rr fill half of the CU's LDS exactly: two workgroups per CU, as many registers
            // each as the accumulators of 20 tiles need — the seismic configuration's pairs of 312 points)
            if (pm) hipLaunchKernelGGL((k_solve_panel<20, 2, true>), grid, dim3(256), 0, s, utp, p, kp.dy);
            else hipLaunchKernelGGL((k_solve_panel<20, 2, false>), grid, dim3(256), 0, s, ut, p, kp.dy);
        } else {
            // the large instantiations exist once each (the walk chosen at run time: see the kernel)
            if (!pm) utp.pm_group = -1;
            if (ut.max_T <= 28) launch_solve_wide(utp, p, kp.dy, grid, s);
            else launch_solve_wide32(utp, p, kp.dy, grid, s);
        }
    }
}

void launch_at(const UnitTab &ut, const Pools &p, hipStream_t s) {
    if (ut.n_ids == 0 || ut.max_T == 0) return;
    // units of more than 1024 points by the split-K GEMM when the launch's largest has more than BIG_AT_GEMM_T tiles per edge (the
    // kernels below then leave them alone): below that a unit's longest part is short enough (9 blocks + 20 pairs of n = 10000:
    // 0.30 ms by k_at, 0.36 by the GEMM, half of whose waves idle on a 64-row tile; ONE block of 10000: 1.16 against 0.28)
    const bool big_gemm = ut.max_T > BIG_AT_GEMM_T;
    const int skip_T = big_gemm ? SMALL_MAX_T : MAX_T;
    if (big_gemm) launch_big_at(ut, p, s);
    // single-unit latency matters while the launch is about one workgroup-round deep (sharded runs); beyond
    // that the wide form's operand reuse wins (C3 on one GPU: 55 vs 58 us, C4: 324 vs 429 us)
    const int cus = device_cus();
    if (ut.n_launch <= cus) {
        hipLaunchKernelGGL(k_at, dim3(xcd_grid(ut.n_ids, (ut.max_T + AT_TILES - 1) / AT_TILES)), dim3(256), 0, s, ut, p, skip_T);
        return;
    }
    // ONE round of at most two workgroups per CU: the second resident of a CU in ASCENDING size (largest with smallest)
    const int first_round = (ut.max_T <= 16 && ut.n_ids > cus && ut.n_ids <= 2 * cus) ? cus : 0;
    hipLaunchKernelGGL(k_at_wide<0>, dim3(xcd_grid(ut.n_ids, (ut.max_T + 15) / 16)), dim3(256), 0, s, ut, p, first_round, skip_T);
}

}  // namespace gprf
