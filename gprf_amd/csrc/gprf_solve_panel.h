// gprf_solve_panel.h — the forward-substitution kernel template k_solve_panel, shared by gprf_solve.hip (instantiations of up
// to 20 tiles per edge) and gprf_solve_wide*.hip (26 / 28 / 32): its fully unrolled instantiations are two thirds of the
// library's compile time, so they are spread over three translation units.
#pragma once
#include "gprf_dev.h"

namespace gprf {

// ------------------------------------------------------------------------------------------------
// Forward substitution  U^T [W | Z] = [I | Y[unit rows]]  (replaces dtrtri/dpotri/dpotrs of gpy_linalg.py:219-253,
// 139-148), right-looking.  Units of up to 32 tiles per edge: k_solve_panel below; larger ones: the blocked path
// (launch_big_solve).  (Rounds 1-5 also had a generic one-workgroup-per-column-block kernel, k_solve, for units of 33 .. 64
// tiles: 2.0 ms for the 25-block run's pairs of ~800 points against 1.07 by the blocked path; removed.)
// ------------------------------------------------------------------------------------------------

constexpr int SOLVE_PANEL_MAXT = 32;  // largest k_solve_panel instantiation (accumulators: 32 tiles x 8 registers, two panels
                                      // of 31 tile columns = 127 KB of LDS, one workgroup per CU: units of up to 512 points)

// k_solve_panel: the same forward substitution with the U row panel of each step staged ONCE per workgroup in LDS
// (cooperative, coalesced loads of panel r+1 overlap step r's MFMAs; one barrier per step), so the four waves
// — four RHS column blocks — share every U tile and each update MFMA costs one conflict-free ds_read.
// Each wave's tiles for all rows stay in MFMA accumulators; the freshly solved tile is already in B-operand layout
// (accumulator register q = rows 4q+lg), so the right-looking updates chain through registers.
// NBUF = 1 (round 4, units of 21 .. 26 tiles): ONE panel buffer — 52 KB + V_rr instead of 104: TWO workgroups per CU where the
// double-buffered form has one; the next panel is requested behind a second barrier (nobody reads the current one any more) and
// its round trip is exposed to this workgroup — the other workgroup of the CU computes meanwhile
// CLS (round 6): 0 = the launch order (UnitTab::srec); 1 / 2 = one of the Cholesky's two device-built lists — the units of more
// than / at most potrf_small_maxT() tiles — each behind ITS Cholesky kernel on that kernel's queue (launch_potrf), the small
// class with an instantiation of its own size at four workgroups per CU.  The large-unit Cholesky's surplus workgroups take
// units from the END of the small list (potrf_reg_body): those units are solved by class 1 too, behind the kernel that
// factored them.
template <int MAXT, int WPS, bool PM, int NBUF = 2, int CLS = 0>
__global__ __launch_bounds__(256, WPS) void k_solve_panel(UnitTab ut, Pools pl, int dy) {
    // panel columns are stored RELATIVE to the first column right of the diagonal tile (16(r+1)): a step loads and
    // keeps only what its updates read.  (LDP/16) odd: lane groups 32 banks apart
    constexpr int LDP = 16 * ((MAXT - 1) | 1);
    constexpr int NCH = (16 * (MAXT - 1) + 127) / 128;
    __shared__ __attribute__((aligned(16))) double panel[NBUF][16 * LDP];
    __shared__ double Vl[2][256];
    static_assert(NBUF == 2 || 16 * MAXT * sizeof(int32_t) <= sizeof(double) * 512, "the row -> point table aliases Vl");
    int slot_, part_;
    int nI = (ut.max_T + 3) >> 2;                // parts 0..nI-1: identity column blocks 4p+wave; part nI: Y blocks
    WgTrace trace(ut, pl, 1);
#ifdef GPRF_PROFILE
    unsigned long long t_kernel0 = __builtin_amdgcn_s_memtime();
#endif
    // (PM instantiations: a negative group size selects the unit-major walk at run time — the large instantiations exist once,
    // their compile time is minutes; the small hot ones keep the walk a template parameter: as a run-time field the same
    // kernel was 5 % slower)
    // (CLS != 0: this class's units in THIS partition; the grid follows the last synchronised partition with slack)
    const ClassList cl = class_list<CLS>(ut);
    const int n_ids = cl.n;
    if (n_ids <= 0) return;
    if (!(PM ? (ut.pm_group >= 0 ? part_major_map(blockIdx.x, n_ids, nI + 1, ut.pm_group, &slot_, &part_)
                                 : xcd_map(blockIdx.x, n_ids, nI + 1, &slot_, &part_))
             : xcd_map(blockIdx.x, n_ids, nI + 1, &slot_, &part_))) return;
    if (CLS != 0 && part_ > nI) return;          // (a grid sized for more units than the partition has)
    // the Y workgroup (every step, a gather in front) is the longest of a unit: it is dispatched first
    part_ = part_ == 0 ? nI : part_ - 1;
    const UnitRef ur = class_unit<CLS>(ut, cl, slot_);
    int u = ur.u;
    int m = ur.m;
    int mp = pad16(m), T = mp >> 4;
    if (T > BIG_LA_T) return;                    // (uniform) the blocked path's units (launch_big_solve)
    int tid = threadIdx.x;
    int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar wave index
    int lr = lane & 15, lg = lane >> 4;
    bool is_y = part_ == nI;
    int cb = is_y ? wave : (part_ * 4 + wave);
    bool live = is_y || (cb < T);                // dead waves still stage panels and hit the barriers
    size_t roff = ur.row_off;
    if (T == 0) {
        if (live && is_y && lane == 0) pl.zzpart[(size_t)u * 4 + cb] = 0.0;
        return;
    }
    // first row any wave of this workgroup needs
    int rmin = is_y ? 0 : part_ * 4;
    if (rmin >= T) return;                        // whole workgroup beyond this unit's columns (uniform)
    const double *__restrict__ U = pl.U + ur.mat_off;
    const double *__restrict__ V = pl.V + roff * 16;
    double *__restrict__ W = pl.W + ur.mat_off;
    double *__restrict__ Z = pl.Z + roff * YPAD;
    const double *__restrict__ Yg = pl.Y;
    int r0 = is_y ? 0 : cb;
    // the Y workgroup gathers its right-hand side through the unit row -> point table: the table goes through LDS
    // first (one coalesced load), so that the gather itself is a single round of independent loads
    // (in the second panel buffer, which the step loop writes only after its first barrier: the two panels + V already
    // fill half of the CU's LDS exactly, and one more kilobyte would halve the occupancy)
    int32_t *s_upt = reinterpret_cast<int32_t *>(NBUF == 2 ? &panel[NBUF - 1][0] : &Vl[0][0]);
    if (is_y) {
        // (every tile of the instantiation; a lane's four rows lg + 4q of a tile next to each other: one 16-byte read)
        for (int e = tid; e < 16 * MAXT; e += 256)
            s_upt[(e & ~15) + 4 * (e & 3) + ((e >> 2) & 3)] = e < m ? ut.upt[roff + e] : -1;
        __syncthreads();
    }

    // (static_for, not "#pragma unroll": the optimizer gives up on the 28-tile instantiation's loops and the
    // accumulators would land in scratch)
    d4 acc[MAXT];
    if (is_y) {
        // Y[unit rows], zero padded: branch-free per element (an invalid row / column loads Y[0] and is masked), so that the
        // table reads and the gather loads of all tiles are issued back to back instead of one round trip at a time
        int col = 16 * cb + lr;
        bool colok = col < dy;
        // (no "r < T" branch either: rows beyond the unit read -1 from the table)
        typedef int i4 __attribute__((ext_vector_type(4)));
        i4 pts[MAXT];
        static_for<0, MAXT>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            pts[r] = *reinterpret_cast<const i4 *>(s_upt + 16 * r + 4 * lg);
        });
        static_for<0, MAXT>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int pt = pts[r][q];
                bool ok = pt >= 0 && colok;
                double v = Yg[ok ? (size_t)pt * dy + col : (size_t)0];     // (32-bit offsets measured SLOWER: 105 vs 89 us)
                acc[r][q] = ok ? v : 0.0;
            }
        });
        if constexpr (NBUF == 1) __syncthreads();      // the table (in Vl) has been read: the step loop may write V_rr there
    } else {
        static_for<0, MAXT>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[r][q] = (live && r == cb && (lg + 4 * q) == lr) ? 1.0 : 0.0;
        });
    }
    // wave w carries rows 4w..4w+3 of the panel in chunks of 128 columns (two per lane: 16 bytes).  Scalar row base + lane
    // offset; a chunk beyond the unit's last column is skipped by a wave-uniform branch, lanes beyond it load nothing
    // (unpredicated loads cost 15 % more time).
    // GPRF_SOLVE_GLDS (default): the chunks go from memory straight into the panel buffer of their step
    // (global_load_lds_dwordx4: 64 lanes x 16 bytes land as one contiguous kilobyte = 128 columns of one row, exactly
    // the panel's layout) — no staging registers, no ds_write pass in front of the barrier; panel r + 1 is requested
    // right after barrier r into the buffer nobody reads any more, and waited for (vmcnt) in front of barrier r + 1.
#ifndef GPRF_SOLVE_GLDS
#define GPRF_SOLVE_GLDS 1
#endif
    constexpr bool GLDS = GPRF_SOLVE_GLDS != 0;
    d2 pre[4][GLDS ? 1 : NCH];
    double prev;
    auto fetch = [&](int r, auto nchc) {
        constexpr int nch = decltype(nchc)::value;
        int ncols = mp - 16 * (r + 1);
        const double *Ur = U + (size_t)(16 * r + 4 * wave) * mp + 16 * (r + 1) + 2 * (unsigned)lane;   // wave-uniform + lane
        double *dst = panel[r & (NBUF - 1)] + (4 * wave) * LDP;                                         // wave-uniform
#pragma unroll
        for (int k = 0; k < nch; ++k) {
            if (128 * k < ncols) {                                                     // uniform
                bool ok = 128 * k + 2 * lane < ncols;
                if constexpr (GLDS) {
                    if (ok) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            __builtin_amdgcn_global_load_lds(
                                (const __attribute__((address_space(1))) void *)(Ur + (size_t)rr * mp + 128 * k),
                                (__attribute__((address_space(3))) void *)(dst + rr * LDP + 128 * k), 16, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        pre[rr][k] = ok ? *reinterpret_cast<const d2 *>(Ur + (size_t)rr * mp + 128 * k) : d2{0.0, 0.0};
                }
            }
        }
        prev = V[(size_t)r * 256 + tid];
    };
    fetch(rmin, std::integral_constant<int, NCH>{});
    double zz = 0.0;
#ifdef GPRF_PROFILE
    // GPRF_SOLVE_STAMPS build: wave 0 of the Y workgroup: cycles in [stage | barrier | fetch | solve tile | updates]
    // (-DGPRF_SOLVE_STAMP_PART=p: the identity workgroup p instead; slots 6 / 7: cycles before the step loop / after it)
    unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#ifdef GPRF_SOLVE_STAMP_PART
    bool stamp = part_ == GPRF_SOLVE_STAMP_PART && wave == 0;
#else
    bool stamp = is_y && wave == 0;
#endif
    unsigned long long t_pro = tprev - t_kernel0, t_loop_end = 0;
#define GPRF_SST(k)                                                       \
    if (stamp) {                                                          \
        unsigned long long tn = __builtin_amdgcn_s_memtime();             \
        tacc[k] += tn - tprev;                                            \
        tprev = tn;                                                       \
    }
#else
#define GPRF_SST(k)
#endif
    static_for<0, MAXT>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if (r >= rmin && r < T) {                 // uniform over the workgroup
            double *buf = panel[r & (NBUF - 1)];
            constexpr int nch_r = (16 * (MAXT - 1 - r) + 127) / 128;          // chunks a unit of MAXT tiles needs at this step
            if constexpr (!GLDS) {
                int ncols = mp - 16 * (r + 1);
#pragma unroll
                for (int k = 0; k < nch_r; ++k) {
                    if (128 * k + 2 * lane < ncols) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            *reinterpret_cast<d2 *>(buf + (4 * wave + rr) * LDP + 128 * k + 2 * lane) = pre[rr][k];
                    }
                }
            }
            (void)nch_r;
            Vl[r & 1][tid] = prev;
            if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this step's panel has landed
            GPRF_SST(0)
            lds_barrier();                        // LDS only (register staging: no wait for the W / Z stores of the step before)
            GPRF_SST(1)
            if constexpr (NBUF == 2) {
                if (r + 1 < T) fetch(r + 1, std::integral_constant<int, (16 * (MAXT - 2 - r) + 127) / 128>{});
            }
            GPRF_SST(2)
            if (live && r >= r0) {
                const double *vl = Vl[r & 1] + lg * 16 + lr;
                d4 w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s) w = mfma(vl[64 * s], acc[r][s], w);
                if (is_y) {
                    double *zp = Z + (size_t)(16 * r + lg) * YPAD + 16 * cb + lr;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        zp[(size_t)(4 * q) * YPAD] = w[q];
                        zz += w[q] * w[q];
                    }
                } else {
                    double *wp = W + (size_t)(16 * r + lg) * mp + 16 * cb + lr;
#pragma unroll
                    for (int q = 0; q < 4; ++q) wp[(size_t)(4 * q) * mp] = w[q];
                }
                GPRF_SST(3)
                const double *pr = buf + lg * LDP + lr;
                static_for<r + 1, MAXT>([&](auto r2c) {
                    constexpr int r2 = decltype(r2c)::value;
                    if (r2 < T) {
                        // (the step's 16 products from zero, then ONE addition into the running tile: "hierarchical
                        // accumulation" above k_potrf_reg — the running tile otherwise rounds 16 times per step at its own magnitude)
                        d4 t16 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int s = 0; s < 4; ++s) t16 = mfma(-pr[(4 * s) * LDP + 16 * (r2 - r - 1)], w[s], t16);
                        acc[r2] += t16;
                    }
                });
                GPRF_SST(4)
            }
            if constexpr (NBUF == 1) {
                if (r + 1 < T) {              // (uniform) the one buffer is free when every wave has finished its updates
                    lds_barrier();
                    fetch(r + 1, std::integral_constant<int, (16 * (MAXT - 2 - r) + 127) / 128>{});
                }
            }
        }
    });
#ifdef GPRF_PROFILE
    t_loop_end = __builtin_amdgcn_s_memtime();
#endif
#undef GPRF_SST
    if (live && is_y) {
        for (int off = 32; off >= 1; off >>= 1) zz += shfl_xor_d(zz, off);
        if (lane == 0) pl.zzpart[(size_t)u * 4 + cb] = zz;
    }
#ifdef GPRF_PROFILE
    if (stamp) {
        __builtin_amdgcn_s_waitcnt(0);            // vmcnt(0): the W / Z stores have left
        unsigned long long t_end = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            for (int k = 0; k < 5; ++k) pl.dbg[(size_t)u * 8 + k] = (double)tacc[k];
            pl.dbg[(size_t)u * 8 + 5] = (double)T;
            pl.dbg[(size_t)u * 8 + 6] = (double)t_pro;
            pl.dbg[(size_t)u * 8 + 7] = (double)(t_end - t_loop_end);
        }
    }
#endif
    trace.done(T * 8 + part_);
}

// the instantiations of more than 20 tiles per edge live in files of their own (utp.pm_group < 0: the unit-major walk)
void launch_solve_wide(const UnitTab &utp, const Pools &p, int dy, dim3 grid, hipStream_t s);       // 21 .. 28 tiles
void launch_solve_wide32(const UnitTab &utp, const Pools &p, int dy, dim3 grid, hipStream_t s);     // 29 .. 32 tiles

}  // namespace gprf
