// gprf_mgrad.hip — the gradient reduce k_mgrad (gprf.py:556-584), k_gx_finalize, the Bethe-weighted assembly k_assemble
// (gprf.py:253-288) and the objective form's k_finish (gprfopt.py:377-417).
#include "gprf_dev.h"

namespace gprf {

// ------------------------------------------------------------------------------------------------
// Gradient reduce: M = At^T At - dy W^T W (= A A^T - dy K^-1) on the lower triangle, reduced against dk/dx and
// dk/dtheta.  By symmetry of M and k a strictly-lower tile (I > J) gives column sums to the points of J and row sums
// to the points of I:  gX[j][d] = sum_i M[i][j] dk(x_j, x_i)/dx_j[d]  (gprf.py:556-573);
// gC partials: tr(M), sum M*k_noise_free, sum M*dk/dl_t (gprf.py:577-584, 362-375).
// ------------------------------------------------------------------------------------------------
constexpr int G2_LD = 144;   // staged chunk row stride in doubles: 128 columns + 16 (lane groups 32 banks apart)

// sum over the 16 lanes of a DPP row (the lr index), result in every lane: rotate-and-add
__device__ __forceinline__ double row16_sum(double v) {
#define GPRF_ROR_ADD(n)                                                                                  \
    {                                                                                                    \
        int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + (n), 0xf, 0xf, false);        \
        int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + (n), 0xf, 0xf, false);        \
        v += __hiloint2double(hi, lo);                                                                   \
    }
    GPRF_ROR_ADD(8) GPRF_ROR_ADD(4) GPRF_ROR_ADD(2) GPRF_ROR_ADD(1)
#undef GPRF_ROR_ADD
    return v;
}

// ------------------------------------------------------------------------------------------------
// k_mgrad: one workgroup per 64x64 block pair (IB >= JB) of one unit.
// MFMA half: M_IJ = At_I^T At_J - dy sum_k W_kI^T W_kJ for its (up to) 16 lower-triangle tiles: 16-row chunks of
// the stacked operand [W ; At] (both row-major, leading dimension mp) are staged in LDS (two register sets keep
// chunks c+1 and c+2 in flight, one LDS-only barrier per chunk; diagonal blocks stage their 64 columns once); a wave
// owns one row tile I of the block (which one rotates with the workgroup) against the four J tiles: the A operands
// are read once per chunk, the B operands of tile jj + 1 before tile jj's four MFMAs; the W chunks come first with
// -dy riding on their A operands, the At chunks continue in the same accumulators.
// Reduction half: the accumulator layout is exactly what the reduction wants (a lane holds 4 rows of one
// column), so the tiles are reduced in place against dk/dx, dk/dtheta — M is never written.  A strictly-lower
// tile gives column sums to the points of J (-> colpart[j][IB]) and row sums to the points of I (-> rowpart[i][JB],
// DPP row reduction); the points' coordinates wait in LDS since kernel start.  k_gx_finalize folds the per-block
// partials in a fixed order.
// ------------------------------------------------------------------------------------------------
// HAVEK (SE only): the k values of strictly-lower tiles are read back from the K pool; false = K was never written
// (k_potrf_reg<.,.,true> generated it on the fly): they are re-evaluated like the diagonal tiles' ones.
// BIG (round 5): the units of more than 1024 points only — their M tiles were made by k_big_gemm (mode 2, 128 x 128 tiles at
// four times this kernel's flops per byte) and wait in the unit's region of the K pool: the chunk loop is skipped, the
// accumulators are loaded, the reductions are the same code.  The plain instantiations leave those units alone.
template <int DIST, int KERN, bool HAVEK, int FAST, bool BIG = false, int CLS = 0>
#ifndef GPRF_MGRAD_LLD_WPC
#define GPRF_MGRAD_LLD_WPC 2
#endif
__global__ __launch_bounds__(256, (DIST == 0 && KERN == 0) ? (FAST ? 4 : 3) : GPRF_MGRAD_LLD_WPC) void k_mgrad(UnitTab ut, Pools pl, KParams kp, int want_gc, int part_major) {
    static_assert(!(BIG && HAVEK), "the big units' K region holds M: their kernel values are re-evaluated");
    __shared__ double chunk[2][16 * G2_LD];
    // the coordinates (or lld records) of the I block's and the J block's points, fetched at kernel start so that the
    // reductions at the end find them in LDS instead of starting with exposed global loads
    __shared__ double xsh[128 * PtRec<DIST>::NREG];
    int TBm = (ut.max_T + 3) >> 2;
    int slot, bp;
    ClassList cl{ut.n_ids, 0, 0};
    WgTrace trace(ut, pl, 3);
    // (part_major: every unit's block pair 0 first, then every unit's pair 1, ...: pairs are in order of descending length)
    if constexpr (BIG) {
        // a unit of more than 1024 points is thousands of block pairs: consecutive workgroups = consecutive pairs of ONE unit,
        // dealt round-robin over the XCDs by the dispatcher (the maps below keep a unit on one XCD, for its L2: the single
        // 10000-point unit's 12403 pairs then ran on 32 of the 256 CUs, 1.2 ms instead of 0.2)
        const int nbp = TBm * (TBm + 1) / 2;
        slot = (int)blockIdx.x / nbp;
        bp = (int)blockIdx.x - slot * nbp;
        if (slot >= ut.n_ids) return;
    } else
    {
        cl = class_list<CLS>(ut);
        if (cl.n <= 0) return;
        if (!(part_major ? part_major_map(blockIdx.x, cl.n, TBm * (TBm + 1) / 2, ut.pm_group, &slot, &bp) : xcd_map(blockIdx.x, cl.n, TBm * (TBm + 1) / 2, &slot, &bp))) return;
        if (CLS != 0 && bp >= TBm * (TBm + 1) / 2) return;      // (a grid sized for more units than the partition has)
    }
    const UnitRef ur = BIG ? unit_ref(ut.srec, slot) : class_unit<CLS>(ut, cl, slot);
    int u = ur.u;
    int m = ur.m;
    int mp = pad16(m), T = mp >> 4;
    if (BIG != (T > SMALL_MAX_T)) return;      // (uniform) the other instantiation's units
    int TB = (T + 3) >> 2;
    // block pair index -> (JB, IB >= JB), enumerated over the launch-wide TBm
    int JB = 0, rem = bp;
    while (rem >= TBm - JB) { rem -= TBm - JB; ++JB; }
    int IB = JB + rem;
    if (IB >= TB) return;
    int tid = threadIdx.x;
    // the wave index as a scalar: everything derived from it (row tile, which tiles exist, diagonal or not) is
    // then uniform for the compiler too and turns into scalar branches instead of EXEC masking
    int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lr = lane & 15, lg = lane >> 4;
    size_t roff = ur.row_off;
    const double *__restrict__ W = pl.W + ur.mat_off;
    const double *__restrict__ At = pl.At + roff * YPAD;
    int J0 = 4 * JB;
    // Which of the block's four row tiles this wave owns rotates with the workgroup: in a diagonal block pair row tile
    // r has r + 1 column tiles, and wave w of every workgroup sits on SIMD w — unrotated, SIMD 3 would issue four
    // times the MFMAs of SIMD 0 in all of them at once.
    const int wrow = (wave + slot + bp) & 3;
    int I = 4 * IB + wrow;
    bool active = I < T;
    bool diagblk = IB == JB;
    double dyd = (double)kp.dy;
    int nchA = (kp.dy + 15) >> 4;
    int nchW = T - 4 * IB;
    int nch = nchW + nchA;

    // staging roles, all wave-uniform (scalar row pointers; the only per-lane part of an address is the lane itself):
    // waves 0/1 carry the even rows of a chunk, waves 2/3 the odd ones; waves 0/2 the I block's 64 columns, waves 1/3
    // the J block's (nothing for a diagonal block pair, whose columns are staged once)
    const int s_row0 = wave >> 1;
    const bool s_isJ = (wave & 1) != 0;
    const int s_col = 64 * (wave & 1) + lane;
    const int scol0 = 64 * (s_isJ ? JB : IB);             // first column of this wave's 64
    const bool s_skip = diagblk && s_isJ;
    const bool scol_ok = (scol0 + lane) < mp && !s_skip;
    int boff = diagblk ? 0 : 64;                          // where the J columns sit in the staged row

    {
        constexpr int XS0 = PtRec<DIST>::STRIDE, XN0 = PtRec<DIST>::NREG;
        const double *Xu0 = pl.Xu + roff * XS0;
        if (tid < 128) {
            int p = (tid < 64) ? 64 * IB + tid : 64 * JB + (tid - 64);
#pragma unroll
            for (int d = 0; d < XN0; ++d) xsh[tid * XN0 + d] = (p < mp) ? Xu0[(size_t)p * XS0 + d] : 0.0;
        }
    }
    d4 acc[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[jj] = d4{0, 0, 0, 0};
    bool need[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) need[jj] = active && (J0 + jj <= I) && (J0 + jj < T);

    double pre0[8], pre1[8];
    auto src_of = [&](int c) -> const double * {      // scalar: row s_row0 of chunk c, at this wave's first column
        return (c < nchW) ? (W + (size_t)(16 * (4 * IB + c) + s_row0) * mp + scol0)
                          : (At + (size_t)(16 * (c - nchW) + s_row0) * mp + scol0);
    };
    auto fetch0 = [&](int c) {
        const double *src = src_of(c);
#pragma unroll
        for (int e = 0; e < 8; ++e) pre0[e] = scol_ok ? src[(size_t)(2 * e) * mp + lane] : 0.0;
    };
    auto fetch1 = [&](int c) {
        const double *src = src_of(c);
#pragma unroll
        for (int e = 0; e < 8; ++e) pre1[e] = scol_ok ? src[(size_t)(2 * e) * mp + lane] : 0.0;
    };
    // need[] is a prefix (both of its conditions are monotone in jj): njj tiles.  The B operands of tile jj + 1 are
    // read from LDS before tile jj's MFMAs issue (left to itself the compiler emits read -> wait -> 2 MFMAs twice per
    // tile: two exposed LDS round trips per 256 cycles of MFMA).
    int njj = 0;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) njj += need[jj] ? 1 : 0;
    auto mma_chunk = [&](const double *buf, double asc) {
        const double *rowp = buf + lg * G2_LD + lr;
        double a[4], b[2][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = rowp[(4 * s) * G2_LD + 16 * wrow];
#pragma unroll
        for (int s = 0; s < 4; ++s) b[0][s] = rowp[(4 * s) * G2_LD + boff];
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] *= asc;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            if (jj < njj) {
                if (jj + 1 < 4 && jj + 1 < njj) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) b[(jj + 1) & 1][s] = rowp[(4 * s) * G2_LD + boff + 16 * (jj + 1)];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[jj] = mfma(a[s], b[jj & 1][s], acc[jj]);
            }
        }
    };
#ifdef GPRF_MGRAD_FINE
    unsigned long long tstep[3] = {0, 0, 0}, tsp = 0;      // [write + barrier | fetch issue | MFMAs]
#define GPRF_SST2(k) { unsigned long long tn = __builtin_amdgcn_s_memtime(); tstep[k] += tn - tsp; tsp = tn; }
#else
#define GPRF_SST2(k)
#endif
    // (chunks by global_load_lds straight into LDS — no staging registers, but only ONE chunk ahead with two LDS buffers,
    // and a third does not fit four workgroups per CU — measured slower: 118 vs 111 us)
    auto step = [&](int c, double (&pre)[8], bool refill_even) {
#ifdef GPRF_MGRAD_FINE
        tsp = __builtin_amdgcn_s_memtime();
#endif
        double *buf = chunk[c & 1];
        if (!s_skip) {
#pragma unroll
            for (int e = 0; e < 8; ++e) buf[(2 * e + s_row0) * G2_LD + s_col] = pre[e];
        }
        // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for chunk c+1's loads, which were
        // issued one step ago precisely so that they need NOT be back yet
        lds_barrier();
        GPRF_SST2(0)
        if (c + 2 < nch) { if (refill_even) fetch0(c + 2); else fetch1(c + 2); }
        GPRF_SST2(1)
        // the -dy of the W part rides on the A operand (4 multiplies per chunk): the accumulators are never
        // rescaled in the middle of the chunk loop
        if (c < nchW) {
            if (active && (4 * IB + c) >= I) mma_chunk(buf, -dyd);
        } else {
            if (active) mma_chunk(buf, 1.0);       // (the last chunk's rows beyond dy are zero padding)
        }
        GPRF_SST2(2)
    };
#ifdef GPRF_PROFILE
    // diagnostic build: cycles of [prologue | chunk loop | reductions] of the unit's first (diagonal, longest) and last
    // (bottom-left) block pair -> Pools::dbg[u][0..3] / [4..7]
    unsigned long long tm0 = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (BIG) {
        const double *__restrict__ Mp = pl.K + ur.mat_off;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            if (need[jj]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[jj][q] = Mp[(size_t)(16 * I + lg + 4 * q) * mp + 16 * (J0 + jj) + lr];
            }
    } else {
        fetch0(0);
        if (nch > 1) fetch1(1);
    }
#ifdef GPRF_PROFILE
    unsigned long long tm1 = __builtin_amdgcn_s_memtime();
#endif
    for (int c = 0; !BIG && c < nch; c += 2) {
        step(c, pre0, true);
        if (c + 1 < nch) step(c + 1, pre1, false);
    }
#ifdef GPRF_PROFILE
    unsigned long long tm2 = __builtin_amdgcn_s_memtime();
    unsigned long long tme[4] = {0, 0, 0, 0};
#define GPRF_MST(k) tme[k] = __builtin_amdgcn_s_memtime();
#else
#define GPRF_MST(k)
#endif
    // ---- the block pair's M tiles are in the accumulators (lane (lg, lr), acc[jj][q] = M[16 I + lg + 4q][16 (J0+jj)
    //      + lr]); reduce them against dk/dx and dk/dtheta right here: M never goes to memory ----
    __syncthreads();                                   // the staging buffer is reused for the reductions
    GPRF_MST(0)
    double (*red)[64][4] = reinterpret_cast<double (*)[64][4]>(&chunk[0][0]);      // [4 waves][64 columns][4]
    double (*gcred)[8] = reinterpret_cast<double (*)[8]>(&chunk[1][0]);            // [4 waves][8]
    const double *__restrict__ Kp = pl.K + ur.mat_off;
    constexpr int XN = PtRec<DIST>::NREG;
    const int tbs = TBm;                               // stride of the per-block partials
    // FAST instantiation (SE kernel, at most two input dimensions, no hyper-parameter gradient — the north-star
    // task): the third coordinate's terms and the theta sums are compiled out of the reductions, whose VALU volume
    // is what bounds this kernel
    constexpr int ND = FAST ? 2 : 3;
    constexpr bool GC = FAST != 1;
    double rowsum[4][3], xi[4][XN];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int i = 16 * I + lg + 4 * q;
#pragma unroll
        for (int d = 0; d < 3; ++d) rowsum[q][d] = 0.0;
#pragma unroll
        for (int d = 0; d < XN; ++d) xi[q][d] = xsh[(16 * wrow + lg + 4 * q) * XN + d];
        (void)i;
    }
    double gc_tr = 0.0, gc_sv = 0.0, gc_l[3] = {0.0, 0.0, 0.0};
    double csum[4][3];
    // SE kernel: dk/dx = -2 delta / ls^2 k, dk/dls = 2 delta^2 / ls^3 k; the factors are applied to the sums
    // (unused dimensions have ls = 0 in KParams: factor 0 there, not inf)
    double fx[3], fl[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        bool used = d < kp.dx;
        fx[d] = used ? -2.0 / (kp.ls[d] * kp.ls[d]) : 0.0;
        fl[d] = used ? 2.0 / (kp.ls[d] * kp.ls[d] * kp.ls[d]) : 0.0;
    }
    // the wave's diagonal tile (diagonal block pairs only; it is tile jj == wrow): k re-evaluated from the
    // coordinates (the pool holds U there), column sums only
    double csd[3] = {0.0, 0.0, 0.0};
    constexpr bool epi = true;
    if (epi && active && diagblk) {                    // wave-uniform
        d4 md = wrow == 0 ? acc[0] : (wrow == 1 ? acc[1] : (wrow == 2 ? acc[2] : acc[3]));
        int j = 16 * I + lr;
        double xj[XN];
#pragma unroll
        for (int d = 0; d < XN; ++d) xj[d] = xsh[(16 * wrow + lr) * XN + d];      // diagonal block: J block = I block
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int i = 16 * I + lg + 4 * q;
            bool ok = (i < m) && (j < m);
            double Mij = ok ? md[q] : 0.0;
            if constexpr (DIST == 0 && KERN == 0) {
                double g = Mij * KernFn<0, 0>::value(kp, xi[q], xj);
                if constexpr (GC) {
                    gc_tr += (i == j) ? Mij : 0.0;
                    gc_sv += g;
                }
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    double delta = xj[d] - xi[q][d];
                    double gd = g * delta;
                    csd[d] += gd;
                    if constexpr (GC) gc_l[d] += gd * delta;
                }
            } else {
                {
                    // (branch-free: padding entries have Mij = 0 and finite derivatives — the four entries of a lane are
                    // independent chains the scheduler can interleave)
                    double dkdxi[3] = {0, 0, 0}, dkdxj[3] = {0, 0, 0}, dkdl[3] = {0, 0, 0};
                    double k = KernFn<DIST, KERN>::pair(kp, xi[q], xj, false, 0.0, dkdxi, dkdxj, dkdl);
                    const double Mo = (i != j) ? Mij : 0.0;
#pragma unroll
                    for (int d = 0; d < 3; ++d) csd[d] += Mo * dkdxj[d];
                    gc_tr += (i == j) ? Mij : 0.0;
                    gc_sv += Mij * k;
#pragma unroll
                    for (int d = 0; d < 3; ++d) gc_l[d] += Mij * dkdl[d];
                }
            }
        }
    }
    GPRF_MST(1)
    // strictly-lower tiles: k read back from the K pool, which holds the 64x64 blocks JB <= IB only: transposed
    // access for an off-diagonal block pair (the four q-loads of a lane cover one 128-byte line);
    // column sums for the points of J, row sums for the points of I, everything counted twice in the theta sums
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        bool mydiag = diagblk && jj == wrow;
        double colsum[3] = {mydiag ? csd[0] : 0.0, mydiag ? csd[1] : 0.0, mydiag ? csd[2] : 0.0};
        if (epi && need[jj] && J0 + jj < I) {          // wave-uniform
            int J = J0 + jj;
            int j = 16 * J + lr;
            double xj[XN], Kv[4];
#pragma unroll
            for (int d = 0; d < XN; ++d) xj[d] = xsh[(64 + 16 * jj + lr) * XN + d];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if constexpr (DIST == 0 && KERN == 0 && HAVEK)
                Kv[q] = diagblk ? Kp[(size_t)(16 * I + lg + 4 * q) * mp + 16 * J + lr]      // diagonal blocks are whole
                                : Kp[(size_t)(16 * J + lr) * mp + 16 * I + lg + 4 * q];     // K(i,j) = K(j,i)
            if constexpr (DIST == 0 && KERN == 0 && !HAVEK) {
                double sq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    sq[q] = se_neg_r2(xi[q][0], xi[q][1], ND > 2 ? xi[q][2] : 0.0, xj[0], xj[1], ND > 2 ? xj[2] : 0.0, kp.inv_ls, ND > 2);
                exp_fast_v<4>(sq, Kv);                  // four chains side by side
#pragma unroll
                for (int q = 0; q < 4; ++q) Kv[q] *= kp.sv;
            }

#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int i = 16 * I + lg + 4 * q;
                bool ok = (i < m) && (j < m);
                double Mij = ok ? acc[jj][q] : 0.0;
                if constexpr (DIST == 0 && KERN == 0) {
                    double g = Mij * Kv[q];
                    if constexpr (GC) gc_sv += 2.0 * g;
#pragma unroll
                    for (int d = 0; d < ND; ++d) {
                        double delta = xj[d] - xi[q][d];
                        double gd = g * delta;
                        colsum[d] += gd;
                        rowsum[q][d] -= gd;
                        if constexpr (GC) gc_l[d] += 2.0 * gd * delta;
                    }
                } else {
                    {
                        // (round 5 also wrote the lane's four pair evaluations step-major, four dependent chains advancing
                        // together, haversine to exp: 240 us against 223 on the seismic shape, 88 bytes of scratch — the
                        // reductions are bound by instruction issue, not by the chains' latency.  Dropped.)
                        double dkdxi[3] = {0, 0, 0}, dkdxj[3] = {0, 0, 0}, dkdl[3] = {0, 0, 0};
                        double k = KernFn<DIST, KERN>::pair(kp, xi[q], xj, false, 0.0, dkdxi, dkdxj, dkdl);
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            colsum[d] += Mij * dkdxj[d];
                            rowsum[q][d] += Mij * dkdxi[d];
                        }
                        gc_sv += 2.0 * Mij * k;
#pragma unroll
                        for (int d = 0; d < 3; ++d) gc_l[d] += 2.0 * Mij * dkdl[d];
                    }
                }
            }
        }
        // column sums of tile column jj over this wave's 16 rows (zero for a wave without that tile)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            double v = 0.0;
            if (d < ND) {
                v = colsum[d];
                if constexpr (DIST == 0 && KERN == 0) v *= fx[d];
                v += shfl_xor_d(v, 16);
                v += shfl_xor_d(v, 32);
            }
            csum[jj][d] = v;
        }
    }
    GPRF_MST(2)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (lg == 0) red[wrow][16 * jj + lr][d] = csum[jj][d];      // by ROW TILE, not by wave: the fold below must
                                                                        // not depend on the launch-dependent rotation
    // row sums of this wave's 16 rows over the block's 64 columns -> rowpart[row][JB]
    if (active) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double rs[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                rs[d] = 0.0;
                if (d < ND) {
                    rs[d] = row16_sum(rowsum[q][d]);
                    if constexpr (DIST == 0 && KERN == 0) rs[d] *= fx[d];
                }
            }
            if (lr < 3) {
                double v = (lr == 0) ? rs[0] : ((lr == 1) ? rs[1] : rs[2]);
                pl.rowpart[((roff + 16 * I + lg + 4 * q) * tbs + JB) * XPAD + lr] = v;
            }
        }
    }
    if constexpr (DIST == 0 && KERN == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) gc_l[d] *= fl[d];
    }
    double gcv[5] = {gc_tr, gc_sv, gc_l[0], gc_l[1], gc_l[2]};
    if constexpr (GC) {
        if (want_gc) {
#pragma unroll
            for (int t = 0; t < 5; ++t)
                for (int off = 32; off >= 1; off >>= 1) gcv[t] += shfl_xor_d(gcv[t], off);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < 5; ++t) gcred[wrow][t] = gcv[t];
    }
    GPRF_MST(3)
    __syncthreads();
    {
        // column partial of this block pair: colpart[column j of block JB][IB]
        int jc = tid >> 2, d = tid & 3;
        int j = 64 * JB + jc;
        if (j < mp) {
            double v = 0.0;
            if (d < 3) v = red[0][jc][d] + red[1][jc][d] + red[2][jc][d] + red[3][jc][d];
            pl.colpart[((roff + j) * tbs + IB) * XPAD + d] = v;
        }
    }
#ifdef GPRF_PROFILE
#ifndef GPRF_MGRAD_FINE
    if (tid == 0 && (bp == 0 || (JB == 0 && IB == TB - 1 && TB > 1))) {
        unsigned long long tm3 = __builtin_amdgcn_s_memtime();
        double *dg = pl.dbg + (size_t)u * 8 + (bp == 0 ? 0 : 4);
        dg[0] = (double)(tm1 - tm0); dg[1] = (double)(tm2 - tm1); dg[2] = (double)(tm3 - tm2); dg[3] = (double)nch;
    }
#endif
#ifdef GPRF_MGRAD_FINE
    // the reductions split [barrier | diagonal tile | lower tiles | row sums, theta sums | last barrier + stores], then
    // the chunk loop and its length; bottom-left pair only (overwrites the record above)
    if (tid == 0 && JB == 0 && IB == TB - 1) {
        double *dg = pl.dbg + (size_t)u * 8;
        unsigned long long tm3 = __builtin_amdgcn_s_memtime();
        dg[0] = (double)(tme[0] - tm2); dg[1] = (double)(tme[1] - tme[0]); dg[2] = (double)(tme[2] - tme[1]);
        dg[3] = (double)(tme[3] - tme[2]); dg[4] = (double)(tm3 - tme[3]); dg[5] = (double)(tm2 - tm1); dg[6] = (double)nch;
#ifdef GPRF_MGRAD_LOOP      // ... or the chunk loop split [LDS write + barrier | fetch issue | MFMAs] in slots 0..2
        dg[0] = (double)tstep[0]; dg[1] = (double)tstep[1]; dg[2] = (double)tstep[2];
#endif
    }
#else
    (void)tme;
#endif
#endif
#undef GPRF_MST
    if (tid < GC_SLOTS) {
        double v = 0.0;
        if (tid < 5) v = gcred[0][tid] + gcred[1][tid] + gcred[2][tid] + gcred[3][tid];
        int pidx = JB * TB - (JB * (JB - 1)) / 2 + (IB - JB);       // block pair index within the unit
        pl.gcpart[((size_t)u * (TBm * (TBm + 1) / 2) + pidx) * GC_SLOTS + tid] = v;
    }
    trace.done(T * 64 + IB * 8 + JB);
}

// gXu[row] = sum_{IB >= B} colpart[row][IB] + sum_{JB <= B} rowpart[row][JB],  B = the row's 64-point block
// (fixed order -> bit-reproducible)
__global__ __launch_bounds__(256) void k_gx_finalize(UnitTab ut, Pools pl, KParams kp, int want_gc) {
    int u = blockIdx.x;
    int m = ut.m[u];
    int mp = pad16(m);
    int r0 = ut.row_off[u];
    int TB = ((mp >> 4) + 3) >> 2;
    int tbs = (ut.max_T + 3) >> 2;
    // the unit's six terms of the final sums (weighted log-likelihood, weighted hyper-parameter gradient), by the last
    // wave while the others fold the gradient slab: k_assemble's single summing workgroup then adds one row per unit
    // instead of walking every unit's partials through chains of dependent loads (C4: 4033 units, 106 -> 20 us)
    if (threadIdx.x >= 192) {
        int t = threadIdx.x - 192;
        if (t < 6) {
            double w = ut.weight[u], v = 0.0;
            if (m > 0) {
                if (t == 0) {
                    const double *zp = pl.zzpart + (size_t)u * 4;
                    double zz = (zp[0] + zp[1]) + (zp[2] + zp[3]);
                    double ll = -0.5 * zz - 0.5 * kp.dy * pl.logdet[u] - 0.5 * kp.dy * m * 1.8378770664093454836 /* log 2pi */;
                    v = w * ll;
                } else if (want_gc) {
                    int nP = TB * (TB + 1) / 2;   // k_mgrad writes one partial per 64x64 block pair
                    double g = 0.0;
                    for (int P = 0; P < nP; ++P)
                        g += pl.gcpart[((size_t)u * (tbs * (tbs + 1) / 2) + P) * GC_SLOTS + (t - 1)];
                    // d/d nv: 1/2 tr(M); d/d sv: 1/2 sum M k / sv; d/d ls_t: 1/2 sum M dk/dls_t
                    v = (t == 2) ? w * 0.5 * g / kp.sv : w * 0.5 * g;
                }
            }
            pl.usum[(size_t)u * 8 + t] = v;
        }
    }
    for (int idx = threadIdx.x; idx < 4 * mp; idx += 256) {
        int local = idx >> 2, d = idx & 3;
        if (d == 3) continue;
        int row = r0 + local;
        int B = local >> 6;
        const double *cp = pl.colpart + (size_t)row * tbs * XPAD + d;
        const double *rp = pl.rowpart + (size_t)row * tbs * XPAD + d;
        double v = 0.0;
        for (int IB = B; IB < TB; ++IB) v += cp[IB * XPAD];
        for (int JB = 0; JB <= B; ++JB) v += rp[JB * XPAD];
        pl.gXu[(size_t)row * XPAD + d] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Assembly (gprf.py:253-288): Bethe-weighted sums, deterministic gather (no float atomics).
// block 0: ll and gradC; blocks >= 1: gradX, one thread per (point, coordinate).
// out = [ll | gradX (n x dx) | gradC (2 + ndfn) | overflow flag | units not PD]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assemble(UnitTab ut, Pools pl, AssembleTab at, KParams kp, int n,
                                                  int want_gx, int want_gc, double *out, int usum_ok, ObjTab ob) {
    int dx = kp.dx;
    if (blockIdx.x == 0) {
        __shared__ double red[256][6];
        double acc[6] = {0, 0, 0, 0, 0, 0};
        __shared__ int s_notpd;
        if (threadIdx.x == 0) s_notpd = 0;
        __syncthreads();
        // Four units per thread and pass, every unit's words requested before the first is used (a thread beyond the last unit
        // re-reads unit 0 and adds nothing): unit by unit — status word, size, then the sums — each unit was two or three
        // dependent memory round trips, and this workgroup is the tail of the evaluation.  Same terms, same order of addition.
        constexpr int NB = 4;
        for (int u0 = threadIdx.x; u0 < ut.n_units; u0 += 256 * NB) {
            int inf[NB], mm[NB];
            double ww[NB], ld[NB], zq[NB][4], us[NB][6];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int u = u0 + 256 * j, uc = u < ut.n_units ? u : 0;
                inf[j] = pl.info[uc];
                if (usum_ok) {
                    const double *up = pl.usum + (size_t)uc * 8;
                    us[j][0] = up[0];
                    if (want_gc)
#pragma unroll
                        for (int t = 1; t < 6; ++t) us[j][t] = up[t];
                } else {
                    mm[j] = ut.m[uc];
                    ww[j] = ut.weight[uc];
                    ld[j] = pl.logdet[uc];
                    const double *zp = pl.zzpart + (size_t)uc * 4;
#pragma unroll
                    for (int t = 0; t < 4; ++t) zq[j][t] = zp[t];
                }
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int u = u0 + 256 * j;
                if (u >= ut.n_units) continue;
                if (inf[j] != 0) atomicAdd(&s_notpd, 1);
                if (usum_ok) {                        // the terms k_gx_finalize left (same values, same order of addition)
                    acc[0] += us[j][0];
                    if (want_gc)
#pragma unroll
                        for (int t = 1; t < 6; ++t) acc[t] += us[j][t];
                    continue;
                }
                const int m = mm[j];
                const double w = ww[j];
                if (m > 0) {
                    double zz = (zq[j][0] + zq[j][1]) + (zq[j][2] + zq[j][3]);
                    double ll = -0.5 * zz - 0.5 * kp.dy * ld[j] -
                                0.5 * kp.dy * m * 1.8378770664093454836 /* log 2pi */;
                    acc[0] += w * ll;
                    if (want_gc) {
                        int T = pad16(m) >> 4;
                        int TB = (T + 3) >> 2, TBm = (ut.max_T + 3) >> 2;
                        int nP = TB * (TB + 1) / 2;   // k_mgrad writes one partial per 64x64 block pair
                        double g[5] = {0, 0, 0, 0, 0};
                        const double *gp0 = pl.gcpart + ((size_t)u * (TBm * (TBm + 1) / 2)) * GC_SLOTS;
                        for (int P = 0; P < nP; ++P) {
                            const double *gp = gp0 + (size_t)P * GC_SLOTS;
                            for (int t = 0; t < 5; ++t) g[t] += gp[t];
                        }
                        acc[1] += w * 0.5 * g[0];             // d/d nv   : 1/2 tr(M)
                        acc[2] += w * 0.5 * g[1] / kp.sv;     // d/d sv   : 1/2 sum M k / sv
                        for (int t = 0; t < 3; ++t) acc[3 + t] += w * 0.5 * g[2 + t];
                    }
                }
            }
        }
        for (int t = 0; t < 6; ++t) red[threadIdx.x][t] = acc[t];
        __syncthreads();
        for (int s = 128; s >= 1; s >>= 1) {
            if (threadIdx.x < s)
                for (int t = 0; t < 6; ++t) red[threadIdx.x][t] += red[threadIdx.x + s][t];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            out[0] = red[0][0];
            int ncov = 2 + kp.ndfn;
            double *gc = out + 1 + (size_t)n * dx;
            for (int t = 0; t < ncov; ++t) gc[t] = want_gc ? red[0][1 + t] : 0.0;
            // two status words behind the result, so that a caller who SUM-all-reduces the vector over ranks learns in
            // the same collective whether any rank has to repeat (workspace outgrown) or to jitter (a unit not PD)
            gc[ncov] = at.ctl[CTL_OVERFLOW] ? 1.0 : 0.0;
            gc[ncov + 1] = (double)s_notpd;
        }
        if (at.mirror_dst) {      // (eight words per thread in flight: element by element it is load, wait, store)
            for (int i0 = threadIdx.x; i0 < at.mirror_n; i0 += 8 * 256) {
                int32_t w8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) w8[q] = at.mirror_src[i0 + 256 * q < at.mirror_n ? i0 + 256 * q : 0];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (i0 + 256 * q < at.mirror_n) at.mirror_dst[i0 + 256 * q] = w8[q];
            }
        }
        return;
    }
    // gradX: 32 points per workgroup, 8 lanes per point — one per unit that contains the point's block, eight at a time
    // (the row lookups are chains of dependent loads: spread over lanes they overlap) — and the terms of a point are
    // added up by ONE lane in ascending unit order (gprf.py:258-273: unary term first, then the pair terms in neighbour
    // order): bit-reproducible, and the same sum whatever the launch looks like.
    __shared__ double term[32][8][3];
    __shared__ int s_maxcnt;
    int t = threadIdx.x, i = t >> 3, e = t & 7;
    int p = (blockIdx.x - 1) * 32 + i;
    // (first CSR entry of the point's block, number of entries): k_scatter_x left them with the partition.  Asked for together
    // with the overflow word, not behind it (a thread without a point re-reads point 0)
    const bool inb = want_gx && p < n;
    const int pc = inb ? p : 0;
    const int over_w = at.ctl[CTL_OVERFLOW];
    const int2 pe2 = reinterpret_cast<const int2 *>(at.pe)[pc];
    const int pos_l = at.posb[pc];
    const bool live = inb && !over_w;
    const int e0 = live ? pe2.x : 0, cnt = live ? pe2.y : 0, pos = live ? pos_l : 0;
    if (t == 0) s_maxcnt = 0;
    __syncthreads();
    if (e == 0 && cnt > 0) atomicMax(&s_maxcnt, cnt);
    __syncthreads();
    int maxcnt = s_maxcnt;
    double v = 0.0;                           // lane e < dx of point i carries coordinate e
    const int tbs_l = (ut.max_T + 3) >> 2;
    // The common shape — units of at most 256 points (four 64-point blocks), a point in at most 16 units, the partials folded
    // here — with every memory round trip of a point's terms taken ONCE: (1) row, weight and block info of both of the lane's
    // entries, (2) all eight partials of both rows.  The loop below is the same arithmetic for any shape; compiled, it is a
    // chain of dependent trips — entry, then its info, then one trip per 64-point block of the column sums, then one per block
    // of the row sums, twice over for a point in nine units: up to fourteen L2 latencies, most of this kernel's 12 us.
    // A term the loop does not add enters here as + 0.0, which changes no bit of a sum that started from + 0.0.
    if (at.fold_gx && tbs_l <= 4 && maxcnt <= 16) {
        typedef double d2v __attribute__((ext_vector_type(2)));
        int rowq[2], infoq[2];
        double wq[2];
        bool okq[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            okq[j] = 8 * j + e < cnt;
            const int idx = okq[j] ? e0 + 8 * j + e : 0;
            const int eb = at.ebase[idx];
            rowq[j] = okq[j] ? eb + pos : 0;      // (an entry nobody wrote — an empty block, an overflowed build — must not form an address)
            wq[j] = at.ewgt[idx];
            infoq[j] = at.einfo[idx];
        }
        // (2) exactly the partials the sums below add — the blocks IB = B .. TB - 1 of the column sums, JB = 0 .. B of the row
        // sums, of the entries this lane has — behind exec masks, consumed only when all of them are in flight
        const d2v zero2 = {0.0, 0.0};
        d2v cq[2][4][2], rq[2][4][2];
        int Bq[2], TBq[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            TBq[j] = infoq[j] & 0x3ff; Bq[j] = ((infoq[j] >> 10) + pos) >> 6;
            const d2v *cp = reinterpret_cast<const d2v *>(pl.colpart + (size_t)rowq[j] * tbs_l * XPAD);
            const d2v *rp = reinterpret_cast<const d2v *>(pl.rowpart + (size_t)rowq[j] * tbs_l * XPAD);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                cq[j][b][0] = zero2; cq[j][b][1] = zero2; rq[j][b][0] = zero2; rq[j][b][1] = zero2;
                if (okq[j] && b >= Bq[j] && b < TBq[j]) { cq[j][b][0] = cp[2 * b]; cq[j][b][1] = cp[2 * b + 1]; }
                if (okq[j] && b <= Bq[j] && b < tbs_l) { rq[j][b][0] = rp[2 * b]; rq[j][b][1] = rp[2 * b + 1]; }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (8 * j < maxcnt) {      // (uniform)
                const int TB = TBq[j], B = Bq[j];
                double v0 = 0.0, v1 = 0.0, v2 = 0.0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const bool in = b >= B && b < TB;
                    v0 += in ? cq[j][b][0][0] : 0.0; v1 += in ? cq[j][b][0][1] : 0.0; v2 += in ? cq[j][b][1][0] : 0.0;
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const bool in = b <= B;
                    v0 += in ? rq[j][b][0][0] : 0.0; v1 += in ? rq[j][b][0][1] : 0.0; v2 += in ? rq[j][b][1][0] : 0.0;
                }
                term[i][e][0] = okq[j] ? wq[j] * v0 : 0.0;
                term[i][e][1] = okq[j] ? wq[j] * v1 : 0.0;
                term[i][e][2] = okq[j] ? wq[j] * v2 : 0.0;
                __syncthreads();
                if (e < dx) {
                    int kn = cnt - 8 * j < 8 ? cnt - 8 * j : 8;
                    for (int k = 0; k < kn; ++k) v += term[i][k][e];
                }
                __syncthreads();
            }
        }
        maxcnt = 0;      // (the loop below has nothing left to do)
    }
    for (int k0 = 0; k0 < maxcnt; k0 += 8) {
        double g0 = 0.0, g1 = 0.0, g2 = 0.0;
        if (k0 + e < cnt) {
            int row = at.ebase[e0 + k0 + e] + pos;      // = row_off[u] + (second block ? off_j[u] : 0) + pos
            double w = at.ewgt[e0 + k0 + e];            // = weight[u]
            if (at.fold_gx) {
                // k_gx_finalize's sum, here: the row's partials over the 64-point blocks of its unit, same order
                const int info = at.einfo[e0 + k0 + e];
                const int TB = info & 0x3ff, B = ((info >> 10) + pos) >> 6;
                const int tbs = (ut.max_T + 3) >> 2;
                const double *cp = pl.colpart + (size_t)row * tbs * XPAD;
                const double *rp = pl.rowpart + (size_t)row * tbs * XPAD;
                double v0 = 0.0, v1 = 0.0, v2 = 0.0;
                for (int IB = B; IB < TB; ++IB) { v0 += cp[IB * XPAD]; v1 += cp[IB * XPAD + 1]; v2 += cp[IB * XPAD + 2]; }
                for (int JB = 0; JB <= B; ++JB) { v0 += rp[JB * XPAD]; v1 += rp[JB * XPAD + 1]; v2 += rp[JB * XPAD + 2]; }
                g0 = w * v0; g1 = w * v1; g2 = w * v2;
            } else {
                const double *gr = pl.gXu + (size_t)row * XPAD;
                g0 = w * gr[0]; g1 = w * gr[1]; g2 = w * gr[2];
            }
        }
        term[i][e][0] = g0; term[i][e][1] = g1; term[i][e][2] = g2;
        __syncthreads();
        if (e < dx) {
            int kn = cnt - k0 < 8 ? cnt - k0 : 8;
            for (int k = 0; k < kn; ++k) v += term[i][k][e];
        }
        __syncthreads();
    }
    // gprf_objective: the optimiser's form of the result — the location prior's gradient -(x - x_obs) / sigma^2 added
    // here (gprfopt.py:172-182, 396-399), signs flipped (gprfopt.py:417); the prior's log-density leaves as one partial
    // sum of ((x - x_obs) / sigma)^2 per workgroup, folded in a fixed order by k_finish
    double r2 = 0.0;
    if (ob.on && p < n && e < dx) {
        if (ob.Xobs) {
            double d = ob.X[(size_t)p * dx + e] - ob.Xobs[(size_t)p * dx + e];
            double r = d / ob.sigma;
            r2 = r * r;
            v += -d / ob.var;
        }
        v = -v;
    }
    if (p < n && e < dx) out[1 + (size_t)p * dx + e] = v;
    if (ob.on && ob.Xobs) {      // (uniform)
        __shared__ double r2w[4];
        for (int off = 32; off >= 1; off >>= 1) r2 += shfl_xor_d(r2, off);
        if ((t & 63) == 0) r2w[t >> 6] = r2;
        __syncthreads();
        if (t == 0) ob.part[blockIdx.x - 1] = (r2w[0] + r2w[1]) + (r2w[2] + r2w[3]);
    }
}

// k_finish (gprf_objective only; takes k_done's place at the end of a host-in / host-out evaluation): out[0] = -(ll +
// location prior), the prior's partial sums folded in a fixed order; xp_const = -1/2 N log(2 pi sigma^2) (gprfopt.py:178).
// extras (may be nullptr) <- [ll of the GPRF terms alone, location prior].
__global__ __launch_bounds__(256) void k_finish(double *out, ObjTab ob, int nparts, double xp_const, double *extras,
                                                int32_t *flag, int32_t seq) {
    __shared__ double red[256];
    int t = threadIdx.x;
    double s = 0.0;
    if (ob.Xobs)
        for (int i = t; i < nparts; i += 256) s += ob.part[i];
    red[t] = s;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
        if (t < h) red[t] += red[t + h];
        __syncthreads();
    }
    if (t == 0) {
        double ll = out[0];
        double xp = ob.Xobs ? -0.5 * red[0] + xp_const : 0.0;
        out[0] = -(ll + xp);
        if (extras) { extras[0] = ll; extras[1] = xp; }
        if (flag) {
            __threadfence_system();
            __atomic_store_n(flag, seq, __ATOMIC_RELEASE);
        }
    }
}

void launch_finish(double *out, const ObjTab &ob, int nparts, double xp_const, double *extras, int32_t *flag, int32_t seq,
                   hipStream_t s) {
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, s, out, ob, nparts, xp_const, extras, flag, seq);
}


void launch_gx_finalize(const UnitTab &ut, const Pools &p, const KParams &kp, int want_gc, hipStream_t s) {
    if (ut.n_units == 0) return;
    hipLaunchKernelGGL(k_gx_finalize, dim3(ut.n_units), dim3(256), 0, s, ut, p, kp, want_gc);
}

// the gradient kernel of one size class (SE kernel, at most two input dimensions: the configurations launch_potrf pipelines by
// class), behind that class's At on its queue; launch-wide part-major walk over the class's list
void launch_grad_class(const UnitTab &ut, const Pools &p, const KParams &kp, int want_gc, int which, hipStream_t s) {
    const int TBm = (ut.max_T + 3) / 4, nbp = TBm * (TBm + 1) / 2;
    UnitTab utp = ut;
    // (a launch more than two rounds of CUs deep is walked in groups of 64 slots, like launch_grad's: L2 residency)
    const int G = ut.n_launch > 2 * device_cus() ? 64 : 0;
    utp.pm_group = G;
    const int n = which == 1 ? ut.grid_big : ut.grid_small;
    if (n <= 0) return;
    dim3 grid(G > 0 ? ((n + G - 1) / G) * G * nbp : xcd_grid(n, nbp));
    if (which == 1) {
        if (want_gc) hipLaunchKernelGGL((k_mgrad<0, 0, false, 2, false, 1>), grid, dim3(256), 0, s, utp, p, kp, want_gc, 1);
        else hipLaunchKernelGGL((k_mgrad<0, 0, false, 1, false, 1>), grid, dim3(256), 0, s, utp, p, kp, want_gc, 1);
    } else {
        if (want_gc) hipLaunchKernelGGL((k_mgrad<0, 0, false, 2, false, 2>), grid, dim3(256), 0, s, utp, p, kp, want_gc, 1);
        else hipLaunchKernelGGL((k_mgrad<0, 0, false, 1, false, 2>), grid, dim3(256), 0, s, utp, p, kp, want_gc, 1);
    }
}

void launch_grad(int dist_id, int kern_id, const UnitTab &ut, const Pools &p, const KParams &kp, int want_gc,
                 bool have_K, hipStream_t s) {
    if (ut.n_ids == 0 || ut.max_T == 0) return;
    int TBm = (ut.max_T + 3) / 4;
    // part by part (longest workgroups first): launch-wide while the launch is at most two rounds of CUs wide; deeper launches
    // in GROUPS of 64 launch slots — a unit's workgroups then run within one L2 residency window and longest first inside the
    // group (C4: 762 -> 738 us; launch-wide there the ten workgroups of a unit run far apart and each fetches the unit's W / At
    // from HBM again: 809).  diag part_major=0: unit by unit.
    const int pm = diag("part_major", 1) != 0 ? 1 : 0;
    // (round 5: the lld / Matern instantiation always in groups — its block pairs re-read W / At at 482 MB per launch walked
    // launch-wide on the seismic shape — and never with fewer than two groups: part_major_map then walks launch-wide)
    const int G_d = diag("mgrad_group", -1);      // (diagnostic: the walk's group size, a multiple of 8; 0 = launch-wide)
    const int G = G_d >= 0 ? G_d : ((ut.n_launch > 2 * device_cus() || (dist_id == 1 && ut.n_launch > 128)) ? 64 : 0);      // (n_launch: see launch_solve)
    const int nbp = TBm * (TBm + 1) / 2;
    dim3 grid(pm && G > 0 ? ((ut.n_ids + G - 1) / G) * G * nbp : xcd_grid(ut.n_ids, nbp));
    UnitTab utp = ut;
    utp.pm_group = G;
    // (a launch with units of more than BIG_LA_T tiles: their regions of the K pool have been the blocked substitution's scratch
    // — every kernel value is re-evaluated)
    if (ut.max_T > BIG_LA_T) have_K = false;
    if (dist_id == 0 && kern_id == 0) {
        // 0: general; 1: at most two input dimensions, no hyper-parameter gradient; 2: two dimensions with it
        const int fast = kp.dx <= 2 ? (want_gc ? 2 : 1) : 0;
        if (have_K) {
            if (fast == 1) hipLaunchKernelGGL((k_mgrad<0, 0, true, 1>), grid, dim3(256), 0, s, utp, p, kp, want_gc, pm);
            else if (fast == 2) hipLaunchKernelGGL((k_mgrad<0, 0, true, 2>), grid, dim3(256), 0, s, utp, p, kp, want_gc, pm);
            else hipLaunchKernelGGL((k_mgrad<0, 0, true, 0>), grid, dim3(256), 0, s, utp, p, kp, want_gc, pm);
        } else {
            if (fast == 1) hipLaunchKernelGGL((k_mgrad<0, 0, false, 1>), grid, dim3(256), 0, s, utp, p, kp, want_gc, pm);
            else if (fast == 2) hipLaunchKernelGGL((k_mgrad<0, 0, false, 2>), grid, dim3(256), 0, s, utp, p, kp, want_gc, pm);
            else hipLaunchKernelGGL((k_mgrad<0, 0, false, 0>), grid, dim3(256), 0, s, utp, p, kp, want_gc, pm);
        }
    } else {
        hipLaunchKernelGGL((k_mgrad<1, 1, false, 0>), grid, dim3(256), 0, s, utp, p, kp, want_gc, pm);
    }
    if (ut.max_T > SMALL_MAX_T) {
        // units of more than 1024 points: M by the LDS-staged GEMM into their K regions, then the reductions alone
        launch_big_mgemm(ut, p, kp.dy, s);
        if (dist_id == 0 && kern_id == 0) hipLaunchKernelGGL((k_mgrad<0, 0, false, 0, true>), grid, dim3(256), 0, s, utp, p, kp, want_gc, pm);
        else hipLaunchKernelGGL((k_mgrad<1, 1, false, 0, true>), grid, dim3(256), 0, s, utp, p, kp, want_gc, pm);
    }
}


void launch_assemble(const UnitTab &ut, const Pools &p, const AssembleTab &at, const KParams &kp, int n,
                     int want_gx, int want_gc, double *out, int usum_ok, const ObjTab &ob, hipStream_t s) {
    int blocks = 1 + (n + 31) / 32;
    hipLaunchKernelGGL(k_assemble, dim3(blocks), dim3(256), 0, s, ut, p, at, kp, n, want_gx, want_gc, out, usum_ok, ob);
}

}  // namespace gprf
