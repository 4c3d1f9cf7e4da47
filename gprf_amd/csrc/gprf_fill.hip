// gprf_fill.hip — the kernel-matrix fill (gprf.py:333-343 -> VectorTree.kernel_matrix + nv I): k_fill<lld, matern32>, k_fill_se.
// Off the north-star path (the register-resident Cholesky generates K itself there).
#include "gprf_dev.h"

namespace gprf {

// ------------------------------------------------------------------------------------------------
// gathers (gprf.py:300-302, 314-326: X[idxs], Y[idxs], vstack) into padded per-unit rows.  The coordinates are
// scattered once per evaluation from the point side (k_scatter_x, with the re-blocking kernels at the end of this
// file: a point writes its record into its row of every unit that contains its block; padding rows are zeroed when
// the tables are built); the outputs never move: the one kernel that needs a unit's Y rows (the forward
// substitution) reads them through upt from the resident n x dy array, which stays in L2 / the Infinity Cache.
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// K fill (gprf.py:333-343 -> VectorTree.kernel_matrix + nv I): the 64x64 blocks ti <= tj of the unit's row-major
// mp x mp matrix (by symmetry nobody reads the others: the Cholesky wants the upper triangle, k_mgrad reads a lower
// block's values transposed from the upper one); lane = column, so every wave-store is 512 contiguous bytes.
// Algorithmic bytes 8 mp^2 per unit, a little over half of them written.
// skip_T: units of at most skip_T tiles per edge are left alone (the register-resident Cholesky generates their kernel
// matrices itself; 0 = fill every unit)
// one workgroup per 64x64 block (ti <= tj), entry by entry through KernFn<DIST, KERN>::value: the fill of the ("lld","matern32")
// kernel, and the SE fill's reference form (GPRF_FILL_VARIANT=0; k_fill_se below is the one that runs)
template <int DIST, int KERN>
__global__ __launch_bounds__(256) void k_fill(UnitTab ut, Pools pl, KParams kp, int skip_T) {
    constexpr int XS = PtRec<DIST>::STRIDE, XN = PtRec<DIST>::NREG;
    __shared__ double xr[64 * XS];
    const UnitRef ur = unit_ref(ut.srec, blockIdx.y);
    int u = ur.u;
    int m = ur.m;
    int mp = pad16(m);
    if ((mp >> 4) <= skip_T) return;
    int nt = (mp + 63) >> 6;
    int pidx = blockIdx.x;
    if (pidx >= nt * (nt + 1) / 2) return;
    int ti = 0, rem = pidx;
    while (rem >= nt - ti) { rem -= nt - ti; ++ti; }
    int tj = ti + rem;
    int r0 = ti * 64, c0 = tj * 64;
    const double *Xu = pl.Xu + (size_t)ur.row_off * XS;
    int t = threadIdx.x;
#pragma unroll
    for (int e = t; e < 64 * XS; e += 256) {
        int rr = r0 + e / XS;
        xr[e] = (rr < mp) ? Xu[(size_t)r0 * XS + e] : 0.0;
    }
    int cl = t & 63;
    int col = c0 + cl;
    double xj[XN];
#pragma unroll
    for (int d = 0; d < XN; ++d) xj[d] = (col < mp) ? Xu[(size_t)col * XS + d] : 0.0;
    __syncthreads();
    double *U = pl.K + ur.mat_off;     // K pool: 64x64 tiles ti <= tj only (diagonal tiles whole)
    double diag_add = kp.nv + ut.jitter[u];
    int rbase = t >> 6;
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
        int rl = rbase + 4 * q;
        int row = r0 + rl;
        double v = 0.0;
        if (row < mp && col < mp) {
            if (row < m && col < m) {
                v = KernFn<DIST, KERN>::value(kp, &xr[rl * XS], xj);
                if (row == col) v = __dadd_rn(v, diag_add);
            } else {
                v = (row == col) ? 1.0 : 0.0;
            }
            U[(size_t)row * mp + col] = v;
        }
    }
}

// k_fill_se (round 4): the SE fill with its vector-ALU work halved.  The rocprofv3 SQ pass of k_fill<0,0> (profiles/
// r04_fill_rocprof_summary.txt) shows what bounds it: 1.155e7 VALU wave-instructions per launch for 1.05e7 values — 70 per
// value, 46 % of the wave cycles stalled on instruction dependencies, 24 % parked at waits, and only 84 MB written in 34 us
// (2.4 TB/s): the vector ALU, not HBM.  Of the 70, the exponential needs 24 and the distance 6; the rest was a run-time
// loop over the dimensions, three data-dependent branches per value and a 64-bit row * mp + col per store.  Here: blocks off
// the diagonal (ti < tj: all their rows are inside the unit, no entry is on the diagonal) evaluate sv * exp(-r^2) with ONE
// per-lane predicate (col < m) hoisted out; diagonal blocks select branch-free; the dimensions are unrolled (a uniform test
// for the third), the store address is a running pointer.  Same arithmetic per entry as KernFn<0,0>::value: the same bits.
__global__ __launch_bounds__(256) void k_fill_se(UnitTab ut, Pools pl, KParams kp, int skip_T) {
    __shared__ double xr[64 * XPAD];
    const UnitRef ur = unit_ref(ut.srec, blockIdx.y);
    int u = ur.u;
    int m = ur.m;
    int mp = pad16(m);
    if ((mp >> 4) <= skip_T) return;
    int nt = (mp + 63) >> 6;
    int pidx = blockIdx.x;
    if (pidx >= nt * (nt + 1) / 2) return;
    int ti = 0, rem = pidx;
    while (rem >= nt - ti) { rem -= nt - ti; ++ti; }
    int tj = ti + rem;
    int r0 = ti * 64, c0 = tj * 64;
    const double *Xu = pl.Xu + (size_t)ur.row_off * XPAD;
    int t = threadIdx.x;
    {
        int rr = r0 + (t >> 2);      // 256 threads = 64 rows x XPAD
        xr[t] = (rr < mp) ? Xu[(size_t)r0 * XPAD + t] : 0.0;
    }
    int cl = t & 63;
    int col = c0 + cl;
    const bool two_d = kp.dx <= 2;
    double xj0 = 0.0, xj1 = 0.0, xj2 = 0.0;
    if (col < mp) {
        xj0 = Xu[(size_t)col * XPAD];
        xj1 = Xu[(size_t)col * XPAD + 1];
        if (!two_d) xj2 = Xu[(size_t)col * XPAD + 2];
    }
    const double diag_add = kp.nv + ut.jitter[u];
    __syncthreads();
    if (col >= mp) return;
    const int rbase = t >> 6;
    double *dst = pl.K + ur.mat_off + (size_t)(r0 + rbase) * mp + col;      // rows r0 + rbase + 4 q: 4 mp apart
    const size_t rstep = (size_t)4 * mp;
    const bool colm = col < m;
    const double sv = kp.sv;
    const bool diag = ti == tj;
    const int nrow = mp - r0 < 64 ? mp - r0 : 64;      // rows of this block inside the padded unit (a multiple of 16)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        if (16 * h >= nrow) break;      // (uniform)
        double sq[4], e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rl = rbase + 4 * (4 * h + q);
            sq[q] = se_neg_r2(xr[rl * XPAD], xr[rl * XPAD + 1], two_d ? 0.0 : xr[rl * XPAD + 2], xj0, xj1, xj2, kp.inv_ls, !two_d);
        }
        exp_fast_v<4>(sq, e);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double v;
            if (!diag) {
                v = colm ? sv * e[q] : 0.0;
            } else {
                const int row = r0 + rbase + 4 * (4 * h + q);
                // (K = k(X, X) first, THEN + nv I, as the reference forms it (gprf.py:337-342): two roundings — no fused
                // multiply-add across the two steps; k_fill<0,0> and the register Cholesky's generation do the same)
                v = sv * e[q];
                v = (row == col) ? __dadd_rn(v, diag_add) : v;
                if (!(row < m && colm)) v = (row == col) ? 1.0 : 0.0;
            }
            *dst = v;
            dst += rstep;
        }
    }
}


void launch_fill(int dist_id, int kern_id, const UnitTab &ut, const Pools &p, const KParams &kp, int skip_T, hipStream_t s) {
    if (ut.n_ids == 0 || ut.max_T == 0 || ut.max_T <= skip_T) return;
    int nt = (16 * ut.max_T + 63) / 64;
    dim3 grid(nt * (nt + 1) / 2, ut.n_ids);
    // Round 4, measured on C3 forced through the pool (stage us; 117 MB algorithmic, 84 MB written): the entry-by-entry form
    // (k_fill<0,0>, gone since round 5) 35.9; one workgroup per 64-row strip walking its blocks (a third of the workgroups, one
    // round of them) 53.2; eight interleaved exp chains per thread instead of four 38.0; non-temporal stores 36.0; k_fill_se,
    // half the vector-ALU instructions per value, 26.2 = 4.5 TB/s algorithmic
    if (dist_id == 0 && kern_id == 0) hipLaunchKernelGGL(k_fill_se, grid, dim3(256), 0, s, ut, p, kp, skip_T);
    else hipLaunchKernelGGL((k_fill<1, 1>), grid, dim3(256), 0, s, ut, p, kp, skip_T);
}

}  // namespace gprf
