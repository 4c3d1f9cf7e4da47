// gprf_big.hip — units beyond one workgroup (513 .. 16384 points): the blocked Cholesky / forward substitution over whole
// launches and the LDS-staged MFMA GEMM k_big_gemm (trailing updates, the sweep, At by split-K, the gradient matrix M).
#include "gprf_dev.h"

namespace gprf {

// ------------------------------------------------------------------------------------------------
// Units of more than 1024 points (round 4).  The reference has no size limit (gprf.py:496-591 is LAPACK on whatever the
// partition gives) and its own experiment matrix uses such units: n = 10000 with 9 blocks or 1 ("the true GP"), n = 80000 with
// 16 / 36 blocks (gprfopt_analyze.py:195, 237-238).  One workgroup per unit cannot hold them; they go through the SAME
// pipeline in 64 x 64 blocks over whole launches, one launch per step and kind of work:
//   Cholesky, right-looking (K's upper blocks are first copied into the U pool):  per block row k
//     k_big_diag    U_kk = chol(C_kk) in LDS (one workgroup per unit), V_kk = U_kk^-1, log-det
//     k_big_apply   U_kj = V_kk^T C_kj                    (row panel, j > k)
//     k_big_update  C_ij -= U_ki^T U_kj                   (trailing blocks k < i <= j)
//   forward substitution U^T [W | Z] = [I | Y[rows]], right-looking in super-blocks S (round 5, second half):
//     k_big_wss_*   W_SS = U_SS^-T, the super-block's diagonal block of W, for EVERY super-block at once (the 64-row steps
//                   V_kk^T / U_ki^T restricted to the super-block's own columns: 2 sup - 1 launches in all)
//     k_big_gemm    mode 4:  [W_Sc | Z_S] = W_SS [R_Sc | R_S]   (c < S; R = the running right-hand side, in the K / At pools)
//                   mode 1:  [R_ic | R_i] -= U_Si^T [W_Sc | Z_S]  (i > S)
// every small product a 64 x 64 x 64 block product in the file's one MFMA form (no transposes: D += SA^T SB with SA, SB
// row-major and k the slow index), a step's products summed from zero and added once (the hierarchical accumulation of the
// small kernels, here for free); everything behind a super-block, At (mode 3) and the gradient matrix M (mode 2) by the
// LDS-staged GEMM k_big_gemm.
// ------------------------------------------------------------------------------------------------
constexpr int BIGB = 64;

struct BigUnit { int u, m, mp, nb; size_t mat_off; size_t row_off; };
// the unit of launch slot `slot` if it is a big one and has a block row kb
// min_T: BIG_LA_T for the Cholesky / substitution kernels, SMALL_MAX_T for what serves At and the gradient (modes 2, 3)
__device__ __forceinline__ bool big_unit(const UnitTab &ut, int slot, int kb, BigUnit *b, int min_T = BIG_LA_T) {
    const UnitRef ur = unit_ref(ut.srec, slot);
    b->u = ur.u; b->m = ur.m; b->mp = pad16(ur.m); b->mat_off = ur.mat_off; b->row_off = (size_t)ur.row_off;
    b->nb = (b->mp + BIGB - 1) / BIGB;
    return (b->mp >> 4) > min_T && kb < b->nb;
}
__device__ __forceinline__ int big_rows(const BigUnit &b, int blk) { int r = b.mp - BIGB * blk; return r < BIGB ? r : BIGB; }
// where a unit's V_kk blocks live in Pools::Vb
__device__ __forceinline__ double *big_vkk(const Pools &pl, const BigUnit &b, int kb) {
    return pl.Vb + (b.row_off + (size_t)BIGB * b.u) * BIGB + (size_t)kb * BIGB * BIGB;
}

// acc[jt] += sum_{k < kn} SA[k][lr] * SB[k][16 jt + lr-th column]  for this wave's 16 output rows: SA points at the wave's
// first column of the k x 64 operand (leading dimension lda), SB at the other operand's block (ldb); nj column tiles.
// kn <= 64, a multiple of 16.  Every operand value of the block product is requested BEFORE the first MFMA (80 loads in
// flight per lane): round 4's loop asked for a k-step's five values, waited, issued four MFMAs, sixteen times over — sixteen
// exposed L2 round trips per block product, and these kernels run between the launches of a 64-row step, where nothing hides them.
__device__ __forceinline__ void big_block_mma(const double *__restrict__ SA, int lda, const double *__restrict__ SB, int ldb, int kn,
                                              int nj, int lane, d4 (&acc)[4]) {
    int lr = lane & 15, lg = lane >> 4;
    const double *pa = SA + (size_t)lg * lda + lr;
    const double *pb = SB + (size_t)lg * ldb + lr;
    double a[16], bv[16][4];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const bool ok = 4 * s < kn;
        a[s] = ok ? pa[(size_t)(4 * s) * lda] : 0.0;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) bv[s][jt] = (ok && jt < nj) ? pb[(size_t)(4 * s) * ldb + 16 * jt] : 0.0;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[jt] = mfma(a[s], bv[s][jt], acc[jt]);
    }
}

// K's upper blocks -> U; W = identity (all of it); Z = Y[unit rows], zero padded.
// grid.x = nbmax * nbmax + nbmax: block (i, j) of the launch-wide block grid, then one workgroup per block row for Z.
// (the gathered outputs go to the At pool, where the substitution's sweep keeps its running right-hand side: launch_big_solve)
__global__ __launch_bounds__(256) void k_big_init(UnitTab ut, Pools pl, int nbmax, int dy) {
    BigUnit b;
    if (!big_unit(ut, blockIdx.y, 0, &b)) return;
    int x = blockIdx.x, t = threadIdx.x;
    const size_t mp = (size_t)b.mp;
    if (x >= nbmax * nbmax) {
        int bi = x - nbmax * nbmax;
        if (bi >= b.nb) return;
        double *Z = pl.At + b.row_off * YPAD;
        const int32_t *upt = ut.upt + b.row_off;
        for (int e = t; e < BIGB * YPAD; e += 256) {
            int row = BIGB * bi + (e >> 6), col = e & 63;
            if (row < b.mp) Z[(size_t)row * YPAD + col] = (row < b.m && col < dy) ? pl.Y[(size_t)upt[row] * dy + col] : 0.0;
        }
        return;
    }
    int i = x / nbmax, j = x - i * nbmax;
    if (i >= b.nb || j >= b.nb) return;
    const double *K = pl.K + b.mat_off;
    double *U = pl.U + b.mat_off, *W = pl.W + b.mat_off;
    for (int e = t; e < BIGB * BIGB; e += 256) {
        int row = BIGB * i + (e >> 6), col = BIGB * j + (e & 63);
        if (row < b.mp && col < b.mp) {
            if (j >= i) U[row * mp + col] = K[row * mp + col];
            // (W = I on EVERY block, the strictly-upper ones too: k_big_gemm walks W in 128-wide tiles that straddle the
            // diagonal, and what is above it must be zero, not what an earlier partition left in the pool)
            W[row * mp + col] = (row == col) ? 1.0 : 0.0;
        }
    }
}

// the diagonal block of block row kb: upper Cholesky and its inverse, in 16 x 16 tiles — the small kernels' arithmetic on a
// 4 x 4 tile grid in LDS.  Step j: wave 0 factors tile (j, j) (diag_factor16_ldl: the root-free pivot chain) and inverts it
// (the column operations of tile_inverse); the row panel U_jk = V_jj^T C_jk and the trailing tiles C_ik -= U_ji^T U_jk are
// four MFMAs each, dealt over the four waves.  The inverse V = U^-1 tile by tile: V_jk = -V_jj sum_{l = j+1..k} U_jl V_lk, by
// distance from the diagonal (three rounds).  ~15 us per block.  (Round 4's form — one pivot at a time over the whole block with
// three workgroup barriers each, the inverse by per-thread back substitution — took ~100 us, a sixth of the blocked
// Cholesky's time at n = 10000; two one-wave forms tried on the way, the block in LDS or a column per lane in registers with
// v_readlane multipliers, took 200 and 57 us.)
__global__ __launch_bounds__(256) void k_big_diag(UnitTab ut, Pools pl, int kb) {
    constexpr int LDA = BIGB + 16;           // = 16 mod 32 doubles: the k-major MFMA operand reads are conflict free
    __shared__ double A[BIGB * LDA];         // the block, row-major: becomes U_kk
    __shared__ double Vl[BIGB * LDA];        // V = U_kk^-1
    __shared__ double Vt[4 * 256];           // the diagonal tiles' inverses V_jj, row-major
    __shared__ double piv[BIGB];             // U's diagonal
    __shared__ int s_bad;
    BigUnit b;
    if (!big_unit(ut, blockIdx.x, kb, &b)) return;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int n = big_rows(b, kb);
    const size_t mp = (size_t)b.mp;
    double *U = pl.U + b.mat_off + ((size_t)BIGB * kb) * mp + (size_t)BIGB * kb;
    for (int e = t; e < BIGB * BIGB; e += 256) {
        const int i = e >> 6, j = e & 63;
        A[i * LDA + j] = (i < n && j < n) ? U[(size_t)i * mp + j] : ((i == j) ? 1.0 : 0.0);      // identity padding
        Vl[i * LDA + j] = 0.0;
    }
    if (t == 0) s_bad = 0;
    __syncthreads();
    for (int j = 0; j < 4; ++j) {
        if (wave == 0) {
            double s16[16], dk, rdk;
#pragma unroll
            for (int r = 0; r < 16; ++r) s16[r] = A[(16 * j + r) * LDA + 16 * j + lr];
            const int bad = diag_factor16_ldl<NoEarly, false>(s16, lr, &dk, &rdk, nullptr);
            if (lane < 16) {
#pragma unroll
                for (int i = 0; i < 16; ++i) A[(16 * j + i) * LDA + 16 * j + lr] = (lr >= i) ? s16[i] : 0.0;
                piv[16 * j + lr] = dk;
                if (bad && lane == 0 && s_bad == 0) s_bad = BIGB * kb + 16 * j + bad;
            }
            double v[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                int lrc = lr;
                asm volatile("" : "+v"(lrc));
                v[c] = (c == lrc) ? 1.0 : 0.0;
            }
            dpp_src_ready(rdk);
            static_for<0, 16>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                v[k] *= bcast16<k>(rdk);
                dpp_src_ready(s16[k]);
                static_for<k + 1, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    fnma_bcast16<i>(v[i], s16[k], v[k]);
                });
            });
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    Vt[j * 256 + lr * 16 + c] = v[c];
                    Vl[(16 * j + lr) * LDA + 16 * j + c] = v[c];
                }
            }
        }
        __syncthreads();
        {   // row panel: U_jk = V_jj^T C_jk
            const int k = j + 1 + wave;
            if (k < 4) {
                d4 r = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) r = mfma(Vt[j * 256 + (4 * q + lg) * 16 + lr], A[(16 * j + 4 * q + lg) * LDA + 16 * k + lr], r);
#pragma unroll
                for (int q = 0; q < 4; ++q) A[(16 * j + lg + 4 * q) * LDA + 16 * k + lr] = r[q];
            }
        }
        __syncthreads();
        {   // trailing tiles (i, k), j < i <= k: a step's products from zero, then one subtraction
            int e = 0;
            for (int i = j + 1; i < 4; ++i)
                for (int k = i; k < 4; ++k, ++e)
                    if ((e & 3) == wave) {
                        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc = mfma(A[(16 * j + 4 * q + lg) * LDA + 16 * i + lr], A[(16 * j + 4 * q + lg) * LDA + 16 * k + lr], acc);
#pragma unroll
                        for (int q = 0; q < 4; ++q) A[(16 * i + lg + 4 * q) * LDA + 16 * k + lr] -= acc[q];
                    }
        }
        __syncthreads();
    }
    // V's off-diagonal tiles by distance d from the diagonal: V_jk = -V_jj (sum_l U_jl V_lk), l = j+1 .. k
    for (int d = 1; d < 4; ++d) {
        const int jj = wave, kk = wave + d;
        if (kk < 4) {
            d4 T = {0.0, 0.0, 0.0, 0.0};
            for (int l = jj + 1; l <= kk; ++l) {
#pragma unroll
                for (int q = 0; q < 4; ++q)      // SA^T = U_jl: a transposed read
                    T = mfma(A[(16 * jj + lr) * LDA + 16 * l + 4 * q + lg], Vl[(16 * l + 4 * q + lg) * LDA + 16 * kk + lr], T);
            }
            d4 R = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) R = mfma(Vt[jj * 256 + lr * 16 + 4 * q + lg], T[q], R);
#pragma unroll
            for (int q = 0; q < 4; ++q) Vl[(16 * jj + lg + 4 * q) * LDA + 16 * kk + lr] = -R[q];
        }
        __syncthreads();
    }
    double *Vk = big_vkk(pl, b, kb);
    for (int e = t; e < BIGB * BIGB; e += 256) {
        const int i = e >> 6, j = e & 63;
        const bool in = i < n && j < n;
        if (in) U[(size_t)i * mp + j] = (j >= i) ? A[i * LDA + j] : 0.0;      // (zeros below the diagonal)
        Vk[e] = (in && j >= i) ? Vl[i * LDA + j] : 0.0;
    }
    if (t < 64) {
        double lg2 = (t < n) ? log(piv[t]) : 0.0;
        for (int off = 32; off >= 1; off >>= 1) lg2 += shfl_xor_d(lg2, off);
        if (t == 0) {
            pl.logdet[b.u] = (kb == 0 ? 0.0 : pl.logdet[b.u]) + 2.0 * lg2;
            if (kb == 0) pl.info[b.u] = 0;
            if (s_bad && pl.info[b.u] == 0) pl.info[b.u] = s_bad;
        }
    }
}

// B <- V_kk^T B for one block (n x 16 nj, leading dimension ldb) of block row kb; the whole workgroup
__device__ __forceinline__ void big_apply_block(const Pools &pl, const BigUnit &b, int kb, double *B, int ldb, int nj) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int n = big_rows(b, kb);
    const double *Vk = big_vkk(pl, b, kb);
    d4 acc[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc[jt] = d4{0.0, 0.0, 0.0, 0.0};
    const bool active = 16 * wave < n;
    // V_kk is upper triangular: rows k beyond this strip's last column contribute nothing
    if (active) big_block_mma(Vk + 16 * wave, BIGB, B, ldb, 16 * (wave + 1) < n ? 16 * (wave + 1) : n, nj, lane, acc);
    __syncthreads();      // every wave has read the whole block before anybody overwrites a row of it
    if (active) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
            if (jt < nj)
#pragma unroll
                for (int q = 0; q < 4; ++q) B[(size_t)(16 * wave + lg + 4 * q) * ldb + 16 * jt + lr] = acc[jt][q];
    }
}
// C -= U_ki^T SB for one block (rows of block i x 16 nj): the whole workgroup
__device__ __forceinline__ void big_update_block(const BigUnit &b, const double *Uk, int kn, int i, double *C, const double *SB, int ldc, int nj) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    if (16 * wave >= big_rows(b, i)) return;
    d4 acc[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc[jt] = d4{0.0, 0.0, 0.0, 0.0};
    // (the 16 values of C requested together, in front of the block product's own loads: element by element, "*cp = *cp - acc"
    // is load, wait, store sixteen times over)
    double cv[4][4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) cv[jt][q] = C[(size_t)(16 * wave + lg + 4 * q) * ldc + (jt < nj ? 16 * jt + lr : lr)];
    big_block_mma(Uk + (size_t)BIGB * i + 16 * wave, b.mp, SB, ldc, kn, nj, lane, acc);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
        if (jt < nj)
#pragma unroll
            for (int q = 0; q < 4; ++q) C[(size_t)(16 * wave + lg + 4 * q) * ldc + 16 * jt + lr] = cv[jt][q] - acc[jt][q];
}

// W_SS = U_SS^-T, the diagonal super-block of W, of EVERY super-block of every unit at once (blockIdx.z = super-block): the
// substitution's 64-row steps restricted to the super-block's own columns — step t of `sup`: block row kb = z sup + t.
// (Round 5, second half: the sweep over the super-blocks then needs no 64-row steps at all — a super-block's rows are ONE product
// with W_SS, k_big_gemm mode 4 — and these 2 sup - 1 launches are made once, not once per super-block.)
__global__ __launch_bounds__(256) void k_big_wss_apply(UnitTab ut, Pools pl, int t, int sup) {
    const int kb = (int)blockIdx.z * sup + t;
    BigUnit b;
    if (!big_unit(ut, blockIdx.y, kb, &b)) return;
    const int c = (int)blockIdx.z * sup + (int)blockIdx.x;      // (blockIdx.x = 0 .. t)
    double *B = pl.W + b.mat_off + ((size_t)BIGB * kb) * b.mp + (size_t)BIGB * c;
    big_apply_block(pl, b, kb, B, b.mp, big_rows(b, c) >> 4);
}
__global__ __launch_bounds__(256) void k_big_wss_update(UnitTab ut, Pools pl, int t, int sup) {
    const int z0 = (int)blockIdx.z * sup, kb = z0 + t;
    BigUnit b;
    if (!big_unit(ut, blockIdx.y, kb, &b)) return;
    const int i = kb + 1 + (int)blockIdx.x / (t + 1), c = z0 + (int)blockIdx.x % (t + 1);      // rows behind kb inside the super-block
    if (i >= b.nb || i >= z0 + sup) return;
    const size_t mp = (size_t)b.mp;
    const double *Uk = pl.U + b.mat_off + ((size_t)BIGB * kb) * mp;
    double *C = pl.W + b.mat_off + ((size_t)BIGB * i) * mp + (size_t)BIGB * c;
    const double *SB = pl.W + b.mat_off + ((size_t)BIGB * kb) * mp + (size_t)BIGB * c;
    big_update_block(b, Uk, big_rows(b, kb), i, C, SB, b.mp, 4);
}

// The Cholesky's row panel inside a super-block: U_kj = V_kk^T C_kj in place, blocks j = kb + 1 .. of block row kb.
__global__ __launch_bounds__(256) void k_big_apply(UnitTab ut, Pools pl, int kb) {
    BigUnit b;
    if (!big_unit(ut, blockIdx.y, kb, &b)) return;
    const int j = kb + 1 + blockIdx.x;
    if (j >= b.nb) return;
    double *B = pl.U + b.mat_off + ((size_t)BIGB * kb) * b.mp + (size_t)BIGB * j;
    big_apply_block(pl, b, kb, B, b.mp, big_rows(b, j) >> 4);
}

// The Cholesky's trailing blocks (i, j), kb < i <= j, i < i_end: U_ij -= U_ki^T U_kj.  blockIdx.x enumerates the launch-wide block
// grid (r = nbmax - kb - 1 rows behind kb).  i_end: only the block rows INSIDE the current super-block; everything behind it takes
// the super-block's whole contribution at once (k_big_gemm)
__global__ __launch_bounds__(256) void k_big_update(UnitTab ut, Pools pl, int kb, int nbmax, int i_end) {
    BigUnit b;
    if (!big_unit(ut, blockIdx.y, kb, &b)) return;
    const size_t mp = (size_t)b.mp;
    const double *Uk = pl.U + b.mat_off + ((size_t)BIGB * kb) * mp;      // block row kb of U
    const int r = nbmax - kb - 1;
    int x = blockIdx.x, di = 0;
    while (x >= r - di) { x -= r - di; ++di; }      // row di of the upper block triangle, x columns in
    const int i = kb + 1 + di, j = i + x;
    if (j >= b.nb || i >= i_end) return;
    double *C = pl.U + b.mat_off + ((size_t)BIGB * i) * mp + (size_t)BIGB * j;
    big_update_block(b, Uk, big_rows(b, kb), i, C, Uk + (size_t)BIGB * j, b.mp, big_rows(b, j) >> 4);
}

// ||Z[:, 16 cb : 16 cb + 16]||_F^2 per column block (the small kernels' zzpart), fixed order: BIG_ZZ_PARTS workgroups per unit sum
// a row range each into a scratch slot (the unit's region of rowpart, which the gradient kernel overwrites later); a second,
// one-wave launch folds the slots in slot order (round 4's single workgroup per unit walked the 10000-point unit's 5 MB
// alone: 0.86 ms)
constexpr int BIG_ZZ_PARTS = 64;
__global__ __launch_bounds__(256) void k_big_zz(UnitTab ut, Pools pl, int tbs) {
    __shared__ double red[256];
    BigUnit b;
    if (!big_unit(ut, blockIdx.y, 0, &b)) return;
    const int t = threadIdx.x, col = t & 63, r0 = t >> 6;
    const double *Z = pl.Z + b.row_off * YPAD;
    const int per = ((b.mp + BIG_ZZ_PARTS - 1) / BIG_ZZ_PARTS + 3) & ~3;
    const int lo = per * (int)blockIdx.x, hi = lo + per < b.mp ? lo + per : b.mp;
    double s = 0.0;
    for (int row = lo + r0; row < hi; row += 4) {
        double z = Z[(size_t)row * YPAD + col];
        s = fma(z, z, s);
    }
    red[t] = s;
    __syncthreads();
    if (t < 64) red[t] = (red[t] + red[t + 64]) + (red[t + 128] + red[t + 192]);
    __syncthreads();
    if (t < 4) {
        double v = 0.0;
        for (int k = 0; k < 16; ++k) v += red[16 * t + k];
        pl.rowpart[b.row_off * (size_t)tbs * XPAD + 4 * blockIdx.x + t] = v;
    }
}
__global__ __launch_bounds__(64) void k_big_zz_fold(UnitTab ut, Pools pl, int tbs) {
    BigUnit b;
    if (!big_unit(ut, blockIdx.x, 0, &b)) return;
    const int t = threadIdx.x;
    if (t < 4) {
        const double *part = pl.rowpart + b.row_off * (size_t)tbs * XPAD;
        double v = 0.0;
        for (int k = 0; k < BIG_ZZ_PARTS; ++k) v += part[4 * k + t];
        pl.zzpart[(size_t)b.u * 4 + t] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// k_big_gemm (round 5): the blocked path's trailing updates as an LDS-staged MFMA GEMM.
// A right-looking factorisation in 64-row steps updates the whole trailing matrix once per step with K = 64: one read and one
// write of every trailing entry per 64 rows — at n = 10000 that is 125 GB through L2 / HBM for 3.3e11 flop, and round 4's
// blocked path ran at 8-12 TFLOP/s because of it.  Now the 64-row steps only run INSIDE a super-block of BG_SUPER = 4 block
// rows (k_big_diag / k_big_apply / k_big_update with i_end); what lies behind the super-block takes its 256 rows' contribution
// in ONE pass,  C -= A^T B  with A, B the super-block's rows of U (or W / Z) — the file's one MFMA form, D += SA^T SB with k the
// slow index, so nothing is transposed:
//   * one workgroup = one 128 x 128 tile of C, four waves of 64 x 64 (16 accumulator tiles each);
//   * the operands' 8-row chunks [8 x 128 | 8 x 128] go global -> registers -> LDS (pitch 144: the k-major operand reads are
//     conflict free), two register sets and two LDS buffers deep, one LDS-only barrier per chunk; a wave reads 4 + 4 operand
//     values per k-step of 16 MFMAs;
//   * a tile's products are summed from zero and enter C with one subtraction (the hierarchical accumulation of the small
//     kernels).
// mode 0: the Cholesky's trailing tiles (upper triangle behind the super-block, 128-tile (di, dj), dj >= di);
// mode 1: the substitution's rows behind the super-block, [W columns up to the super-block's end | Z];
// mode 2: M = At^T At - dy W^T W on the lower triangle of 128-tiles, WRITTEN to the unit's region of the K pool (nobody reads
//         K there any more) for k_mgrad<.., BIG> to reduce — the gradient kernel's own 64 x 64 block pairs re-read W at 8 flop
//         per byte: 49 of the 120 ms of the 10000-point unit.
// ------------------------------------------------------------------------------------------------
constexpr int BG_LD = 144, BG_KC = 8, BG_SUPER = 4;      // (BGT = 128, the tile edge: gprf_dev.h)
constexpr int BG_ATSEG = 512;      // rows of [Z | W] per partial product of At (mode 3)

// a_trans: the A operand is given transposed — element (k, i) at A[i * lda + k] (mode 4: W_SS read through its transpose)
struct BgOp { const double *A, *B; int lda, ldb, K; double scale; bool a_trans = false; };

// acc[ii][jj] += scale * sum_k A[k][64 wr + 16 ii + .] B[k][64 wc + 16 jj + .]   (wr, wc = this wave's quadrant)
__device__ __forceinline__ void bg_accumulate(const BgOp &op, int a_ext, int b_ext, double *sm, d4 (&acc)[4][4], bool compute) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    // staging roles: waves 0 / 1 the A chunk's columns 0..63 / 64..127, waves 2 / 3 the B chunk's
    const bool isB = wave >= 2;
    const int col = 64 * (wave & 1) + lane;
    const bool col_ok = col < (isB ? b_ext : a_ext);
    // (branch-free: a lane beyond the operand's edge re-reads column 0 — its values only reach accumulator rows / columns
    // that are never stored — and K is a multiple of BG_KC on every path (multiples of 16).  Written with a select per value,
    // "col_ok && row < K ? load : 0", the compiler fenced every pair of loads with exec branches and s_waitcnt vmcnt(0):
    // four serialised round trips per chunk, the loop ran on the latency of its own prefetch)
    const int colc = col_ok ? col : 0;
    const bool tr = !isB && op.a_trans;
    const double *src0 = isB ? op.B + colc : (tr ? op.A + (size_t)colc * op.lda : op.A + colc);
    const size_t ld = tr ? (size_t)1 : (size_t)(isB ? op.ldb : op.lda);      // distance between consecutive k
    const int nch = op.K / BG_KC;
    double pre0[BG_KC], pre1[BG_KC];
    auto fetch = [&](int c, double (&pre)[BG_KC]) {
        const double *src = src0 + (size_t)(BG_KC * c) * ld;
#pragma unroll
        for (int e = 0; e < BG_KC; ++e) pre[e] = src[(size_t)e * ld];
    };
    auto step = [&](int c, double (&pre)[BG_KC]) {
        double *buf = sm + (c & 1) * (2 * BG_KC * BG_LD);
        double *dst = buf + (isB ? BG_KC * BG_LD : 0) + col;
#pragma unroll
        for (int e = 0; e < BG_KC; ++e) dst[e * BG_LD] = pre[e];
        lds_barrier();
        if (c + 2 < nch) fetch(c + 2, pre);
        if (compute) {
            const double *pa = buf + lg * BG_LD + 64 * (wave >> 1) + lr;
            const double *pb = buf + BG_KC * BG_LD + lg * BG_LD + 64 * (wave & 1) + lr;
#pragma unroll
            for (int s = 0; s < BG_KC / 4; ++s) {
                double a[4], bb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[q] = pa[(4 * s) * BG_LD + 16 * q] * op.scale;
                    bb[q] = pb[(4 * s) * BG_LD + 16 * q];
                }
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) acc[ii][jj] = mfma(a[ii], bb[jj], acc[ii][jj]);
            }
        }
    };
    if (nch > 0) fetch(0, pre0);
    if (nch > 1) fetch(1, pre1);
    for (int c = 0; c < nch; c += 2) {
        step(c, pre0);
        if (c + 1 < nch) step(c + 1, pre1);
    }
    lds_barrier();      // (a second call reuses the buffers)
}

__global__ __launch_bounds__(256, 2) void k_big_gemm(UnitTab ut, Pools pl, int mode, int sb0, int sb1, int ntmax, double dy) {
    __shared__ double sm[2 * 2 * BG_KC * BG_LD];
    BigUnit b;
    if (!big_unit(ut, blockIdx.y, 0, &b, (mode == 2 || mode == 3) ? SMALL_MAX_T : BIG_LA_T)) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int mp = b.mp;
    const size_t mps = (size_t)mp;
    const int r0 = BIGB * sb1, k0 = BIGB * sb0;
    double *U = pl.U + b.mat_off, *W = pl.W + b.mat_off;
    int i0, j0, ldc, a_ext, b_ext;
    double *C;
    bool skip = false;      // this wave's quadrant is not wanted
    int store = mode >= 2 ? 1 : 0;      // how the tile enters C: 0 = C -= acc, 1 = C = acc, 2 = C = -acc
    d4 acc[4][4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[ii][jj] = d4{0.0, 0.0, 0.0, 0.0};
    int x = blockIdx.x;
    if (mode == 0) {
        if (r0 >= mp) return;
        const int nt = (mp - r0 + BGT - 1) / BGT;
        int di = 0;
        while (x >= ntmax - di) { x -= ntmax - di; ++di; }
        const int dj = di + x;
        if (dj >= nt) return;
        i0 = r0 + BGT * di; j0 = r0 + BGT * dj;
        a_ext = mp - i0 < BGT ? mp - i0 : BGT; b_ext = mp - j0 < BGT ? mp - j0 : BGT;
        skip = di == dj && (wave >> 1) == 1 && (wave & 1) == 0;      // below the diagonal
        BgOp op{U + (size_t)k0 * mps + i0, U + (size_t)k0 * mps + j0, mp, mp, r0 - k0, 1.0};
        bg_accumulate(op, a_ext, b_ext, sm, acc, !skip);
        C = U + (size_t)i0 * mps + j0; ldc = mp;
    } else if (mode == 1) {
        if (r0 >= mp) return;
        const int nt = (mp - r0 + BGT - 1) / BGT, ncol = sb1 / 2 + 1;
        const int di = x / ncol, c = x - di * ncol;
        if (di >= nt) return;
        i0 = r0 + BGT * di;
        a_ext = mp - i0 < BGT ? mp - i0 : BGT;
        // (round 5, second half: the RUNNING right-hand sides R live outside the result pools — the W columns' in the unit's
        // region of the K pool, which the factorisation has left, the Z columns' in the At pool, which nobody needs before the
        // substitution is over — because a super-block's rows are now solved by ONE product with its inverse diagonal block
        // (mode 4), which cannot run in place.  A tile of R whose columns belong to THIS super-block has no earlier term: it
        // is written, not updated — nobody has to zero 800 MB first.)
        BgOp op{U + (size_t)k0 * mps + i0, nullptr, mp, mp, r0 - k0, 1.0};
        if (c < ncol - 1) {
            j0 = BGT * c; b_ext = BGT;
            op.B = W + (size_t)k0 * mps + j0;
            C = pl.K + b.mat_off + (size_t)i0 * mps + j0; ldc = mp;
            if (j0 >= k0) store = 2;
        } else {
            j0 = 0; b_ext = YPAD;
            op.B = pl.Z + (b.row_off + (size_t)k0) * YPAD; op.ldb = YPAD;
            C = pl.At + (b.row_off + (size_t)i0) * YPAD; ldc = YPAD;
            skip = (wave & 1) == 1;
        }
        bg_accumulate(op, a_ext, b_ext, sm, acc, !skip);
    } else if (mode == 4) {
        // the super-block's own rows of [W | Z]:  X_S = W_SS R_S  with W_SS = U_SS^-T, the super-block's diagonal block of W
        // (k_big_wss_*: every super-block's at once, before the sweep) — row tile ra of the super-block (128 rows) x column tile
        // c of the columns in front of it (c = ncol: the Z columns).  W_SS is lower triangular: K = the rows up to this tile's last.
        if (k0 >= mp) return;
        const int rows = (r0 < mp ? r0 : mp) - k0;                 // of this unit's super-block
        const int ncol = k0 / BGT, nrt = (BIGB * (sb1 - sb0) + BGT - 1) / BGT;
        const int ra = x % nrt, c = x / nrt;
        if (BGT * ra >= rows || c > ncol) return;
        i0 = k0 + BGT * ra;
        a_ext = rows - BGT * ra < BGT ? rows - BGT * ra : BGT;
        BgOp op{W + (size_t)i0 * mps + k0, nullptr, mp, mp, BGT * ra + a_ext, 1.0, true};
        if (c < ncol) {
            j0 = BGT * c; b_ext = BGT;
            op.B = pl.K + b.mat_off + (size_t)k0 * mps + j0;
            C = W + (size_t)i0 * mps + j0; ldc = mp;
        } else {
            j0 = 0; b_ext = YPAD;
            op.B = pl.At + (b.row_off + (size_t)k0) * YPAD; op.ldb = YPAD;
            C = pl.Z + (b.row_off + (size_t)i0) * YPAD; ldc = YPAD;
            skip = (wave & 1) == 1;
        }
        bg_accumulate(op, a_ext, b_ext, sm, acc, !skip);
    } else if (mode == 3) {
        // At = Z^T W (64 x mp), split over the rows: column tile tj of At, segment sg of BG_ATSEG rows of [Z | W] from the tile's
        // first row on (W is lower triangular: nothing above) — a partial product per (tile, segment) into the unit's region of
        // the K pool (free between the substitution and mode 2), slab sg = rows [64 sg, 64 sg + 64) x mp; k_big_at_fold adds the
        // slabs in segment order.  Only the tile's upper half (64 rows of At) exists: waves 2 and 3 stage and do not compute.
        const int nsegmax = (BGT * ntmax + BG_ATSEG - 1) / BG_ATSEG;
        const int tj = x / nsegmax, sg = x - tj * nsegmax;
        j0 = BGT * tj;
        const int k_lo = j0 + BG_ATSEG * sg;
        if (j0 >= mp || k_lo >= mp) return;
        i0 = 0;
        a_ext = YPAD; b_ext = mp - j0 < BGT ? mp - j0 : BGT;
        skip = (wave >> 1) == 1;
        BgOp op{pl.Z + (b.row_off + (size_t)k_lo) * YPAD, W + (size_t)k_lo * mps + j0, YPAD, mp, mp - k_lo < BG_ATSEG ? mp - k_lo : BG_ATSEG, 1.0};
        bg_accumulate(op, a_ext, b_ext, sm, acc, !skip);
        C = pl.K + b.mat_off + (size_t)sg * YPAD * mps + j0; ldc = mp;
    } else {
        const int nt = (mp + BGT - 1) / BGT;
        int tj = 0;
        while (x >= ntmax - tj) { x -= ntmax - tj; ++tj; }
        const int ti = tj + x;      // I >= J
        if (ti >= nt) return;
        i0 = BGT * ti; j0 = BGT * tj;
        a_ext = mp - i0 < BGT ? mp - i0 : BGT; b_ext = mp - j0 < BGT ? mp - j0 : BGT;
        skip = ti == tj && (wave >> 1) == 0 && (wave & 1) == 1;      // above the diagonal
        const double *At = pl.At + b.row_off * YPAD;
        BgOp opa{At + i0, At + j0, mp, mp, YPAD, 1.0};
        bg_accumulate(opa, a_ext, b_ext, sm, acc, !skip);
        // W is lower triangular: its columns of tile I are zero above row i0
        BgOp opw{W + (size_t)i0 * mps + i0, W + (size_t)i0 * mps + j0, mp, mp, mp - i0, -dy};
        bg_accumulate(opw, a_ext, b_ext, sm, acc, !skip);
        C = pl.K + b.mat_off + (size_t)i0 * mps + j0; ldc = mp;
    }
    if (skip) return;
    const int rbase = 64 * (wave >> 1), cbase = 64 * (wave & 1);
    // C -= acc, one row of four tiles at a time: its 16 loads all in flight, then 16 stores (an element-wise "*cp = *cp - acc"
    // compiles to load, s_waitcnt vmcnt(0), store, 64 times over: 64 exposed memory round trips per lane and tile); lanes
    // beyond the tile's edge read the tile's first element and store nothing
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        double cv[4][4];
        if (store == 0) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int cc = cbase + 16 * jj + lr;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rr = rbase + 16 * ii + lg + 4 * q;
                    const bool ok = cc < b_ext && rr < a_ext;
                    cv[jj][q] = C[ok ? (size_t)rr * ldc + cc : (size_t)0];
                }
            }
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int cc = cbase + 16 * jj + lr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = rbase + 16 * ii + lg + 4 * q;
                if (cc < b_ext && rr < a_ext)
                    C[(size_t)rr * ldc + cc] = store == 1 ? acc[ii][jj][q] : (store == 2 ? -acc[ii][jj][q] : cv[jj][q] - acc[ii][jj][q]);
            }
        }
    }
}

// (Round 5 also ran the forward substitution BESIDE the Cholesky — super-block S of the substitution needs U's rows of S and
// nothing behind them: a second queue, one event per super-block.  One block of 10000 points: 50.5 ms against 40.0 one after the
// other — the Cholesky's small, latency-critical launches (k_big_diag needs 90 KB of LDS) then wait for a CU to drain behind
// the other queue's GEMM workgroups.  Removed.)
// At[i][j] = sum over the segments of column tile j / 128 of the partial products k_big_gemm (mode 3) left in the K pool, in
// segment order (fixed: the result does not depend on the launch).  grid = (64 rows x column chunks of 256, launch slots)
__global__ __launch_bounds__(256) void k_big_at_fold(UnitTab ut, Pools pl) {
    BigUnit b;
    if (!big_unit(ut, blockIdx.y, 0, &b, SMALL_MAX_T)) return;
    const int i = (int)blockIdx.x & (YPAD - 1), j = 256 * ((int)blockIdx.x >> 6) + (int)threadIdx.x;
    if (j >= b.mp) return;
    const size_t mps = (size_t)b.mp;
    const int j0 = j & ~(BGT - 1);
    const int nseg = (b.mp - j0 + BG_ATSEG - 1) / BG_ATSEG;
    const double *slab = pl.K + b.mat_off + (size_t)i * mps + j;
    double v = 0.0;
    for (int s0 = 0; s0 < nseg; s0 += 8) {      // (eight slabs in flight)
        double t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = slab[(size_t)(s0 + q < nseg ? s0 + q : s0) * YPAD * mps];
#pragma unroll
        for (int q = 0; q < 8; ++q) v += s0 + q < nseg ? t[q] : 0.0;
    }
    pl.At[b.row_off * YPAD + (size_t)i * mps + j] = v;
}
// At = Z^T W of the units of more than 1024 points: split-K partial products by the GEMM kernel, then the fold
// block rows per super-block: 4 (256 rows), 8 beyond 4096 points — a GEMM pass has a fixed cost per tile (first fetch, the
// read-modify-write of C: ~20 % of a K = 256 pass), the 64-row steps inside a super-block grow with its square: one block of
// 10000 points 31.4 / 30.6 / 30.4 / 30.5 ms at 4 / 6 / 8 / 12, 9 blocks + 20 pairs 10.62 / 10.64 / 10.72 / 11.21 (diag big_super=<n>)
static int big_super(int max_T) { static const int v = diag("big_super", 0); return v > 0 ? (v + 1) & ~1 : (max_T > 256 ? 2 * BG_SUPER : BG_SUPER); }      // (even: a super-block starts on a 128-column tile)
void launch_big_at(const UnitTab &ut, const Pools &p, hipStream_t s) {
    if (ut.n_ids == 0 || ut.max_T <= SMALL_MAX_T) return;
    const int nt = (16 * ut.max_T + BGT - 1) / BGT, nsegmax = (BGT * nt + BG_ATSEG - 1) / BG_ATSEG;
    hipLaunchKernelGGL(k_big_gemm, dim3(nt * nsegmax, ut.n_ids), dim3(256), 0, s, ut, p, 3, 0, 0, nt, 0.0);
    hipLaunchKernelGGL(k_big_at_fold, dim3(YPAD * ((16 * ut.max_T + 255) / 256), ut.n_ids), dim3(256), 0, s, ut, p);
}

void launch_big_mgemm(const UnitTab &ut, const Pools &p, int dy, hipStream_t s) {
    if (ut.n_ids == 0 || ut.max_T <= SMALL_MAX_T) return;
    const int nt = (16 * ut.max_T + BGT - 1) / BGT;
    hipLaunchKernelGGL(k_big_gemm, dim3(nt * (nt + 1) / 2, ut.n_ids), dim3(256), 0, s, ut, p, 2, 0, 0, nt, (double)dy);
}

void launch_big_potrf(const UnitTab &ut, const Pools &p, const KParams &kp, hipStream_t s) {
    if (ut.n_ids == 0 || ut.max_T <= BIG_LA_T) return;
    const int nbmax = (16 * ut.max_T + BIGB - 1) / BIGB;
    dim3 blk(256);
    hipLaunchKernelGGL(k_big_init, dim3(nbmax * nbmax + nbmax, ut.n_ids), blk, 0, s, ut, p, nbmax, kp.dy);
    for (int sb0 = 0; sb0 < nbmax; sb0 += big_super(ut.max_T)) {
        const int sb1 = sb0 + big_super(ut.max_T) < nbmax ? sb0 + big_super(ut.max_T) : nbmax;
        for (int kb = sb0; kb < sb1; ++kb) {
            hipLaunchKernelGGL(k_big_diag, dim3(ut.n_ids), blk, 0, s, ut, p, kb);
            const int r = nbmax - kb - 1, rin = sb1 - kb - 1;
            if (r > 0) hipLaunchKernelGGL(k_big_apply, dim3(r, ut.n_ids), blk, 0, s, ut, p, kb);
            // the rows inside the super-block: the first rin rows of the upper block triangle behind kb
            if (rin > 0) hipLaunchKernelGGL(k_big_update, dim3(rin * r - rin * (rin - 1) / 2, ut.n_ids), blk, 0, s, ut, p, kb, nbmax, sb1);
        }
        if (sb1 < nbmax) {
            const int nt = (BIGB * (nbmax - sb1) + BGT - 1) / BGT;
            hipLaunchKernelGGL(k_big_gemm, dim3(nt * (nt + 1) / 2, ut.n_ids), blk, 0, s, ut, p, 0, sb0, sb1, nt, 0.0);
        }
    }
}

void launch_big_solve(const UnitTab &ut, const Pools &p, hipStream_t s) {
    if (ut.n_ids == 0 || ut.max_T <= BIG_LA_T) return;
    const int nbmax = (16 * ut.max_T + BIGB - 1) / BIGB;
    dim3 blk(256);
    // The sweep's super-blocks (block rows of 64; even): deeper than the factorisation's where the units are large — nothing
    // inside a super-block costs launches here.  Substitution of ONE block of 10000 points / 9 blocks + 20 pairs / one block of
    // 4000, ms: 12.8 / 3.75 / 2.17 at 2, 10.6 / 3.35 / 1.80 at 4, 9.85 / 3.27 / 1.73 at 6, 9.6 / 3.5 / 1.63 at 8, 9.25 / 3.43 /
    // 1.71 at 12, 9.7 / 3.8 / 1.58 at 16 (diag big_super_solve=<n>); with the 64-row steps inside every super-block (the round's
    // first half, super-blocks as the factorisation's): 10.34 / 3.54 / 1.72.
    static const int sup_diag = diag("big_super_solve", 0);
    const int sup = sup_diag > 0 ? (sup_diag + 1) & ~1 : (ut.max_T > 512 ? 12 : (ut.max_T > 192 ? 8 : 6));
    {
        // W_SS = U_SS^-T of every super-block at once (2 sup - 1 small launches in all), then the sweep: per super-block ONE
        // product X_S = W_SS R_S (mode 4; R in the K pool, the Z columns' in the At pool) and the update of everything behind it
        const int nsb = (nbmax + sup - 1) / sup;
        for (int t = 0; t < sup; ++t) {
            hipLaunchKernelGGL(k_big_wss_apply, dim3(t + 1, ut.n_ids, nsb), blk, 0, s, ut, p, t, sup);
            if (t + 1 < sup) hipLaunchKernelGGL(k_big_wss_update, dim3((sup - 1 - t) * (t + 1), ut.n_ids, nsb), blk, 0, s, ut, p, t, sup);
        }
        for (int sb0 = 0; sb0 < nbmax; sb0 += sup) {
            const int sb1 = sb0 + sup < nbmax ? sb0 + sup : nbmax;
            const int nrt = (BIGB * (sb1 - sb0) + BGT - 1) / BGT, ncol = BIGB * sb0 / BGT;
            hipLaunchKernelGGL(k_big_gemm, dim3(nrt * (ncol + 1), ut.n_ids), blk, 0, s, ut, p, 4, sb0, sb1, 0, 0.0);
            if (sb1 < nbmax) {
                const int nt = (BIGB * (nbmax - sb1) + BGT - 1) / BGT;
                hipLaunchKernelGGL(k_big_gemm, dim3(nt * (sb1 / 2 + 1), ut.n_ids), blk, 0, s, ut, p, 1, sb0, sb1, nt, 0.0);
            }
        }
        const int tbs = (ut.max_T + 3) / 4;
        hipLaunchKernelGGL(k_big_zz, dim3(BIG_ZZ_PARTS, ut.n_ids), blk, 0, s, ut, p, tbs);
        hipLaunchKernelGGL(k_big_zz_fold, dim3(ut.n_ids), dim3(64), 0, s, ut, p, tbs);
    }
}

}  // namespace gprf
