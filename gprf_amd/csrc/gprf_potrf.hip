// gprf_potrf.hip — Cholesky of the units of up to 512 points, one workgroup per unit (gpy_linalg.py:77-97 jitchol -> dpotrf,
// logdet :234): the generic kernel k_potrf and the register-resident kernels k_potrf_reg8 / reg8w / reg2, and launch_potrf.
#include "gprf_dev.h"

namespace gprf {

// ------------------------------------------------------------------------------------------------
// Blocked upper Cholesky K = U^T U, one workgroup per unit (gpy_linalg.py:77-97 jitchol -> dpotrf;
// logdet gpy_linalg.py:234).  Per 16-row panel j:
//   (a) wave 0 factors the 16x16 diagonal tile in registers (lane = column, cross-lane by v_readlane)
//       and inverts it (V_jj = U_jj^-1, kept for the triangular solves);
//   (b) row panel  U_jk = V_jj^T C_jk  by MFMA, staged into LDS (k-major rows, conflict-free);
//   (c) trailing update  C_ik -= U_ji^T U_jk  by MFMA with both operands read from the LDS panel.
// ------------------------------------------------------------------------------------------------
constexpr int POTRF_WAVES = 8;

// ------------------------------------------------------------------------------------------------
// k_potrf: the same blocked upper Cholesky, re-scheduled around its critical path
//     diag(j) -> row panel(j) -> update of tile (j+1,j+1) -> diag(j+1) -> ...
// * look-ahead: once row panel j is in LDS, wave 0 alone updates tile (j+1,j+1) and factors it while
//   waves 1..7 apply the rest of the trailing update (MFMA, both operands from the LDS panel);
// * the 16x16 diagonal factor keeps one column per lane and broadcasts with v_readlane; it scales by the
//   reciprocal of the pivot's root, as LAPACK's dpotf2 does;
// * the row panel U_jk = U_jj^-T C_jk is a true forward substitution on the vector ALU (one matrix column
//   per lane, U_jj broadcast from LDS) — no explicit inverse on the critical path, and backward stable;
// * V_jj = U_jj^-1 (wanted by the triangular-solve kernel's MFMA form) is built after the loop, four tiles
//   per wave at once, by the column operations that reduce U_jj to I;
// * log|K| = 2 sum log U_kk (gpy_linalg.py:234) from the stored diagonal, in parallel, fixed order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(POTRF_WAVES * 64, 4) void k_potrf(UnitTab ut, Pools pl, int stamps, int reg_maxT) {
    extern __shared__ double lds[];
    __shared__ int s_fail;
    __shared__ double lred[POTRF_WAVES];
    const UnitRef ur = unit_ref(ut.srec, blockIdx.x);
    int u = ur.u;
    int m = ur.m;
    int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar wave index
    int lr = lane & 15, lg = lane >> 4;
    if (m == 0) {
        if (threadIdx.x == 0) { pl.logdet[u] = 0.0; pl.info[u] = 0; }
        return;
    }
    int mp = pad16(m), T = mp >> 4;
    if (T <= reg_maxT || T > BIG_LA_T) return;               // k_potrf_reg's units; the blocked path's (k_big_*)
    int ldp = mp + ((T & 1) ? 0 : 16);
    double *P = lds;                      // [16][ldp] row panel j of U
    double *Ud = P + 16 * ldp;            // [16][16]  U_jj
    double *rdt = Ud + 256;               // [16]      1 / diag(U_jj)
    double *Tt = rdt + 16;                // [16][17]  look-ahead tile, row-major
    double *Vd = Tt + 16 * 17;            // [16][16]  V_jj = U_jj^-1, row-major (the row panel's operand)
    double *dvals = Vd + 256;             // [mp]      diagonal of U
    double *U = pl.U + ur.mat_off;
    const double *Kp = pl.K + ur.mat_off;   // every tile is first read from the K pool (all of them in step 0)
    double *V = pl.V + (size_t)ur.row_off * 16;
    if (threadIdx.x == 0) s_fail = 0;
    __syncthreads();

    // publish a factored diagonal tile (wave 0): global U, LDS dvals, and V_jj = U_jj^-1 (LDS + the V pool) — the column
    // operations of the register kernels' tile_inverse, the same arithmetic in the same order
    (void)Ud; (void)rdt;
    auto publish = [&](double (&s)[16], double dk, double rdk, int jt, int bad) {
        if (lane < 16) {
            double *Ujj = U + (size_t)(16 * jt) * mp + 16 * jt;
#pragma unroll
            for (int i = 0; i < 16; ++i) Ujj[(size_t)i * mp + lr] = s[i];      // the factor left 0 below the diagonal
            dvals[16 * jt + lr] = dk;
            if (bad && lane == 0) s_fail = 16 * jt + bad;
        }
        double v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            int lrc = lr;
            asm volatile("" : "+v"(lrc));       // (opaque: sixteen loop-invariant doubles would be kept alive across the step loop)
            v[c] = (c == lrc) ? 1.0 : 0.0;
        }
        dpp_src_ready(rdk);
        static_for<0, 16>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            v[k] *= bcast16<k>(rdk);
            dpp_src_ready(s[k]);
            static_for<k + 1, 16>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                fnma_bcast16<i>(v[i], s[k], v[k]);
            });
        });
        if (lane < 16) {
            double *Vj = V + (size_t)jt * 256 + lr * 16;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                Vd[lr * 16 + c] = v[c];
                Vj[c] = v[c];
            }
        }
    };
    if (wave == 0) {
        double s[16], dk, rdk;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = Kp[(size_t)i * mp + lr];
        int bad = diag_factor16_ldl<NoEarly, false>(s, lr, &dk, &rdk, nullptr);
        publish(s, dk, rdk, 0, bad);
    }
    __syncthreads();

    // diagnostic stamps (GPRF_POTRF_STAMPS=1): cycles wave 0 spends in [row panel | barrier | factor | barrier]
    unsigned long long tacc[4] = {0, 0, 0, 0}, tprev = 0;
    bool stamp = stamps && threadIdx.x == 0;
#define GPRF_STAMP(k)                                                     \
    if (stamp) {                                                          \
        unsigned long long tn = __builtin_amdgcn_s_memtime();             \
        tacc[k] += tn - tprev;                                            \
        tprev = tn;                                                       \
    }
    if (stamp) tprev = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < T; ++j) {
        if (s_fail) {
            if (threadIdx.x == 0) { pl.info[u] = s_fail; pl.logdet[u] = 0.0; }
            return;
        }
        int ntr = T - j - 1;
        if (ntr == 0) break;
        // ---- row panel on the matrix pipe (round 4, as in the register kernels: the same bits): U_jk = V_jj^T C_jk, a tile per
        // wave task, four MFMAs each; the next tile's values are in flight while this one's MFMAs run ----
        const double *Csrc = (j == 0) ? Kp : U;   // the trailing matrix: K itself in step 0, U's pool afterwards
        {
            double vp[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) vp[t] = Vd[64 * t + 16 * lg + lr];      // V[4 t + lg][lr]: A = V^T
            int k = j + 1 + wave;
            const double *Cr = Csrc + (size_t)(16 * j + lg) * mp + 16 * k + lr;
            double *Cc = U + (size_t)(16 * j + lg) * mp + 16 * k + lr;
            d4 cur = {0.0, 0.0, 0.0, 0.0};
            if (k < T) {
#pragma unroll
                for (int q = 0; q < 4; ++q) cur[q] = Cr[(size_t)(4 * q) * mp];
            }
            for (; k < T; k += POTRF_WAVES) {
                d4 nxt = {0.0, 0.0, 0.0, 0.0};
                if (k + POTRF_WAVES < T) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) nxt[q] = Cr[(size_t)(4 * q) * mp + 16 * POTRF_WAVES];
                }
                d4 r = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) r = mfma(vp[q], cur[q], r);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    Cc[(size_t)(4 * q) * mp] = r[q];
                    P[(4 * q + lg) * ldp + 16 * k + lr] = r[q];
                }
                Cr += 16 * POTRF_WAVES;
                Cc += 16 * POTRF_WAVES;
                cur = nxt;
            }
        }
        GPRF_STAMP(0)
        __syncthreads();
        GPRF_STAMP(1)
        if (wave == 0) {
            // look-ahead: tile (j+1, j+1) -> LDS (row-major) -> one column per lane -> factor
            int i = j + 1;
            const double *Cii = Csrc + (size_t)(16 * i + lg) * mp + 16 * i + lr;
            d4 acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = Cii[(size_t)(4 * q) * mp];
            // (the step's 16 products from zero, then ONE addition into the running tile — the hierarchical accumulation of
            // the register kernels, see above k_potrf_reg: this kernel factors the units of more than 256 points and the
            // lld / Matérn ones, and was left with the sequential order and its 1.2x LAPACK's error)
            d4 sacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                double a = P[(4 * s + lg) * ldp + 16 * i + lr];
                sacc = mfma(-a, a, sacc);
            }
            acc += sacc;
#pragma unroll
            for (int q = 0; q < 4; ++q) Tt[(lg + 4 * q) * 17 + lr] = acc[q];
            __builtin_amdgcn_wave_barrier();    // same wave, LDS is in order: the reads below see the tile
            double s[16], dk, rdk;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = Tt[r * 17 + lr];
            int bad = diag_factor16_ldl<NoEarly, false>(s, lr, &dk, &rdk, nullptr);
            publish(s, dk, rdk, i, bad);
        } else {
            // trailing update without tile (j+1,j+1): tile rows i = j+1 .. T-1 dealt cyclically to waves 1..7; along
            // a row the A operand (column block i of the panel) is read once, the pointer just advances by one
            // tile, and the next tile's C values are in flight while the current tile's MFMAs run
            {
                for (int i = j + 1 + (wave - 1); i < T; i += POTRF_WAVES - 1) {
                    double a[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) a[s] = -P[(4 * s + lg) * ldp + 16 * i + lr];
                    int k = (i == j + 1) ? i + 1 : i;
                    if (k >= T) continue;
                    double *Cik = U + (size_t)(16 * i + lg) * mp + 16 * k + lr;
                    const double *Rik = Csrc + (size_t)(16 * i + lg) * mp + 16 * k + lr;   // read side
                    const double *Pk = P + lg * ldp + 16 * k + lr;
                    d4 cur;
#pragma unroll
                    for (int q = 0; q < 4; ++q) cur[q] = Rik[(size_t)(4 * q) * mp];
                    for (; k < T; ++k) {
                        d4 nxt = {0.0, 0.0, 0.0, 0.0};
                        if (k + 1 < T) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) nxt[q] = Rik[(size_t)(4 * q) * mp + 16];
                        }
                        d4 t16 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int s = 0; s < 4; ++s) t16 = mfma(a[s], Pk[(4 * s) * ldp], t16);
                        cur += t16;
#pragma unroll
                        for (int q = 0; q < 4; ++q) Cik[(size_t)(4 * q) * mp] = cur[q];
                        Cik += 16;
                        Rik += 16;
                        Pk += 16;
                        cur = nxt;
                    }
                }
            }
        }
        GPRF_STAMP(2)
        __syncthreads();
        GPRF_STAMP(3)
    }
    if (stamp) {
        for (int k = 0; k < 4; ++k) pl.dbg[(size_t)u * 8 + k] = (double)tacc[k];
        pl.dbg[(size_t)u * 8 + 4] = (double)T;
    }
#undef GPRF_STAMP
    if (s_fail) {
        if (threadIdx.x == 0) { pl.info[u] = s_fail; pl.logdet[u] = 0.0; }
        return;
    }
    potrf_epilogue<POTRF_WAVES, false>(U, V, P, dvals, lred, mp, T, u, pl);      // (V_jj went out tile by tile)
}

// ------------------------------------------------------------------------------------------------
// k_potrf_reg's accumulator tiles live in EXPLICITLY NUMBERED AGPRs: tile S = a[8S : 8S+7], D layout (lane
// (lg, lr), register pair q = element [lg + 4q][lr]).  They are invisible to the compiler on purpose: as C++
// values it copies them between the VGPR and AGPR halves around every use (240 v_accvgpr_read per step) or
// spills them.  Every access is one of the volatile asm blocks below (volatile asm keeps program order); the
// kernel declares the range with atile_reserve() and holds no other AGPR values (tests/ checks the ISA).
// Inline asm is invisible to the hazard recogniser, so the wait states are written out:
//   * VALU write (v_accvgpr_write, operand moves) -> MFMA read: 2          -> s_nop 1 before the MFMAs
//   * MFMA f64 16x16x4 result -> same-tuple srcC of the next MFMA: 0        (back-to-back accumulate)
//   * MFMA f64 16x16x4 result -> VALU / LDS read: 18                        -> atile_settle() / trailing s_nop's
// ------------------------------------------------------------------------------------------------
template <int SLOTS>
__device__ __forceinline__ void atile_reserve() {
    static_assert(SLOTS == 32 || SLOTS == 20, "one clobber list per instantiation");
    if constexpr (SLOTS == 32) asm volatile("; accumulator tiles: a[0:255]" ::: "a0", "a1", "a254", "a255");
    else asm volatile("; accumulator tiles: a[0:159]" ::: "a0", "a1", "a158", "a159");
}
__device__ __forceinline__ void atile_settle() { asm volatile("s_nop 15\n\ts_nop 3"); }

template <int S>
__device__ __forceinline__ void atile_set(const double (&v)[4]) {
    asm volatile("v_accvgpr_write_b32 a[%8], %0\n\tv_accvgpr_write_b32 a[%9], %1\n\t"
                 "v_accvgpr_write_b32 a[%10], %2\n\tv_accvgpr_write_b32 a[%11], %3\n\t"
                 "v_accvgpr_write_b32 a[%12], %4\n\tv_accvgpr_write_b32 a[%13], %5\n\t"
                 "v_accvgpr_write_b32 a[%14], %6\n\tv_accvgpr_write_b32 a[%15], %7"
                 :
                 : "v"(__double2loint(v[0])), "v"(__double2hiint(v[0])), "v"(__double2loint(v[1])),
                   "v"(__double2hiint(v[1])), "v"(__double2loint(v[2])), "v"(__double2hiint(v[2])),
                   "v"(__double2loint(v[3])), "v"(__double2hiint(v[3])), "n"(8 * S), "n"(8 * S + 1), "n"(8 * S + 2),
                   "n"(8 * S + 3), "n"(8 * S + 4), "n"(8 * S + 5), "n"(8 * S + 6), "n"(8 * S + 7));
}
template <int S>
__device__ __forceinline__ void atile_get(double (&v)[4]) {
    int w[8];
    asm volatile("v_accvgpr_read_b32 %0, a[%8]\n\tv_accvgpr_read_b32 %1, a[%9]\n\t"
                 "v_accvgpr_read_b32 %2, a[%10]\n\tv_accvgpr_read_b32 %3, a[%11]\n\t"
                 "v_accvgpr_read_b32 %4, a[%12]\n\tv_accvgpr_read_b32 %5, a[%13]\n\t"
                 "v_accvgpr_read_b32 %6, a[%14]\n\tv_accvgpr_read_b32 %7, a[%15]"
                 : "=v"(w[0]), "=v"(w[1]), "=v"(w[2]), "=v"(w[3]), "=v"(w[4]), "=v"(w[5]), "=v"(w[6]), "=v"(w[7])
                 : "n"(8 * S), "n"(8 * S + 1), "n"(8 * S + 2), "n"(8 * S + 3), "n"(8 * S + 4), "n"(8 * S + 5),
                   "n"(8 * S + 6), "n"(8 * S + 7));
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = __hiloint2double(w[2 * q + 1], w[2 * q]);
}
// tile S += sum_t a[t]^T b[t]  (four chained MFMAs), in two pieces: the caller puts the next tile's operand
// fetch (scalar decode, address adds, LDS reads) between them, where it issues for free while the first MFMA
// occupies the pipe — with one wave per SIMD nothing else would hide it
// (`dep` is tied through the block without being touched: whatever the caller derives from it afterwards — the
// next tile's slot decode — cannot be scheduled in front of this MFMA)
template <int S>
__device__ __forceinline__ void atile_mfma_first(const double (&a)[4], const double (&b)[4], int &dep) {
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f64_16x16x4_f64 a[%3:%4], %1, %2, a[%3:%4]"
                 : "+v"(dep)
                 : "v"(a[0]), "v"(b[0]), "n"(8 * S), "n"(8 * S + 7));
}
template <int S>
__device__ __forceinline__ void atile_mfma_rest(const double (&a)[4], const double (&b)[4]) {
    asm volatile("s_nop 1\n\t"          // (a compiler-inserted copy of an operand may sit right in front)
                 "v_mfma_f64_16x16x4_f64 a[%6:%7], %0, %1, a[%6:%7]\n\t"
                 "v_mfma_f64_16x16x4_f64 a[%6:%7], %2, %3, a[%6:%7]\n\t"
                 "v_mfma_f64_16x16x4_f64 a[%6:%7], %4, %5, a[%6:%7]"
                 :
                 : "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3]), "n"(8 * S), "n"(8 * S + 7));
}
// ---- hierarchical accumulation ----
// A trailing entry used to take its 16 products per step one fused multiply-add after the other, each rounding at the
// magnitude of the running entry: measured against an 80-bit factorisation that sequential chain is what made the device's
// factor 1.2x as far from the truth as LAPACK's (whose blocked updates sum a block's products from zero first); a numpy
// emulation of both orders on the north-star pair units reproduces the 1.2x and gives 0.67x for this one
// (tests/diag/cpu_accumulation_order.py).  Now: the 16 products of a step are summed FROM ZERO in a temporary VGPR tile
// (the same four MFMAs, srcC = 0 for the first) and enter the accumulator with ONE addition — on the vector ALU, in the
// shadow of the next tile's MFMAs (8 accumulator reads, 4 adds, 8 writes: ~90 issue cycles against 256 of matrix pipe).
__device__ __forceinline__ void ttile_mfma_first(d4 &t, const double (&a)[4], const double (&b)[4], int &dep) {
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %2, %3, 0"
                 : "=&v"(t), "+v"(dep)
                 : "v"(a[0]), "v"(b[0]));
}
__device__ __forceinline__ void ttile_mfma_rest(d4 &t, const double (&a)[4], const double (&b)[4]) {
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %3, %4, %0\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %5, %6, %0"
                 : "+v"(t)
                 : "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3]));
}
// tile S += t   (t must be settled: at least 18 wait states behind the MFMA that wrote it — in the chain below the next
// tile's four MFMAs and its operand fetch lie in between)
template <int S>
__device__ __forceinline__ void atile_add(const d4 &t) {
    double v[4];
    atile_get<S>(v);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += t[q];
    atile_set<S>(v);
}

// the same on a VGPR tile (the diagonal tiles, staged through LDS): c += sum_t a[t]^T b[t]; the result is
// settled (readable) on return
__device__ __forceinline__ void mfma4_vgpr(d4 &c, const double (&a)[4], const double (&b)[4]) {
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %3, %4, %0\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %5, %6, %0\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %7, %8, %0\n\t"
                 "s_nop 15\n\ts_nop 3"
                 : "+v"(c)
                 : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3]));
}

constexpr int POTRF_REG_MAXT_C = 16;  // largest unit edge in tiles the four-wave register-resident kernels take

// the eight-wave instantiation (one workgroup per CU) takes units of up to 20 tiles per edge (320 points: the seismic
// configuration's block pairs): 8 x 20 accumulator slots hold 160 of a 20-tile unit's 190 strictly-upper tiles, the FIRST
// 30 in row-major order (row 0 and part of row 1: they retire first and are updated at most once) wait in LDS
constexpr int POTRF_REG8_MAXT = 20;
constexpr int POTRF_REG8_LDP = 336;  // >= 16 * 20, = 16 mod 32
// ... and 21 .. 32 tiles (GW): the tiles beyond the 160 accumulator slots wait in the U pool, at their own place
constexpr int POTRF_REG8W_MAXT = 32;
constexpr int POTRF_REG8W_LDP = 528; // >= 16 * 32, = 16 mod 32
constexpr int POTRF_REG2_LDP = 240;  // the two-per-CU instantiation: >= 16 * 13, = 16 mod 32
// ------------------------------------------------------------------------------------------------
// k_potrf_reg<SLOTS>: the same factorisation for units whose whole upper triangle of 16x16 tiles fits on
// chip (T <= reg_maxT tiles per edge).  The trailing matrix never goes back to memory: the strictly-upper
// tiles, enumerated row-major, are dealt cyclically to the worker waves 1..3 (and, for the largest units, the
// first few rows also to wave 0) and live in explicitly numbered AGPR tiles, so that every row panel and every
// trailing update is spread over the workers; the T diagonal tiles live in LDS (Dt).  Step j:
//   waves 1..7: their tiles of row j -> LDS panel -> one column per lane -> forward substitution (DPP
//               broadcast of U_jj) -> LDS panel + global U
//   barrier
//   wave 0    : look-ahead — Dt[j+1] -= P_{j+1}^T P_{j+1}, factor, publish U_{j+1,j+1}
//   waves 1..7: acc[slot] -= P_i^T P_k for their live tiles and Dt[i] -= P_i^T P_i for i >= j+2 (tile i by wave
//               1 + i%7), both MFMA operands from the LDS panel
//   barrier
// Global traffic is one read of K's upper triangle and one write of U; the per-step chain is
// substitution + factor with no memory latency in it.
// ------------------------------------------------------------------------------------------------
// GEN: the kernel matrix is not read from the K pool but GENERATED here from the unit's coordinates (SE kernel):
// k_fill does not run at all, K never exists in HBM, and the prologue's burst of tile loads (every resident unit
// at once) becomes arithmetic spread over the launch; k_mgrad<.,.,false> re-evaluates the values it needs.
// Every wave has 256 registers (20 tile slots = a[0:159] + 96 VGPRs).  RW = 4: units of up to 13 tiles per edge, TWO
// workgroups per CU — a unit's factorisation is a latency chain that keeps its SIMDs a quarter busy, so two of them side by
// side nearly double the CU's throughput; RW = 8: one workgroup per CU, units of up to 20 (GW: 32) tiles.  Units outside
// [min_T, reg_maxT] are left alone.  (Rounds 1-4 also had a four-wave form with 512 registers per wave, a run-ahead step
// loop without workgroup barriers and ("lld","matern32") generation in here: each measured slower than what is left —
// DESIGN.md section 4 — and removed in round 5.)
// GW (eight-wave instantiation, K from the pool): units of up to 32 tiles per edge — the (up to 336) tiles beyond the
// accumulator slots wait in GLOBAL memory instead of LDS: in the U pool, each at its own final place (nobody else touches a
// tile of U before its row is solved), read and written through the CU's L1 / the L2 like the generic kernel's whole trailing
// matrix — a fraction of that kernel's traffic (the first rows only, and only until they retire).  Waves of one workgroup
// share the CU's L1: a store is visible to the other waves behind s_waitcnt vmcnt(0) + the workgroup barrier.
template <int RW, int SLOTS, bool GEN, bool GW = false>
__device__ __forceinline__ void potrf_reg_body(const UnitTab &ut, const Pools &pl, int stamps, int reg_maxT, const KParams &kp,
                                               int which, int min_T = 0) {
    static_assert(8 * SLOTS <= 256, "atile_reserve() covers a[0:255]");
    static_assert(!GW || (RW == 8 && !GEN), "waiting tiles in the U pool: the eight-wave kernel reading the K pool");
    extern __shared__ double lds[];
    __shared__ int s_fail;
    __shared__ double lred[RW];
#ifdef GPRF_WGTRACE
    __shared__ double s_tr0;       // (WgTrace itself does not survive this kernel's register discipline)
    if (threadIdx.x == 0) s_tr0 = (double)__builtin_amdgcn_s_memrealtime();
#endif
#ifdef GPRF_PROFILE
    unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
    // which: 0 = every unit of the launch order; 1 / 2 = the device-built list of large / small units
    // The large-unit launch has grid_big >= |big_list| workgroups; its surplus ones must not idle (a 512-register
    // workgroup can only be scheduled on an EMPTY CU: waiting for one to drain behind the two-per-CU kernel's residents,
    // just to exit, would hold back this kernel's completion): they take units from the END of the small list (the
    // smallest ones; this instantiation handles every size), and the small-unit launch leaves those to them.
    // fork of the two Cholesky queues (launch_potrf): this kernel has started, so everything in front of it on the main
    // queue is complete — tell the side queue, whose small-unit kernel waits for this word
    if (which == 1 && ut.fork_flag != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(ut.fork_flag, ut.fork_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // (the record of this workgroup's most likely slot is loaded alongside the list lengths, not behind them)
    UnitRef ur;
    if (which == 0) {
        ur = unit_ref(ut.srec, blockIdx.x);
    } else {
        int bid = blockIdx.x;
        ur = unit_ref(which == 1 ? ut.big_rec : ut.small_rec, bid);
        int nb = ut.ctl[CTL_NBIG], ns = ut.ctl[CTL_NSMALL];
        int surplus = ut.grid_big > nb ? ut.grid_big - nb : 0;
        if (surplus > ns) surplus = ns;
        if (which == 1) {
            if (bid >= nb) {
                if (bid - nb < surplus) ur = unit_ref(ut.small_rec, ns - 1 - (bid - nb));
                else return;
            }
        } else {
            if (bid >= ns - surplus) return;
        }
    }
    const int u = ur.u;
    int m = ur.m;
    int mp = pad16(m), T = mp >> 4;
    if (T > reg_maxT || T < min_T) return;      // k_potrf's units; another instantiation's
    if (m == 0) {
        if (threadIdx.x == 0) { pl.logdet[u] = 0.0; pl.info[u] = 0; }
        return;
    }
    int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lr = lane & 15, lg = lane >> 4;
    // (round 4 measured alternating which hardware wave is the factor wave between co-resident workgroups — its ~250 DPP fp64
    // multiply-adds per step would otherwise pile up on one SIMD: no change, 110.6 vs 109.6 us, C4 664 vs 666: they do not)
    // fixed panel pitch (an odd multiple of 16 doubles: the k-major MFMA operand reads are conflict free):
    // every LDS row offset below is an instruction immediate
    // ONE panel buffer: pitch 240 for the two-per-CU form (two workgroups share the CU's 160 KB; units of up to 13 tiles), the
    // wide pitches for the eight-wave one; the solved panel goes to global memory from inside the row solve
    constexpr int MT = RW == 8 ? (GW ? POTRF_REG8W_MAXT : POTRF_REG8_MAXT) : POTRF_REG_MAXT_C;
    constexpr int ldp = RW == 8 ? (GW ? POTRF_REG8W_LDP : POTRF_REG8_LDP) : POTRF_REG2_LDP;
    double *P0 = lds;                     // [16][ldp] row panel j of U
    double *Ud = P0 + 16 * ldp;           // [16][16]  U_jj
    double *rdt = Ud + 256;               // [16]      1 / diag(U_jj)
    double *Gd = rdt + 16;                // [16][16]  rows of G = D^-1 U_jj (unit triangular: the substitution's operand)
    double *dvals = Gd + 256;             // [16 T]    diagonal of U
    double *Dt = dvals + 16 * MT;         // [T][16][16] diagonal tiles of the trailing matrix
    double *U = pl.U + ur.mat_off;
    const double *Kp = pl.K + ur.mat_off;   // read once (upper triangle); U goes to its own pool, K stays for k_mgrad
    double *V = pl.V + (size_t)ur.row_off * 16;
    if (threadIdx.x == 0) s_fail = 0;
    unsigned glane = (unsigned)(lg * mp + lr);
    int dlane = lg * 16 + lr;             // lane's element of a row-major 16x16 tile, rows lg + 4q at + 64 q

    // Dealing the strictly-upper tiles (row-major index idx) to the waves.  While the three workers' 3 * SLOTS slots
    // hold everything (T <= 14) they take the tiles cyclically (idx % 3) and wave 0 only factors.  A larger unit
    // has ov = total - 3 SLOTS tiles too many: its FIRST 4 ov tiles are dealt to all four waves (idx % 4, wave 0
    // taking idx % 4 == 3), the rest to the workers as before.  Wave 0's tiles then lie in the first rows: they are
    // the first to retire, so its trailing work (which runs after its factor, on the critical path) is over after
    // a few steps instead of staying a quarter of everything.
    // (RW waves: NW = RW - 1 workers + the factor wave; the text above is RW = 4.  RW = 8 — eight waves of 256 registers,
    // ONE workgroup per CU, seven workers x 20 slots for every unit of up to 16 tiles, one kernel and one launch — was built
    // and measured in round 3: a unit finishes 20 % sooner (T = 15: 89 vs 104-113 us, T = 13: 71 vs 97) but holds a whole CU,
    // and CU-time is what the stage is short of: 140 us against 123 with the two instantiations)
    static_assert(RW == 4 || RW == 8, "RW - 1 workers + the factor wave");
    constexpr int NW = RW - 1;
    // n_lds: tiles beyond ALL RW * SLOTS accumulator slots (the eight-wave kernel, T = 19, 20): the first n_lds tiles in
    // row-major order stay in LDS (Ot) — solved from there when their row comes up, updated there until then; the others
    // (real index n_lds + idx) are dealt as before
    // FRONT (round 4, the eight-wave kernel too): stamps on a 20-tile unit — wave 0, dealt every eighth tile of ALL rows, spent
    // 12.8 k cycles per step in its phase (7.5 k of chain + its share of every trailing update) and the workers 5.7 k of their
    // 15.6 k waiting for it; with its 20 tiles front-loaded (the rows right behind the waiting tiles) it is a pure factor wave
    // from step 3 on: the seismic shape's Cholesky stage 146 -> 134 us, at paper scale 1464 -> 1413.  (Units of up to 17 tiles have
    // no such tiles: nothing changes for them.  The four-wave two-per-CU kernel front-loaded: C3 109.2 -> 110, C4 662 -> 670: no.)
    constexpr bool FRONT = RW == 8;
    const int total_all = T * (T - 1) / 2;
    const int n_lds = (RW == 8 && total_all > RW * SLOTS) ? total_all - RW * SLOTS : 0;      // (RW == 4: a constant 0)
    const int total = total_all - n_lds;
    const int ov = total > NW * SLOTS ? total - NW * SLOTS : 0;      // (a larger share for wave 0 — total / 6 .. / 14 — measured: no change)
    const int head = FRONT ? ov : (RW * ov < total ? RW * ov : total);
    const bool w0busy = ov > 0;                        // wave 0 owns tiles too
    const bool mine = wave > 0 || w0busy;
    const int wpos = wave == 0 ? NW : wave - 1;        // position in the RW-way deal; workers: also in the NW-way
    const int nhead = FRONT ? (wave == 0 ? ov : 0)
                         : (head - wpos + NW < 0 ? 0 : (head - wpos + NW) / RW);      // this wave's tiles of the RW-way part
    // tiles of this wave among idx < r
    auto cnt = [&](int r_all) {
        const int r = r_all > n_lds ? r_all - n_lds : 0;      // (row boundaries come as real tile indices)
        if constexpr (FRONT) {
            if (wave == 0) return r < ov ? r : ov;
            int c = r - ov - wpos + NW - 1;
            return (r <= ov || c < 0) ? 0 : c / NW;
        }
        if (r <= head) {
            int c = r - wpos + NW;
            return c < 0 ? 0 : c / RW;
        }
        if (wave == 0) return nhead;
        int c = r - head - wpos + NW - 1;
        return nhead + (c < 0 ? 0 : c / NW);
    };
    // lane s: slot s -> tile, 32 * tile row + tile column, or -1 (fetched with v_readlane / a shuffle where
    // needed: 30-odd live SGPRs would crowd out the row pointers);  lane j: s_hi of step j = this wave's tiles
    // in rows 0..j
    int pkv = -1, shv = 0;
    {
        int sl = lane;
        int idx = FRONT ? (wave == 0 ? (sl < ov ? sl : total) : ov + NW * sl + wpos)
                     : (sl < nhead ? RW * sl + wpos : (wave == 0 ? total : head + NW * (sl - nhead) + wpos));
        int i = 0, rs = 0, rl = T - 1;
        const int idr = idx + n_lds;                    // the tile's real row-major index
        while (rl > 0 && idr >= rs + rl) { rs += rl; --rl; ++i; }
        if (mine && rl > 0 && idx < total && lane < SLOTS) pkv = 32 * i + i + 1 + (idr - rs);
        int jj = lane < T - 1 ? lane : T - 1;           // rows 0..jj end at tile index (jj+1) T - (jj+1)(jj+2)/2
        shv = mine ? cnt((jj + 1) * T - ((jj + 1) * (jj + 2)) / 2) : 0;
        if (shv > SLOTS) shv = SLOTS;
    }
#define PK(s) __builtin_amdgcn_readlane(pkv, s)
    atile_reserve<SLOTS>();
    // GEN: K(row, col) of this unit, exactly k_fill's definition (identity in the padding, noise + jitter on the
    // diagonal); the unit's coordinates wait in LDS
    double *xs = Dt + 256 * (T < reg_maxT ? T : reg_maxT);      // [mp][XS], GEN only (the launcher sizes the LDS)
    constexpr int XS = XPAD;
    double *Ot = xs + (GEN ? 16 * (T < reg_maxT ? T : reg_maxT) * XS : 0);      // [n_lds][16][16] the tiles that wait in LDS
    const double diag_add = kp.nv + ut.jitter[u];
    // NT tiles (pk = 32 * tile row + tile column) side by side, branch-free: this wave is alone on its SIMD, so the
    // only thing that hides the latency of one exp()'s dependent chain is the other 4 NT - 1 evaluations
    auto kgen = [&](auto ntc, const int *pk, double (*out)[4], double sign) {
        constexpr int NT = decltype(ntc)::value;
        // (round 4: in the two-per-CU instantiation two workgroups' generating waves share every SIMD and the prologue is bound
        // by instruction issue — a fifth to a quarter of a unit's time: the third coordinate's three instructions go when
        // dx <= 2 (adding (0 - 0)^2 changes no bit), and a tile whose 16 columns all lie inside the unit and off the diagonal —
        // all strictly-upper tiles but those of the last tile column — skips the diagonal / padding selects: the same bits)
        double sq[NT * 4], e[NT * 4];
        const bool two_d = kp.dx <= 2;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int col = 16 * (pk[t] & 31) + lr;
            double xj[3] = {xs[col * XPAD], xs[col * XPAD + 1], two_d ? 0.0 : xs[col * XPAD + 2]};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int row = 16 * (pk[t] >> 5) + 4 * q + lg;
                sq[4 * t + q] = se_neg_r2(xs[row * XPAD], xs[row * XPAD + 1], two_d ? 0.0 : xs[row * XPAD + 2], xj[0], xj[1], xj[2],
                                          kp.inv_ls, !two_d);
            }
        }
        exp_fast_v<NT * 4>(sq, e);
        const double ssv = sign * kp.sv;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int col = 16 * (pk[t] & 31) + lr;
            const bool interior = (pk[t] >> 5) != (pk[t] & 31) && 16 * (pk[t] & 31) + 16 <= m && 16 * (pk[t] >> 5) + 16 <= m;      // (uniform)
            if (interior) {
#pragma unroll
                for (int q = 0; q < 4; ++q) out[t][q] = ssv * e[4 * t + q];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int row = 16 * (pk[t] >> 5) + 4 * q + lg;
                    double v = kp.sv * e[4 * t + q];
                    v = (row == col) ? __dadd_rn(v, diag_add) : v;      // (two roundings, like the fill and the reference)
                    if (!(row < m && col < m)) v = (row == col) ? 1.0 : 0.0;
                    out[t][q] = sign * v;
                }
            }
        }
    };
    if constexpr (GEN) {
        const double *Xu = pl.Xu + (size_t)ur.row_off * XS;
        for (int e = threadIdx.x; e < mp * XS; e += RW * 64) xs[e] = Xu[e];
        __syncthreads();
    }
    // diagonal tiles -> LDS
    for (int i = wave; i < T; i += RW) {
        if constexpr (GEN) {
            double kv[1][4];
            int pk[1] = {33 * i};
            kgen(std::integral_constant<int, 1>{}, pk, kv, 1.0);
#pragma unroll
            for (int q = 0; q < 4; ++q) Dt[i * 256 + 64 * q + dlane] = kv[0][q];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double *Cs = Kp + (size_t)(16 * i + 4 * q) * mp + 16 * i;
                Dt[i * 256 + 64 * q + dlane] = Cs[glane];
            }
        }
    }
    // the tiles that wait in LDS (n_lds > 0: units of 19, 20 tiles per edge in the eight-wave kernel), as they are (not negated)
    // (GW: in the U pool, at their own place)
    for (int t = wave; t < n_lds; t += RW) {
        int i = 0, rs = 0, rl = T - 1;
        while (t >= rs + rl) { rs += rl; --rl; ++i; }
        const int k = i + 1 + (t - rs);
        if constexpr (GW) {
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = Kp[(size_t)(16 * i + 4 * q) * mp + 16 * k + glane];
#pragma unroll
            for (int q = 0; q < 4; ++q) U[(size_t)(16 * i + 4 * q) * mp + 16 * k + glane] = v[q];
        } else if constexpr (GEN) {
            double kv[1][4];
            int pk[1] = {32 * i + k};
            kgen(std::integral_constant<int, 1>{}, pk, kv, 1.0);
#pragma unroll
            for (int q = 0; q < 4; ++q) Ot[t * 256 + 64 * q + dlane] = kv[0][q];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double *Cs = Kp + (size_t)(16 * i + 4 * q) * mp + 16 * k;
                Ot[t * 256 + 64 * q + dlane] = Cs[glane];
            }
        }
    }

    // ---- round 4: the row panel on the matrix pipe ----
    // U_jk = U_jj^-T C_jk used to be a forward substitution on the vector ALU: the tile dumped to LDS, reloaded one column
    // per lane, 120 DPP fp64 multiply-adds per pass of (at most) four tiles — and a DPP fp64 FMA issues at ~16 cycles, four
    // times a plain one: 1.9 k cycles of SIMD time per pass however few tiles it holds, a quarter of a tile-owning wave's
    // step.  Two workgroups share every SIMD of a CU in the two-per-CU instantiation and the eight-wave one has two waves per
    // SIMD too: these kernels are bound by the SIMDs' instruction issue, not by their dependency chains (the run-ahead form,
    // which removes every barrier wait, runs in the same time).  Now wave 0 follows the factor of tile j with V_jj = U_jj^-1
    // (the column operations the epilogue used to do for all tiles at the end — the triangular-solve kernels want V_jj
    // anyway — one tile at a time here) and the tile owners form U_jk = V_jj^T C_jk with four MFMAs per tile: the accumulator
    // registers ARE the B operand (register pair q = rows 4q + lg), the product lands in D layout and goes straight to the
    // LDS panel and to global U.  No dump, no reload, no DPP on the tile owners, no copy pass.
    double *Vd0 = Gd;                     // V_jj in LDS (Ud: U_jj staged for its way to global)
    (void)rdt;
    // wave 0, lanes = columns of U_jj (s[k] = row k of U, rdk = 1 / U_kk of this lane's column): row lr of V_jj -> Vb (LDS,
    // row-major) and the V pool
    // (lro: the lane's column index again, for the store addresses only)
    auto tile_inverse = [&](double (&s)[16], double rdk, int jt, double *Vb, int lro) {
        double v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            int lrc = lr;
            asm volatile("" : "+v"(lrc));       // keep the 16 lane masks from living in SGPRs all at once
            v[c] = (c == lrc) ? 1.0 : 0.0;
        }
        dpp_src_ready(rdk);
        static_for<0, 16>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            v[k] *= bcast16<k>(rdk);
            dpp_src_ready(s[k]);                // (written by the factor's selects)
            static_for<k + 1, 16>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                fnma_bcast16<i>(v[i], s[k], v[k]);
            });
        });
        if (lane < 16) {
            double *Vj = V + (size_t)jt * 256 + lro * 16;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                Vb[lro * 16 + c] = v[c];
                Vj[c] = v[c];
            }
        }
    };
    // wave 0: factor tile jt (in Dt, row-major) and publish it: V_jj / dvals in LDS, U_jj staged for its way to global
    auto factor_publish = [&](int jt) {
        __builtin_amdgcn_wave_barrier();
        double s[16], dk, rdk;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = Dt[jt * 256 + r * 16 + lr];
        int bad = diag_factor16_ldl<NoEarly, false>(s, lr, &dk, &rdk, nullptr);
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) Ud[i * 16 + lr] = s[i];   // the factor left 0 below the diagonal
            dvals[16 * jt + lr] = dk;
            if (bad && lane == 0) s_fail = 16 * jt + bad;
        }
        tile_inverse(s, rdk, jt, Vd0, lr);
    };
    // Dt[i] -= P_i^T P_i
    double *P = P0;                       // the current step's panel buffer
    // (pl_ / dl_: this lane's offsets lg * ldp + lr into a panel row group and lg * 16 + lr into a tile — the step loop
    // passes copies it has made opaque inside the step, so that the addresses built from them are not kept alive across
    // the whole loop: at the 96-register cap the compiler parked exactly those in a0 / a1, i.e. in tile slot 0)
    auto diag_update = [&](int i, int pl_, int dl_) {
        d4 t;
        double a[4], na[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = Dt[i * 256 + 64 * q + dl_];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] = P[(4 * k) * ldp + 16 * i + pl_];
            na[k] = -a[k];
        }
        d4 sacc = {0.0, 0.0, 0.0, 0.0};      // (the step's 16 products from zero, then ONE addition: see "hierarchical accumulation")
        mfma4_vgpr(sacc, na, a);
#pragma unroll
        for (int q = 0; q < 4; ++q) Dt[i * 256 + 64 * q + dl_] = t[q] + sacc[q];
    };
    auto load_tiles = [&]() {
        if constexpr (GEN) {
            // one tile at a time, straight into its numbered accumulator: a runtime loop over this wave's slots (one
            // copy of the exp() code) and a jump on the wave-uniform slot number (the single panel buffer is too small
            // to stage batches in)
#pragma unroll 1
            for (int sl = 0; sl < SLOTS; ++sl) {
                int pk[1] = {__builtin_amdgcn_readlane(pkv, sl)};
                if (pk[0] < 0) break;                // slots are filled from 0 up
                double kv[1][4];
                kgen(std::integral_constant<int, 1>{}, pk, kv, -1.0);      // MINUS the trailing tile
                static_for<0, SLOTS>([&](auto sc) {
                    constexpr int S = decltype(sc)::value;
                    if (sl == S) atile_set<S>(kv[0]);
                });
            }
            return;
        }
        // tiles -> accumulators, PRO_BATCH slots at a time: all the batch's loads are issued before the first
        // (volatile) accumulator write, which nothing is moved across
        constexpr int PRO_BATCH = 4;      // (96 VGPRs)
        static_for<0, (SLOTS + PRO_BATCH - 1) / PRO_BATCH>([&](auto bc) {
            constexpr int B0 = decltype(bc)::value * PRO_BATCH;
            double kv[PRO_BATCH][4];
    #pragma unroll
            for (int i = 0; i < PRO_BATCH; ++i) {
                // uniform row pointer + one 32-bit lane offset: global_load with an SGPR base; unconditional (a
                // branch per slot serialises the loads)
                int pks = (B0 + i < SLOTS) ? PK(B0 + i < SLOTS ? B0 + i : 0) : -1;
                int pc = pks < 0 ? 0 : pks;
    #pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double *Cs = Kp + (size_t)(16 * (pc >> 5) + 4 * q) * mp + 16 * (pc & 31);
                    kv[i][q] = -Cs[glane];            // the accumulators hold MINUS the trailing tile
                }
            }
            static_for<0, PRO_BATCH>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                if constexpr (B0 + I < SLOTS) atile_set<B0 + I>(kv[I]);
            });
        });
    };
    __syncthreads();
    // wave 0 factors the first diagonal tile while the workers fetch their tiles
    if (wave == 0) factor_publish(0);
    if (mine) load_tiles();
    __syncthreads();

    // diagnostic builds only (GPRF_BUILD_DEFS=-DGPRF_PROFILE; the stamps cost registers):
    // GPRF_POTRF_STAMPS=1: wave 0's [idle | barrier | look-ahead + factor | barrier];
    // GPRF_POTRF_STAMPS=2: wave 1's [dump | panel loads | substitution | stores | barrier | trailing | barrier]
#ifdef GPRF_PROFILE
    unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    bool stamp = stamps == 1 && wave == 0;      // wave-uniform: the accumulators stay in SGPRs
    bool stamp2 = stamps == 2 && wave == 1;
    bool stamp3 = stamps == 3 && wave == 1;     // wave 1's phase 2: [panel copy | diagonal tiles | trailing chain | rest]
#define GPRF_STAMPX(on, k)                                                \
    if (on) {                                                             \
        unsigned long long tn = __builtin_amdgcn_s_memtime();             \
        tacc[k] += tn - tprev;                                            \
        tprev = tn;                                                       \
    }
#define GPRF_STAMP(k) GPRF_STAMPX(stamp, k)
#define GPRF_STAMP2(k) GPRF_STAMPX(stamp2, k)
#define GPRF_STAMP3(k) GPRF_STAMPX(stamp3, k)
#else
    constexpr bool stamp = false, stamp2 = false, stamp3 = false;
    unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    (void)tacc; (void)tprev; (void)stamps;
#define GPRF_STAMP(k)
#define GPRF_STAMP2(k)
#define GPRF_STAMP3(k)
#endif
    if (stamp || stamp2 || stamp3) tprev = __builtin_amdgcn_s_memtime();
#ifdef GPRF_PROFILE
    unsigned long long t_loop = tprev;
#endif
#ifdef GPRF_WGTRACE
    __shared__ double s_tr1, s_tr2;       // start / end of the step loop
    if (threadIdx.x == 0) s_tr1 = (double)__builtin_amdgcn_s_memrealtime();
#endif
    const int s_end = T >= 2 ? __builtin_amdgcn_readlane(shv, T - 2) : 0;
    // this wave's tiles of row j: U_jk = V_jj^T C_jk on the matrix pipe, straight from the accumulators (they hold MINUS the
    // trailing tile: the A operand is -V_jj) into the LDS panel and global U.  Slots s_lo .. s_hi-1 (row-major tile order);
    // static walk in groups of 8 slots, like the trailing chain: the slot number must be a compile-time constant for the
    // register numbers.  The products of one tile settle under the next tile's MFMAs.
    auto solve_rows = [&](int j, int s_lo, int s_hi, int lb, int dl, const double *Vb) {
        double va[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) va[t] = -Vb[dl + 64 * t];            // -V[4 t + lg][lr]
        // byte offset of this lane's first row (16 j + lg) in U, column lr (a unit's matrix is at most 512 KB: 32 bits)
        const unsigned ub = ((unsigned)(16 * j + (dl >> 4)) * (unsigned)mp + (unsigned)(dl & 15)) * 8u;
        const unsigned rstep = 32u * (unsigned)mp;                       // four rows down, in bytes
        d4 tt[2];
        int pend = -1;                          // tile column of the product still settling in tt[parity]
        int par = 0;
        auto flush = [&](int k, d4 &t) {
            // (tied to the value: at least 18 wait states between the MFMA that wrote it and its first reader, wherever the
            // scheduler puts these stores)
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(t));
#pragma unroll
            for (int q = 0; q < 4; ++q) P[lb + (4 * q) * ldp + 16 * k] = t[q];
            unsigned off = ub + 128u * (unsigned)k;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(off), "v"(t[q]), "s"(U) : "memory");
                off += rstep;
            }
        };
        static_for<0, (SLOTS + 7) / 8>([&](auto gc) {
            constexpr int G = decltype(gc)::value;
            int lo = s_lo, hi = s_hi;
            asm volatile("" : "+s"(lo), "+s"(hi));
            if (hi > 8 * G && lo < 8 * G + 8) {
                static_for<0, 8>([&](auto sc) {
                    constexpr int S = 8 * G + decltype(sc)::value;
                    if constexpr (S < SLOTS) {
                        int lo2 = lo, hi2 = hi;
                        asm volatile("" : "+s"(lo2), "+s"(hi2));
                        if (S >= lo2 && S < hi2) {
                            asm volatile("s_nop 1\n\t"
                                         "v_mfma_f64_16x16x4_f64 %0, %1, a[%5:%6], 0\n\t"
                                         "v_mfma_f64_16x16x4_f64 %0, %2, a[%7:%8], %0\n\t"
                                         "v_mfma_f64_16x16x4_f64 %0, %3, a[%9:%10], %0\n\t"
                                         "v_mfma_f64_16x16x4_f64 %0, %4, a[%11:%12], %0"
                                         : "=&v"(tt[S & 1])
                                         : "v"(va[0]), "v"(va[1]), "v"(va[2]), "v"(va[3]), "n"(8 * S), "n"(8 * S + 1), "n"(8 * S + 2),
                                           "n"(8 * S + 3), "n"(8 * S + 4), "n"(8 * S + 5), "n"(8 * S + 6), "n"(8 * S + 7));
                            // the tile before this one has settled behind these four MFMAs
                            if (pend >= 0) flush(pend, tt[(S & 1) ^ 1]);
                            pend = PK(S) & 31;
                            par = S & 1;
                        }
                    }
                });
            }
        });
        if (pend >= 0) {
            if (par) flush(pend, tt[1]);
            else flush(pend, tt[0]);
        }
        // row j's tiles that waited in LDS (not negated: +V_jj), dealt over all the waves
        const int rsj = j * T - (j * (j + 1)) / 2;            // first tile of row j, row-major
        if (rsj < n_lds) {      // (uniform)
            const int t1 = rsj + (T - 1 - j) < n_lds ? rsj + (T - 1 - j) : n_lds;
            double vp[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) vp[t] = -va[t];
            if constexpr (GW) {
                const unsigned gl = (unsigned)((dl >> 4) * mp + (dl & 15));      // (from the step's opaque copy: see lb / dl)
                for (int t = rsj + wave; t < t1; t += RW) {
                    const double *Cr = U + (size_t)(16 * j) * mp + 16 * (j + 1 + (t - rsj)) + gl;
                    double b[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) b[q] = Cr[(size_t)(4 * q) * mp];
                    d4 r = {0.0, 0.0, 0.0, 0.0};
                    mfma4_vgpr(r, vp, b);
                    flush(j + 1 + (t - rsj), r);
                }
            } else
            for (int t = rsj + wave; t < t1; t += RW) {
                double b[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) b[q] = Ot[t * 256 + 64 * q + dl];
                d4 r = {0.0, 0.0, 0.0, 0.0};
                mfma4_vgpr(r, vp, b);
                flush(j + 1 + (t - rsj), r);
            }
        }
    };
    // the trailing update of step j on the tiles that wait in LDS (rows > j), dealt over the workers
    auto update_lds_tiles = [&](int j, int lb, int dl) {
        const int rs1 = (j + 1) * T - ((j + 1) * (j + 2)) / 2;      // first tile of row j + 1
        if (rs1 >= n_lds || wave == 0) return;
        if constexpr (GW) {
            int i = j + 1, r0_ = rs1, rl = T - 2 - j;
            const unsigned gl = (unsigned)((dl >> 4) * mp + (dl & 15));
            for (int t = rs1 + (wave - 1); t < n_lds; t += NW) {
                while (t >= r0_ + rl) { r0_ += rl; --rl; ++i; }
                const int k = i + 1 + (t - r0_);
                double *Cp = U + (size_t)(16 * i) * mp + 16 * k + gl;
                double c[4], a[4], b[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) c[q] = Cp[(size_t)(4 * q) * mp];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[q] = -P[(4 * q) * ldp + 16 * i + lb];
                    b[q] = P[(4 * q) * ldp + 16 * k + lb];
                }
                d4 sacc = {0.0, 0.0, 0.0, 0.0};
                mfma4_vgpr(sacc, a, b);
#pragma unroll
                for (int q = 0; q < 4; ++q) Cp[(size_t)(4 * q) * mp] = c[q] + sacc[q];
            }
            return;
        }
        for (int t = rs1 + (wave - 1); t < n_lds; t += NW) {
            int i = j + 1, r0_ = rs1, rl = T - 2 - j;
            while (t >= r0_ + rl) { r0_ += rl; --rl; ++i; }
            const int k = i + 1 + (t - r0_);
            d4 c;
            double a[4], na[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) c[q] = Ot[t * 256 + 64 * q + dl];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = P[(4 * q) * ldp + 16 * i + lb];
                b[q] = P[(4 * q) * ldp + 16 * k + lb];
                na[q] = -a[q];
            }
            d4 sacc = {0.0, 0.0, 0.0, 0.0};
            mfma4_vgpr(sacc, na, b);
#pragma unroll
            for (int q = 0; q < 4; ++q) Ot[t * 256 + 64 * q + dl] = c[q] + sacc[q];
        }
    };
    // the trailing update of step j on this wave: the diagonal tiles beyond the look-ahead one (tile i by worker
    // 1 + i % NW), then its live tiles, slots s_hi .. s_end-1; the MFMA operands of slot S+1 are fetched from the LDS
    // panel before slot S's four MFMAs issue
    auto trailing_update = [&](int j, int s_hi, int lb, int dl) {
        if (wave > 0)
            for (int i = j + 2 + (wave - 1 + NW * T - (j + 2)) % NW; i < T; i += NW) diag_update(i, lb, dl);
        if (n_lds > 0) update_lds_tiles(j, lb, dl);
        GPRF_STAMP3(1)
        auto opnd_load = [&](int pks, double (&oa)[4], double (&ob)[4]) {
            int pc = pks < 0 ? 0 : pks;
            const double *Pa = P + lb + 16 * (pc >> 5);
            const double *Pk = P + lb + 16 * (pc & 31);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                oa[t] = Pa[(4 * t) * ldp];
                ob[t] = Pk[(4 * t) * ldp];
            }
        };
        if (s_hi < s_end) {
            // walked from the LAST slot down: the slot index stays a compile-time constant (register numbers)
            // in straight-line code, the dead slots (below s_hi) are never visited, and one compare per tile
            // ends the walk
            double oa[2][4], ob[2][4];
            d4 tt[2];      // the two temporary product tiles (slot parity)
            tt[0] = tt[1] = d4{0.0, 0.0, 0.0, 0.0};
            opnd_load(__builtin_amdgcn_readlane(pkv, s_end - 1), oa[0], ob[0]);
#pragma unroll
            for (int t = 0; t < 4; ++t) { oa[1][t] = oa[0][t]; ob[1][t] = ob[0][t]; }
            bool done = false;
            // in groups of 8 slots, so that the slots above s_end (small units) and below s_hi (late steps)
            // cost one compare per group instead of one per slot
            static_for<0, (SLOTS + 7) / 8>([&](auto gc) {
                constexpr int G = (SLOTS + 7) / 8 - 1 - decltype(gc)::value;
                int hi = s_hi, end = s_end;
                asm volatile("" : "+s"(hi), "+s"(end));     // (keeps the compares from being hoisted)
                if (!done && end > 8 * G) {
                    static_for<0, 8>([&](auto sc) {
                        constexpr int S = 8 * G + 7 - decltype(sc)::value;
                        if constexpr (S < SLOTS) {
                            int hi2 = hi, end2 = end;
                            asm volatile("" : "+s"(hi2), "+s"(end2));
                            if (!done && S < end2) {
                                if (S < hi2) {
                                    done = true;
                                } else {
                                    ttile_mfma_first(tt[S & 1], oa[S & 1], ob[S & 1], pkv);
                                    if constexpr (S > 0) opnd_load(PK(S - 1), oa[(S - 1) & 1], ob[(S - 1) & 1]);
                                    ttile_mfma_rest(tt[S & 1], oa[S & 1], ob[S & 1]);
                                    // the slot before this one in the walk (S + 1, when it was live): its products have
                                    // settled by now — into its accumulator, behind this slot's MFMAs (issued piecewise
                                    // BETWEEN the MFMAs it was slower: 126 vs 123 us)
                                    if constexpr (S + 1 < SLOTS) {
                                        if (S + 1 < end2) atile_add<S + 1>(tt[(S + 1) & 1]);
                                    }
                                }
                            }
                        }
                    });
                }
            });
            // the last slot of the walk (s_hi): wait for its products, then into its accumulator
            atile_settle();
            static_for<0, SLOTS>([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                if (S == s_hi) atile_add<S>(tt[S & 1]);
            });
        }
    };
    for (int j = 0; j + 1 < T; ++j) {
        if (s_fail) break;
        // keep the per-slot tile coordinates and LDS addresses from being hoisted out of the step loop (they are
        // loop invariant, and 18 slots of them would push the accumulators out of the register file)
        int lb = lg * ldp + lr, dl = dlane;
        asm volatile("" : "+v"(lb));
        asm volatile("" : "+v"(dl));
        asm volatile("" : "+v"(pkv));
        // this worker's slots [s_lo, s_hi) hold tiles of row j, [s_hi, s_end) the live tiles below it
        asm volatile("" : "+v"(shv));
        const int s_lo = j > 0 ? __builtin_amdgcn_readlane(shv, j - 1) : 0;
        const int s_hi = __builtin_amdgcn_readlane(shv, j);
        if (wave == 0) {
            // U_jj (published in LDS by the last look-ahead) -> global, off the critical path
            for (int e = dl; e < 256; e += 64) U[(size_t)(16 * j + (e >> 4)) * mp + 16 * j + (e & 15)] = Ud[e];
        }
        if (mine) {
            solve_rows(j, s_lo, s_hi, lb, dl, Vd0);
            GPRF_STAMP2(2)
        }
        GPRF_STAMP(0)
        GPRF_STAMP2(3)
        if constexpr (GW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the waiting tiles' stores: visible to the other waves
        lds_barrier();
        GPRF_STAMP(1)
        GPRF_STAMP2(4)
        if (wave == 0) {
            diag_update(j + 1, lb, dl);
            factor_publish(j + 1);
        }
        GPRF_STAMP3(3)
        if (mine) {
            GPRF_STAMP3(0)
            trailing_update(j, s_hi, lb, dl);
        }
        GPRF_STAMP(2)
        GPRF_STAMP2(5)
        GPRF_STAMP3(2)
        if constexpr (GW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        GPRF_STAMP(3)
        GPRF_STAMP2(6)
    }
#ifdef GPRF_PROFILE
    unsigned long long t_loopend = __builtin_amdgcn_s_memtime();
#endif
#ifdef GPRF_WGTRACE
    if (threadIdx.x == 0) s_tr2 = (double)__builtin_amdgcn_s_memrealtime();
#endif
    if (stamp && lane == 0) {
        for (int k = 0; k < 4; ++k) pl.dbg[(size_t)u * 8 + k] = (double)tacc[k];
        pl.dbg[(size_t)u * 8 + 4] = (double)T;
    }
    if (stamp3 && lane == 0) {
        for (int k = 0; k < 4; ++k) pl.dbg[(size_t)u * 8 + k] = (double)tacc[k];
        pl.dbg[(size_t)u * 8 + 4] = (double)T;
    }
    if (stamp2 && lane == 0) {
        for (int k = 0; k < 7; ++k) pl.dbg[(size_t)u * 8 + k] = (double)tacc[k];
        pl.dbg[(size_t)u * 8 + 7] = (double)T;
    }
#undef GPRF_STAMP
#undef GPRF_STAMP2
#undef GPRF_STAMP3
#undef PK
    __syncthreads();
    if (s_fail) {
        if (threadIdx.x == 0) { pl.info[u] = s_fail; pl.logdet[u] = 0.0; }
        return;
    }
    if (wave == 0) {
        int jt = T - 1;
        for (int e = lane; e < 256; e += 64) U[(size_t)(16 * jt + (e >> 4)) * mp + 16 * jt + (e & 15)] = Ud[e];
    }
    __syncthreads();    // the epilogue reads U_jj back from global
    potrf_epilogue<RW, false>(U, V, P0, dvals, lred, mp, T, u, pl);      // (V_jj went out tile by tile)
#ifdef GPRF_PROFILE
    if (stamp && lane == 0) {   // [5] prologue, [6] epilogue cycles
        pl.dbg[(size_t)u * 8 + 5] = (double)(t_loop - t_start);
        pl.dbg[(size_t)u * 8 + 6] = (double)(__builtin_amdgcn_s_memtime() - t_loopend);
    }
#endif
#ifdef GPRF_WGTRACE
    if ((RW == 8 ? 4 : 5) == GPRF_WGTRACE && threadIdx.x == 0 && (int)blockIdx.x < GPRF_WGTRACE_MAX) {
        double *rec = pl.dbg + (size_t)(ut.n_units > 1 ? ut.n_units : 1) * 8 + (size_t)blockIdx.x * 4;
        unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        rec[0] = s_tr0;
        rec[1] = (double)__builtin_amdgcn_s_memrealtime();
        rec[2] = (double)(((unsigned long long)(xcc & 0xf) << 32) | hw);
        // tag + 1000 * (ticks before the step loop) + 1e7 * (ticks inside it): scripts/gpu_wg_trace.py PHASES=1
        rec[3] = (double)(T * 8 + which) + 1000.0 * (s_tr1 - s_tr0) + 1e7 * (s_tr2 - s_tr1);
    }
#endif
}

// the kernels around the body (an attribute cannot depend on a template parameter).
// Eight waves of 256 registers, ONE workgroup per CU (seven workers x 20 slots): a unit finishes 20 % sooner than it did with
// four waves of 512 registers (T = 15: 89 vs 104-113 us) — as the only kernel it loses (a whole CU per unit: 140 vs 123 us), as
// the kernel of the LARGEST units, which are what the stage waits for, it is in.
template <int SLOTS, bool GEN>
__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_num_vgpr(96))) void k_potrf_reg8(UnitTab ut, Pools pl, int stamps,
                                                                                           int reg_maxT, KParams kp, int which) {
    potrf_reg_body<8, SLOTS, GEN>(ut, pl, stamps, reg_maxT, kp, which);
}
// ... units of 21 .. 32 tiles per edge (and, in a launch that has such units, every smaller one too), K from the pool: the
// tiles beyond the accumulator slots wait in the U pool (GW)
template <int SLOTS>
__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_num_vgpr(96))) void k_potrf_reg8w(UnitTab ut, Pools pl, int stamps,
                                                                                            int reg_maxT, KParams kp, int which, int min_T) {
    potrf_reg_body<8, SLOTS, false, true>(ut, pl, stamps, reg_maxT, kp, which, min_T);
}
// four waves, TWO workgroups per CU: units of up to 13 tiles per edge, K generated
template <int RW, int SLOTS, bool GEN>
__global__ __launch_bounds__(RW * 64, 2) __attribute__((amdgpu_num_vgpr(96))) void k_potrf_reg2(UnitTab ut, Pools pl, int stamps,
                                                                                                int reg_maxT, KParams kp, int which) {
    potrf_reg_body<RW, SLOTS, GEN>(ut, pl, stamps, reg_maxT, kp, which);
}


constexpr int POTRF_REG_WAVES = 4;      // the two-per-CU instantiation: one wave per SIMD and workgroup
constexpr int POTRF_SMALL_SLOTS = 20;   // 4 waves x 20 slots >= 13*12/2 strictly-upper tiles: units up to 208 points
constexpr int POTRF_SMALL_MAXT = 13;
// (the eight-wave instantiation — the large-unit kernel of the two-queue SE path, and the non-generating register kernel —
// takes units of up to POTRF_REG8_MAXT tiles per edge: 140 .. 160 tiles in accumulators, up to 30 more waiting in LDS)

// Where a unit's kernel matrix comes from is decided PER UNIT: units of at most potrf_gen_maxT() tiles per edge (320
// points) have it generated inside the register-resident Cholesky (SE kernel) — k_fill skips them and K never exists in
// HBM for them — larger ones are filled into the K pool.  (Round 2 decided per launch: one pair growing past the limit
// during an optimisation sent all 442 units of the north-star configuration through the K pool: +36 us fill, +12 us in the
// Cholesky.)  k_mgrad re-evaluates the values it needs in both cases.
// diag potrf_reg=0: every unit through the generic kernel (tests: the register kernels against it, bit for bit)
static bool potrf_use_reg() { return diag("potrf_reg", 1) != 0; }
// units of 21 .. 32 tiles per edge on the eight-wave kernel with its waiting tiles in the U pool (diag potrf_gw=0: the generic
// kernel)
static bool potrf_gw() { return diag("potrf_gw", 1) != 0; }
bool potrf_generates_K(int dist_id, int kern_id, const UnitTab &ut) {
    // (diag fused_fill=0: always fill the K pool.  ("lld","matern32") generation inside the register kernel was built in
    // round 3 and measured a loss — the unary blocks' kernel spent 90 us generating, haversine + asin + two square roots + exp
    // per entry on four lone waves, in front of the generic kernel instead of 52 us reading: Cholesky stage 250 -> 289 us —
    // and is gone since round 5)
    const bool se = dist_id == 0 && kern_id == 0;
    // a launch with units of more than 20 tiles per edge goes through the K pool as a whole: ONE eight-wave kernel then takes
    // every unit of up to 32 tiles (waiting tiles in the U pool) — behind the generating kernels it would run by itself, a
    // unit's whole chain later (measured, 49 blocks of ~184 points + 156 pairs of 20-27 tiles: fill + Cholesky 32 + 281 us
    // against 28 + 439)
    // — when such units are MANY (an eighth of the launch, at least 16).  A few (one pair of a north-star-shaped partition
    // growing past 320 points) leave the others generated, as round 2 decided per unit: they are filled, and take the
    // eight-wave kernel (up to 32 tiles) or the generic one behind the generating kernels.
    if (ut.max_T > 20 && potrf_gw() && ut.n_wide >= 16 && 8 * ut.n_wide >= ut.n_ids) return false;
    return diag("fused_fill", 1) != 0 && se && ut.n_ids > 0 && potrf_use_reg();
}
int potrf_gen_maxT(int) { return POTRF_REG8_MAXT; }
int potrf_small_maxT() { return POTRF_SMALL_MAXT; }

static void launch_reg8w(dim3 grid, hipStream_t s, const UnitTab &ut, const Pools &p, int stamps, const KParams &kp, int min_T) {
    const int capT = ut.max_T < POTRF_REG8W_MAXT ? ut.max_T : POTRF_REG8W_MAXT;
    const size_t lds = (size_t)(16 * POTRF_REG8W_LDP + 256 + 16 + 256 + 16 * POTRF_REG8W_MAXT + 256 * capT) * sizeof(double);
    if (lds_needs_optin(9, lds))
        (void)hipFuncSetAttribute((const void *)k_potrf_reg8w<POTRF_SMALL_SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_potrf_reg8w<POTRF_SMALL_SLOTS>), grid, dim3(512), lds, s, ut, p, stamps, POTRF_REG8W_MAXT, kp, 0, min_T);
}
// LDS of the eight-wave instantiation for units of up to capT tiles per edge (doubles)
static size_t potrf_reg8_lds(int capT, int xs_stride) {
    const int total = capT * (capT - 1) / 2, n_lds = total > 8 * 20 ? total - 8 * 20 : 0;
    return (size_t)(16 * POTRF_REG8_LDP + 256 + 16 + 256 + 16 * POTRF_REG8_MAXT + 256 * capT + 16 * capT * xs_stride + 256 * n_lds);
}
static void launch_reg2(dim3 grid, size_t lds, hipStream_t s, const UnitTab &ut, const Pools &p, int stamps, int maxT,
                        const KParams &kp, int which) {
    if (lds_needs_optin(3, lds))
        (void)hipFuncSetAttribute((const void *)k_potrf_reg2<POTRF_REG_WAVES, POTRF_SMALL_SLOTS, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_potrf_reg2<POTRF_REG_WAVES, POTRF_SMALL_SLOTS, true>), grid, dim3(POTRF_REG_WAVES * 64), lds, s, ut,
                       p, stamps, maxT, kp, which);
}
static void launch_reg8(dim3 grid, size_t lds, hipStream_t s, const UnitTab &ut, const Pools &p, int stamps, int maxT,
                        const KParams &kp, int which, bool gen) {
    if (!gen) {
        if (lds_needs_optin(8, lds))
            (void)hipFuncSetAttribute((const void *)k_potrf_reg8<POTRF_SMALL_SLOTS, false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_potrf_reg8<POTRF_SMALL_SLOTS, false>), grid, dim3(512), lds, s, ut, p, stamps, maxT, kp, which);
    } else {
        if (lds_needs_optin(4, lds))
            (void)hipFuncSetAttribute((const void *)k_potrf_reg8<POTRF_SMALL_SLOTS, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_potrf_reg8<POTRF_SMALL_SLOTS, true>), grid, dim3(512), lds, s, ut, p, stamps, maxT, kp, which);
    }
}

// Any environment that may serialise dispatches across queues — a profiler or debug agent loaded into the runtime,
// serialised / blocking launches — gets the fork and the join of the two Cholesky queues as EVENTS: dependencies the runtime
// itself resolves (slower: stage 131 vs 110 us), where a stream wait on a word that a kernel of the other queue writes would
// never return.  (finish_eval bounds its wait all the same.)  Reported by gprf_runtime_config(), so that a trace taken under
// a tool is labelled with the launch structure it shows.
bool potrf_tool_env() {
    static const bool tool_env = [] {
        // (diag tool_env=0: a tool is loaded but known not to serialise the queues — rocprofv3 --kernel-trace without counters:
        // the trace then shows the product's own launch structure, scripts/profile_run.sh)
        if (diag("tool_env", -1) == 0) return false;
        for (const char *v : {"HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "AMD_SERIALIZE_KERNEL",
                              "HIP_LAUNCH_BLOCKING", "ROCPROF_COUNTER_COLLECTION", "GPRF_SIDE_EVENTS"}) {
            const char *e = getenv(v);
            if (e && e[0] && !(e[0] == '0' && e[1] == 0)) return true;
        }
        return false;
    }();
    return tool_env;
}
// 4 = the large-unit kernel's first workgroup writes the word the side queue waits for + join by stream memory operation (the
// product path); 0 = events both ways (under a tool; diag side_events=1).  (Round 2 measured the mixtures — memory operations
// both ways 177 us, fork by memory operation + join by event 181, fork by event + join by memory operation 121 — gone.)
int potrf_side_mode() { return (potrf_tool_env() || diag("side_events", 0)) ? 0 : 4; }
// rocprofv3 collecting hardware counters serialises the dispatches of ALL queues, and the stream-memory-operation wait that
// joins the two queues in front of the solve would never see its value written (observed: the run hangs): both
// instantiations then go one after the other on the main queue (diag one_queue=1: the same, for standalone durations)
static bool potrf_one_queue() {
    static const bool counters = [] {
        const char *c = getenv("ROCPROF_COUNTER_COLLECTION");
        return c && c[0] && c[0] != '0' && c[0] != 'F' && c[0] != 'f';
    }();
    return counters || diag("one_queue", 0) != 0;
}

int launch_potrf(const UnitTab &ut, const Pools &p, const KParams &kp, bool gen, hipStream_t s, const SideQueue &side, int class_stages,
                 int want_gc, hipStream_t *tail) {
    if (tail) *tail = s;
    if (ut.n_ids == 0) return 0;
    hipStream_t s2 = side.s2;
    const int stamps = diag("potrf_stamps", 0);      // diagnostic builds (-DGPRF_PROFILE): in-kernel cycle stamps into Pools::dbg
    // every unit of up to 20 tiles per edge on the register-resident kernels (28 with its waiting tiles in the U pool), the
    // generic kernel above that
    int reg_maxT = potrf_use_reg() ? POTRF_REG8_MAXT : 0;
    // units of more than reg_maxT tiles per edge: the generic kernel, from the K pool (its workgroups leave the others alone)
    auto launch_generic = [&]() {
        // (units beyond BIG_LA_T tiles belong to the blocked path: with the register kernels reaching that limit nothing is
        // left for k_potrf — every workgroup of such a launch would start, with ~70 KB of LDS, only to leave)
        auto nothing_left = [&]() { return (ut.max_T < BIG_LA_T ? ut.max_T : BIG_LA_T) <= reg_maxT; };
        if (nothing_left()) return;
        // the few units of 21 .. 32 tiles per edge of a generating launch: the eight-wave kernel with its waiting tiles in the
        // U pool, from the K pool (they were filled), behind the generating kernels; the generic kernel above that
        if (gen && potrf_gw() && reg_maxT == POTRF_REG8_MAXT) {
            launch_reg8w(dim3(ut.n_ids), s, ut, p, stamps, kp, POTRF_REG8_MAXT + 1);
            reg_maxT = POTRF_REG8W_MAXT;
            if (nothing_left()) return;
        }
        const int capG = ut.max_T < BIG_LA_T ? ut.max_T : BIG_LA_T;      // (larger units: launch_big_potrf)
        size_t ldsg = (size_t)(16 * (16 * capG + 16) + 256 + 16 + 16 * 17 + 256 + 16 * capG) * sizeof(double);
        if (lds_needs_optin(1, ldsg))
            (void)hipFuncSetAttribute((const void *)k_potrf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsg);
        hipLaunchKernelGGL(k_potrf, dim3(ut.n_ids), dim3(POTRF_WAVES * 64), ldsg, s, ut, p, stamps, reg_maxT);
    };
    if (!reg_maxT) {
        launch_generic();
        return 0;
    }
    const int capT = ut.max_T < reg_maxT ? ut.max_T : reg_maxT;
    if (!gen) {
        // the K pool's units of up to 20 tiles, eight waves a unit, one launch over the launch order (longest units first)
        if (ut.max_T > POTRF_REG8_MAXT && potrf_gw()) {
            // (a launch with units above 20 tiles: ONE instantiation for everything of up to 28 — two launches on one
            // stream would run one after the other)
            launch_reg8w(dim3(ut.n_ids), s, ut, p, stamps, kp, 0);
            reg_maxT = POTRF_REG8W_MAXT;
        } else
            launch_reg8(dim3(ut.n_ids), potrf_reg8_lds(capT, 0) * sizeof(double), s, ut, p, stamps, reg_maxT, kp, 0, false);
        launch_generic();
        return 0;
    }
    if (ut.max_T <= POTRF_SMALL_MAXT) {      // every unit has at most 13 tiles: the two-per-CU kernel alone
        size_t ldsS = (size_t)(16 * POTRF_REG2_LDP + 256 + 16 + 256 + 16 * POTRF_REG_MAXT_C + 256 * capT + 16 * capT * XPAD) * sizeof(double);
        launch_reg2(dim3(ut.n_ids), ldsS, s, ut, p, stamps, POTRF_SMALL_MAXT, kp, 0);
        return 0;
    }
    // two instantiations side by side on two queues: units of up to 13 tiles per edge two to a CU, the larger ones one to a
    // CU; each over its own device-built list (an early-exit workgroup of the eight-wave kernel still needs an EMPTY CU to
    // be scheduled and would stall behind the two-per-CU kernel's residents: the grids follow the list lengths of the last
    // synchronised partition with a little slack)
    if (potrf_one_queue() || !s2) s2 = s;
    const int capS = POTRF_SMALL_MAXT;
    const size_t ldsS = (size_t)(16 * POTRF_REG2_LDP + 256 + 16 + 256 + 16 * POTRF_REG_MAXT_C + 256 * capS + 16 * capS * XPAD) * sizeof(double);
    // measured on the north-star configuration (stage time, us): events both ways 131; no fork command — the large-unit
    // kernel's first workgroup writes the word the side queue waits for — + join by memory operation: 110 (an event fork
    // costs 12 us, all of it in front of the small-unit kernel, which finishes last)
    const bool values = side.words && potrf_side_mode() == 4;
    const bool fork_kernel = values && ut.grid_big > 0 && s2 != s;      // (only when that kernel is really launched)
    UnitTab utb = ut;
    if (fork_kernel) { utb.fork_flag = side.words + 2; utb.fork_seq = side.seq; }
    if (s2 != s && !fork_kernel) {      // fork: the side queue starts when everything enqueued on s so far is done
        (void)hipEventRecord(side.ev_fork, s);
        (void)hipStreamWaitEvent(s2, side.ev_fork, 0);
    }
    // (the large-unit kernel must go FIRST and on the main queue: launched behind the two-per-CU kernel it waits for whole
    // CUs to drain — measured: stage 178-264 us instead of 121)
    // round 6: each class's forward substitution behind its own Cholesky kernel, on that kernel's queue; the join below then
    // waits for both (nothing else of this launch may be left for the generic kernel: max_T <= 16)
    // (how deep: diag class_depth=<1..3>; the gradient kernel by class exists for the SE kernel's two-dimensional instantiations)
    int depth = (s2 != s && solve_by_class(ut)) ? class_stages : 0;
    {
        // (a launch more than two rounds of CUs deep — C4: 4033 units — keeps its gradient kernel launch-wide: that kernel
        // walks such launches in groups of 64 slots for its L2 residency, and by class it measured 2.60-2.64 ms per evaluation
        // against 2.55-2.56 with the first two stages alone and 2.64-2.72 with none)
        const int cap = diag("class_depth", ut.n_launch <= 2 * device_cus() ? 3 : 2);
        if (depth > cap) depth = cap;
        if (depth > 2 && kp.dx > 2) depth = 2;
    }
    const bool by_class = depth >= 1;
    if (ut.grid_big > 0)
        launch_reg8(dim3(ut.grid_big), potrf_reg8_lds(capT, XPAD) * sizeof(double), s, utb, p, stamps, reg_maxT, kp, 1, true);
    if (fork_kernel) {
        // (should that launch ever be refused, nothing would write the word the side queue waits for)
        if (hipPeekAtLastError() != hipSuccess) (void)hipStreamWriteValue32(s, side.words + 2, side.seq, 0);
        // The wait goes in BEHIND the kernel that satisfies it, in host order: HIP streams share a few hardware queues, which
        // drain in submission order — a wait submitted ahead of its writer blocks the writer whenever the two streams land on
        // the same hardware queue (observed: ten contexts enqueued back to back hang).  Every wait in this file depends on
        // something submitted earlier.
        (void)hipStreamWaitValue32(s2, side.words + 2, side.seq, hipStreamWaitValueGte, 0xffffffffu);
    }
    if (ut.grid_small > 0) launch_reg2(dim3(ut.grid_small), ldsS, s2, ut, p, stamps, POTRF_SMALL_MAXT, kp, 2);
    // (BOTH Cholesky kernels are submitted before any later stage: a launch costs the host 3-5 us, and with the large class's
    // three later kernels submitted in front of it the side queue's Cholesky started 25 us behind the main queue's —
    // rocprofv3 kernel trace, scripts/trace_eval_timeline.py.  The small class is the longer pipeline: its stages go in first.)
    if (by_class) launch_solve_class(ut, p, kp.dy, 2, s2);
    if (by_class) launch_solve_class(ut, p, kp.dy, 1, s);
    if (depth >= 2) launch_at_class(ut, p, 2, s2);
    if (depth >= 2) launch_at_class(ut, p, 1, s);
    if (depth >= 3) launch_grad_class(ut, p, kp, want_gc, 2, s2);
    if (depth >= 3) launch_grad_class(ut, p, kp, want_gc, 1, s);
    if (s2 != s && tail && depth >= 3 && values && diag("tail_swap", 1) != 0) {
        // the join INTO the side queue (see the declaration): the wait is submitted behind the write that satisfies it
        (void)hipStreamWriteValue32(s, side.words + 11, side.seq, 0);
        (void)hipStreamWaitValue32(s2, side.words + 11, side.seq, hipStreamWaitValueGte, 0xffffffffu);
        *tail = s2;
        return depth;      // (nothing of a by-class launch is left for the generic kernel)
    }
    if (s2 != s) {      // join
        if (values) {
            (void)hipStreamWriteValue32(s2, side.words + 1, side.seq, 0);
            (void)hipStreamWaitValue32(s, side.words + 1, side.seq, hipStreamWaitValueGte, 0xffffffffu);
        } else {
            (void)hipEventRecord(side.ev_join, s2);
            (void)hipStreamWaitEvent(s, side.ev_join, 0);
        }
    }
    launch_generic();
    return depth;
}

}  // namespace gprf
