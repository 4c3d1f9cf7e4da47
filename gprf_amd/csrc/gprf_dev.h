// gprf_dev.h — device-side helpers shared by the kernel files (gprf_fill / gprf_potrf / gprf_solve / gprf_mgrad / gprf_big /
// gprf_tables .hip): the one MFMA form, wave shuffles, the launch-slot record, the grid maps, the covariance functions
// (treegp side of gprf.py:333-375) and the 16 x 16 diagonal-tile factor / inverse every Cholesky kernel shares.  Everything
// here is __device__ __forceinline__ or constexpr; the few host helpers the launchers share are declared at the end and
// defined in gprf_tables.hip.  gfx950 only.
//
//
// One "unit" = one block or one concatenated neighbouring block pair (gprf.py:299-330).  Per unit the
// reference computes (gprf.py:496-591)  K = k(X,X)+nv I ; chol ; K^-1 ; A = K^-1 Y ; ll ; gradX ; gradC.
// Here every unit matrix is padded to mp = 16*T rows and the whole pipeline is written in the one GEMM
// form the f64 MFMA (v_mfma_f64_16x16x4_f64) consumes without any transposition:
//
//        D[i][j] (+)= sum_k SA[k][i] * SB[k][j]        SA, SB, D all ROW-major, k = slow index
//
// lane l of a wave (lr = l & 15, lg = l >> 4) feeds  a = SA[4s+lg][lr],  b = SB[4s+lg][lr]  for k-step s and
// owns D[lg + 4q][lr], q = 0..3.  Consecutive lanes therefore always touch consecutive doubles (128-B
// segments from HBM/L2, conflict-free 256-B rows from LDS), and an accumulator register q IS the B operand
// of k-step q of the next product (rows 4q+lg) — tiles chain through registers.
//
// In that form:   K = U^T U            (upper Cholesky, U row-major)              k_potrf
//                 W = U^-T, Z = U^-T Y (forward substitution on [I | Y])          k_solve_panel (k_solve for m > 288)
//                 At = Z^T W = (K^-1 Y)^T                                         k_at
//                 M = At^T At - dy * W^T W  ( = A A^T - dy K^-1 ), lower-triangle tiles,      k_mgrad
//                 M reduced against dk/dx and dk/dtheta into gradX / gradC partials (same kernel)
// Reference identities:  gX[p,i] = sum_q M[p,q] dk(x_p,x_q)/dx_p[i]  (gprf.py:556-573),
//                        gC[t]   = 1/2 sum_pq M[p,q] dK_pq/dtheta_t  (gprf.py:577-584),
//                        ll      = -1/2 ||Z||_F^2 - dy sum log U_kk - 1/2 dy m log 2pi (gprf.py:542-544).
#pragma once
#include "gprf_kernels.h"
#include <type_traits>

#include <cstdlib>
#include <cstring>

namespace gprf {

int diag(const char *key, int dflt);      // GPRF_DIAG="key=value,...": see the definition

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ double readlane_d(double x, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// wave shuffles from the thread index (workgroups are one-dimensional multiples of 64 here): HIP's __shfl* derive
// the lane from v_mbcnt, which the compiler hoists out of loops and keeps alive across them — in the register-starved
// Cholesky instantiation it parked that value in an accumulator register (tests/test_isa_invariants.py)
__device__ __forceinline__ int shfl_i(int v, int src_lane) {
    return __builtin_amdgcn_ds_bpermute(src_lane << 2, v);
}
__device__ __forceinline__ double shfl_xor_d(double x, int mask) {
    int idx = (((int)threadIdx.x & 63) ^ mask) << 2;
    int lo = __builtin_amdgcn_ds_bpermute(idx, __double2loint(x));
    int hi = __builtin_amdgcn_ds_bpermute(idx, __double2hiint(x));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int pad16(int m) { return (m + 15) & ~15; }

// Workgroup barrier that orders LDS traffic only: waits for this wave's LDS operations (lgkmcnt) but NOT for
// its outstanding global stores (vmcnt), which __syncthreads() would also drain (~1 us of store-acknowledge
// latency per barrier on a loaded chip).  Use only where nothing written to global memory before the barrier
// is read back inside the kernel.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// XCD-aware 1-D grid -> (unit slot, part): workgroups are dealt round-robin over the 8 XCDs (each with its own
// 4 MiB L2), so all `nparts` workgroups of one unit are given linear ids that are equal mod 8: they land on one
// XCD and the unit's matrices are pulled from HBM once.  Launch with xcd_grid(n_ids, nparts) workgroups.
// The unit of a launch slot from its 16-byte record (see SlotRec): one load, everything uniform.
struct UnitRef { int u, m, row_off; size_t mat_off; };
__device__ __forceinline__ UnitRef unit_ref(const SlotRec *__restrict__ rec, int slot) {
    typedef int i4 __attribute__((ext_vector_type(4)));
    i4 r = *reinterpret_cast<const i4 *>(rec + slot);
    UnitRef x;
    x.u = __builtin_amdgcn_readfirstlane(r.x);
    x.m = __builtin_amdgcn_readfirstlane(r.y);
    x.row_off = __builtin_amdgcn_readfirstlane(r.z);
    x.mat_off = (size_t)(unsigned)__builtin_amdgcn_readfirstlane(r.w) << 8;
    return x;
}

// The Cholesky's two size classes as launch lists of their own (round 6).  The tables are built on the device with the
// partition: big_rec / small_rec hold the units of more than / at most potrf_small_maxT() tiles in launch order, their
// lengths are ctl[CTL_NBIG] / ctl[CTL_NSMALL]; the large-unit Cholesky kernel's surplus workgroups (grid_big beyond its list)
// take units from the END of the small list.  CLS 1 = the large list + those surplus units (everything the main queue's
// Cholesky kernel factors), CLS 2 = the rest of the small list (the side queue's), CLS 0 = the whole launch order.
struct ClassList { int n, nb, ns; };
template <int CLS>
__device__ __forceinline__ ClassList class_list(const UnitTab &ut) {
    ClassList c{ut.n_ids, 0, 0};
    if constexpr (CLS != 0) {
        c.nb = ut.ctl[CTL_NBIG];
        c.ns = ut.ctl[CTL_NSMALL];
        int surplus = ut.grid_big > c.nb ? ut.grid_big - c.nb : 0;
        if (surplus > c.ns) surplus = c.ns;
        c.n = CLS == 1 ? c.nb + surplus : c.ns - surplus;
    }
    return c;
}
template <int CLS>
__device__ __forceinline__ UnitRef class_unit(const UnitTab &ut, const ClassList &c, int slot) {
    if constexpr (CLS == 0) return unit_ref(ut.srec, slot);
    else if constexpr (CLS == 2) return unit_ref(ut.small_rec, slot);
    else return slot < c.nb ? unit_ref(ut.big_rec, slot) : unit_ref(ut.small_rec, c.ns - 1 - (slot - c.nb));
}

// Diagnostic builds (-DGPRF_WGTRACE=<id>: 1 solve, 2 at, 3 mgrad, 4 / 5 the Cholesky's 512- / 256-register kernel): every workgroup of that kernel records when and where
// it ran — (start, end) of the constant-rate counter, the HW_ID / XCC_ID registers — for scripts/gpu_wg_trace.py.
struct WgTrace {
#ifdef GPRF_WGTRACE
    unsigned long long t0;
    double *rec;
    __device__ __forceinline__ WgTrace(const UnitTab &ut, const Pools &pl, int id) {
        rec = nullptr;
        if (id == GPRF_WGTRACE && threadIdx.x == 0 && (int)blockIdx.x < GPRF_WGTRACE_MAX)
            rec = pl.dbg + (size_t)(ut.n_units > 1 ? ut.n_units : 1) * 8 + (size_t)blockIdx.x * 4;
        t0 = __builtin_amdgcn_s_memrealtime();
    }
    __device__ __forceinline__ void done(int tag) {
        if (rec) {
            unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            rec[0] = (double)t0;
            rec[1] = (double)__builtin_amdgcn_s_memrealtime();
            rec[2] = (double)(((unsigned long long)(xcc & 0xf) << 32) | hw);
            rec[3] = (double)tag;
        }
    }
#else
    __device__ __forceinline__ WgTrace(const UnitTab &, const Pools &, int) {}
    __device__ __forceinline__ void done(int) {}
#endif
};

// The same grid walked part by part: all units' part 0, then all units' part 1, ... (a unit's workgroups still land on one
// XCD: the slot count is padded to a multiple of 8).  For launches several rounds of workgroups deep whose parts differ in
// length: with the longest kind of part first across ALL units the launch order is longest-first by workgroup, not by unit,
// and the tail of the launch is made of short workgroups.
// G > 0 (a multiple of 8): part by part inside GROUPS of G launch slots, group after group — the parts of one unit then run
// within about one round of workgroups of each other and find the unit's matrices still in their XCD's L2 (launch-wide, a
// unit's parts are a whole round apart and every one fetches them again: 2.4x the algorithmic bytes in the gradient kernel)
__device__ __forceinline__ bool part_major_map(int linear, int n_ids, int nparts, int G, int *slot, int *part) {
    int n8 = (n_ids + 7) & ~7;
    if (G <= 0 || G >= n8) {
        *part = linear / n8;
        *slot = linear - *part * n8;
        return *slot < n_ids;
    }
    int per = G * nparts;
    int g = linear / per, rem = linear - g * per;
    *part = rem / G;
    *slot = g * G + (rem - *part * G);
    return *slot < n_ids;
}
__device__ __forceinline__ bool xcd_map(int linear, int n_ids, int nparts, int *slot, int *part) {
    int grp = linear / (8 * nparts);
    int rem = linear - grp * (8 * nparts);
    *part = rem >> 3;
    *slot = 8 * grp + (rem & 7);
    return *slot < n_ids;
}

// ------------------------------------------------------------------------------------------------
// distance / covariance functions (treegp side of gprf.py:333-375; definitions SURVEY.md §8a)
// ------------------------------------------------------------------------------------------------
constexpr double EARTH_R_KM = 6371.0;  // run_seismic.py:52
constexpr double DEG2RAD = 0.017453292519943295769;
constexpr double SQRT3 = 1.7320508075688772935;

template <int DIST, int KERN>
struct KernFn;

// exp(x) for the covariance functions' arguments (x <= 0 in exact arithmetic; any finite x works): n = rint(x log2 e),
// r = x - n ln2 (two-piece ln2, |r| <= 0.347), Taylor polynomial of degree 13 (truncation 4e-18 relative) summed as
// 1 + (r + r^2 q(r)), ldexp.
// The library exp() spends half of its ~40 instructions moving polynomial coefficients into VGPRs for v_fmac; here
// every Horner step is one v_fma_f64 with the coefficient as a scalar operand — 20 instructions.  That matters where
// a wave is alone on its SIMD and generates kernel values itself (k_potrf_reg<.,.,true>).  NaN stays NaN, anything
// below -745.2 (the smallest subnormal's logarithm) is 0, including -inf.
__device__ __forceinline__ double fma_sc(double a, double b, double c_scalar) {
    double d;
    // (volatile: the step-major order of exp_fast_v is the point, the scheduler would re-serialise the chains)
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_scalar));
    return d;
}
// N independent arguments, written step-major: the N dependent chains advance together (a lone wave has nothing
// else to hide the latency of a v_fma_f64 behind; the compiler does not interleave them by itself)
template <int N>
__device__ __forceinline__ void exp_fast_v(const double (&x)[N], double (&y)[N]) {
    double n[N], r[N], p[N];
#pragma unroll
    for (int i = 0; i < N; ++i) n[i] = __builtin_rint(x[i] * 1.4426950408889634074);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_fma(n[i], -6.93147180369123816490e-01, x[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_fma(n[i], -1.90821492927058770002e-10, r[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = fma_sc(1.0 / 6227020800.0, r[i], 1.0 / 479001600.0);
#define GPRF_EXP_STEP(c)              \
    _Pragma("unroll") for (int i = 0; i < N; ++i) p[i] = fma_sc(p[i], r[i], c);
    GPRF_EXP_STEP(1.0 / 39916800.0)
    GPRF_EXP_STEP(1.0 / 3628800.0)
    GPRF_EXP_STEP(1.0 / 362880.0)
    GPRF_EXP_STEP(1.0 / 40320.0)
    GPRF_EXP_STEP(1.0 / 5040.0)
    GPRF_EXP_STEP(1.0 / 720.0)
    GPRF_EXP_STEP(1.0 / 120.0)
    GPRF_EXP_STEP(1.0 / 24.0)
    GPRF_EXP_STEP(1.0 / 6.0)
    GPRF_EXP_STEP(0.5)
#undef GPRF_EXP_STEP
    // e^r = 1 + (r + r^2 q(r)): the Horner chain's rounding enters scaled by r^2 <= 0.12, the last two roundings are of
    // r + r^2 q (|.| <= 0.41) and of the final sum — under 1 ulp in all (the plain Horner form's last two steps, each
    // rounding a value near 1, left up to 4)
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = __builtin_fma(r[i] * r[i], p[i], r[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = 1.0 + p[i];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double v = __builtin_ldexp(p[i], (int)n[i]);
        y[i] = x[i] < -745.2 ? 0.0 : v;
    }
}
__device__ __forceinline__ double exp_fast(double x) {
    double xa[1] = {x}, ya[1];
    exp_fast_v<1>(xa, ya);
    return ya[0];
}

// ("euclidean","se"):  r^2 = sum(((a-b)/l)^2),  k = sv exp(-r^2).
// treegp forms d = sqrt(r^2) with a divide per coordinate and then exp(-d*d); here the scaled differences use the
// host-rounded reciprocal lengthscales and r^2 goes straight into exp: at most ~2 ulp apart in the exponent's
// argument (relative 2e-16 * r^2 in k), the same size as the exp implementations' own disagreement.
// -(r^2) with ONE order of roundings wherever an SE kernel value is made — the fill, the generation inside the register
// Cholesky, the re-evaluation in the gradient kernel, neighbour discovery: d_i = (a_i - b_i) * (1 / l_i), d_0^2 rounded, the
// others added fused, in order.  Written with explicit operations: under -ffp-contract=fast "d0 * d0 + d1 * d1" may fuse
// EITHER product, and two kernels that spell the same sum differently came out 1 ulp apart in 15 % of the arguments — up to
// 16 ulp in exp(-r^2) (round 4: k_fill_se against k_fill<0,0>, tests/diag/gpu_fill_compare.py).  Unused coordinates are 0
// in every caller's records; three_d = false skips the third term (adding (0 - 0)^2 changes no bit).
__device__ __forceinline__ double se_neg_r2(double a0, double a1, double a2, double b0, double b1, double b2, const double (&inv)[3],
                                            bool three_d) {
    double d0 = __dmul_rn(__dsub_rn(a0, b0), inv[0]);
    double d1 = __dmul_rn(__dsub_rn(a1, b1), inv[1]);
    double sq = __dmul_rn(d0, d0);
    sq = __builtin_fma(d1, d1, sq);
    if (three_d) {
        double d2 = __dmul_rn(__dsub_rn(a2, b2), inv[2]);
        sq = __builtin_fma(d2, d2, sq);
    }
    return -sq;
}

template <>
struct KernFn<0, 0> {
    // (xi, xj: three coordinates each, unused ones 0)
    __device__ static __forceinline__ double value(const KParams &p, const double *xi, const double *xj) {
        return p.sv * exp_fast(se_neg_r2(xi[0], xi[1], xi[2], xj[0], xj[1], xj[2], p.inv_ls, p.dx > 2));
    }
    // k, d k(xj, xi)/d xj[d], d k / d ls[t]
    __device__ static __forceinline__ double full(const KParams &p, const double *xi, const double *xj,
                                                  double *dkdxj, double *dkdl) {
        double k = value(p, xi, xj);
        for (int d = 0; d < p.dx; ++d) {
            double delta = xj[d] - xi[d];
            double l = p.ls[d];
            dkdxj[d] = -2.0 * delta / (l * l) * k;
            dkdl[d] = 2.0 * delta * delta / (l * l * l) * k;
        }
        return k;
    }
    // both ends' derivatives from a kernel value already in hand (have_k) or recomputed
    __device__ static __forceinline__ double pair(const KParams &p, const double *xi, const double *xj, bool have_k,
                                                  double kval, double *dkdxi, double *dkdxj, double *dkdl) {
        double k = have_k ? kval : value(p, xi, xj);
        for (int d = 0; d < p.dx; ++d) {
            double delta = xj[d] - xi[d];
            double l = p.ls[d];
            double t = -2.0 * delta / (l * l) * k;
            dkdxj[d] = t;
            dkdxi[d] = -t;
            dkdl[d] = 2.0 * delta * delta / (l * l * l) * k;
        }
        return k;
    }
};

// ("lld","matern32"):  r = sqrt((g/l0)^2 + (dz/l1)^2),  k = sv (1 + sqrt3 r) exp(-sqrt3 r),  g = great-circle km
// (run_seismic.py:19-63: haversine).  The gather stage turns every point into the record
//     { sin(lat/2), cos(lat/2), sin(lon/2), cos(lon/2), depth }      (angles in radians, GEO_* below)
// once per evaluation, so that a point PAIR needs no sin/cos at all: the half-difference sines and cosines of the
// haversine come from the angle-difference identities (products of the two records; the cancellation happens before
// the squaring, so a pair 1 km apart still has g to ~1e-12 relative), cos/sin(lat) from the double-angle ones.
// What is left per pair is one sqrt + asin for g, one sqrt for r and one exp.
constexpr int GEO_SLH = 0, GEO_CLH = 1, GEO_SNH = 2, GEO_CNH = 3, GEO_Z = 4, GEO_N = 5;
constexpr int GEO_STRIDE = 8;       // doubles per gathered row of the lld instantiation (XPAD for the Euclidean one)

struct Hav {
    double a, g2, ggp, s1, c1, s2, c2, cli, clj, sli, slj;      // g2 = g^2, ggp = g dg/da (km^2)
};
// Round 4: g = 2 R asin(sqrt(a)) is never needed by itself — the kernel wants g^2 (in r^2) and the gradient g dg/da — and both
// are analytic in a:  asin(sqrt a)^2 = a Q(a) = 1/2 sum_{n>=1} (4a)^n / (n^2 C(2n,n)),  d/da = asin(sqrt a) / sqrt(a (1 - a)) = D(a).
// For a <= 0.04 (great-circle distance <= 23 degrees = 2560 km: every pair inside a block or between neighbouring blocks of a
// regional catalogue) two degree-11 Taylor polynomials (exact rational coefficients rounded once; truncation < 1e-17
// relative) replace a square root + asin (+ a second square root and a division in the gradient): 12 / 24 multiply-adds
// instead of ~100 / ~190 instructions.  Farther pairs take the closed form (a wave-uniform branch skips it when no lane
// needs it).
constexpr double HAV_A0 = 0.04;
// Round 5: a <= 1.5e-3 (great-circle distance <= 490 km: every pair of a block or of neighbouring blocks at the seismic
// configuration's block size) needs the first SIX terms only — the seventh is 0.05 a^6 < 6e-19 of Q, 0.34 a^6 < 4e-18 of D — half
// the multiply-adds of the gradient kernel's two polynomials.  The degree is chosen per pair by its own a, so a pair's value
// is the same wherever it is evaluated (fill, gradient, neighbour discovery).
constexpr double HAV_A1 = 1.5e-3;
// (the coefficient of a Horner step as a SCALAR operand of v_fma_f64, like exp_fast's: left to the compiler every step is a
// v_fmac_f64 whose addend — the coefficient — is first moved into the destination register pair, two v_mov_b32 per step, a
// fifth of the vector instructions of a pair evaluation in k_mgrad<1,1>; the same arithmetic, the same bits)
__device__ __forceinline__ double fma_sc_free(double a, double b, double c_scalar) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_scalar));
    return d;
}
template <int DEG>
__device__ static __forceinline__ double hav_poly(double a, const double (&c)[12]) {
    double r = c[DEG];
#pragma unroll
    for (int n = DEG - 1; n >= 0; --n) r = fma_sc_free(r, a, c[n]);
    return r;
}
template <int DEG>
__device__ static __forceinline__ double hav_Q(double a) {
    const double c[12] = {0x1.0000000000000p+0, 0x1.5555555555555p-2, 0x1.6c16c16c16c17p-3, 0x1.d41d41d41d41dp-4,
                          0x1.4ce19ae67b348p-4, 0x1.f85d955d36cbbp-5, 0x1.8f0ef795b5337p-5, 0x1.45e5d2ba42ea0p-5,
                          0x1.10a57fc5a815cp-5, 0x1.d0ef1a8f09124p-6, 0x1.928a4e67e4640p-6, 0x1.60f3b40d2e48ep-6};
    return hav_poly<DEG>(a, c);
}
template <int DEG>
__device__ static __forceinline__ double hav_D(double a) {
    const double c[12] = {0x1.0000000000000p+0, 0x1.5555555555555p-1, 0x1.1111111111111p-1, 0x1.d41d41d41d41dp-2,
                          0x1.a01a01a01a01ap-2, 0x1.7a463005e918cp-2, 0x1.5d2d18a2fe8d0p-2, 0x1.45e5d2ba42ea0p-2,
                          0x1.32ba2fbe5d188p-2, 0x1.2295709965ab6p-2, 0x1.14bf15e76d04cp-2, 0x1.08b6c709e2b6ap-2};
    return hav_poly<DEG>(a, c);
}
template <bool GRAD>
__device__ static __forceinline__ Hav haversine(const double *gi, const double *gj) {
    Hav h;
    h.s1 = gj[GEO_SLH] * gi[GEO_CLH] - gj[GEO_CLH] * gi[GEO_SLH];      // sin((lat_j - lat_i) / 2)
    h.s2 = gj[GEO_SNH] * gi[GEO_CNH] - gj[GEO_CNH] * gi[GEO_SNH];      // sin((lon_j - lon_i) / 2)
    h.cli = gi[GEO_CLH] * gi[GEO_CLH] - gi[GEO_SLH] * gi[GEO_SLH];
    h.clj = gj[GEO_CLH] * gj[GEO_CLH] - gj[GEO_SLH] * gj[GEO_SLH];
    if constexpr (GRAD) {
        h.c1 = gj[GEO_CLH] * gi[GEO_CLH] + gj[GEO_SLH] * gi[GEO_SLH];
        h.c2 = gj[GEO_CNH] * gi[GEO_CNH] + gj[GEO_SNH] * gi[GEO_SNH];
        h.sli = 2.0 * gi[GEO_SLH] * gi[GEO_CLH];
        h.slj = 2.0 * gj[GEO_SLH] * gj[GEO_CLH];
    }
    double a = h.s1 * h.s1 + h.cli * h.clj * h.s2 * h.s2;
    if (a > 1.0) a = 1.0;
    h.a = a;
    h.ggp = 0.0;
    if (__builtin_expect(a <= HAV_A1, 1)) {
        h.g2 = (4.0 * EARTH_R_KM * EARTH_R_KM) * (a * hav_Q<5>(a));
        if constexpr (GRAD) h.ggp = (2.0 * EARTH_R_KM * EARTH_R_KM) * hav_D<5>(a);
    } else if (a <= HAV_A0) {
        h.g2 = (4.0 * EARTH_R_KM * EARTH_R_KM) * (a * hav_Q<11>(a));
        if constexpr (GRAD) h.ggp = (2.0 * EARTH_R_KM * EARTH_R_KM) * hav_D<11>(a);
    } else {
        double g = 2.0 * asin(sqrt(a)) * EARTH_R_KM;
        h.g2 = g * g;
        // g dg/da with dg/da = R / sqrt(a (1 - a)); zero at antipodal points
        if constexpr (GRAD) h.ggp = a < 1.0 ? g * (EARTH_R_KM / sqrt(a * (1.0 - a))) : 0.0;
    }
    return h;
}

// sqrt of x >= 0 from the hardware reciprocal-square-root seed, two Newton steps and a residual correction (the pivot chain's
// sqrt_and_rsqrt without the reciprocal): 9 instructions against the library sqrt's ~20 with its range scaling — x is a
// squared scaled distance here, 0 or O(1e-12 .. 1e4); 0 stays 0 (the seed of 0 is inf: guarded)
__device__ static __forceinline__ double sqrt_nn(double x) {
    double xs = x > 1e-280 ? x : 1e-280;
    double y = __builtin_amdgcn_rsq(xs);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        double t = xs * y;
        double e = fma(-t, y, 1.0);
        y = fma(0.5 * y, e, y);
    }
    double d = xs * y;
    double r = fma(-d, d, xs);
    d = fma(r, 0.5 * y, d);
    return x > 1e-280 ? d : 0.0;
}

template <>
struct KernFn<1, 1> {
    __device__ static __forceinline__ double value(const KParams &p, const double *gi, const double *gj) {
        Hav h = haversine<false>(gi, gj);
        double il0 = p.inv_ls[0];
        double dd = (gi[GEO_Z] - gj[GEO_Z]) * p.inv_ls[1];
        double r = sqrt_nn(h.g2 * (il0 * il0) + dd * dd);
        double s3r = SQRT3 * r;
        return p.sv * (1.0 + s3r) * exp_fast(-s3r);
    }
    // k(x_i, x_j) with the derivatives with respect to both ends and to the two lengthscales.  The great-circle
    // derivatives are not antisymmetric in the two ends (d a / d lat has the other point's cos(lat) in it), but
    // everything up to them — a, g, r, exp — is shared.
    __device__ static __forceinline__ double pair(const KParams &p, const double *gi, const double *gj, bool, double,
                                                  double *dkdxi, double *dkdxj, double *dkdl) {
        Hav h = haversine<true>(gi, gj);
        double il0 = p.inv_ls[0], il1 = p.inv_ls[1];
        double il02 = il0 * il0, il12 = il1 * il1;
        double dz = gj[GEO_Z] - gi[GEO_Z];
        double dd = dz * il1;
        double r = sqrt_nn(h.g2 * il02 + dd * dd);
        double s3r = SQRT3 * r;
        double e = exp_fast(-s3r);
        double k = p.sv * (1.0 + s3r) * e;
        double c = -3.0 * p.sv * e;  // dk/dr = c * r ; r cancels against d r/d(.) = (.)/r
        // d k / d(lon, lat) = c (g dg/da) (da / d.) / l0^2, angles in degrees (every da / d. below vanishes at coincident points)
        double w = c * h.ggp * (DEG2RAD * il02);
        double s22 = h.s2 * h.s2, s1c1 = h.s1 * h.c1;
        double da_dlon = h.cli * h.clj * h.s2 * h.c2;
        dkdxj[0] = w * da_dlon;
        dkdxi[0] = -w * da_dlon;
        dkdxj[1] = w * (s1c1 - h.slj * h.cli * s22);
        dkdxi[1] = w * (-s1c1 - h.sli * h.clj * s22);
        double tz = c * dz * il12;
        dkdxj[2] = tz;
        dkdxi[2] = -tz;
        dkdl[0] = -c * h.g2 * (il02 * il0);
        dkdl[1] = -c * dz * dz * (il12 * il1);
        return k;
    }
};

// what a kernel instantiation keeps per point: row stride in the gathered pool and values held in registers
template <int DIST> struct PtRec { static constexpr int STRIDE = XPAD, NREG = 3; };
template <> struct PtRec<1> { static constexpr int STRIDE = GEO_STRIDE, NREG = GEO_N; };

// ------------------------------------------------------------------------------------------------
// the 16 x 16 diagonal tile: factor, inverse, log-det epilogue (k_potrf, the register-resident kernels, k_big_diag)
// ------------------------------------------------------------------------------------------------
// d = sqrt(p) and rd = 1/sqrt(p) from ONE Newton chain on the hardware reciprocal-square-root seed (the 16
// pivots of a diagonal tile are a serial dependency: this halves the dependent instruction count of
// sqrt() followed by 1.0/d).  p is a pivot of a kernel matrix, O(1e-8 .. 1e1): no range scaling needed.
// d carries the usual final residual correction (correctly rounded except in rare halfway cases);
// rd is accurate to ~1 ulp.
__device__ __forceinline__ void sqrt_and_rsqrt(double p, double *d_out, double *rd_out) {
    double y = __builtin_amdgcn_rsq(p);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        double t = p * y;
        double e = fma(-t, y, 1.0);
        y = fma(0.5 * y, e, y);
    }
    double d = p * y;
    double r = fma(-d, d, p);
    d = fma(r, 0.5 * y, d);
    double e2 = fma(-d, y, 1.0);
    *rd_out = fma(e2, y, y);
    *d_out = d;
}

// compile-time counted loop: f(std::integral_constant<int, i>) for i in [B, E) — the DPP lane selectors below
// are instruction immediates
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// 64-bit DPP (gfx90a+: DP-ALU DPP, row_newbcast only): every lane reads lane L of ITS row of 16 lanes.
// s_nop 1 = the two wait states a DPP read needs after a VALU write of the source register (the assembler
// does not see into inline asm, so the hazard is covered here).
template <int L>
__device__ __forceinline__ double bcast16(double src) {
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(L));
    return r;
}
// acc -= (lane L's src) * mul   in one instruction.  No wait states inside: the caller guarantees that `src`
// was not written by a VALU instruction in the two issue slots before (LDS / memory loads are covered by
// s_waitcnt; after a VALU definition use dpp_src_ready()).
template <int L>
__device__ __forceinline__ void fnma_bcast16(double &acc, double src, double mul) {
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(L));
}
// two wait states after the VALU definition of a value that DPP instructions are about to read
__device__ __forceinline__ void dpp_src_ready(double &src) { asm volatile("s_nop 1" : "+v"(src)); }

// upper Cholesky of one 16x16 tile held one column per lane (s[i] = C[i][lr], replicated in the wave's four
// rows of 16 lanes); returns the first bad pivot (1-based row within the tile) or 0; *dk / *rdk = this lane's
// diagonal entry and its reciprocal.  Pivot k: every lane fetches the pivot by DPP broadcast and computes its
// root redundantly; the rank-1 update s[i] -= U[k][i] U[k][lr] takes U[k][i] from lane i by DPP inside the
// FMA — no v_readlane, no SGPR traffic on the 16-pivot chain.
__device__ __forceinline__ int diag_factor16(double (&s)[16], int lr_in, double *dk, double *rdk) {
    double myrd = 1.0;
    static_for<0, 16>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        // an opaque copy of the lane index per pivot: otherwise the 32 lane masks (lr > k, lr == k) are all
        // computed up front, hoisted out of the caller's step loop and spilled (64 SGPRs)
        int lr = lr_in;
        asm volatile("" : "+v"(lr));
        // a non-positive (or NaN) pivot turns into NaN here and poisons every later pivot: found after the loop
        double pkk = bcast16<k>(s[k]);
        double d, rd;
        sqrt_and_rsqrt(pkk, &d, &rd);
        double ukc = (lr > k) ? s[k] * rd : ((lr == k) ? d : 0.0);
        dpp_src_ready(ukc);
        s[k] = ukc;
        static_for<k + 1, 16>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            fnma_bcast16<i>(s[i], ukc, ukc);
        });
        myrd = (lr == k) ? rd : myrd;
    });
    // this lane's diagonal entry: row lr of its own column
    double mydiag = s[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        int lr = lr_in;
        asm volatile("" : "+v"(lr));
        mydiag = (lr == i) ? s[i] : mydiag;
    }
    *dk = mydiag;
    *rdk = myrd;
    // first pivot that failed = lowest lane (of the 16 columns) whose diagonal is not a positive number
    unsigned long long badmask = __ballot(!(mydiag > 0.0)) & 0xffffull;
    return badmask ? __builtin_ctzll(badmask) + 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// diag_factor16_ldl: the same 16x16 factor with the square roots taken OFF the pivot chain.
// A dependent fp64 VALU operation costs ~16 cycles of latency here (4 to issue) and a tile's 16 pivots are one serial
// chain: with  p -> rsqrt(p) (seed, two Newton steps, corrections) -> scale -> update  that chain was 22 dependent
// operations per pivot (380 cycles measured, 6.1 k per tile: the longest item of a Cholesky step).  The root-free
// ordering keeps the pivot ROW unscaled:
//     r_k = row k of the trailing tile (p_k = r_kk),   w_k = r_k / p_k,   s[i][j] -= w_ki r_kj    (i, j > k)
// so the chain per pivot is  p -> 1/p (seed + two Newton steps) -> w -> first update : 8 dependent operations, and the
// independent updates of pivot k-1 are issued in its shadows — everything is volatile asm in exactly that order (left
// to itself the scheduler packs independent work in FRONT of a dependent chain, not into it).  The roots are taken
// once, behind the loop, for all 16 pivots in parallel (lane k owns p_k):  U_kj = r_kj / sqrt(p_k).
// Rounding: an update term w_ki r_kj carries ONE rounded quotient (the scaled form's u_ki u_kj carries two); a stored
// factor entry is r_kj times a reciprocal root, as in LAPACK's dpotf2.
// Leaves the rows w_k of G = D^-1 U in LDS (Gd[k][lane]; meaningful right of the diagonal): the row panel's forward
// substitution with the UNIT triangular G has one fused multiply-add per step on its chain instead of three operations.
// On return s[k] = row k of U (the part LEFT of the diagonal is unspecified: nobody reads it), *dk / *rdk = this lane's
// diagonal entry and its reciprocal; the result is the first bad pivot (1-based) or 0.
// ------------------------------------------------------------------------------------------------
// dst = (lane index == K) ? src : dst, compare and selects in ONE ordered block: neither a lane mask kept in SGPRs from
// far ahead nor a copy of the lane index per use (both are what the compiler makes of sixteen of these in a row)
template <int K>
__device__ __forceinline__ void select_lane(double &dst, double src, int lr) {
    int dlo = __double2loint(dst), dhi = __double2hiint(dst);
    asm volatile("v_cmp_eq_u32_e32 vcc, %4, %5\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_cndmask_b32_e32 %1, %1, %3, vcc"
                 : "+v"(dlo), "+v"(dhi)
                 : "v"(__double2loint(src)), "v"(__double2hiint(src)), "n"(K), "v"(lr)
                 : "vcc");
    dst = __hiloint2double(dhi, dlo);
}
template <int L>
__device__ __forceinline__ void fnma_bcast16_ordered(double &acc, double src, double mul) {      // fnma_bcast16, kept in program order
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(L));
}
template <int K, int LO, int HI>
__device__ __forceinline__ void ldl_pending(double (&s)[16], double wprev) {
    // updates of pivot K-1 still owed to rows LO .. HI-1:  s[i][lane] -= w_{K-1}[i] * r_{K-1}[lane]
    static_for<LO, (HI < 16 ? HI : 16)>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        fnma_bcast16_ordered<i>(s[i], wprev, s[K - 1]);
    });
}

// `early(d, rd, mypiv)`: called once the pivots' roots are known and BEFORE the rows are scaled into U — everything the row
// panel's substitution needs (the rows of G in LDS, 1 / U_kk = rd) exists at that point; the run-ahead pipeline publishes
// there and scales U off the critical chain.
struct NoEarly { __device__ __forceinline__ void operator()(double, double, double) const {} };
// WRITE_G = false: nobody wants the rows of G (the register kernels' row panel is V_jj^T C_jk on the matrix pipe, round 4).
// (Round 4 also built the tile's inverse INSIDE this pivot loop — V = G^-1 D^-1/2, the 120 DPP multiply-adds of the column
// operations one pivot behind the factor, in the empty issue slots of its latency chain instead of 2.2 k cycles behind it:
// correct, but the sixteen extra doubles do not fit the 96-register instantiations — the compiler parked kernel state in
// a0..a5, i.e. in tile slot 0, tests/test_isa_invariants.py — and an inverse by another formula in some instantiations only
// would break their bit-for-bit agreement.  Dropped.)
template <class Early = NoEarly, bool WRITE_G = true>
__device__ __forceinline__ int diag_factor16_ldl(double (&s)[16], int lr_in, double *dk, double *rdk, double *Gd, Early early = Early()) {
    double w[2] = {0.0, 0.0};
    // (LDS byte address of this lane's column of G: the rows are stored from inside the ordered sequence)
    unsigned ga = WRITE_G ? (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)(Gd + lr_in) : 0u;
    static_for<0, 16>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int NF = k >= 1 ? 15 - k : 0;      // rows k+1 .. 15 still owed pivot k-1's update (row k had it on the chain)
        constexpr int PER = (NF + 4) / 5;            // ... dealt over the five gaps of this pivot's chain
        constexpr int B = k + 1;
        constexpr int KP = k >= 1 ? k : 1;
        double pk, y, e;
        asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(pk) : "v"(s[k]), "n"(k));
        if constexpr (k >= 1) ldl_pending<KP, B, B + PER>(s, w[(k - 1) & 1]);
        asm volatile("v_rcp_f64 %0, %1" : "=v"(y) : "v"(pk));
        if constexpr (k >= 1) ldl_pending<KP, B + PER, B + 2 * PER>(s, w[(k - 1) & 1]);
        asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(pk), "v"(y));
        if constexpr (k >= 1) ldl_pending<KP, B + 2 * PER, B + 3 * PER>(s, w[(k - 1) & 1]);
        asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(y) : "v"(e));
        if constexpr (k >= 1) ldl_pending<KP, B + 3 * PER, B + 4 * PER>(s, w[(k - 1) & 1]);
        asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(pk), "v"(y));
        if constexpr (k >= 1) ldl_pending<KP, B + 4 * PER, 16>(s, w[(k - 1) & 1]);
        asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(y) : "v"(e));
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(w[k & 1]) : "v"(s[k]), "v"(y));
        // (two wait states between the VALU write of w and its first DPP read)
        if constexpr (k < 15)
            asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                         : "+v"(s[k + 1])
                         : "v"(w[k & 1]), "v"(s[k]), "n"(k + 1));
        // off the chain: the row of G (ordered too: a store the compiler is free to delay keeps its value alive, and the
        // two-per-CU instantiation has 96 registers)
        unsigned ga_k = ga;      // (a C++ use: inline-asm operands alone do not make a generic lambda capture a variable)
        if constexpr (WRITE_G) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(ga_k), "v"(w[k & 1]), "n"(k * 128) : "memory");
        (void)ga_k;
    });
    // this lane's own pivot p_lr = r_lr,lr: row lr has not changed since it was the pivot row
    double mypiv = s[0];
    static_for<1, 16>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        select_lane<i>(mypiv, s[i], lr_in);
    });
    // the roots, all pivots at once
    double d, rd;
    sqrt_and_rsqrt(mypiv, &d, &rd);
    early(d, rd, mypiv);
    dpp_src_ready(rd);
    static_for<0, 16>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        double bc;      // (volatile: one broadcast value alive at a time — sixteen hoisted ones would not fit the 96-register kernel)
        asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(bc) : "v"(rd), "n"(k));
        double u = s[k] * bc;
        select_lane<k>(u, d, lr_in);
        s[k] = u;
    });
    *dk = d;
    *rdk = rd;
    unsigned long long badmask = __ballot(!(mypiv > 0.0)) & 0xffffull;
    return badmask ? __builtin_ctzll(badmask) + 1 : 0;
}

// shared tail of the Cholesky kernels: V_jj = U_jj^-1 for every diagonal tile (wanted by the triangular-solve
// kernels' MFMA form; 4 tiles per wave at a time, lane (lg, lr) = row lr of tile 4*grp + lg, by the column
// operations that reduce U_jj to I) and log|K| = 2 sum log U_kk (gpy_linalg.py:234) in a fixed order.
// `stage` is >= 256*T doubles of LDS that are free by now.
template <int NWAVES, bool WITH_V = true>
__device__ __forceinline__ void potrf_epilogue(const double *U, double *V, double *stage, const double *dvals,
                                               double *lred, int mp, int T, int u, const Pools &pl) {
    int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar wave index
    int lr = lane & 15, lg = lane >> 4;
    for (int grp = wave; WITH_V && 4 * grp < T; grp += NWAVES) {
        int jt = 4 * grp + lg;
        double *Us = stage + jt * 256;
        if (jt < T) {
            const double *Ujj = U + (size_t)(16 * jt) * mp + 16 * jt;
#pragma unroll
            for (int i = 0; i < 16; ++i) Us[i * 16 + lr] = Ujj[(size_t)i * mp + lr];
        }
        __builtin_amdgcn_wave_barrier();
        if (jt < T) {
            // the same column operations as the row-panel substitution, on the identity: lane lr holds column lr
            // of its tile's U_jj (uc) and row lr of V; U[k][i] reaches the FMA by DPP broadcast from lane i
            double v[16], uc[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                int lrc = lr;
                asm volatile("" : "+v"(lrc));       // keep the 16 lane masks from living in SGPRs all at once
                v[c] = (c == lrc) ? 1.0 : 0.0;
                uc[c] = Us[c * 16 + lr];
            }
            double rdl = 1.0 / Us[lr * 16 + lr];
            dpp_src_ready(rdl);
            static_for<0, 16>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                v[k] *= bcast16<k>(rdl);
                static_for<k + 1, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    fnma_bcast16<i>(v[i], uc[k], v[k]);
                });
            });
            double *Vj = V + (size_t)jt * 256 + lr * 16;
#pragma unroll
            for (int c = 0; c < 16; ++c) Vj[c] = v[c];
        }
    }
    double part = 0.0;
    for (int r = threadIdx.x; r < mp; r += NWAVES * 64) part += log(dvals[r]);
    for (int off = 32; off >= 1; off >>= 1) part += shfl_xor_d(part, off);
    if (lane == 0) lred[wave] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < NWAVES; ++w) t += lred[w];
        pl.logdet[u] = 2.0 * t;
        pl.info[u] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// shared by the launchers of more than one file
// ------------------------------------------------------------------------------------------------
constexpr int BGT = 128;            // k_big_gemm's tile edge (gprf_big.hip)
constexpr int BIG_AT_GEMM_T = 192;  // launches whose largest unit has more tiles per edge (3072 points) form At by the GEMM's mode 3
// dynamic LDS above 48 KB has to be opted into per kernel AND per device: remembers the largest size already granted for
// (kernel slot, current device).  Slots: 1 k_potrf, 3 k_potrf_reg2, 4 / 8 k_potrf_reg8 gen / pool, 9 k_potrf_reg8w
bool lds_needs_optin(int kernel_slot, size_t lds);
int device_cus();                       // compute units of the current device
int xcd_grid(int n_ids, int nparts);    // workgroups of an xcd_map / part_major_map launch
void launch_big_at(const UnitTab &ut, const Pools &p, hipStream_t s);
// M = At^T At - dy W^T W of the units of more than 1024 points by the LDS-staged GEMM, into their K regions (launch_grad)
void launch_big_mgemm(const UnitTab &ut, const Pools &p, int dy, hipStream_t s);

}  // namespace gprf
