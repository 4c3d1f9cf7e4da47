// gprf_kernels.hip — gfx950 (MI355X, CDNA4) kernels for the GPRF block-local log-likelihood/gradient path.
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
template <int MAXT, int WPS, bool PM, int NBUF = 2>
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
    if (!(PM ? (ut.pm_group >= 0 ? part_major_map(blockIdx.x, ut.n_ids, nI + 1, ut.pm_group, &slot_, &part_)
                                 : xcd_map(blockIdx.x, ut.n_ids, nI + 1, &slot_, &part_))
             : xcd_map(blockIdx.x, ut.n_ids, nI + 1, &slot_, &part_))) return;
    // the Y workgroup (every step, a gather in front) is the longest of a unit: it is dispatched first
    part_ = part_ == 0 ? nI : part_ - 1;
    const UnitRef ur = unit_ref(ut.srec, slot_);
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

// k_at_wide: the throughput form of k_at (many units per CU): one workgroup per 16 column tiles of the unit; wave w owns the column tiles
// I = I0 + w, 7-w, 8+w, 15-w (W is lower triangular: column tile I has T - I row tiles, and wave w always lands on SIMD w —
// dealt w, w+4, w+8, w+12, wave 0 of every workgroup on a CU carried 40 tile steps of a 16-tile unit against wave 3's 28;
// the snake gives 34 each) and all four 16-row blocks of At for each (16 accumulators).  The k-loop runs
// DOWN from the last row tile so the four waves need the same Z chunk at the same time (shared through L1):
// per k-tile 16 Z operands are loaded once and reused for up to four column tiles.
__global__ __launch_bounds__(256, 2) void k_at_wide(UnitTab ut, Pools pl, int first_round, int skip_T) {
    int slot_, part_;
    WgTrace trace(ut, pl, 2);
    if (!xcd_map(blockIdx.x, ut.n_ids, (ut.max_T + 15) >> 4, &slot_, &part_)) return;
    // A launch of at most two workgroups per CU is resident all at once: workgroup first_round + j (first_round = the CUs)
    // becomes the second resident of the CU that took workgroup j.  The launch order is largest unit first, so the CU of the
    // largest unit also got the largest of the rest, and the launch lasted as long as those two sharing four SIMDs; with the
    // second round in ASCENDING size the largest unit is paired with the smallest.
    if (first_round > 0 && slot_ >= first_round) slot_ = ut.n_ids - 1 - (slot_ - first_round);
    const UnitRef ur = unit_ref(ut.srec, slot_);
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
template <int DIST, int KERN, bool HAVEK, int FAST, bool BIG = false>
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
    if (!(part_major ? part_major_map(blockIdx.x, ut.n_ids, TBm * (TBm + 1) / 2, ut.pm_group, &slot, &bp) : xcd_map(blockIdx.x, ut.n_ids, TBm * (TBm + 1) / 2, &slot, &bp))) return;
    const UnitRef ur = unit_ref(ut.srec, slot);
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
constexpr int BGT = 128, BG_LD = 144, BG_KC = 8, BG_SUPER = 4;
constexpr int BG_ATSEG = 512;      // rows of [Z | W] per partial product of At (mode 3)
constexpr int BIG_AT_GEMM_T = 192;  // launches whose largest unit has more tiles per edge (3072 points) form At by mode 3

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

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// dynamic LDS above 48 KB has to be opted into per kernel AND per device: remembers the largest size already
// granted for (kernel slot, current device)
static bool lds_needs_optin(int kernel_slot, size_t lds) {
    static size_t granted[12][64] = {};
    if (lds <= 48 * 1024) return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (lds <= granted[kernel_slot][dev]) return false;
    granted[kernel_slot][dev] = lds;
    return true;
}

// compute units of the current device (kernel variants are picked by how many workgroup rounds a launch is deep)
static int device_cus() {
    static int n_cus = 0;
    if (n_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    return n_cus;
}

static int xcd_grid(int n_ids, int nparts) { return ((n_ids + 7) / 8) * 8 * nparts; }
// ONE diagnostic switch for everything that selects a launch structure: GPRF_DIAG="key=value,key=value".  The product path
// sets none of them; tests/test_gpu_variants.py compares the forms they select bit for bit, scripts/ time them.
//   fused_build=0   table build + coordinate scatter as three launches      gx_fold=0     k_gx_finalize as a launch of its own
//   one_queue=1     both Cholesky instantiations on the main queue          side_events=1 fork / join of the two queues by events
//   part_major=0/1  solve / gradient grids unit by unit / part by part      potrf_reg=0   every unit through the generic Cholesky
//   fused_fill=0    K always through the pool (k_fill)                      potrf_gw=0    units of 21-32 tiles on the generic kernel
//   pipe=<percent>  solve / At / gradient as two pipelines (off)            max_unit=<points>  a lower GPRF_MAX_UNIT (refusal-path tests)
//   potrf_stamps=1..3  which wave's cycle stamps a -DGPRF_PROFILE build records    grid_hint=0  k_assign scans every centre
//   big_super=<n>   block rows of 64 per super-block of the blocked path (read once per process); big_super_solve=<n>: the sweep's
//   big_beside=0    the blocked path's Cholesky / substitution behind the one-workgroup kernels instead of beside them (mixed launches)
// Read at every call (a handful of string searches per evaluation): a test may change it between two contexts of one process.
int diag(const char *key, int dflt) {
    const char *e = getenv("GPRF_DIAG");
    if (!e || !e[0]) return dflt;
    const size_t kl = strlen(key);
    for (const char *q = e; (q = strstr(q, key)) != nullptr; q += kl) {
        if ((q == e || q[-1] == ',') && q[kl] == '=') return atoi(q + kl + 1);
    }
    return dflt;
}

// ------------------------------------------------------------------------------------------------
// Re-blocking on the device (gprf.py:169-174: update_X re-runs block_fn before every evaluation).
//
// partition_tail: what both partition kernels (nearest centre, split tree) end with.  One workgroup (one wave) = one
// chunk of CHUNK = 64 consecutive points (n / 64 workgroups spread a small problem over enough CUs).  Besides the new block of its point each thread leaves
//   rank[p]       = points of the same block earlier in the chunk,
//   cnt[chunk][b] = points of block b in the chunk (written by the block's last point of the chunk; the workgroup
//                   zeroes its own row first),
// from which k_build (both passes) / k_scatter_x derive every table — the points of a block keep ascending index
// order, exactly `all_idxs[blocks == i]` (block_clustering.py:21-24).  A point that changes block stamps
// ctl[CTL_CHANGED] with this evaluation's epoch (no reset needed between evaluations).
// ------------------------------------------------------------------------------------------------
// partition_head: the part of it that depends on nothing — the point's block of the last evaluation is asked for and the
// workgroup's row of cnt is zeroed at the START of the kernel, under the latency of the coordinates' own load (they may come
// from pinned host memory) instead of as two more exposed round trips behind the search.
__device__ __forceinline__ int partition_head(int p, int n, const BuildTab &bt) {
    int *row = bt.cnt + (size_t)blockIdx.x * bt.n_blocks;
    const int old = p < n ? bt.assign[p] : -1;
    for (int b = threadIdx.x; b < bt.n_blocks; b += CHUNK) row[b] = 0;
    return old;
}
__device__ __forceinline__ void partition_tail(int p, int n, int best, int old, const BuildTab &bt, int epoch,
                                               int *keys /* LDS [CHUNK], 16-byte aligned */) {
    int t = threadIdx.x;
    int *row = bt.cnt + (size_t)blockIdx.x * bt.n_blocks;
    keys[t] = p < n ? best : -1;
    __syncthreads();      // (also: the row's zeroes of partition_head are in memory before anybody writes a count)
    if (p >= n) return;
    int before = 0, total = 0;
    const int4 *k4 = reinterpret_cast<const int4 *>(keys);
#pragma unroll
    for (int q4 = 0; q4 < CHUNK / 4; ++q4) {
        int4 k = k4[q4];                           // wave-uniform address: LDS broadcast
        int q = 4 * q4;
        int s0 = k.x == best, s1 = k.y == best, s2 = k.z == best, s3 = k.w == best;
        total += s0 + s1 + s2 + s3;
        before += (q < t ? s0 : 0) + (q + 1 < t ? s1 : 0) + (q + 2 < t ? s2 : 0) + (q + 3 < t ? s3 : 0);
    }
    bt.rank[p] = before;
    if (before == total - 1) row[best] = total;
    if (old != best) {
        bt.assign[p] = best;
        bt.ctl[CTL_CHANGED] = epoch;        // benign race: every writer stores the same value
    }
}

// k_assign: nearest cluster centre of every point (block_clustering.py:4-5,15-17), one thread per point, the centres
// (structure of arrays + squared norms) staged through LDS a tile at a time.  Same arithmetic as the host helper
// gprf_nearest_center — radicand x2 - 2 x.c + c2 accumulated in the same order with no FMA contraction; numpy's argmin
// over sqrt(radicand): the FIRST negative radicand (NaN distance) wins, otherwise the first minimum — so the two agree
// bit for bit.
constexpr int ASSIGN_TILE = 512;
// (DX is a template parameter: the coordinate loops unroll and x[] stays in registers — indexed by a runtime loop it
// would live in scratch memory.)  Xcopy: the kernel's own copy of the points in HBM for the kernels that follow (X
// itself may be pinned host memory read over the fabric).
// Round 5, GridHint: the reference's own block function is a g x g grid of centres (gprfopt.py:519-523).  The argmin over all
// g^2 centres is decided among the 3 x 3 around the point's cell: the radicands of the others exceed the minimum by at least
// 1.75 h^2, ten orders of magnitude above the formula's rounding for |x| <= 1e3 — so the fast path evaluates the SAME
// radicand expression on those nine (fewer at the border), in ascending centre index, with the same first-negative /
// first-minimum rule: the same block, bit for bit (tests/test_gpu_parity.py, test_gpu_reference_partitions.py: points on
// centres, exact ties, points outside the square).  A wave with a point beyond 1e3 (or NaN) takes the full scan.
template <int DX>
__global__ __launch_bounds__(CHUNK) void k_assign(const double *__restrict__ X, double *__restrict__ Xcopy,
                                                const double *__restrict__ cs, const double *__restrict__ c2, int nc,
                                                GridHint gh, BuildTab bt, int epoch) {
    __shared__ __attribute__((aligned(16))) int keys[CHUNK];
    __shared__ double scs[DX * ASSIGN_TILE], sc2[ASSIGN_TILE];
    int n = bt.n;
    int p = blockIdx.x * CHUNK + threadIdx.x;
    double x[DX], x2 = 0.0;
    // (branch-free: a thread beyond n re-reads point n - 1 and stores nothing)
    const int pl = p < n ? p : n - 1;
#pragma unroll
    for (int d = 0; d < DX; ++d) x[d] = X[(size_t)pl * DX + d];
    const int old = partition_head(p, n, bt);
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        x[d] = p < n ? x[d] : 0.0;
        x2 = __dadd_rn(x2, __dmul_rn(x[d], x[d]));
    }
    if (Xcopy && p < n) {
#pragma unroll
        for (int d = 0; d < DX; ++d) Xcopy[(size_t)p * DX + d] = x[d];
    }
    int best = 0, neg_k = -1;
    double bestv = 0.0;
    bool grid_done = false;
    if constexpr (DX == 2) {
        const bool near = p >= n || (fabs(x[0]) <= 1e3 && fabs(x[1]) <= 1e3);      // (false for NaN)
        if (gh.g > 0 && __all(near)) {
            const int g = gh.g;
            int ix = (int)floor(__builtin_fma(x[0] - gh.a0, gh.inv_ha, 0.5)), iy = (int)floor(__builtin_fma(x[1] - gh.b0, gh.inv_hb, 0.5));
            ix = ix < 0 ? 0 : (ix > g - 1 ? g - 1 : ix);
            iy = iy < 0 ? 0 : (iy > g - 1 ? g - 1 : iy);
            const int ix0 = ix > 0 ? ix - 1 : 0, ix1 = ix < g - 1 ? ix + 1 : g - 1;
            const int iy0 = iy > 0 ? iy - 1 : 0, iy1 = iy < g - 1 ? iy + 1 : g - 1;
            // all 27 centre values requested at once (a cell beyond the border re-reads its clamped neighbour and is left out of
            // the comparison): written "if (inside) { load; compare }" every one of the nine was a branch, three loads and an
            // s_waitcnt vmcnt(0) — nine memory round trips one after the other in a kernel that is nothing but latency
            double c0v[9], c1v[9], c2v[9];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const int kx = ix0 + a < ix1 ? ix0 + a : ix1, ky = iy0 + b < iy1 ? iy0 + b : iy1;
                    const int k = kx * g + ky;
                    c0v[3 * a + b] = cs[k];
                    c1v[3 * a + b] = cs[(size_t)nc + k];
                    c2v[3 * a + b] = c2[k];
                }
            }
            bool first = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const int kx = ix0 + a, ky = iy0 + b;
                    const bool inside = kx <= ix1 && ky <= iy1;
                    const int k = kx * g + ky;
                    double r = 0.0;
                    r = __dadd_rn(r, __dmul_rn(x[0], c0v[3 * a + b]));
                    r = __dadd_rn(r, __dmul_rn(x[1], c1v[3 * a + b]));
                    const double v = __dadd_rn(__dsub_rn(x2, __dmul_rn(2.0, r)), c2v[3 * a + b]);
                    if (inside && first) { bestv = v; best = k; first = false; }
                    if (inside && v < 0.0 && neg_k < 0) neg_k = k;
                    if (inside && v < bestv) { best = k; bestv = v; }
                }
            }
            grid_done = true;
        }
    }
    for (int k0 = 0; !grid_done && k0 < nc; k0 += ASSIGN_TILE) {
        int kn = nc - k0 < ASSIGN_TILE ? nc - k0 : ASSIGN_TILE;
        __syncthreads();
        for (int e = threadIdx.x; e < kn; e += CHUNK) {
#pragma unroll
            for (int d = 0; d < DX; ++d) scs[d * ASSIGN_TILE + e] = cs[(size_t)d * nc + k0 + e];
            sc2[e] = c2[k0 + e];
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < kn; ++k) {
            double r = 0.0;
#pragma unroll
            for (int d = 0; d < DX; ++d) r = __dadd_rn(r, __dmul_rn(x[d], scs[d * ASSIGN_TILE + k]));
            double v = __dadd_rn(__dsub_rn(x2, __dmul_rn(2.0, r)), sc2[k]);
            if (k0 + k == 0) bestv = v;
            if (v < 0.0 && neg_k < 0) neg_k = k0 + k;
            if (v < bestv) { best = k0 + k; bestv = v; }
        }
    }
    if (neg_k >= 0) best = neg_k;
    partition_tail(p, n, best, old, bt, epoch, keys);
}

// k_route: the seismic driver's re-blocking (pdtree_clustering.py:65-94 via gprf.py:171-172): every point descends
// the principal-direction tree — (x - center_k) . vec_k < split_k ? left : right — one thread per point, the
// longitude first moved to [-22, 338) like the reference's `(lon + 22) % 360 - 22`.  The projection is accumulated
// column by column with separately rounded multiplies and adds, which is how gprf_amd/seismic.py builds and routes
// (numpy element-wise ops): bit-identical decisions, including the median point whose projection equals the split.
__global__ __launch_bounds__(CHUNK) void k_route(const double *__restrict__ X, double *__restrict__ Xcopy, int dx, int dim,
                                               int lon_wrap, const double *__restrict__ vec,
                                               const double *__restrict__ center, const double *__restrict__ split,
                                               const int32_t *__restrict__ left, const int32_t *__restrict__ right,
                                               const int32_t *__restrict__ leaf_block, BuildTab bt, int epoch) {
    __shared__ __attribute__((aligned(16))) int keys[CHUNK];
    int n = bt.n;
    int p = blockIdx.x * CHUNK + threadIdx.x;
    int best = 0;
    const int old = partition_head(p, n, bt);
    if (p < n) {
        double x[3] = {0.0, 0.0, 0.0};              // dx <= 3 (gprf_create); fixed-bound loops keep x[] in registers
#pragma unroll
        for (int d = 0; d < 3; ++d) x[d] = X[(size_t)p * dx + (d < dx ? d : 0)];      // (branch-free: three loads in flight)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (d < dx && Xcopy) Xcopy[(size_t)p * dx + d] = x[d];
            x[d] = d < dx ? x[d] : 0.0;
        }
        if (lon_wrap) {
            double r = fmod(__dadd_rn(x[0], 22.0), 360.0);          // numpy's %: the result takes the divisor's sign
            if (r != 0.0) { if (r < 0.0) r = __dadd_rn(r, 360.0); } else r = 0.0;
            x[0] = __dsub_rn(r, 22.0);
        }
        // one memory round trip per tree level: everything about node k is requested together (every array has an entry for
        // every node, leaves included) — "while (left[k] >= 0) { ... }" asked for left[k], waited, then for the rest
        int k = 0;
        for (;;) {
            const int lk = left[k], rk = right[k];
            const double sp = split[k];
            double c[3] = {0.0, 0.0, 0.0}, v[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int dd = d < dim ? d : 0;
                c[d] = center[(size_t)k * dim + dd]; v[d] = vec[(size_t)k * dim + dd];
            }
            if (lk < 0) break;
            double a = __dmul_rn(__dsub_rn(x[0], c[0]), v[0]);
#pragma unroll
            for (int d = 1; d < 3; ++d)
                if (d < dim) a = __dadd_rn(a, __dmul_rn(__dsub_rn(x[d], c[d]), v[d]));
            k = (a < sp) ? lk : rk;
        }
        best = leaf_block[k];
    }
    partition_tail(p, n, best, old, bt, epoch, keys);
}

void launch_assign(const double *X, double *Xcopy, int dx, const double *cs, const double *c2, int nc, const GridHint &gh,
                   const BuildTab &bt, int epoch, hipStream_t s) {
    if (bt.n == 0) return;
    dim3 g(bt.n_chunks), b(CHUNK);
    if (dx == 1) hipLaunchKernelGGL((k_assign<1>), g, b, 0, s, X, Xcopy, cs, c2, nc, gh, bt, epoch);
    else if (dx == 2) hipLaunchKernelGGL((k_assign<2>), g, b, 0, s, X, Xcopy, cs, c2, nc, gh, bt, epoch);
    else hipLaunchKernelGGL((k_assign<3>), g, b, 0, s, X, Xcopy, cs, c2, nc, gh, bt, epoch);
}

void launch_route(const double *X, double *Xcopy, int dx, int dim, int lon_wrap, const double *vec, const double *center,
                  const double *split, const int32_t *left, const int32_t *right, const int32_t *leaf_block,
                  const BuildTab &bt, int epoch, hipStream_t s) {
    if (bt.n == 0) return;
    hipLaunchKernelGGL(k_route, dim3(bt.n_chunks), dim3(CHUNK), 0, s, X, Xcopy, dx, dim, lon_wrap, vec, center, split, left,
                       right, leaf_block, bt, epoch);
}

// whether this evaluation rebuilds the tables: asked to (force), or the partition kernel stamped a change
__device__ __forceinline__ bool rebuilding(const BuildTab &bt, int force, int epoch) {
    return force || bt.ctl[CTL_CHANGED] == epoch;
}

// k_build: the unit tables from the partition, in two launches of the same kernel.
//  (1) from_chunks = 1: per block (one wave each, four per workgroup), the exclusive prefix of cnt over the chunks (in
//      place) and the block size;
//  (2) from_chunks = 0, ONE workgroup: the unit scan: per local unit m = |block i| (+ |block j|), off_j = |block i|, and the running offsets row_off = sum mp,
//      mat_off = sum mp^2 (mp = m rounded up to 16) — what rebuild_units did on the host (gprf.py:236-239 order); the
//      unit's padding rows of the coordinate pool are zeroed; the totals are checked against the workspace the host
//      reserved and the max_T the evaluation's kernels will be launched with: on overflow every unit gets m = 0 (each
//      kernel then has nothing to do) and ctl says so; the host grows the workspace and repeats the evaluation.
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;

// exclusive prefix sums of a and b over the workgroup; returns the totals through ta / tb
// inclusive prefix sum over the 64 lanes of a wave, in registers: four row shifts and two row broadcasts (DPP) — through
// __shfl_up it is six rounds of ds_bpermute, and with two 64-bit values per call 24 dependent LDS round trips
__device__ __forceinline__ int wave_incl_scan(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);      // row_shr:1 (lanes without a source keep 0)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);      // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);      // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);      // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2 and 3
    return x;
}
// (a, b: non-negative, at most 2^20 each — a unit's padded size and its square in units of 256 elements (mp is a multiple of
// 16; up to 16384 points per unit) — so the sums of one call fit 32 bits)
__device__ __forceinline__ void wg_exscan2(long long &a, long long &b, long long *sh /* LDS [2][4] */, long long *ta,
                                           long long *tb) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long ia = wave_incl_scan((int)a), ib = wave_incl_scan((int)b);
    if (lane == 63) { sh[wave] = ia; sh[4 + wave] = ib; }
    __syncthreads();
    long long pa = 0, pb = 0, sa = 0, sb = 0;
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        if (w < wave) { pa += sh[w]; pb += sh[4 + w]; }
        sa += sh[w]; sb += sh[4 + w];
    }
    __syncthreads();
    a = pa + ia - a;
    b = pb + ib - b;
    *ta = sa;
    *tb = sb;
}

// the unit scan of one workgroup of SCAN_THREADS threads (k_build's second launch; the table workgroup of k_build_scatter):
// bsz = the block sizes (global or LDS), s_m / s_ro / s_mo = LDS scratch of M_LDS words each
// pre (may be nullptr): unit_bi / unit_bj / ids of units t and t + 256, loaded by the caller ahead of time
template <int M_LDS>
__device__ __forceinline__ void unit_tables(const BuildTab &bt, const int *bsz, int *s_m, int *s_ro, unsigned *s_mo,
                                            long long *sh /* LDS [8] */, int *s_maxm_p /* LDS */, const int (*pre)[2] = nullptr) {
    int t = threadIdx.x;
    const int builds_before = t == 0 ? bt.ctl[CTL_BUILDS] : 0;      // (asked for now: at the end it would be one more exposed round trip)
    if (t == 0) *s_maxm_p = 0;
    __syncthreads();
    long long rows = 0, mat = 0;
    for (int l0 = 0; l0 < bt.n_local; l0 += SCAN_THREADS) {
        int l = l0 + t;
        int m = 0, mi = 0;
        if (l < bt.n_local) {
            int bi = (pre && l0 < 512) ? pre[0][l0 >> 8] : bt.unit_bi[l];
            int bj = (pre && l0 < 512) ? pre[1][l0 >> 8] : bt.unit_bj[l];
            mi = bsz[bi];
            m = mi + (bj >= 0 ? bsz[bj] : 0);
        }
        long long mp = (m + 15) & ~15;
        long long a = mp, b = (mp * mp) >> 8, ta, tb;      // (matrix elements in units of 256: see wg_exscan2)
        wg_exscan2(a, b, sh, &ta, &tb);
        b <<= 8;
        tb <<= 8;
        if (l < bt.n_local) {
            long long r0 = rows + a;
            bt.m[l] = m;
            bt.off_j[l] = mi;
            bt.row_off[l] = (int32_t)r0;
            bt.mat_off[l] = mat + b;
            if (l < M_LDS) { s_m[l] = m; s_ro[l] = (int32_t)r0; s_mo[l] = (unsigned)((mat + b) >> 8); }
            atomicMax(s_maxm_p, m);
        }
        rows += ta;
        mat += tb;
    }
    __syncthreads();
    int maxm = *s_maxm_p;
    int maxT = ((maxm + 15) & ~15) >> 4;
    bool over = rows > bt.cap_rows || mat > bt.cap_mat || maxT > bt.maxT_bound || maxm > MAX_MP;
    // the launch-slot records (SlotRec) in launch order, and the Cholesky's two launch lists (units of more than
    // small_maxT tiles one to a CU, the others two to a CU): a stable partition of the launch order by THIS partition's
    // sizes, as unit ids and as records
    long long nbig = 0;
    for (int k0 = 0; k0 < bt.n_local; k0 += SCAN_THREADS) {
        int k = k0 + t;
        SlotRec r = {0, 0, 0, 0u};
        long long big = 0, one = 0, tb_, to_;
        if (k < bt.n_local) {
            int u = (pre && k0 < 512) ? pre[2][k0 >> 8] : bt.ids[k];
            r.u = u;
            r.m = over ? 0 : (u < M_LDS ? s_m[u] : bt.m[u]);
            r.row_off = u < M_LDS ? s_ro[u] : bt.row_off[u];
            r.mat256 = u < M_LDS ? s_mo[u] : (unsigned)(bt.mat_off[u] >> 8);
            bt.srec[k] = r;
            big = (bt.small_maxT > 0 && ((r.m + 15) >> 4) > bt.small_maxT) ? 1 : 0;
            one = 1;
        }
        if (bt.small_maxT > 0) {
            long long isbig = big, pos = one;
            wg_exscan2(big, pos, sh, &tb_, &to_);
            if (k < bt.n_local) {
                if (isbig) {
                    bt.big_list[nbig + big] = r.u;
                    bt.big_rec[nbig + big] = r;
                } else {
                    bt.small_list[(k0 - nbig) + (pos - big)] = r.u;
                    bt.small_rec[(k0 - nbig) + (pos - big)] = r;
                }
            }
            nbig += tb_;
        }
    }
    if (bt.small_maxT > 0 && !over && (nbig > bt.grid_big || (bt.n_local - nbig) > bt.grid_small)) {
        over = true;      // a list outgrew its launch: like every other overflow, nothing of this build may be used
        __syncthreads();
        for (int l = t; l < bt.n_local; l += SCAN_THREADS) { bt.srec[l].m = 0; bt.big_rec[l].m = 0; bt.small_rec[l].m = 0; }
    }
    if (over)
        for (int l = t; l < bt.n_local; l += SCAN_THREADS) { bt.m[l] = 0; bt.row_off[l] = 0; bt.mat_off[l] = 0; bt.off_j[l] = 0; }
    if (t == 0) {
        bt.ctl[CTL_OVERFLOW] = over ? 1 : 0;
        bt.ctl[CTL_NBIG] = over ? 0 : (int32_t)nbig;
        bt.ctl[CTL_NSMALL] = over ? 0 : (int32_t)(bt.n_local - nbig);
        bt.ctl[CTL_ROWS] = (int32_t)rows;
        bt.ctl[CTL_MAXT] = maxT;
        bt.ctl[CTL_MAXM] = maxm;
        bt.ctl[CTL_MAT_LO] = (int32_t)(mat & 0xffffffffll);
        bt.ctl[CTL_MAT_HI] = (int32_t)(mat >> 32);
        bt.ctl[CTL_BUILDS] = builds_before + 1;
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_build(BuildTab bt, int from_chunks, int force, int epoch) {
    __shared__ long long sh[8];
    __shared__ int s_maxm;
    if (!rebuilding(bt, force, epoch)) return;
    int t = threadIdx.x, lane = t & 63;
    if (from_chunks) {
        int b = blockIdx.x * 4 + (t >> 6);
        if (b < bt.n_blocks) {
            int run = 0;
            for (int c0 = 0; c0 < bt.n_chunks; c0 += 64) {
                int c = c0 + lane;
                int *e = bt.cnt + (size_t)c * bt.n_blocks + b;
                int v = c < bt.n_chunks ? *e : 0;
                int inc = v;
                for (int off = 1; off < 64; off <<= 1) {
                    int u = __shfl_up(inc, off, 64);
                    if (lane >= off) inc += u;
                }
                if (c < bt.n_chunks) *e = run + inc - v;
                run += __shfl(inc, 63, 64);
            }
            if (lane == 0) bt.bsize[b] = run;
        }
        return;
    }
    // (the unit sizes stay in LDS for the second pass: every global round trip of this one-workgroup kernel is exposed)
    constexpr int M_LDS = 8192;
    __shared__ int s_m[M_LDS], s_ro[M_LDS];
    __shared__ unsigned s_mo[M_LDS];
    unit_tables<M_LDS>(bt, bt.bsize, s_m, s_ro, s_mo, sh, &s_maxm);
}

// one point's coordinate record into its row of every local unit that contains its block b (position pos inside the
// block); unit_of(unit) = {row_off, off_j, m} of the unit
struct UnitRows { int row_off, off_j, m; };
// x = the point's raw coordinates (x[d] for d < dx), ent_of(e) = bu_ent[e]
template <class UnitOf, class EntOf>
__device__ __forceinline__ void scatter_rows(const BuildTab &bt, const double (&x)[3], int geo, int p, int e_first, int e_end,
                                             int pos, bool rebuild, UnitOf unit_of, EntOf ent_of) {
    double r0, r1, r2, r3, r4 = 0.0;
    if (geo) {
        // lld: (lon, lat, depth) -> half-angle record, see KernFn<1,1>
        double lon = x[0], lat = x[1], z = x[2];
        double hl = lat * DEG2RAD / 2.0, hn = lon * DEG2RAD / 2.0;
        r0 = sin(hl); r1 = cos(hl); r2 = sin(hn); r3 = cos(hn); r4 = z;      // GEO_SLH, GEO_CLH, GEO_SNH, GEO_CNH, GEO_Z
    } else {
        r0 = x[0]; r1 = x[1]; r2 = x[2]; r3 = 0.0;
    }
    typedef double d2v __attribute__((ext_vector_type(2)));
    if (rebuild) { bt.pe[2 * p] = e_first; bt.pe[2 * p + 1] = e_end - e_first; }      // k_assemble's shortcuts
    for (int e = e_first; e < e_end; ++e) {
        int ent = ent_of(e);
        int u = ent >> 1;
        const UnitRows ur = unit_of(u);
        const int local0 = (ent & 1) ? ur.off_j : 0;
        int row = ur.row_off + local0 + pos;
        if (rebuild) {
            bt.upt[row] = p;
            if (pos == 0) {      // the block's first row inside this unit (one writer per entry)
                bt.ebase[e] = row;
                bt.einfo[e] = (local0 << 10) | ((((ur.m + 15) >> 4) + 3) >> 2);      // (local0 < 2^15, 64-point blocks <= 256)
            }
        }
        d2v *dst = reinterpret_cast<d2v *>(bt.Xu + (size_t)row * (geo ? GEO_STRIDE : XPAD));      // 32- / 64-byte rows
        dst[0] = d2v{r0, r1};
        dst[1] = d2v{r2, r3};
        if (geo) {
            dst[2] = d2v{r4, 0.0};
            dst[3] = d2v{0.0, 0.0};
        }
    }
}
// the point's raw coordinates, unused dimensions 0
__device__ __forceinline__ void load_point(const double *__restrict__ X, int dx, int p, double (&x)[3]) {
    x[0] = X[(size_t)p * dx];
    x[1] = dx > 1 ? X[(size_t)p * dx + 1] : 0.0;
    x[2] = dx > 2 ? X[(size_t)p * dx + 2] : 0.0;
}

// k_scatter_x (every evaluation): a point's coordinate record into its row of every local unit that contains its
// block — position posb of the block, rows of block j after block i's (gprf.py:322-326) — and, when the tables are
// being rebuilt, the position itself (from the chunk ranks) and the unit row -> point table.
__global__ __launch_bounds__(256) void k_scatter_x(BuildTab bt, const double *__restrict__ X, int dx, int geo,
                                                   int from_chunks, int force, int epoch) {
    bool rebuild = rebuilding(bt, force, epoch);
    int nblk_pts = (bt.n + 255) / 256;
    if ((int)blockIdx.x >= nblk_pts) {
        // the workgroups behind the points': when the tables were rebuilt, the units' padding rows (m .. mp) of the
        // coordinate pool are zeroed — 16 lanes per unit (one single-workgroup kernel doing this was store-issue bound)
        if (!rebuild || bt.ctl[CTL_OVERFLOW]) return;
        int idx = ((int)blockIdx.x - nblk_pts) * 256 + threadIdx.x;
        int u = idx >> 4, r = idx & 15;
        if (u >= bt.n_local) return;
        int m = bt.m[u];
        if (m + r < ((m + 15) & ~15)) {
            double *xr = bt.Xu + (size_t)(bt.row_off[u] + m + r) * bt.xstride;
            for (int e = 0; e < bt.xstride; ++e) xr[e] = 0.0;
        }
        return;
    }
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= bt.n) return;
    int b = bt.assign[p];
    if (b < 0) {
        if (rebuild) { bt.pe[2 * p] = 0; bt.pe[2 * p + 1] = 0; }
        return;
    }
    int pos;
    if (rebuild && from_chunks) {
        pos = bt.cnt[(size_t)(p / CHUNK) * bt.n_blocks + b] + bt.rank[p];
        bt.posb[p] = pos;
    } else {
        pos = bt.posb[p];
    }
    if (bt.ctl[CTL_OVERFLOW]) return;
    double x[3];
    load_point(X, dx, p, x);
    scatter_rows(bt, x, geo, p, bt.bu_ptr[b], bt.bu_ptr[b + 1], pos, rebuild,
                 [&](int u) { return UnitRows{bt.row_off[u], bt.off_j[u], rebuild ? bt.m[u] : 0}; },
                 [&](int e) { return bt.bu_ent[e]; });
}

// k_build_scatter: k_build (both launches) and k_scatter_x as ONE launch for a partition that came from k_assign / k_route
// in this evaluation.  Three dependent launches of tiny kernels cost 28 us of a 430 us evaluation, nearly all of it launch
// ramps, drains and exposed memory round trips; an arrival ticket between them was no better (a grid-wide wait is a launch
// boundary by another name).  Here nobody waits for anybody: EVERY workgroup derives what it needs by itself, in LDS — a
// histogram of the whole block assignment (n words, read once as int4: the block sizes, and how many points of each block
// come before the workgroup's own 256), then the unit scan (sizes, row offsets) — and scatters its 256 points; one more
// workgroup (the last) only writes the tables (unit_tables).  The redundant work is a few thousand integer operations per
// workgroup.  Limits (else the three-launch path): FB_MAX_BLOCKS blocks, FB_MAX_UNITS local units, FB_MAX_POINTS points,
// FB_MAX_ENT entries of the block -> units CSR.
constexpr int FB_MAX_BLOCKS = 1024, FB_MAX_UNITS = 2048, FB_MAX_POINTS = 1 << 15, FB_MAX_ENT = 4096;
__global__ __launch_bounds__(256) void k_build_scatter(BuildTab bt, const double *__restrict__ X, int dx, int geo, int force,
                                                       int epoch) {
    static_assert(SCAN_THREADS == 256 && CHUNK == 64, "four chunks per workgroup");
    __shared__ long long sh[8];
    __shared__ int s_maxm;
    __shared__ int s_lo[FB_MAX_BLOCKS], s_bsize[FB_MAX_BLOCKS], s_pref[4][FB_MAX_BLOCKS];
    __shared__ int s_m[FB_MAX_UNITS], s_ro[FB_MAX_UNITS];
    __shared__ unsigned s_x[FB_MAX_UNITS];      // table workgroup: mat_off >> 8; the others: off_j
    __shared__ int s_buptr[FB_MAX_BLOCKS + 1], s_buent[FB_MAX_ENT];
    const int npw = (bt.n + 255) / 256;         // point workgroups; workgroup npw writes the tables
    const int t = threadIdx.x;
    const bool table_wg = (int)blockIdx.x == npw;
    const int nb = bt.n_blocks, p0 = 256 * (int)blockIdx.x, n = bt.n;
    const int p = p0 + t;
    // Round trip 1 — everything that depends on nothing, asked for at once (each dependent load of this kernel is an
    // exposed trip to HBM: with the control word, the point's block, the CSR range of the block and its entries read one
    // after the other the kernel took 17.7 us): the control words, the point's own words and coordinates, the first batch
    // of the assignment histogram, the unit scan's block ids, and the static block -> units CSR (into LDS).
    const int ctl_changed = bt.ctl[CTL_CHANGED], ctl_over = bt.ctl[CTL_OVERFLOW];
    const int b = p < n ? bt.assign[p] : -1;
    const int rank_p = p < n ? bt.rank[p] : 0, posb_p = p < n ? bt.posb[p] : 0;
    double x[3] = {0.0, 0.0, 0.0};
    if (p < n) load_point(X, dx, p, x);
    int pre[3][2] = {{0, 0}, {-1, -1}, {0, 0}};      // unit_bi, unit_bj, ids (the last for the table workgroup)
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (t + 256 * q < bt.n_local) {
            pre[0][q] = bt.unit_bi[t + 256 * q];
            pre[1][q] = bt.unit_bj[t + 256 * q];
            if (table_wg) pre[2][q] = bt.ids[t + 256 * q];
        }
    const int4 *a4 = reinterpret_cast<const int4 *>(bt.assign);
    const int n4 = n >> 2;
    int4 hv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) hv[q] = t + 256 * q < n4 ? a4[t + 256 * q] : int4{-1, -1, -1, -1};
    if (!table_wg) {
        int bp[(FB_MAX_BLOCKS + 256) / 256], be[FB_MAX_ENT / 256];
#pragma unroll
        for (int q = 0; q < (FB_MAX_BLOCKS + 256) / 256; ++q) bp[q] = t + 256 * q <= nb ? bt.bu_ptr[t + 256 * q] : 0;
#pragma unroll
        for (int q = 0; q < FB_MAX_ENT / 256; ++q) be[q] = t + 256 * q < bt.n_ent ? bt.bu_ent[t + 256 * q] : 0;
#pragma unroll
        for (int q = 0; q < (FB_MAX_BLOCKS + 256) / 256; ++q)
            if (t + 256 * q <= nb) s_buptr[t + 256 * q] = bp[q];
#pragma unroll
        for (int q = 0; q < FB_MAX_ENT / 256; ++q)
            if (t + 256 * q < bt.n_ent) s_buent[t + 256 * q] = be[q];
    }
    const bool rebuild = force || ctl_changed == epoch;
    if (table_wg && !rebuild) return;
    bool over = false;
    if (rebuild) {
        for (int k = t; k < nb; k += 256) {
            s_lo[k] = 0; s_bsize[k] = 0;        // (s_bsize: the points from p0 on, until the two are added)
            s_pref[0][k] = 0; s_pref[1][k] = 0; s_pref[2][k] = 0; s_pref[3][k] = 0;
        }
        __syncthreads();
        {
            auto count = [&](const int4 &v, int idx) {
                const int bb[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (bb[k] >= 0) atomicAdd(idx + k < p0 ? &s_lo[bb[k]] : &s_bsize[bb[k]], 1);
            };
#pragma unroll
            for (int q = 0; q < 8; ++q) count(hv[q], 4 * (t + 256 * q));
            for (int i0 = t + 8 * 256; i0 < n4; i0 += 8 * 256) {      // (more than 8192 points: eight loads in flight per thread)
                int4 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = i0 + 256 * q < n4 ? a4[i0 + 256 * q] : int4{-1, -1, -1, -1};
#pragma unroll
                for (int q = 0; q < 8; ++q) count(v[q], 4 * (i0 + 256 * q));
            }
            if (4 * n4 + t < n) {
                int bb = bt.assign[4 * n4 + t];
                if (bb >= 0) atomicAdd(4 * n4 + t < p0 ? &s_lo[bb] : &s_bsize[bb], 1);
            }
            if (!table_wg && b >= 0) atomicAdd(&s_pref[t >> 6][b], 1);      // the workgroup's own four chunks
        }
        __syncthreads();
        for (int k = t; k < nb; k += 256) {
            // s_pref[j] = points of block k before chunk j of this workgroup
            int lo = s_lo[k], o0 = s_pref[0][k], o1 = s_pref[1][k], o2 = s_pref[2][k];
            s_pref[0][k] = lo; s_pref[1][k] = lo + o0; s_pref[2][k] = lo + o0 + o1; s_pref[3][k] = lo + o0 + o1 + o2;
            s_bsize[k] += lo;
        }
        __syncthreads();
        if (table_wg) {
            for (int k = t; k < nb; k += 256) bt.bsize[k] = s_bsize[k];
            unit_tables<FB_MAX_UNITS>(bt, s_bsize, s_m, s_ro, s_x, sh, &s_maxm, pre);
            return;
        }
        // the unit scan again, for this workgroup's own use: sizes, first rows, where the second block starts
        if (t == 0) s_maxm = 0;
        __syncthreads();
        long long rows = 0, mat = 0;
        for (int l0 = 0; l0 < bt.n_local; l0 += 256) {
            int l = l0 + t;
            int m = 0, mi = 0;
            if (l < bt.n_local) {
                int bi = l0 < 512 ? pre[0][l0 >> 8] : bt.unit_bi[l];
                int bj = l0 < 512 ? pre[1][l0 >> 8] : bt.unit_bj[l];
                mi = s_bsize[bi];
                m = mi + (bj >= 0 ? s_bsize[bj] : 0);
            }
            long long mp = (m + 15) & ~15;
            long long a = mp, b2 = (mp * mp) >> 8, ta, tb;
            wg_exscan2(a, b2, sh, &ta, &tb);
            tb <<= 8;
            if (l < bt.n_local) {
                s_m[l] = m;
                s_ro[l] = (int32_t)(rows + a);
                s_x[l] = (unsigned)mi;
                atomicMax(&s_maxm, m);
            }
            rows += ta;
            mat += tb;
        }
        __syncthreads();
        const int maxm = s_maxm;
        over = rows > bt.cap_rows || mat > bt.cap_mat || (((maxm + 15) & ~15) >> 4) > bt.maxT_bound || maxm > MAX_MP;
        // (a launch list outgrowing its grid is found by the table workgroup alone; the rows written here are inside the
        // workspace all the same, and the evaluation is repeated)
        // the units' padding rows (m .. mp) of the coordinate pool, dealt over the point workgroups
        for (int idx = (int)blockIdx.x * 256 + t; !over && idx < bt.n_local * 16; idx += npw * 256) {
            int u = idx >> 4, r = idx & 15;
            int m = s_m[u];
            if (m + r < ((m + 15) & ~15)) {
                double *xr = bt.Xu + (size_t)(s_ro[u] + m + r) * bt.xstride;
                for (int e = 0; e < bt.xstride; ++e) xr[e] = 0.0;
            }
        }
    } else {
        if (ctl_over) return;
        __syncthreads();      // the CSR copy in LDS
    }
    if (p >= n) return;
    if (b < 0) {
        if (rebuild) { bt.pe[2 * p] = 0; bt.pe[2 * p + 1] = 0; }
        return;
    }
    const int e_first = s_buptr[b], e_end = s_buptr[b + 1];
    if (rebuild) {
        int pos = s_pref[t >> 6][b] + rank_p;
        bt.posb[p] = pos;      // (also when the partition does not fit: the repeated evaluation builds from posb / bsize)
        if (over) return;
        scatter_rows(bt, x, geo, p, e_first, e_end, pos, true, [&](int u) { return UnitRows{s_ro[u], (int)s_x[u], s_m[u]}; },
                     [&](int e) { return s_buent[e]; });
    } else {
        scatter_rows(bt, x, geo, p, e_first, e_end, posb_p, false, [&](int u) { return UnitRows{bt.row_off[u], bt.off_j[u], 0}; },
                     [&](int e) { return s_buent[e]; });
    }
}

// whether the single-launch form applies to this partition
bool build_scatter_fits(const BuildTab &bt) {
    return diag("fused_build", 1) != 0 && bt.n > 0 && bt.n_blocks > 0 && bt.n_blocks <= FB_MAX_BLOCKS && bt.n_local <= FB_MAX_UNITS &&
           bt.n <= FB_MAX_POINTS && bt.n_ent <= FB_MAX_ENT;
}
void launch_build_scatter(const BuildTab &bt, const double *X, int dx, int dist_id, int force, int epoch, hipStream_t s) {
    hipLaunchKernelGGL(k_build_scatter, dim3((bt.n + 255) / 256 + 1), dim3(256), 0, s, bt, X, dx, dist_id == 1 ? 1 : 0, force, epoch);
}

void launch_build_tables(const BuildTab &bt, int from_chunks, int force, int epoch, hipStream_t s) {
    // (two launches: merged into one with an arrival ticket they took 19.8 us against 4.9 + 13.3 — the unit scan is a
    // chain of dependent memory round trips either way)
    if (from_chunks && bt.n_blocks > 0)
        hipLaunchKernelGGL(k_build, dim3((bt.n_blocks + 3) / 4), dim3(SCAN_THREADS), 0, s, bt, 1, force, epoch);
    hipLaunchKernelGGL(k_build, dim3(1), dim3(SCAN_THREADS), 0, s, bt, 0, force, epoch);
}

void launch_scatter_x(const BuildTab &bt, const double *X, int dx, int dist_id, int from_chunks, int force, int epoch,
                      hipStream_t s) {
    if (bt.n == 0 && bt.n_local == 0) return;
    hipLaunchKernelGGL(k_scatter_x, dim3((bt.n + 255) / 256 + (bt.n_local * 16 + 255) / 256), dim3(256), 0, s, bt, X, dx, dist_id == 1 ? 1 : 0, from_chunks,
                       force, epoch);
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

void launch_potrf(const UnitTab &ut, const Pools &p, const KParams &kp, bool gen, hipStream_t s, const SideQueue &side) {
    if (ut.n_ids == 0) return;
    hipStream_t s2 = side.s2;
    const int stamps = diag("potrf_stamps", 0);      // diagnostic builds (-DGPRF_PROFILE): in-kernel cycle stamps into Pools::dbg
    // every unit of up to 20 tiles per edge on the register-resident kernels (28 with its waiting tiles in the U pool), the
    // generic kernel above that
    int reg_maxT = potrf_use_reg() ? POTRF_REG8_MAXT : 0;
    // units of more than reg_maxT tiles per edge: the generic kernel, from the K pool (its workgroups leave the others alone)
    auto launch_generic = [&]() {
        if (ut.max_T <= reg_maxT) return;
        // the few units of 21 .. 32 tiles per edge of a generating launch: the eight-wave kernel with its waiting tiles in the
        // U pool, from the K pool (they were filled), behind the generating kernels; the generic kernel above that
        if (gen && potrf_gw() && reg_maxT == POTRF_REG8_MAXT) {
            launch_reg8w(dim3(ut.n_ids), s, ut, p, stamps, kp, POTRF_REG8_MAXT + 1);
            reg_maxT = POTRF_REG8W_MAXT;
            if (ut.max_T <= reg_maxT) return;
        }
        const int capG = ut.max_T < BIG_LA_T ? ut.max_T : BIG_LA_T;      // (larger units: launch_big_potrf)
        size_t ldsg = (size_t)(16 * (16 * capG + 16) + 256 + 16 + 16 * 17 + 256 + 16 * capG) * sizeof(double);
        if (lds_needs_optin(1, ldsg))
            (void)hipFuncSetAttribute((const void *)k_potrf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsg);
        hipLaunchKernelGGL(k_potrf, dim3(ut.n_ids), dim3(POTRF_WAVES * 64), ldsg, s, ut, p, stamps, reg_maxT);
    };
    if (!reg_maxT) {
        launch_generic();
        return;
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
        return;
    }
    if (ut.max_T <= POTRF_SMALL_MAXT) {      // every unit has at most 13 tiles: the two-per-CU kernel alone
        size_t ldsS = (size_t)(16 * POTRF_REG2_LDP + 256 + 16 + 256 + 16 * POTRF_REG_MAXT_C + 256 * capT + 16 * capT * XPAD) * sizeof(double);
        launch_reg2(dim3(ut.n_ids), ldsS, s, ut, p, stamps, POTRF_SMALL_MAXT, kp, 0);
        return;
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
}

void launch_solve(const UnitTab &ut_all, const Pools &p, const KParams &kp, hipStream_t s) {
    if (ut_all.n_ids == 0) return;
    // PM: the grid walked part by part (part_major_map) — launches at most two rounds of CUs wide; diag part_major=0 / 1 forces
    const int pm_d = diag("part_major", -1);
    const bool pm = pm_d >= 0 ? pm_d == 1 : ut_all.n_ids <= 2 * device_cus();
    // (GPRF_ONLY_POTRF: tests/test_isa_invariants.py compiles this file for the Cholesky kernels' ISA alone — the dozen
    // unrolled k_solve_panel / k_mgrad instantiations are two thirds of the compile time)
#ifndef GPRF_ONLY_POTRF
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
            if (ut.max_T <= 26)           // (one panel buffer, two workgroups per CU: the paper-scale catalogue's pairs of 390 points)
                hipLaunchKernelGGL((k_solve_panel<26, 2, true, 1>), grid, dim3(256), 0, s, utp, p, kp.dy);
            else if (ut.max_T <= 28)      // (448 points: the seismic configuration's pairs at every block size below 210; 11 % faster
                                          // there than the 32-tile instantiation)
                hipLaunchKernelGGL((k_solve_panel<28, 1, true>), grid, dim3(256), 0, s, utp, p, kp.dy);
            else                          // (units of up to 512 points, one workgroup per CU)
                hipLaunchKernelGGL((k_solve_panel<SOLVE_PANEL_MAXT, 1, true>), grid, dim3(256), 0, s, utp, p, kp.dy);
        }
    }
#endif
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
    hipLaunchKernelGGL(k_at_wide, dim3(xcd_grid(ut.n_ids, (ut.max_T + 15) / 16)), dim3(256), 0, s, ut, p, first_round, skip_T);
}

void launch_gx_finalize(const UnitTab &ut, const Pools &p, const KParams &kp, int want_gc, hipStream_t s) {
    if (ut.n_units == 0) return;
    hipLaunchKernelGGL(k_gx_finalize, dim3(ut.n_units), dim3(256), 0, s, ut, p, kp, want_gc);
}

void launch_grad(int dist_id, int kern_id, const UnitTab &ut, const Pools &p, const KParams &kp, int want_gc,
                 bool have_K, hipStream_t s) {
    if (ut.n_ids == 0 || ut.max_T == 0) return;
#ifndef GPRF_ONLY_POTRF
    int TBm = (ut.max_T + 3) / 4;
    // part by part (longest workgroups first): launch-wide while the launch is at most two rounds of CUs wide; deeper launches
    // in GROUPS of 64 launch slots — a unit's workgroups then run within one L2 residency window and longest first inside the
    // group (C4: 762 -> 738 us; launch-wide there the ten workgroups of a unit run far apart and each fetches the unit's W / At
    // from HBM again: 809).  diag part_major=0: unit by unit.
    const int pm = diag("part_major", 1) != 0 ? 1 : 0;
    // (round 5: the lld / Matern instantiation always in groups — its block pairs re-read W / At at 482 MB per launch walked
    // launch-wide on the seismic shape — and never with fewer than two groups: part_major_map then walks launch-wide)
    const int G = (ut.n_ids > 2 * device_cus() || (dist_id == 1 && ut.n_ids > 128)) ? 64 : 0;
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
        const int nt = (16 * ut.max_T + BGT - 1) / BGT;
        hipLaunchKernelGGL(k_big_gemm, dim3(nt * (nt + 1) / 2, ut.n_ids), dim3(256), 0, s, ut, p, 2, 0, 0, nt, (double)kp.dy);
        if (dist_id == 0 && kern_id == 0) hipLaunchKernelGGL((k_mgrad<0, 0, false, 0, true>), grid, dim3(256), 0, s, utp, p, kp, want_gc, pm);
        else hipLaunchKernelGGL((k_mgrad<1, 1, false, 0, true>), grid, dim3(256), 0, s, utp, p, kp, want_gc, pm);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// k_pair_max: threshold neighbour discovery (gprf.py:119-150): for a candidate block pair (i, j) the largest
// |k(x_p, x_q)| / signal_var over p in block i, q in block j — one workgroup per candidate, 64 x 64 point tiles (the
// coordinates / great-circle records of the tile's points wait in LDS), the pair is decided as soon as one tile holds
// a value above the threshold (want_max = 0), exactly the reference's `np.max(np.abs(K / wfn_var)) > threshold`.
// ------------------------------------------------------------------------------------------------
template <int DIST, int KERN>
__global__ __launch_bounds__(256) void k_pair_max(const double *__restrict__ X, int dx, const int64_t *__restrict__ blk_ptr,
                                                  const int32_t *__restrict__ blk_pts, const int32_t *__restrict__ cand,
                                                  KParams kp, double thr, int want_max, int32_t *__restrict__ keep,
                                                  double *__restrict__ max_out) {
    constexpr int XN = PtRec<DIST>::NREG;
    __shared__ double xi[64][XN], xj[64][XN];
    __shared__ double wred[4];
    int c = blockIdx.x;
    int bi = cand[2 * c], bj = cand[2 * c + 1];
    int64_t i0 = blk_ptr[bi], i1 = blk_ptr[bi + 1], j0 = blk_ptr[bj], j1 = blk_ptr[bj + 1];
    int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    auto load = [&](double (*dst)[XN], int64_t p0, int64_t p1) {
        if (t < 64) {
            int64_t k = p0 + t;
            double r[XN];
#pragma unroll
            for (int d = 0; d < XN; ++d) r[d] = 0.0;
            if (k < p1) {
                const double *x = X + (size_t)blk_pts[k] * dx;
                if constexpr (DIST == 1) {
                    double hl = x[1] * DEG2RAD / 2.0, hn = x[0] * DEG2RAD / 2.0;
                    r[GEO_SLH] = sin(hl); r[GEO_CLH] = cos(hl); r[GEO_SNH] = sin(hn); r[GEO_CNH] = cos(hn); r[GEO_Z] = x[2];
                } else {
#pragma unroll
                    for (int d = 0; d < 3; ++d) if (d < dx) r[d] = x[d];
                }
            }
#pragma unroll
            for (int d = 0; d < XN; ++d) dst[t][d] = r[d];
        }
    };
    double best = 0.0;
    const double inv_sv = 1.0 / kp.sv;
    for (int64_t a = i0; a < i1; a += 64) {
        __syncthreads();
        load(xi, a, i1);
        for (int64_t b = j0; b < j1; b += 64) {
            __syncthreads();
            load(xj, b, j1);
            __syncthreads();
            double m = 0.0;
            if (b + lane < j1) {
                double xq[XN];
#pragma unroll
                for (int d = 0; d < XN; ++d) xq[d] = xj[lane][d];
#pragma unroll 4
                for (int q = 0; q < 16; ++q) {
                    int r = wave + 4 * q;
                    if (a + r < i1) {
                        double v = fabs(KernFn<DIST, KERN>::value(kp, xi[r], xq) / kp.sv);      // |K / wfn_var| (gprf.py:141)
                        m = v > m ? v : m;
                    }
                }
            }
            (void)inv_sv;
            best = m > best ? m : best;
            if (!want_max && __syncthreads_or(m > thr)) {
                if (t == 0) keep[c] = 1;
                return;
            }
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        double o = shfl_xor_d(best, off);
        best = o > best ? o : best;
    }
    if (lane == 0) wred[wave] = best;
    __syncthreads();
    if (t == 0) {
        double mx = wred[0];
        for (int w = 1; w < 4; ++w) mx = wred[w] > mx ? wred[w] : mx;
        keep[c] = mx > thr ? 1 : 0;
        if (max_out) max_out[c] = mx;
    }
}

void launch_pair_max(int dist_id, int kern_id, const double *X, int dx, const int64_t *blk_ptr, const int32_t *blk_pts,
                     const int32_t *cand, int n_cand, const KParams &kp, double thr, int want_max, int32_t *keep,
                     double *max_out, hipStream_t s) {
    if (n_cand == 0) return;
    if (dist_id == 0 && kern_id == 0)
        hipLaunchKernelGGL((k_pair_max<0, 0>), dim3(n_cand), dim3(256), 0, s, X, dx, blk_ptr, blk_pts, cand, kp, thr, want_max, keep, max_out);
    else
        hipLaunchKernelGGL((k_pair_max<1, 1>), dim3(n_cand), dim3(256), 0, s, X, dx, blk_ptr, blk_pts, cand, kp, thr, want_max, keep, max_out);
}

// k_done: the last kernel of a host-in / host-out evaluation: everything before it on the stream has completed
// (kernel boundary), so one store of the evaluation's sequence number into pinned host memory tells a polling host
// that the result is there — a few microseconds instead of the runtime's stream-synchronisation path.
// (Round 4 measured the word written from INSIDE the assembly instead — every workgroup fences its result stores at system
// scope and takes a ticket, the last one stores the word; one launch less: 0.409 ms per step against 0.387 — 314 workgroups'
// system-scope fences cost four times what the 4 us launch does.  Dropped.)
__global__ void k_done(int32_t *flag, int32_t seq) {
    if (threadIdx.x == 0) {
        __atomic_store_n(flag, seq, __ATOMIC_RELEASE);
    }
}

void launch_done(int32_t *flag, int32_t seq, hipStream_t s) { hipLaunchKernelGGL(k_done, dim3(1), dim3(64), 0, s, flag, seq); }

// k_sum_parts (single-process multi-device evaluation): the member contexts' partial result vectors — written by their
// assembly kernels straight into device 0's memory (peer stores over xGMI) — added in member order (fixed: reproducible)
// into the front context's pinned host vector.  [ll | gradX | gradC | s0 | s1]: every word is a sum.
__global__ __launch_bounds__(256) void k_sum_parts(const double *__restrict__ slots, int n_parts, size_t stride, size_t nvec,
                                                   double *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    double acc = slots[i];
    for (int k = 1; k < n_parts; ++k) acc += slots[(size_t)k * stride + i];
    out[i] = acc;
}

void launch_sum_parts(const double *slots, int n_parts, size_t stride, size_t nvec, double *out, hipStream_t s) {
    hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, s, slots, n_parts, stride, nvec, out);
}

void launch_assemble(const UnitTab &ut, const Pools &p, const AssembleTab &at, const KParams &kp, int n,
                     int want_gx, int want_gc, double *out, int usum_ok, const ObjTab &ob, hipStream_t s) {
    int blocks = 1 + (n + 31) / 32;
    hipLaunchKernelGGL(k_assemble, dim3(blocks), dim3(256), 0, s, ut, p, at, kp, n, want_gx, want_gc, out, usum_ok, ob);
}

}  // namespace gprf
