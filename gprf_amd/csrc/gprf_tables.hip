// gprf_tables.hip — everything around the math: the launchers' shared host helpers and the GPRF_DIAG switch, re-blocking on
// the device (k_assign / k_route; gprf.py:169-174), the unit tables and the coordinate scatter (k_build*, k_scatter_x;
// gprf.py:299-330), neighbour discovery (k_pair_max; gprf.py:119-150), k_done, k_sum_parts.
#include <cstdlib>
#include <cstring>
#include "gprf_dev.h"

namespace gprf {

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// dynamic LDS above 48 KB has to be opted into per kernel AND per device: remembers the largest size already
// granted for (kernel slot, current device)
bool lds_needs_optin(int kernel_slot, size_t lds) {
    static size_t granted[12][64] = {};
    if (lds <= 48 * 1024) return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (lds <= granted[kernel_slot][dev]) return false;
    granted[kernel_slot][dev] = lds;
    return true;
}

// compute units of the current device (kernel variants are picked by how many workgroup rounds a launch is deep)
int device_cus() {
    static int n_cus = 0;
    if (n_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    return n_cus;
}

int xcd_grid(int n_ids, int nparts) { return ((n_ids + 7) / 8) * 8 * nparts; }
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
//   solve_class=0   every stage behind the Cholesky as ONE launch over all units (rounds 1-5); class_depth=<1..3>: how many of them run by size class
//   tail_swap=0     the by-class pipelines joined into the main queue (default: into the side queue, which carries the longer pipeline)
//   tool_env=0      a tool is loaded but does not serialise the queues (rocprofv3 --kernel-trace without counters): keep the product's
//                   launch structure (scripts/profile_run.sh)          mgrad_group=<G>  the gradient grid walked in groups of G launch slots (0 = launch-wide)
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

}  // namespace gprf
