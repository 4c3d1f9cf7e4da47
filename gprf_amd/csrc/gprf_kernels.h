// gprf_kernels.h — shared declarations between the HIP kernels (gprf_<stage>.hip, device helpers in gprf_dev.h) and the C-ABI host
// layer (gprf_capi.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gprf {

constexpr int TILE = 16;          // MFMA f64 16x16x4 tile edge
constexpr int MAX_MP = 16384;      // largest padded unit (GPRF_MAX_UNIT)
constexpr int MAX_T = MAX_MP / TILE;
constexpr int SMALL_MAX_T = 64;    // units of up to 64 tiles per edge (1024 points) run one workgroup per unit and stage; larger
                                   // ones ("big" units) go through the blocked multi-launch path (k_big_*)
constexpr int BIG_LA_T = 32;       // units of more than 32 tiles per edge (512 points): Cholesky and forward substitution by the blocked
                                   // path already (k_solve_panel's limit) — 25 blocks + 72 pairs of n = 10000 (pairs of ~800 points):
                                   // 5.3 -> 3.4 ms per evaluation; their At and gradient stay with k_at / k_mgrad up to SMALL_MAX_T
constexpr int YPAD = 64;          // dy padded to 4 column tiles
constexpr int XPAD = 4;           // dx padded (dx <= 3)
constexpr int GC_SLOTS = 8;       // per-(unit, column-tile) hyper-gradient partials
constexpr int CHUNK = 64;         // points per workgroup of the partition kernels (= per row of BuildTab::cnt)

// Kernel hyper-parameters, passed by value.  theta = [nv, sv, ls...] (gprf.py:160-164).
struct KParams {
    double nv, sv;
    double ls[3];
    double inv_ls[3];   // 1 / ls[d], rounded once on the host
    int dx, ndfn, dy;
};

// Per-unit tables (device pointers), one entry per LOCAL unit.  m / row_off / mat_off / off_j / upt are written ON THE
// DEVICE from the partition (k_build, k_scatter_x): nothing about a re-blocking returns to the host.
// One 16-byte record per launch slot: everything a workgroup needs to find its unit, in ONE load instead of the chain
// ids[slot] -> m[u] -> row_off[u] -> mat_off[u] (four dependent memory round trips at the head of every workgroup of
// every kernel: 12-17 k cycles before the first useful instruction, measured).  Written by k_build next to the tables.
struct SlotRec {
    int32_t u;          // local unit id
    int32_t m;          // points (0 when the build overflowed)
    int32_t row_off;    // = row_off[u]
    uint32_t mat256;    // = mat_off[u] / 256 (mp is a multiple of 16, so every offset is a multiple of 256)
};

struct UnitTab {
    const SlotRec *srec;     // [n_ids] in launch order (ids)
    const SlotRec *big_rec, *small_rec;   // the same records in the order of big_list / small_list
    const int32_t *m;        // points in the unit (0 for every unit when the build overflowed the workspace: ctl)
    const int32_t *row_off;  // padded-row offset: sum of mp over previous units
    const int64_t *mat_off;  // element offset of the unit's mp x mp matrices in the U / W pools
    const double *weight;    // Bethe weight: 1 - deg(i) for unaries, 1 for pairs
    const double *jitter;    // extra diagonal (jitchol retry)
    const int32_t *upt;      // [total padded rows] global point index of each unit row; rows >= m are NOT written
    const int32_t *ids;      // the local unit ids in launch order (largest first at the last host build) ...
    int n_ids;               // ... and how many
    int n_launch;            // the evaluation's WHOLE launch (= n_ids unless this table is one part of a split launch: the
                             // pipelined halves of enqueue_eval): what picks a kernel FORM, so that a unit's arithmetic does not
                             // depend on how the launch was split
    int n_units;
    int max_T;               // launch-wide bound on mp/16 (the largest unit at the last synchronised build)
    // the Cholesky's two launch lists, built on the device with the tables (k_build): units of more than / at most
    // potrf_small_maxT() tiles, each in the order of ids; their lengths are ctl[CTL_NBIG] / ctl[CTL_NSMALL]
    const int32_t *big_list, *small_list;
    const int32_t *ctl;
    int grid_big, grid_small;   // workgroups the two lists are launched with (the build reports a list that is longer)
    int pm_group;               // part-major grids (solve, gradient): group size of the walk, 0 = launch-wide (part_major_map)
    int n_wide;                 // units of more than 20 tiles per edge at the last synchronised partition (potrf_generates_K)
    // fork of the two Cholesky queues: the first workgroup of the large-unit kernel stores fork_seq here as its first
    // instruction; the side queue's small-unit kernel sits behind a stream wait for that value (nullptr: none)
    uint32_t *fork_flag;
    uint32_t fork_seq;
};

// What the device-side table build works from and leaves behind (all device pointers).
struct BuildTab {
    int32_t *assign;         // [n] block of every point, -1 = in no block
    int32_t *posb;           // [n] position of the point inside its block
    int32_t *rank;           // [n] scratch: position among the points of the same block within its CHUNK-point chunk
    int32_t *cnt;            // [n_chunks][n_blocks] points of a block per chunk -> (k_build) exclusive prefix over chunks
    int32_t *bsize;          // [n_blocks] points per block
    const int32_t *unit_bi;  // [n_local] first block of the unit
    const int32_t *unit_bj;  // [n_local] second block, -1 for a unary unit
    const int32_t *bu_ptr;   // [n_blocks + 1] CSR: block -> the local units that contain it, ascending unit id ...
    const int32_t *bu_ent;   // ... as 2 * unit + side (side 1 = the block's rows come second in the unit)
    const int32_t *ids;      // [n_local] launch order (largest first at the last host build)
    int32_t *big_list, *small_list;   // [n_local] each: the units of more than / at most small_maxT tiles, in ids order
    SlotRec *srec, *big_rec, *small_rec;   // [n_local] each: the launch-slot records (UnitTab)
    int32_t *pe, *ebase;     // AssembleTab's per-point / per-entry shortcuts, rebuilt with the partition
    int32_t *einfo;          // per entry: (the block's first row inside the unit, local) << 8 | the unit's 64-point blocks
    int small_maxT;          // 0 = no split
    int grid_big, grid_small;         // launch sizes the lists must fit
    int32_t *m;              // the UnitTab columns this build writes
    int32_t *row_off;
    int64_t *mat_off;
    int32_t *off_j;          // [n_local] rows of the unit's first block (= where the second block's rows start)
    int32_t *upt;
    double *Xu;              // the coordinate pool (k_scatter_x) and its row stride in doubles
    int xstride;
    int32_t *ctl;            // control / result words, CTL_* below
    int n, n_blocks, n_local, n_chunks;
    int n_ent;               // entries of the block -> units CSR (= bu_ptr[n_blocks])
    int64_t cap_rows, cap_mat;   // workspace capacities the build must stay within
    int maxT_bound;              // the launch-wide max_T the evaluation kernels will be launched with
};
constexpr int CTL_CHANGED = 0;     // = the epoch of the last evaluation in which some point changed block (k_assign / k_route)
constexpr int CTL_OVERFLOW = 1;    // the partition does not fit the workspace / the launch bound: every unit got m = 0
constexpr int CTL_ROWS = 2;        // total padded rows of the partition
constexpr int CTL_MAXT = 3;        // its largest unit in tiles
constexpr int CTL_MAXM = 4;        // ... in points
constexpr int CTL_MAT_LO = 5;      // total matrix elements (64 bit, two words)
constexpr int CTL_MAT_HI = 6;
constexpr int CTL_NBIG = 8;        // units of more than small_maxT tiles per edge in this partition (length of big_list)
constexpr int CTL_NSMALL = 9;      // the others (length of small_list)
constexpr int CTL_BUILDS = 7;      // table builds since the context was created (tests)
constexpr int CTL_WORDS = 10;

// spare doubles behind the last unit's matrix in the U / W / K pools (row-panel loads address whole 64-column
// chunks; the lanes beyond the unit's edge are masked off, the slack keeps even an unmasked variant in bounds)
constexpr size_t GPRF_POOL_SLACK = 512;

// diagnostic builds (-DGPRF_WGTRACE=<kernel id>): workgroup records behind the per-unit stamps in Pools::dbg
constexpr int GPRF_WGTRACE_MAX = 1 << 15;

struct Pools {
    double *K;     // kernel matrices, row-major mp x mp per unit: k_fill writes the 64x64 blocks ti <= tj only
                   // (diagonal blocks whole); read by the Cholesky once and by k_mgrad
    double *U;     // upper Cholesky factors (same layout; the strictly-lower part is never written)
    double *W;     // U^-T (lower triangular, row-major)
    double *V;     // inverses of U's 16x16 diagonal tiles: T tiles per unit at 16*row_off
    double *Vb;    // big units (> 1024 points): inverses of U's 64x64 diagonal blocks, at (row_off + 64 u) * 64 (k_big_diag)
    double *Xu;    // gathered unit coordinates per padded row: XPAD doubles (euclidean) or the 8-double half-angle
                   // record of the lld distance (k_scatter_x)
    const double *Y;  // the outputs, n x dy row-major (gathered through upt where a kernel needs unit rows)
    double *Z;     // U^-T Y[unit rows], YPAD per padded row
    double *At;    // (K^-1 Y[unit rows])^T : per unit YPAD x mp at YPAD*row_off
    double *gXu;   // per-unit-row gradient slab, XPAD per padded row
    double *rowpart;  // k_mgrad: per padded row, TBm x XPAD partial row sums (one per column block JB <= its block)
    double *colpart;  // k_mgrad: per padded row, TBm x XPAD partial column sums (one per row block IB >= its block)
    double *logdet;   // per unit
    double *zzpart;   // per unit x 4 : partial sums of ||Z||_F^2 per Y column block
    double *usum;     // per unit x 8 : the unit's weighted terms of ll and gradC (k_gx_finalize -> k_assemble)
    double *gcpart;   // per unit x TBm (TBm + 1) / 2 block pairs x GC_SLOTS,  TBm = ceil(max_T / 4)
    int32_t *info;    // per unit: 0 ok, k>0 = non-positive pivot at row k-1
    double *dbg;      // per unit x 8: in-kernel cycle stamps of diagnostic builds (GPRF_POTRF_ABLATE & 16)
};

struct AssembleTab {
    const int32_t *assign;     // [n] block of every point
    const int32_t *posb;       // [n] its position inside the block
    const int32_t *bu_ptr;     // block -> local units CSR (BuildTab)
    const int32_t *bu_ent;
    const int32_t *off_j;
    const int32_t *ctl;
    // per point (first CSR entry of its block, number of entries) and per CSR entry (first unit row of the block inside
    // that unit; the unit's Bethe weight): the point's rows in two dependent loads instead of four
    const int32_t *pe;         // [2 n], written by k_scatter_x with the partition
    const int32_t *ebase;      // [entries], written by k_scatter_x with the partition
    const int32_t *einfo;      // [entries], likewise: (local row of the block's first point) << 8 | 64-point blocks of the unit
    int fold_gx;               // 1: k_gx_finalize did not run — the gradient partials are folded here, per (point, unit)
    const double *ewgt;        // [entries], static
    // the context's result words [ctl | info | bsize] are mirrored by the assembly kernel into (host-visible) memory,
    // so that a host-in / host-out evaluation needs no copy command at all; mirror_dst = nullptr: no mirror
    const int32_t *mirror_src;
    int32_t *mirror_dst;
    int mirror_n;
};

constexpr int GX_FOLD_MAX_UNITS = 1024;      // local units up to which k_assemble folds the gradient partials itself

// The optimiser-facing form of the result (gprf_objective; gprfopt.py:377-417): the assembly adds the location prior's
// terms and flips the signs, so that what comes down in the evaluation's one download is what scipy's minimiser consumes.
struct ObjTab {
    int on;                  // 0: the plain (ll, gradX, gradC)
    const double *X;         // the points of this evaluation (n x dx)
    const double *Xobs;      // prior means; nullptr: this context adds no location prior (none set, or not the owner rank)
    double sigma, var;       // obs_std and its square
    double *part;            // [gradX workgroups of k_assemble] partial sums of ((x - x_obs) / sigma)^2
};

// re-blocking: nearest centre / split-tree descent of every point, with the per-chunk ranks and counts the table
// build starts from (bt.assign / rank / cnt; ctl[CTL_CHANGED] = epoch when somebody moved)
// GridHint: the centres are a separable, uniformly spaced g x g grid (centre ix * g + iy = (a0 + ix ha, b0 + iy hb) to 1e-9 of
// the spacing; dx = 2): k_assign then evaluates the reference's radicand on the 3 x 3 centres around a point's cell only —
// the same values in the same order as the full scan, whose other centres are farther by at least 1.75 h^2 (g = 0: no grid)
struct GridHint { int g; double a0, inv_ha, b0, inv_hb; };
void launch_assign(const double *X, double *Xcopy, int dx, const double *cs, const double *c2, int nc, const GridHint &gh,
                   const BuildTab &bt, int epoch, hipStream_t s);
void launch_route(const double *X, double *Xcopy, int dx, int dim, int lon_wrap, const double *vec, const double *center,
                  const double *split, const int32_t *left, const int32_t *right, const int32_t *leaf_block,
                  const BuildTab &bt, int epoch, hipStream_t s);
// unit tables from the partition (k_build): block sizes from the chunk counts, unit sizes and offsets.  from_chunks: the partition came from launch_assign / launch_route in this evaluation, otherwise
// bt.posb / bt.bsize are already on the device.  force = 0: only when ctl[CTL_CHANGED] == epoch.
void launch_build_tables(const BuildTab &bt, int from_chunks, int force, int epoch, hipStream_t s);
// every evaluation: coordinates into the unit rows (+ positions and unit row -> point when rebuilding)
void launch_scatter_x(const BuildTab &bt, const double *X, int dx, int dist_id, int from_chunks, int force, int epoch,
                      hipStream_t s);
// both of the above as one launch, for a partition made by launch_assign / launch_route in this evaluation
// (build_scatter_fits: small enough for every workgroup to redo the scans in LDS)
bool build_scatter_fits(const BuildTab &bt);
void launch_build_scatter(const BuildTab &bt, const double *X, int dx, int dist_id, int force, int epoch, hipStream_t s);
// skip_T: units of at most that many tiles per edge are not filled (0 = all)
void launch_fill(int dist_id, int kern_id, const UnitTab &ut, const Pools &p, const KParams &kp, int skip_T, hipStream_t s);
// whether the register-resident Cholesky generates the kernel matrices of its units (at most potrf_gen_maxT() tiles per
// edge) itself; larger units are filled into the K pool and factored by the generic kernel
bool potrf_generates_K(int dist_id, int kern_id, const UnitTab &ut);
int potrf_gen_maxT(int dist_id);   // (20)
int potrf_small_maxT();         // the register-resident Cholesky runs as two instantiations side by side: units of at most this
                                // many tiles per edge two to a CU
int diag(const char *key, int dflt);   // GPRF_DIAG="key=value,...": the one diagnostic switch (gprf_tables.hip)
int potrf_side_mode();          // how the two queues fork / join (launch_potrf): 4 = kernel-written fork word + memory-op join, 0 = events
bool potrf_tool_env();          // a profiler / serialising launch mode is in the environment
// The second queue for the instantiation that runs beside the main one, and how the two queues wait for each other:
// events (11 us per dependency measured, scripts/stream_dep_latency.hip) and, for the join, a stream memory operation on a
// device word (hipStreamWriteValue32 / hipStreamWaitValue32: 4 us) where the device supports it.  s2 = nullptr: one launch.
struct SideQueue {
    hipStream_t s2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    uint32_t *words = nullptr;      // [3] device words: fork, join, fork written by the large-unit kernel itself
    uint32_t seq = 0;               // value of this evaluation (monotonic)
};
// gen: the register kernels generate K themselves (SE)
// class_stages: how many of the stages behind the Cholesky the caller wants next (1 solve, 2 + At, 3 + gradient) — a two-queue
// launch then runs each size class's stages behind that class's Cholesky kernel on its queue and joins the queues behind
// them (round 6); returns how many it ran (the caller skips those launches)
// tail (may be nullptr): where the caller may continue.  When all three stages run by class the SIDE queue's pipeline is the longer
// one; with tail != nullptr the queues then join INTO the side queue (the main queue writes a word behind its last kernel, the
// side queue waits for it — it has been written long before) and *tail = that queue: the assembly follows the critical
// pipeline without a cross-queue hop.  Otherwise *tail = s.
int launch_potrf(const UnitTab &ut, const Pools &p, const KParams &kp, bool gen, hipStream_t s, const SideQueue &side, int class_stages,
                 int want_gc, hipStream_t *tail);
void launch_solve(const UnitTab &ut, const Pools &p, const KParams &kp, hipStream_t s);
// the forward substitution of ONE of the Cholesky's two size classes (which = 1: the large units + the surplus units of the
// small list; 2: the small list), and whether a launch is split that way (diag solve_class=0: never)
void launch_solve_class(const UnitTab &ut, const Pools &p, int dy, int which, hipStream_t s);
void launch_at_class(const UnitTab &ut, const Pools &p, int which, hipStream_t s);
void launch_grad_class(const UnitTab &ut, const Pools &p, const KParams &kp, int want_gc, int which, hipStream_t s);
bool solve_by_class(const UnitTab &ut);
// units of more than 1024 points: blocked Cholesky and forward substitution over whole launches (no-ops when the launch has none)
void launch_big_potrf(const UnitTab &ut, const Pools &p, const KParams &kp, hipStream_t s);
void launch_big_solve(const UnitTab &ut, const Pools &p, hipStream_t s);
void launch_at(const UnitTab &ut, const Pools &p, hipStream_t s);
void launch_grad(int dist_id, int kern_id, const UnitTab &ut, const Pools &p, const KParams &kp, int want_gc,
                 bool have_K, hipStream_t s);
void launch_gx_finalize(const UnitTab &ut, const Pools &p, const KParams &kp, int want_gc, hipStream_t s);
void launch_assemble(const UnitTab &ut, const Pools &p, const AssembleTab &at, const KParams &kp, int n,
                     int want_gx, int want_gc, double *out, int usum_ok, const ObjTab &ob, hipStream_t s);
void launch_finish(double *out, const ObjTab &ob, int nparts, double xp_const, double *extras, int32_t *flag, int32_t seq,
                   hipStream_t s);
void launch_done(int32_t *flag, int32_t seq, hipStream_t s);
// out[i] = sum over the n_parts partial vectors slots[k * stride + i], in member order
void launch_sum_parts(const double *slots, int n_parts, size_t stride, size_t nvec, double *out, hipStream_t s);
// threshold neighbour discovery: keep[c] = max |k| / sv over candidate block pair c > thr (early out unless want_max)
void launch_pair_max(int dist_id, int kern_id, const double *X, int dx, const int64_t *blk_ptr, const int32_t *blk_pts,
                     const int32_t *cand, int n_cand, const KParams &kp, double thr, int want_max, int32_t *keep,
                     double *max_out, hipStream_t s);

}  // namespace gprf
