// gprf_capi.hip — host side of libgprf_hip.so: context, unit tables, workspace pools, the C ABI of
// include/gprf_hip.h.  No torch, no Python; plain pointers and sizes.
//
// Who builds what.  The STATIC tables depend on the neighbour list, the shard and the jitter only: unit -> (block i,
// block j), Bethe weight, block -> units CSR, launch order; they are built here on the host (rebuild_static) and
// uploaded in one staged copy.  Everything that depends on the PARTITION — unit sizes, row / matrix offsets, the unit
// row -> point table, a point's position inside its block — is built on the device (k_build, k_scatter_x) from the
// block of every point, which is either computed on the device too (k_assign / k_route: the re-blocking the
// reference's drivers do before every evaluation, gprf.py:169-174) or uploaded (gprf_set_blocks).  An evaluation that
// re-partitions therefore never returns to the host in the middle: one upload of X, one download of the result
// (+ a few control words: sizes, status), one synchronisation.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

#include "../../include/gprf_hip.h"
#include "gprf_kernels.h"

using namespace gprf;

namespace {

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n, double slack = 1.25) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = (size_t)(n * slack) + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

template <typename T>
struct PinBuf {
    T *p = nullptr;      // host address
    T *d = nullptr;      // the same memory as the device sees it (kernels read / write it directly: zero-copy I/O)
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = d = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 64;
        // pinned, mapped into the device's address space, coherent (fine-grained): what a kernel wrote is visible to
        // the host once the stream has been synchronised
        // (portable: a multi-device group's members on other devices read / write the front context's buffers)
        hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable);
        if (e != hipSuccess) return e;
        e = hipHostGetDevicePointer((void **)&d, (void *)p, 0);
        if (e != hipSuccess) { (void)hipHostFree(p); p = d = nullptr; return e; }
        cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = d = nullptr;
        cap = 0;
    }
};

}  // namespace

struct gprf_ctx {
    int n = 0, dx = 0, dy = 0, dist_id = 0, kern_id = 0, device = 0, ndfn = 0, ncov = 0;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;        // second queue: the Cholesky instantiation that runs beside the main one
    hipStream_t stream3 = nullptr;        // third queue, LOW priority: the second half of the solve -> At -> gradient pipeline
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    DevBuf<uint32_t> d_side;               // fork / join words of the side queue (stream memory operations)
    uint32_t side_seq = 0;
    bool caller_pipelines = false;        // gprf_set_stream_pipelines: the by-class pipelines on a caller's stream too
    GridHint grid_hint = {0, 0.0, 0.0, 0.0, 0.0};      // the centres are a uniform g x g grid (gprf_set_centers): k_assign's fast path
    bool side_values = false;             // the device supports hipStreamWaitValue32
    std::string err;

    // host-side model state (what the reference keeps on the GPRF object)
    std::vector<double> theta;
    bool have_Y = false, have_theta = false, have_blocks = false;
    int n_blocks = 0, n_pairs = 0, n_chunks = 0;
    std::vector<int32_t> pairs;           // (i, j) rows
    std::vector<double> unit_jitter;      // global unit ids
    // the partition as the host knows it: block sizes (exact after every synchronised evaluation); the block of every
    // point / its position inside the block only while they were given by the host (gprf_set_blocks)
    std::vector<int32_t> h_bsize, h_assign_host, h_posb_host;
    bool host_blocks_dirty = false;       // h_assign_host / h_posb_host / h_bsize wait to be uploaded
    bool static_dirty = true;             // neighbour list / shard / block count changed
    bool owner_dirty = true;              // ... in a way that re-deals the units over the ranks
    std::vector<int> owner;               // rank of every unit (global ids)
    bool jitter_dirty = false;
    bool need_build = true;               // the device tables must be rebuilt at the next enqueue, changed or not
    bool assign_valid = false;            // d_assign holds the partition the device tables were built from

    // local units (static part; sizes / offsets mirror the device tables only on demand: refresh_host_units)
    int n_local = 0, max_T = 0;           // max_T: the launch-wide bound
    int shrink_votes = 0;                 // consecutive re-partitions whose largest unit was below the bound
    int64_t cap_rows = 0, cap_mat = 0;
    std::vector<int32_t> l_global, l_bi, l_bj, l_m, l_rowoff;
    std::vector<int64_t> l_matoff;
    int64_t cur_rows = 0, cur_mat = 0;
    double work_flops = 0, work_fill_bytes = 0;

    // device state
    DevBuf<double> d_X, d_Y, d_out;
    template <typename T> struct View { T *p = nullptr; };
    View<int32_t> d_ids, d_unit_bi, d_unit_bj, d_bu_ptr, d_bu_ent;
    View<double> d_ewgt;                  // per CSR entry: its unit's Bethe weight (static)
    int n_ent = 0;                        // entries of the block -> units CSR
    bool gxu_pending = false;             // the last evaluation left gXu to be made on demand (k_gx_finalize)
    int gxu_want_gc = 0;
    DevBuf<int32_t> d_einfo;              // k_assemble's per-entry (local first row, 64-point blocks) word
    DevBuf<int32_t> d_pe, d_ebase;        // k_assemble's per-point / per-entry shortcuts (k_scatter_x)
    DevBuf<int32_t> d_big_list, d_small_list;
    DevBuf<SlotRec> d_srec, d_big_rec, d_small_rec;
    int grid_big = 0, grid_small = 0;     // launch sizes of the Cholesky's two lists (list lengths at the last sync + slack)
    int n_wide = 0;                       // local units of more than 20 tiles per edge at the last sync (potrf_generates_K)
    int n_la_big = 0, n_la_small = 0;     // ... of more than / at most BIG_LA_T tiles (the blocked path's / the one-workgroup kernels')
    View<double> d_weight, d_jitter;
    DevBuf<char> d_tab;
    PinBuf<char> h_tab;
    DevBuf<int32_t> d_m, d_rowoff, d_offj, d_upt, d_assign, d_posb, d_rank, d_cnt;
    DevBuf<int64_t> d_matoff;
    DevBuf<int32_t> d_res;                // [ctl (CTL_WORDS) | info (n_local) | bsize (n_blocks)] : one download per evaluation
    PinBuf<int32_t> h_res, h_up;          // ... and the staging buffer of an uploaded partition
    size_t res_words = 0;
    // device re-blocking (gprf_set_centers / gprf_set_split_tree)
    DevBuf<double> d_cs, d_c2;            // centres as structure of arrays [dx][nc] and their squared norms
    int n_centers = 0;
    DevBuf<double> d_tvec, d_tcenter, d_tsplit;
    DevBuf<int32_t> d_tleft, d_tright, d_tleaf;
    int tree_nodes = 0, tree_dim = 0, tree_wrap = 0;
    int last_stop_after = 6;              // stage the last gprf_debug_run stopped after
    bool debug_mode = false;              // inside gprf_debug_run: also store the per-unit-row gradient slab (k_gx_finalize)
    DevBuf<double> d_Vb;                  // big units: inverses of the 64 x 64 diagonal blocks (k_big_diag)
    DevBuf<double> d_K, d_U, d_W, d_V, d_Xu, d_Z, d_At, d_gXu, d_logdet, d_zzpart, d_usum, d_gcpart, d_rowpart, d_colpart, d_dbg;
    PinBuf<double> h_X, h_out;
    PinBuf<int32_t> h_done;               // [0]: sequence number of the last finished host-io evaluation (k_done)
    int32_t done_seq = 0;
    bool poll_pending = false;            // the pending evaluation ends with k_done: finish_eval may poll
    bool spin = true;                     // GPRF_SYNC=block: hipStreamSynchronize instead of polling
    // how a host-in / host-out evaluation moves its data (GPRF_IO_MODE, A/B diagnostics):
    //   0 zero-copy: the kernels read X from, and write the result to, pinned host memory; completion = a polled word
    //   1 copies:    X by hipMemcpyAsync H2D, result in HBM + hipMemcpyAsync D2H, hipStreamSynchronize
    //   2 mixed:     X zero-copy in, result in HBM + hipMemcpyAsync D2H, completion = a polled word behind the copy
    int io_mode = 0;
    bool poisoned = false;                // an evaluation timed out: the queues may never drain — never wait for them again

    // the optimiser-facing form (gprf_objective): location prior N(X_obs, obs_std^2) and the log-space hyper-parameters
    bool xprior = false;
    double obs_std = 0.0;
    DevBuf<double> d_Xobs, d_xpart;
    int hyper_mode = 0;                   // GPRF_HYPER_NONE / TIED / FULL
    double cov_scale = 1.0, hp_mean = 0.0, hp_std = 1.0, fixed_nv = 0.0, fixed_sv = 1.0;
    bool objective_call = false;          // the evaluation being enqueued leaves in the optimiser's form

    // single-process multi-device group (gprf_create_multi): this context is the FRONT of kids.size() member contexts, one
    // per (logical) device, member k evaluating shard (k, N) of the units; the members' assembly kernels write their
    // partial vectors into slots in the front device's memory, k_sum_parts adds them into the front's pinned host vector
    std::vector<gprf_ctx *> kids;
    std::vector<double> h_Xobs_front;     // the front's copy of the prior means (gprf_objective's parts_out)
    // the members' slots: FINE-GRAINED memory of the front device (peer stores over xGMI, coherent at system scope: a
    // member's stores are visible to the front device's summing kernel once that member's stream event has completed) —
    // or, when some member's device cannot reach the front device's memory, pinned host memory (every member stores
    // over its own host link; the summing kernel reads it zero-copy): slots_on_host
    double *slots = nullptr;              // as the devices see it
    void *slots_base = nullptr;           // what to free
    bool slots_on_host = false;
    std::string slots_why;                // why the host-staged form was chosen ("" = peer stores)
    size_t slot_stride = 0;
    hipStream_t red_stream = nullptr;
    std::vector<hipEvent_t> ev_kid;

    // host-side phases of the host-in / host-out evaluation (GPRF_HOST_TRACE=1: printed when the context is destroyed)
    double host_us[4] = {0, 0, 0, 0};     // copy X in | enqueue | wait | copy result out
    uint64_t host_n = 0;

    // timing: a ring of event sets so that evaluations can be timed back to back without a host sync;
    // a slot's elapsed times are folded into the running totals when the slot is about to be reused
    static constexpr int RING = 32;
    bool timing = false;
    hipEvent_t ev[RING][GPRF_N_STAGES + 1] = {};
    bool ev_valid = false;
    bool slot_pending[RING] = {};
    uint64_t n_timed = 0;          // evaluations recorded
    uint64_t n_folded = 0;         // evaluations folded into stage_ms_sum
    double stage_ms_sum[GPRF_N_STAGES] = {};
    double stage_ms_last[GPRF_N_STAGES] = {};
    bool eval_pending = false;
    bool pending_reblocked = false;       // the pending evaluation ran the partition kernel
    int epoch = 0, pending_epoch = 0;     // evaluation counter the partition kernels stamp ctl[CTL_CHANGED] with
    bool last_reblocked = false;          // ... and the last finished one changed the partition
    hipStream_t last_stream = nullptr;    // stream of the last enqueue (gprf_eval_status synchronises it)
    hipEvent_t ev_tables = nullptr;       // recorded on the context stream after a table upload
    hipEvent_t ev_last = nullptr;         // recorded behind the last evaluation on whatever stream it went to
};

namespace {

// why the last gprf_create / gprf_create_multi of this thread failed (there is no context to carry the text then):
// gprf_last_error(NULL) returns it
thread_local std::string g_create_err;

int fail(gprf_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}
int create_fail(int code, const std::string &msg) {
    g_create_err = msg;
    return code;
}

#define HIP_TRY(c, expr)                                                                           \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail((c), GPRF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));    \
    } while (0)

inline int pad16(int m) { return (m + 15) & ~15; }

// the largest unit accepted: GPRF_MAX_UNIT, or less through GPRF_DIAG max_unit=<points> (tests of the refusal path: a unit of
// 16385 points costs 4e12 flop to get to)
int max_unit_limit() {
    int v = diag("max_unit", 0);
    return (v > 0 && v < GPRF_MAX_UNIT) ? v : GPRF_MAX_UNIT;
}

int32_t *res_ctl(gprf_ctx *c) { return c->d_res.p; }
int32_t *res_info(gprf_ctx *c) { return c->d_res.p + CTL_WORDS; }
int32_t *res_bsize(gprf_ctx *c) { return c->d_res.p + CTL_WORDS + std::max(c->n_local, 0); }

UnitTab make_tab(gprf_ctx *c) {
    UnitTab t;
    t.m = c->d_m.p;
    t.row_off = c->d_rowoff.p;
    t.mat_off = c->d_matoff.p;
    t.weight = c->d_weight.p;
    t.jitter = c->d_jitter.p;
    t.upt = c->d_upt.p;
    t.n_units = c->n_local;
    t.max_T = c->max_T;
    t.ids = c->d_ids.p;
    t.n_ids = c->n_local;
    t.n_launch = c->n_local;
    t.big_list = c->d_big_list.p; t.small_list = c->d_small_list.p; t.ctl = c->d_res.p;
    t.srec = c->d_srec.p; t.big_rec = c->d_big_rec.p; t.small_rec = c->d_small_rec.p;
    t.grid_big = c->grid_big; t.grid_small = c->grid_small;
    t.fork_flag = nullptr; t.fork_seq = 0;
    t.pm_group = 0;
    t.n_wide = c->n_wide;
    return t;
}

BuildTab make_build(gprf_ctx *c) {
    BuildTab b;
    b.assign = c->d_assign.p; b.posb = c->d_posb.p; b.rank = c->d_rank.p; b.cnt = c->d_cnt.p;
    b.bsize = res_bsize(c);
    b.unit_bi = c->d_unit_bi.p; b.unit_bj = c->d_unit_bj.p; b.bu_ptr = c->d_bu_ptr.p; b.bu_ent = c->d_bu_ent.p;
    b.ids = c->d_ids.p; b.big_list = c->d_big_list.p; b.small_list = c->d_small_list.p;
    b.srec = c->d_srec.p; b.big_rec = c->d_big_rec.p; b.small_rec = c->d_small_rec.p;
    b.pe = c->d_pe.p; b.ebase = c->d_ebase.p; b.einfo = c->d_einfo.p;
    b.small_maxT = (c->dist_id == GPRF_DIST_EUCLIDEAN && c->kern_id == GPRF_KERN_SE) ? potrf_small_maxT() : 0;
    b.grid_big = c->grid_big; b.grid_small = c->grid_small;
    b.m = c->d_m.p; b.row_off = c->d_rowoff.p; b.mat_off = c->d_matoff.p; b.off_j = c->d_offj.p; b.upt = c->d_upt.p;
    b.Xu = c->d_Xu.p; b.xstride = c->dist_id == GPRF_DIST_LLD ? 8 : XPAD;
    b.ctl = res_ctl(c);
    b.n = c->n; b.n_blocks = c->n_blocks; b.n_local = c->n_local; b.n_chunks = c->n_chunks;
    b.n_ent = c->n_ent;
    b.cap_rows = c->cap_rows; b.cap_mat = c->cap_mat; b.maxT_bound = c->max_T;
    return b;
}

Pools make_pools(gprf_ctx *c) {
    Pools p;
    p.Vb = c->d_Vb.p;
    p.K = c->d_K.p; p.U = c->d_U.p; p.W = c->d_W.p; p.V = c->d_V.p; p.Xu = c->d_Xu.p; p.Y = c->d_Y.p; p.Z = c->d_Z.p;
    p.At = c->d_At.p; p.gXu = c->d_gXu.p; p.logdet = c->d_logdet.p; p.zzpart = c->d_zzpart.p; p.usum = c->d_usum.p;
    p.gcpart = c->d_gcpart.p; p.info = res_info(c); p.rowpart = c->d_rowpart.p; p.colpart = c->d_colpart.p; p.dbg = c->d_dbg.p;
    return p;
}

KParams make_kparams(gprf_ctx *c) {
    KParams k;
    k.nv = c->theta[0];
    k.sv = c->theta[1];
    for (int i = 0; i < 3; ++i) k.ls[i] = (i < c->ndfn) ? c->theta[2 + i] : 1.0;
    for (int i = 0; i < 3; ++i) k.inv_ls[i] = 1.0 / k.ls[i];
    k.dx = c->dx;
    k.ndfn = c->ndfn;
    k.dy = c->dy;
    return k;
}

// sizes / offsets of the local units as the device derives them (k_build), recomputed from the block sizes
void refresh_host_units(gprf_ctx *c) {
    const int nl = c->n_local;
    c->l_m.resize(nl); c->l_rowoff.resize(nl); c->l_matoff.resize(nl);
    int64_t rows = 0, mat = 0;
    double flops = 0, fbytes = 0;
    for (int l = 0; l < nl; ++l) {
        int m = c->h_bsize[c->l_bi[l]] + (c->l_bj[l] >= 0 ? c->h_bsize[c->l_bj[l]] : 0);
        int mp = pad16(m);
        c->l_m[l] = m; c->l_rowoff[l] = (int32_t)rows; c->l_matoff[l] = mat;
        rows += mp; mat += (int64_t)mp * mp;
        flops += (double)m * m * m + 4.0 * m * m * c->dy;
        fbytes += 8.0 * m * m;
    }
    c->cur_rows = rows; c->cur_mat = mat;
    c->work_flops = flops; c->work_fill_bytes = fbytes;
    // launch sizes of the Cholesky's two lists: the present lengths + slack, followed down only when far above
    int nbig = 0, nwide = 0;
    for (int l = 0; l < nl; ++l) {
        nbig += pad16(c->l_m[l]) / 16 > potrf_small_maxT() ? 1 : 0;
        nwide += pad16(c->l_m[l]) / 16 > 20 ? 1 : 0;      // units the generating Cholesky kernels do not take
    }
    c->n_wide = nwide;
    c->n_la_big = c->n_la_small = 0;
    for (int l = 0; l < nl; ++l) (pad16(c->l_m[l]) / 16 > BIG_LA_T ? c->n_la_big : c->n_la_small)++;
    // (surplus workgroups of the large-unit launch take small units, see potrf_reg_body: slack costs nothing there; the
    // small-unit launch simply covers every unit)
    // (round 6: slack 32 -> 16.  The surplus workgroups take the SMALLEST units of the small list, each alone on a CU of its
    // own: measured on the north star with the by-class pipelines, evaluations/s at slack 8 / 32 / 64 / 100: 2846, 2837 / 2829,
    // 2806 / 2784, 2783 / 2766; 16 keeps room for the list to grow between two synchronised partitions)
    if (nbig + 4 > c->grid_big || nbig + 48 < c->grid_big) c->grid_big = std::min(nl, nbig + 16);
    c->grid_small = nl;
}

// workspace for `rows` padded rows, `mat` matrix elements and units of up to maxT tiles per edge
int reserve_workspace(gprf_ctx *c, int64_t rows, int64_t mat, int maxT) {
    size_t nl1 = (size_t)std::max(c->n_local, 1);
    size_t tbm = (size_t)std::max((maxT + 3) / 4, 1);      // 64-point blocks per edge of the largest local unit
    HIP_TRY(c, c->d_logdet.reserve(nl1));
    HIP_TRY(c, c->d_zzpart.reserve(nl1 * 4));
    HIP_TRY(c, c->d_usum.reserve(nl1 * 8));
#ifdef GPRF_WGTRACE
    HIP_TRY(c, c->d_dbg.reserve(nl1 * 8 + 4 * GPRF_WGTRACE_MAX));     // + one (start, end, hw id, block) record per workgroup
#else
    HIP_TRY(c, c->d_dbg.reserve(nl1 * 8));
#endif
    HIP_TRY(c, c->d_gcpart.reserve(nl1 * (tbm * (tbm + 1) / 2) * GC_SLOTS));
    HIP_TRY(c, c->d_rowpart.reserve((size_t)rows * tbm * XPAD + 1, 1.0));
    HIP_TRY(c, c->d_colpart.reserve((size_t)rows * tbm * XPAD + 1, 1.0));
    HIP_TRY(c, c->d_K.reserve((size_t)mat + GPRF_POOL_SLACK, 1.0));
    HIP_TRY(c, c->d_U.reserve((size_t)mat + GPRF_POOL_SLACK, 1.0));
    HIP_TRY(c, c->d_W.reserve((size_t)mat + GPRF_POOL_SLACK, 1.0));
    HIP_TRY(c, c->d_V.reserve((size_t)rows * 16 + 1, 1.0));
    if (maxT > BIG_LA_T) HIP_TRY(c, c->d_Vb.reserve(((size_t)rows + 64 * nl1 + 64) * 64, 1.0));
    HIP_TRY(c, c->d_Xu.reserve((size_t)rows * 8 + 1, 1.0));      // XPAD, or 8 for the lld record
    HIP_TRY(c, c->d_Z.reserve((size_t)rows * YPAD + 1, 1.0));
    HIP_TRY(c, c->d_At.reserve((size_t)rows * YPAD + 1, 1.0));
    HIP_TRY(c, c->d_gXu.reserve((size_t)rows * XPAD + 1, 1.0));
    HIP_TRY(c, c->d_upt.reserve((size_t)rows + 1, 1.0));
    return GPRF_OK;
}

// Workspace capacities for the partitions to come.  Padded rows have a partition-independent bound (a point
// appears once in every unit that contains its block); the matrix pools get headroom over the present partition and
// the launch-wide tile bound is the present largest unit: k_build reports a partition that exceeds any of them,
// the host grows them and repeats that one evaluation (run_checked).
int size_workspace(gprf_ctx *c, int64_t min_rows, int64_t min_mat, int min_maxT) {
    std::vector<int> per_block((size_t)std::max(c->n_blocks, 1), 0);
    for (int l = 0; l < c->n_local; ++l) {
        per_block[c->l_bi[l]]++;
        if (c->l_bj[l] >= 0) per_block[c->l_bj[l]]++;
    }
    int mx = 0;
    for (int v : per_block) mx = std::max(mx, v);
    int64_t rows_bound = (int64_t)c->n * mx + 15ll * c->n_local;
    // (capped at twice the present need so that a degenerate neighbour list does not reserve n * n_blocks rows)
    int64_t rows = std::max<int64_t>(std::max(min_rows, c->cur_rows),
                                     std::min<int64_t>(rows_bound, 2 * std::max<int64_t>(c->cur_rows, 1) + 4096));
    int64_t mat = std::max<int64_t>(min_mat, c->cur_mat + c->cur_mat / 4 + 65536);
    int maxT = min_maxT;
    for (int l = 0; l < c->n_local; ++l) maxT = std::max(maxT, pad16(c->l_m[l]) / 16);
    if (rows > c->cap_rows || mat > c->cap_mat || maxT != c->max_T) {
        if (c->ev_last) HIP_TRY(c, hipEventSynchronize(c->ev_last));      // an evaluation may still use the old pools
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        int rc = reserve_workspace(c, std::max(rows, c->cap_rows), std::max(mat, c->cap_mat), maxT);
        if (rc != GPRF_OK) return rc;
        c->cap_rows = std::max(rows, c->cap_rows);
        c->cap_mat = std::max(mat, c->cap_mat);
        c->max_T = maxT;
    }
    return GPRF_OK;
}

// (Re)build the static tables after neighbours / shard / block count (or a host partition's sizes) changed.
// Units: blocks 0..n_blocks-1 (gprf.py:236), then pairs in the caller's order (gprf.py:239).
int rebuild_static(gprf_ctx *c) {
    const int nb = c->n_blocks, np = c->n_pairs;
    const int nu = nb + np;
    if ((int)c->h_bsize.size() != nb) return fail(c, GPRF_ERR_STATE, "block sizes unknown");
    std::vector<int32_t> um(nu);
    std::vector<int> deg(nb, 0);
    for (int b = 0; b < nb; ++b) um[b] = c->h_bsize[b];
    for (int q = 0; q < np; ++q) {
        int i = c->pairs[2 * q], j = c->pairs[2 * q + 1];
        if (i < 0 || i >= nb || j < 0 || j >= nb || i == j)
            return fail(c, GPRF_ERR_ARG, "neighbor pair refers to a block out of range");
        um[nb + q] = um[i] + um[j];
        deg[i]++;
        deg[j]++;
    }
    for (int u = 0; u < nu; ++u)
        if (um[u] > max_unit_limit()) {
            char buf[200];
            snprintf(buf, sizeof buf, "unit %d has %d points; the kernels accept at most %d per unit (GPRF_MAX_UNIT)", u,
                     um[u], max_unit_limit());
            return fail(c, GPRF_ERR_ARG, buf);
        }
    // shard: longest-processing-time-first over cost m^3 + 4 m^2 dy, on the sizes of THIS partition;
    // the ownership then stays until the next static rebuild (later re-blockings on the device keep it)
    // (a rebuild that only refreshes the launch lists — a unit changed size class — keeps the ownership: ranks do not
    // rebuild in lockstep then)
    std::vector<int> &owner = c->owner;
    if (c->owner_dirty || (int)owner.size() != nu) {
        owner.assign((size_t)nu, 0);
        if (c->world > 1 && nu > 0) gprf_partition_units(nu, um.data(), c->dy, c->world, owner.data());
        c->owner_dirty = false;
    }
    c->l_global.clear(); c->l_bi.clear(); c->l_bj.clear();
    std::vector<double> weight, jitter;
    for (int u = 0; u < nu; ++u) {
        if (owner[u] != c->rank) continue;
        c->l_global.push_back(u);
        c->l_bi.push_back(u < nb ? u : c->pairs[2 * (u - nb)]);
        c->l_bj.push_back(u < nb ? -1 : c->pairs[2 * (u - nb) + 1]);
        weight.push_back(u < nb ? (double)(1 - deg[u]) : 1.0);
        jitter.push_back((size_t)u < c->unit_jitter.size() ? c->unit_jitter[u] : 0.0);
    }
    const int nl = (int)c->l_global.size();
    c->n_local = nl;
    refresh_host_units(c);
    // block -> local units, ascending unit id; entry = 2 * unit + side
    std::vector<int32_t> bu_ptr((size_t)nb + 1, 0), bu_ent;
    for (int l = 0; l < nl; ++l) {
        bu_ptr[c->l_bi[l] + 1]++;
        if (c->l_bj[l] >= 0) bu_ptr[c->l_bj[l] + 1]++;
    }
    for (int b = 0; b < nb; ++b) bu_ptr[b + 1] += bu_ptr[b];
    bu_ent.resize((size_t)bu_ptr[nb]);
    {
        std::vector<int32_t> cur(bu_ptr.begin(), bu_ptr.end() - 1);
        for (int l = 0; l < nl; ++l) {
            bu_ent[cur[c->l_bi[l]]++] = 2 * l;
            if (c->l_bj[l] >= 0) bu_ent[cur[c->l_bj[l]]++] = 2 * l + 1;
        }
    }
    std::vector<double> ewgt(bu_ent.size());
    for (size_t e = 0; e < bu_ent.size(); ++e) ewgt[e] = weight[(size_t)(bu_ent[e] >> 1)];
    // launch order: largest units first, so that the long factorisations start first
    std::vector<int32_t> ids(nl);
    std::iota(ids.begin(), ids.end(), 0);
    std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) { return c->l_m[a] > c->l_m[b]; });

    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    // an evaluation enqueued on a caller's stream may still read the old tables
    if (c->ev_last) HIP_TRY(c, hipEventSynchronize(c->ev_last));
    HIP_TRY(c, hipStreamSynchronize(s));
    size_t nl1 = (size_t)std::max(nl, 1);
    c->n_chunks = (c->n + CHUNK - 1) / CHUNK;
    HIP_TRY(c, c->d_m.reserve(nl1));
    HIP_TRY(c, c->d_pe.reserve(2 * (size_t)c->n + 2, 1.0));
    HIP_TRY(c, c->d_ebase.reserve(bu_ent.size() + 1, 1.0));
    HIP_TRY(c, c->d_einfo.reserve(bu_ent.size() + 1, 1.0));
    c->n_ent = (int)bu_ent.size();
    HIP_TRY(c, c->d_rowoff.reserve(nl1));
    HIP_TRY(c, c->d_offj.reserve(nl1));
    HIP_TRY(c, c->d_matoff.reserve(nl1));
    HIP_TRY(c, c->d_big_list.reserve(nl1));
    HIP_TRY(c, c->d_small_list.reserve(nl1));
    HIP_TRY(c, c->d_srec.reserve(nl1));
    HIP_TRY(c, c->d_big_rec.reserve(nl1));
    HIP_TRY(c, c->d_small_rec.reserve(nl1));
    HIP_TRY(c, c->d_assign.reserve((size_t)c->n + 1, 1.0));
    HIP_TRY(c, c->d_posb.reserve((size_t)c->n + 1, 1.0));
    HIP_TRY(c, c->d_rank.reserve((size_t)c->n + 1, 1.0));
    HIP_TRY(c, c->d_cnt.reserve((size_t)std::max(c->n_chunks, 1) * std::max(nb, 1) + 1, 1.0));
    // the result words [ctl | info | bsize] move when n_local / n_blocks change
    size_t words = (size_t)CTL_WORDS + nl1 + (size_t)std::max(nb, 1);
    if (words > c->d_res.cap) {
        int32_t builds = 0;
        if (c->d_res.p) HIP_TRY(c, hipMemcpy(&builds, res_ctl(c) + CTL_BUILDS, sizeof(int32_t), hipMemcpyDeviceToHost));
        c->d_res.release();
        HIP_TRY(c, c->d_res.reserve(words));
        HIP_TRY(c, hipMemset(c->d_res.p, 0, c->d_res.cap * sizeof(int32_t)));
        HIP_TRY(c, hipMemcpy(res_ctl(c) + CTL_BUILDS, &builds, sizeof(int32_t), hipMemcpyHostToDevice));
    }
    HIP_TRY(c, c->h_res.reserve(words));
    c->res_words = (size_t)CTL_WORDS + nl + nb;
    HIP_TRY(c, hipMemset(c->d_res.p, 0, CTL_BUILDS * sizeof(int32_t)));
    // block sizes as the host knows them (exact: the device build re-derives them whenever the partition changes)
    if (nb > 0) HIP_TRY(c, hipMemcpy(res_bsize(c), c->h_bsize.data(), (size_t)nb * sizeof(int32_t), hipMemcpyHostToDevice));
    {
        // ONE staged upload: every table is packed (256-byte aligned) into one pinned buffer and copied with a single
        // asynchronous H2D on the context stream
        struct Seg { const void *src; size_t bytes; void **dst; };
        Seg segs[] = {
            {ids.data(), (size_t)nl * sizeof(int32_t), (void **)&c->d_ids.p},
            {c->l_bi.data(), (size_t)nl * sizeof(int32_t), (void **)&c->d_unit_bi.p},
            {c->l_bj.data(), (size_t)nl * sizeof(int32_t), (void **)&c->d_unit_bj.p},
            {weight.data(), (size_t)nl * sizeof(double), (void **)&c->d_weight.p},
            {jitter.data(), (size_t)nl * sizeof(double), (void **)&c->d_jitter.p},
            {bu_ptr.data(), bu_ptr.size() * sizeof(int32_t), (void **)&c->d_bu_ptr.p},
            {bu_ent.data(), bu_ent.size() * sizeof(int32_t), (void **)&c->d_bu_ent.p},
            {ewgt.data(), ewgt.size() * sizeof(double), (void **)&c->d_ewgt.p},
        };
        size_t total = 256;
        for (auto &sg : segs) total += (sg.bytes + 255) & ~(size_t)255;
        HIP_TRY(c, c->d_tab.reserve(total));
        HIP_TRY(c, c->h_tab.reserve(total));
        size_t off = 0;
        for (auto &sg : segs) {
            if (sg.bytes) memcpy(c->h_tab.p + off, sg.src, sg.bytes);
            *sg.dst = (void *)(c->d_tab.p + off);
            off += (sg.bytes + 255) & ~(size_t)255;
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_tab.p, c->h_tab.p, std::max<size_t>(off, 256), hipMemcpyHostToDevice, s));
    }
    int rc = size_workspace(c, 0, 0, 0);
    if (rc != GPRF_OK) return rc;
    if (!c->ev_tables) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_tables, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->ev_tables, s));
    c->static_dirty = false;
    c->jitter_dirty = false;
    c->need_build = true;
    return GPRF_OK;
}

// the jitter vector alone (jitchol's retries): one small upload, no rebuild
int upload_jitter(gprf_ctx *c) {
    std::vector<double> jitter((size_t)std::max(c->n_local, 1), 0.0);
    for (int l = 0; l < c->n_local; ++l) {
        size_t u = (size_t)c->l_global[l];
        jitter[l] = u < c->unit_jitter.size() ? c->unit_jitter[u] : 0.0;
    }
    if (c->ev_last) HIP_TRY(c, hipEventSynchronize(c->ev_last));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->n_local > 0)
        HIP_TRY(c, hipMemcpy(c->d_jitter.p, jitter.data(), (size_t)c->n_local * sizeof(double), hipMemcpyHostToDevice));
    c->jitter_dirty = false;
    return GPRF_OK;
}

// a partition given by the host (gprf_set_blocks): block of every point, position inside the block -> device
int upload_host_partition(gprf_ctx *c, hipStream_t s) {
    size_t n = (size_t)c->n;
    HIP_TRY(c, c->h_up.reserve(2 * n + 1));
    if (c->ev_last) HIP_TRY(c, hipEventSynchronize(c->ev_last));
    HIP_TRY(c, hipStreamSynchronize(c->stream));           // the staging buffer of the previous upload
    if (n) {
        memcpy(c->h_up.p, c->h_assign_host.data(), n * sizeof(int32_t));
        memcpy(c->h_up.p + n, c->h_posb_host.data(), n * sizeof(int32_t));
        HIP_TRY(c, hipMemcpyAsync(c->d_assign.p, c->h_up.p, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(c->d_posb.p, c->h_up.p + n, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    }
    if (c->n_blocks > 0)
        HIP_TRY(c, hipMemcpyAsync(res_bsize(c), c->h_bsize.data(), (size_t)c->n_blocks * sizeof(int32_t), hipMemcpyHostToDevice, s));
    c->host_blocks_dirty = false;
    c->assign_valid = true;
    c->need_build = true;
    return GPRF_OK;
}

int check_ready(gprf_ctx *c) {
    if (!c) return GPRF_ERR_ARG;
    if (!c->kids.empty()) {
        int rc = check_ready(c->kids[0]);
        if (rc != GPRF_OK) c->err = c->kids[0]->err;
        return rc;
    }
    if (!c->have_Y) return fail(c, GPRF_ERR_STATE, "gprf_set_Y has not been called");
    if (!c->have_theta) return fail(c, GPRF_ERR_STATE, "gprf_set_theta has not been called");
    if (!c->have_blocks) return fail(c, GPRF_ERR_STATE, "gprf_set_blocks has not been called");
    return GPRF_OK;
}

// fold one finished event slot into the running per-stage totals (waits for the slot's last event)
int fold_slot(gprf_ctx *c, int slot) {
    HIP_TRY(c, hipEventSynchronize(c->ev[slot][GPRF_N_STAGES]));
    for (int i = 0; i < GPRF_N_STAGES; ++i) {
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[slot][i], c->ev[slot][i + 1]));
        c->stage_ms_sum[i] += ms;
        c->stage_ms_last[i] = ms;
    }
    c->n_folded++;
    c->slot_pending[slot] = false;
    return GPRF_OK;
}

// everything host-side an enqueue needs: static tables, jitter, an uploaded partition
int prepare(gprf_ctx *c, hipStream_t s) {
    if (c->static_dirty) {
        int rc = rebuild_static(c);
        if (rc != GPRF_OK) return rc;
    }
    if (c->jitter_dirty) {
        int rc = upload_jitter(c);
        if (rc != GPRF_OK) return rc;
    }
    if (c->host_blocks_dirty) {
        int rc = upload_host_partition(c, c->stream);
        if (rc != GPRF_OK) return rc;
        if (!c->ev_tables) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_tables, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->ev_tables, c->stream));
    }
    if (s != c->stream && c->ev_tables) HIP_TRY(c, hipStreamWaitEvent(s, c->ev_tables, 0));
    return GPRF_OK;
}

// partition kernel for the points d_X on stream s (centres or tree), leaving ranks / per-chunk counts for the build
int enqueue_partition(gprf_ctx *c, const double *d_X, hipStream_t s) {
    if (c->n_centers < 1) return fail(c, GPRF_ERR_STATE, "gprf_set_centers or gprf_set_split_tree first");
    if (c->n_centers != c->n_blocks) return fail(c, GPRF_ERR_STATE, "the centres / tree leaves do not match the block count");
    BuildTab bt = make_build(c);
    if (!c->assign_valid) HIP_TRY(c, hipMemsetAsync(c->d_assign.p, 0xff, (size_t)c->n * sizeof(int32_t), s));
    c->epoch = c->epoch >= 0x3fffffff ? 1 : c->epoch + 1;
    // (the partition kernel leaves a copy of the points in d_X when they came from somewhere else — pinned host
    // memory in the host-in / host-out form: the kernels behind it read HBM)
    double *xcopy = d_X == c->d_X.p ? nullptr : c->d_X.p;
    if (c->tree_nodes > 0)
        launch_route(d_X, xcopy, c->dx, c->tree_dim, c->tree_wrap, c->d_tvec.p, c->d_tcenter.p, c->d_tsplit.p, c->d_tleft.p,
                     c->d_tright.p, c->d_tleaf.p, bt, c->epoch, s);
    else
        launch_assign(d_X, xcopy, c->dx, c->d_cs.p, c->d_c2.p, c->n_centers, c->grid_hint, bt, c->epoch, s);
    c->assign_valid = true;
    return GPRF_OK;
}

constexpr int PIPE_DEFAULT_PCT = 0;       // the first (main-queue) part of the split solve -> At -> gradient launches, in per cent (0: off)
constexpr int PIPE_MIN_UNITS = 128;       // ... which is not worth its fork and join below this many units

// enqueue one evaluation on stream s reading d_X, writing d_out; stop_after < 6 truncates (debug); reblock: first
// re-partition the points on the device (update_X's block_fn, gprf.py:171-172)
// host_io: 0 = d_X / d_out are the caller's device buffers; 1 = the zero-copy host-in / host-out form (d_X / d_out are the
// pinned host buffers as the device sees them, completion by k_done's polled word); 2 = host-in / host-out through HBM:
// d_out = the context's device vector, copied down (with the result words) behind the kernels — polled (io_mode 2) or
// stream-synchronised (io_mode 1) by finish_eval
int enqueue_eval(gprf_ctx *c, const double *d_X, int want_gx, int want_gc, double *d_out, hipStream_t s,
                 int stop_after, bool reblock, int host_io = 0) {
    int rc = prepare(c, s);
    if (rc != GPRF_OK) return rc;
    c->gxu_pending = false;      // (whatever the last evaluation left to be finalized on demand is overwritten from here on)
    bool tm = c->timing;
    if (tm && !c->ev_valid) {
        for (int r = 0; r < gprf_ctx::RING; ++r)
            for (int i = 0; i <= GPRF_N_STAGES; ++i) HIP_TRY(c, hipEventCreate(&c->ev[r][i]));
        c->ev_valid = true;
    }
    int slot = (int)(c->n_timed % gprf_ctx::RING);
    if (tm) {
        if (c->slot_pending[slot]) {
            rc = fold_slot(c, slot);
            if (rc != GPRF_OK) return rc;
        }
        c->slot_pending[slot] = true;
        c->n_timed++;
    }
    int stage = 0;
    auto mark = [&]() { if (tm) (void)hipEventRecord(c->ev[slot][stage], s); ++stage; };
    mark();      // stage "gather" = re-partition + table build (when asked for) + the coordinate gather
    int from_chunks = 0, force = c->need_build ? 1 : 0;
    bool fused_build = false;      // table build + coordinate scatter as one launch (k_build_scatter)
    if (reblock) {
        rc = enqueue_partition(c, d_X, s);
        if (rc != GPRF_OK) return rc;
        if (d_X != c->d_X.p) d_X = c->d_X.p;      // the partition kernel's copy
        from_chunks = 1;
        fused_build = build_scatter_fits(make_build(c));
        if (!fused_build) launch_build_tables(make_build(c), 1, force, c->epoch, s);
    } else if (c->need_build) {
        launch_build_tables(make_build(c), 0, 1, c->epoch, s);
    }
    c->need_build = false;
    c->pending_epoch = c->epoch;
    UnitTab ut = make_tab(c);
    Pools pl = make_pools(c);
    KParams kp = make_kparams(c);
    // host_io: d_X / d_out are the pinned host buffers themselves (read / written by the kernels over the fabric) and
    // the result words are mirrored into pinned memory by the assembly kernel: no copy command in the evaluation
    AssembleTab at{c->d_assign.p, c->d_posb.p, c->d_bu_ptr.p, c->d_bu_ent.p, c->d_offj.p, res_ctl(c),
                   c->d_pe.p, c->d_ebase.p, c->d_einfo.p, 0, c->d_ewgt.p,
                   c->d_res.p, host_io == 1 ? c->h_res.d : nullptr, (int)c->res_words};
    bool do_grad = stop_after >= 4 && (want_gx || want_gc);
    bool fold_gx = false;
    if (fused_build) launch_build_scatter(make_build(c), d_X, c->dx, c->dist_id, force, c->epoch, s);
    else launch_scatter_x(make_build(c), d_X, c->dx, c->dist_id, from_chunks, force, c->epoch, s);
    mark();
    // (the K pool exists only when somebody reads it: a fill-only debug run, the generic Cholesky, big units)
    bool gen = stop_after >= 1 && potrf_generates_K(c->dist_id, c->kern_id, ut);
    bool beside = false;      // the blocked path's Cholesky / substitution beside the small units' (below)
    int solved = 0;           // stages behind the Cholesky that launch_potrf ran itself (by size class, on its two queues): 1 solve, 2 + At, 3 + gradient
    launch_fill(c->dist_id, c->kern_id, ut, pl, kp, gen ? potrf_gen_maxT(c->dist_id) : 0, s);
    mark();
    if (stop_after >= 1) {
        SideQueue side;
        side.s2 = c->stream2; side.ev_fork = c->ev_fork; side.ev_join = c->ev_join;
        side.words = c->side_values ? c->d_side.p : nullptr;
        side.seq = ++c->side_seq;
        // A launch with units on BOTH sides of BIG_LA_T (the reference's 25- / 36-block partitions: unaries of 280-400 points,
        // pairs of 560-870): the blocked path's launches run BESIDE the one-workgroup-per-unit kernels, on the third queue —
        // independent units, and neither side fills the chip alone (the few units of 21-32 tiles live 300-400 us on their CUs
        // while the blocked path issues its ~40 us steps)
        beside = c->n_la_big > 0 && c->n_la_small > 0 && ut.max_T > BIG_LA_T && c->stream3 && c->side_values && s == c->stream &&
                 !potrf_tool_env() && !tm && diag("big_beside", 1) != 0;
        uint32_t *w = c->d_side.p;
        if (beside) {
            HIP_TRY(c, hipStreamWriteValue32(s, w + 6, side.seq, 0));      // (the fill is done)
            HIP_TRY(c, hipStreamWaitValue32(c->stream3, w + 6, side.seq, hipStreamWaitValueGte, 0xffffffffu));
            launch_big_potrf(ut, pl, kp, c->stream3);
            HIP_TRY(c, hipStreamWriteValue32(c->stream3, w + 7, side.seq, 0));
        }
        // (round 6: a two-queue launch runs each size class's substitution, At and gradient behind that class's Cholesky kernel on
        // its queue and joins the queues behind them — the plain launch-wide structure only: not beside the blocked path, not
        // under the split pipelines, and not under the per-stage timers, which time one launch-wide stage after the other)
        // (... and on the library's own stream: ten contexts enqueued back to back on ONE caller's stream — bench.py's
        // device-resident figure — share a few hardware queues between their side streams, and four more kernels on each
        // cost that form 12 %)
        const bool plain = !beside && !tm && (s == c->stream || c->caller_pipelines) && !(diag("pipe", PIPE_DEFAULT_PCT) > 0 && diag("pipe", PIPE_DEFAULT_PCT) < 100);
        const int want_stages = !plain ? 0 : (do_grad ? 3 : (stop_after >= 3 ? 2 : (stop_after >= 2 ? 1 : 0)));
        // (the zero-copy, polled host path on the library's own stream may continue on whichever queue carries the longer
        // pipeline: the completion word is polled, nothing is waited for on a stream, and the next evaluation is not enqueued
        // before this one has been seen to finish)
        const bool may_move = host_io == 1 && stop_after >= 5 && c->spin && c->h_done.p && s == c->stream;
        hipStream_t tail = s;
        solved = launch_potrf(ut, pl, kp, gen, s, side, want_stages, want_gc, may_move ? &tail : nullptr);
        s = tail;
        if (beside) HIP_TRY(c, hipStreamWaitValue32(s, w + 7, side.seq, hipStreamWaitValueGte, 0xffffffffu));
        else launch_big_potrf(ut, pl, kp, s);
    }
    mark();
    // ---- round 5, measured and left OFF: the three GEMM-shaped stages as TWO pipelines side by side (GPRF_DIAG pipe=<percent>) ----
    // solve -> At -> gradient are launch-wide stages while the dependency is per unit: each of them ends in a tail of 10-25 us
    // in which a few long workgroups run on a mostly idle chip, and per-unit flags at agent scope were
    // measured unaffordable in round 2.  Here the launch order (largest units first) is cut in two: the first part runs its
    // three kernels on the main queue, the rest runs the SAME three kernels on a third, low-priority queue, fork and join by
    // stream memory operations (each wait submitted behind its writer).  The same kernel instantiations (UnitTab::n_launch
    // picks the forms), the same arithmetic per unit: bit-identical to the launch-wide form (tests/test_gpu_variants.py).
    // Measured on the north-star configuration (profiles/r05_pipeline_ab.txt, ms per step, medians of 7 x 200 steps):
    // launch-wide 0.3782; cut at 25 % 0.3799, 35 % 0.3780, 50 % 0.3785, 65 % 0.3755, 80 % 0.3816 — nothing.  Why: the three
    // kernels have different register footprints (157 / 256 / 128 VGPRs: three, two, four workgroups per CU), a slot freed by
    // one kernel's workgroup does not fit the next kernel's, so the dispatcher keeps filling freed slots with the OTHER
    // queue's workgroups of the same kernel and the first part's next stage starts no sooner than before; and two half-size
    // launches lose the launch-wide longest-workgroup-first order that shortened these tails in round 3.
    const int pipe_pct = diag("pipe", PIPE_DEFAULT_PCT);
    const bool pipe = pipe_pct > 0 && pipe_pct < 100 && !tm && stop_after >= 5 && do_grad && c->stream3 && c->side_values &&
                      s == c->stream && !potrf_tool_env() && ut.max_T <= SMALL_MAX_T && ut.n_ids >= PIPE_MIN_UNITS;
    if (pipe) {
        int nA = ((ut.n_ids * pipe_pct / 100) + 7) & ~7;
        if (nA >= ut.n_ids) nA = ut.n_ids - 1;
        UnitTab ua = ut, ub = ut;
        ua.n_ids = nA;
        ub.srec = ut.srec + nA; ub.ids = ut.ids + nA; ub.n_ids = ut.n_ids - nA;
        hipStream_t s3 = c->stream3;
        uint32_t *w = c->d_side.p;
        const uint32_t seq = c->side_seq;
        HIP_TRY(c, hipStreamWriteValue32(s, w + 3, seq, 0));      // (everything in front — the Cholesky's join included — is done)
        HIP_TRY(c, hipStreamWaitValue32(s3, w + 3, seq, hipStreamWaitValueGte, 0xffffffffu));
        launch_solve(ua, pl, kp, s);
        launch_solve(ub, pl, kp, s3);
        launch_at(ua, pl, s);
        launch_at(ub, pl, s3);
        launch_grad(c->dist_id, c->kern_id, ua, pl, kp, want_gc, !gen, s);
        launch_grad(c->dist_id, c->kern_id, ub, pl, kp, want_gc, !gen, s3);
        HIP_TRY(c, hipStreamWriteValue32(s3, w + 4, seq, 0));
        HIP_TRY(c, hipStreamWaitValue32(s, w + 4, seq, hipStreamWaitValueGte, 0xffffffffu));
        mark(); mark();
    } else {
    if (stop_after >= 2) {
        if (beside) {
            uint32_t *w = c->d_side.p;
            const uint32_t seq = c->side_seq;
            HIP_TRY(c, hipStreamWriteValue32(s, w + 8, seq, 0));      // (both Cholesky sides are done: the join above)
            HIP_TRY(c, hipStreamWaitValue32(c->stream3, w + 8, seq, hipStreamWaitValueGte, 0xffffffffu));
            launch_big_solve(ut, pl, c->stream3);
            HIP_TRY(c, hipStreamWriteValue32(c->stream3, w + 9, seq, 0));
            launch_solve(ut, pl, kp, s);
            HIP_TRY(c, hipStreamWaitValue32(s, w + 9, seq, hipStreamWaitValueGte, 0xffffffffu));
        } else {
            if (solved < 1) launch_solve(ut, pl, kp, s);
            launch_big_solve(ut, pl, s);
        }
    }
    mark();
    if (stop_after >= 3 && solved < 2) launch_at(ut, pl, s);
    mark();
    if (do_grad && solved < 3) launch_grad(c->dist_id, c->kern_id, ut, pl, kp, want_gc, !gen, s);      // (re-evaluates k whenever K was generated for some units)
    }
    if (do_grad) {
        // k_gx_finalize is a launch of its own for sums the assembly can do on the way (10 us of a 430 us evaluation) —
        // up to GX_FOLD_MAX_UNITS units; beyond that the assembly's single summing workgroup would walk every unit's
        // partials itself (C4: 106 us against 20).  The per-unit gradient (gprf_debug_fetch) is then made on demand.
        fold_gx = diag("gx_fold", 1) != 0 && stop_after >= 5 && c->n_local <= GX_FOLD_MAX_UNITS;
        if (!fold_gx) launch_gx_finalize(ut, pl, kp, want_gc, s);
        c->gxu_pending = fold_gx;
        c->gxu_want_gc = want_gc;
    }
    at.fold_gx = fold_gx ? 1 : 0;
    mark();
    // gprf_objective: the result leaves in the optimiser's form; the location prior is added by ONE context of a
    // sharded job (rank 0), so that the all-reduce of the partial vectors counts it once
    ObjTab ob{0, nullptr, nullptr, 0.0, 0.0, nullptr};
    const bool objective = c->objective_call && stop_after >= 5;
    if (objective) {
        ob.on = 1;
        ob.X = d_X;
        ob.Xobs = (c->xprior && c->rank == 0) ? c->d_Xobs.p : nullptr;
        ob.sigma = c->obs_std;
        ob.var = c->obs_std * c->obs_std;
        ob.part = c->d_xpart.p;
    }
    // control words, unit status, block sizes -> pinned host: one download (or the assembly kernel's mirror)
    c->poll_pending = false;
    const bool poll = host_io && stop_after >= 5 && c->spin && c->h_done.p && !(host_io == 2 && c->io_mode == 1);
    if (poll) c->done_seq = c->done_seq >= 0x3fffffff ? 1 : c->done_seq + 1;
    if (stop_after >= 5) launch_assemble(ut, pl, at, kp, c->n, want_gx, want_gc, d_out, (do_grad && !fold_gx) ? 1 : 0, ob, s);
    mark();
    if (objective) {
        const double nel = (double)c->n * c->dx;
        const double xp_const = -0.5 * nel * std::log(2.0 * M_PI * (c->obs_std * c->obs_std));
        size_t nout = 1 + (size_t)c->n * c->dx + c->ncov + 2;
        launch_finish(d_out, ob, (c->n + 31) / 32, xp_const, host_io ? d_out + nout : nullptr,
                      (poll && host_io == 1) ? c->h_done.d : nullptr, c->done_seq, s);
    }
    HIP_TRY(c, hipGetLastError());
    if (host_io == 2 && stop_after >= 5) {
        // result and result words by DMA behind the kernels; the completion word (when polled) behind the copies
        size_t nout = 1 + (size_t)c->n * c->dx + c->ncov + 2;
        HIP_TRY(c, hipMemcpyAsync(c->h_out.p, d_out, (nout + (objective ? 2 : 0)) * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(c->h_res.p, c->d_res.p, c->res_words * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (poll) {
            launch_done(c->h_done.d, c->done_seq, s);
            c->poll_pending = true;
        }
    } else if (!host_io || stop_after < 5) {
        HIP_TRY(c, hipMemcpyAsync(c->h_res.p, c->d_res.p, c->res_words * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    } else if (poll) {
        // (round 5: the completion word by a stream memory operation on the pinned word instead of the one-thread kernel —
        // no launch, the command processor writes it behind the assembly: 0.3773 vs 0.3765 ms per step, nothing; removed)
        if (!objective) launch_done(c->h_done.d, c->done_seq, s);
        c->poll_pending = true;
    }
    if (!c->ev_last) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_last, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->ev_last, s));
    c->eval_pending = true;
    c->pending_reblocked = reblock;
    c->last_stream = s;
    return GPRF_OK;
}

// the control words of a finished (re-partitioning) evaluation: take over the sizes; an overflow grows the workspace
// and returns GPRF_RETRY
int absorb_control_words(gprf_ctx *c, bool reblocked_run, int32_t *reblocked) {
    const int32_t *ctl = c->h_res.p, *bsz = c->h_res.p + CTL_WORDS + c->n_local;
    bool changed = reblocked_run && ctl[CTL_CHANGED] == c->pending_epoch;
    if (reblocked) *reblocked = changed ? 1 : 0;
    if (changed || ctl[CTL_OVERFLOW]) {
        c->h_bsize.assign(bsz, bsz + c->n_blocks);
        c->h_assign_host.clear();           // the host's copy of an uploaded partition is history now
        c->h_posb_host.clear();
        refresh_host_units(c);
    }
    if (ctl[CTL_OVERFLOW]) {
        int64_t mat = ((int64_t)ctl[CTL_MAT_HI] << 32) | (uint32_t)ctl[CTL_MAT_LO];
        c->need_build = true;
        if (ctl[CTL_MAXM] > max_unit_limit()) {
            char buf[200];
            snprintf(buf, sizeof buf, "after re-blocking a unit has %d points; the kernels accept at most %d per unit "
                     "(GPRF_MAX_UNIT)", ctl[CTL_MAXM], max_unit_limit());
            return fail(c, GPRF_ERR_ARG, buf);
        }
        int rc = size_workspace(c, ctl[CTL_ROWS], mat + mat / 4, ctl[CTL_MAXT]);
        if (rc != GPRF_OK) return rc;
        return GPRF_RETRY;
    }
    // keep the launch-wide bound at the present largest unit (it only has to grow through the overflow path)
    // ... and follows it down only after a run of smaller partitions: an optimiser whose iterates hover around a
    // tile boundary would otherwise pay a repeated evaluation every time the largest unit crosses it upwards
    if (changed) {
        int maxT = 0;
        for (int l = 0; l < c->n_local; ++l) maxT = std::max(maxT, pad16(c->l_m[l]) / 16);
        c->shrink_votes = maxT < c->max_T ? c->shrink_votes + 1 : 0;
        if (c->shrink_votes >= 8) {
            c->shrink_votes = 0;
            int rc = size_workspace(c, 0, 0, 0);
            if (rc != GPRF_OK) return rc;
        }
    }
    return GPRF_OK;
}

// after the stream has been synchronised: GPRF_OK / GPRF_NOT_PD / GPRF_RETRY (the partition outgrew the workspace:
// it has been grown, enqueue the evaluation again without re-partitioning)
// Wait for stream s without ever blocking for good: an evaluation whose two Cholesky queues wait for each other through
// stream memory operations cannot finish under a tool that serialises the dispatches of all queues (launch_potrf picks
// events when it recognises such an environment; this is the net under the ones it does not).  GPRF_EVAL_TIMEOUT_S
// (default 120) bounds it.
int bounded_stream_wait(gprf_ctx *c, hipStream_t s) {
    if (c->poisoned) return fail(c, GPRF_ERR_HIP, "an earlier evaluation of this context timed out; the context is unusable");
    // GPRF_SYNC=block: a true blocking wait (the host thread sleeps in the runtime) — except under a tool that may
    // serialise the queues, where only the bounded form below can get out of a stuck evaluation
    if (!c->spin && !potrf_tool_env()) {
        hipError_t e = hipStreamSynchronize(s);
        return e == hipSuccess ? GPRF_OK : fail(c, GPRF_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    }
    static const double limit = [] { const char *e = getenv("GPRF_EVAL_TIMEOUT_S"); double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 120.0; }();
    auto t0 = std::chrono::steady_clock::now();
    for (long it = 0;; ++it) {
        hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) return GPRF_OK;
        if (q != hipErrorNotReady) return fail(c, GPRF_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
        double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el > limit) {
            char buf[256];
            snprintf(buf, sizeof buf, "the evaluation did not finish within %.0f s (GPRF_EVAL_TIMEOUT_S): a tool that serialises "
                     "dispatches across queues? set GPRF_POTRF_DUAL=2 (one queue) or GPRF_SIDE_EVENTS=1", limit);
            // the work is still queued (or deadlocked): nothing of this context may ever be waited for again —
            // gprf_destroy then skips every synchronisation and leaks the buffers the device may still touch
            c->poisoned = true;
            return fail(c, GPRF_ERR_HIP, buf);
        }
        if (el > 2e-3) std::this_thread::sleep_for(std::chrono::microseconds(el > 0.1 ? 1000 : 50));
        else __builtin_ia32_pause();
    }
}

int finish_eval(gprf_ctx *c, hipStream_t s, int32_t *first_bad_unit, int32_t *reblocked) {
    if (c->last_stream) s = c->last_stream;      // (an evaluation may end on the side queue: launch_potrf's tail)
    if (c->poll_pending) {
        // spin on the sequence number k_done stores into pinned memory (bounded: fall back to querying the stream)
        volatile int32_t *flag = c->h_done.p;
        auto t0 = std::chrono::steady_clock::now();
        long spins = 0;
        while (*flag != c->done_seq) {
            __builtin_ia32_pause();
            if ((++spins & 0xfff) == 0 &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.05) {
                int rc = bounded_stream_wait(c, s);
                if (rc != GPRF_OK) { c->poll_pending = false; c->eval_pending = false; return rc; }
                break;
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        c->poll_pending = false;
    } else if (c->io_mode == 1 && !c->poisoned && !potrf_tool_env()) {
        // the copy-based form's own completion: the runtime's stream synchronisation
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) {
            c->eval_pending = false;
            c->poll_pending = false;
            return fail(c, GPRF_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
        }
    } else {
        int rc = bounded_stream_wait(c, s);
        if (rc != GPRF_OK) { c->eval_pending = false; return rc; }
    }
    c->eval_pending = false;
    if (first_bad_unit) *first_bad_unit = -1;
    int32_t rb = 0;
    int rc = absorb_control_words(c, c->pending_reblocked, &rb);
    c->last_reblocked = rb != 0;
    if (reblocked) *reblocked = rb;
    if (rc != GPRF_OK) return rc;
    const int32_t *info = c->h_res.p + CTL_WORDS;
    int bad = -1;
    for (int l = 0; l < c->n_local; ++l)
        if (info[l] != 0) { bad = c->l_global[l]; break; }
    if (first_bad_unit) *first_bad_unit = bad;
    if (bad >= 0) {
        char buf[128];
        snprintf(buf, sizeof buf, "unit %d: kernel matrix not positive definite", bad);
        c->err = buf;
        return GPRF_NOT_PD;
    }
    return GPRF_OK;
}

// every member: the call, first failure reported through the front context
#define GROUP_BROADCAST(c, call)                                        \
    if (!(c)->kids.empty()) {                                           \
        for (gprf_ctx *k : (c)->kids) {                                 \
            int rc__ = (call);                                          \
            if (rc__ != GPRF_OK) { (c)->err = k->err; return rc__; }    \
        }                                                               \
    }

#define GROUP_FORWARD(c, call)                                          \
    if ((c) && !(c)->kids.empty()) {                                    \
        GROUP_BROADCAST(c, call)                                        \
        return GPRF_OK;                                                 \
    }
#define GROUP_REFUSE(c, what)                                           \
    if ((c) && !(c)->kids.empty())                                      \
        return fail((c), GPRF_ERR_STATE, what " is not available on a multi-device group (it evaluates through gprf_eval / gprf_update_eval / gprf_objective)");

// One evaluation over a multi-device group, driven from this one host thread (the reference's single-process drivers,
// gprfopt.py:377-422, with the fan-out inside llgrad, gprf.py:218-233): X once into the front's pinned buffer, which every
// member's kernels read; every member enqueues its shard on its own device and stream, its assembly kernel storing the
// partial vector into its slot on the front device; k_sum_parts behind all of them; one completion flag.
int group_run(gprf_ctx *c, const double *X, int want_gx, int want_gc, double *ll_out, double *gradX_out, double *gradC_out,
              int32_t *first_bad_unit, bool reblock, int32_t *reblocked, bool objective) {
    const int N = (int)c->kids.size();
    size_t nx = (size_t)c->n * c->dx;
    size_t nout = 1 + nx + c->ncov + 2;
    memcpy(c->h_X.p, X, nx * sizeof(double));
    int any_reblocked = 0;
    if (first_bad_unit) *first_bad_unit = -1;
    for (int attempt = 0; attempt < 3; ++attempt) {
        for (int k = 0; k < N; ++k) {
            gprf_ctx *m = c->kids[k];
            HIP_TRY(c, hipSetDevice(m->device));
            m->objective_call = objective;
            int rc = enqueue_eval(m, c->h_X.d, want_gx, want_gc, c->slots + (size_t)k * c->slot_stride, m->stream, 6, reblock);
            m->objective_call = false;
            if (rc != GPRF_OK) {
                // the members enqueued so far still run: let them finish (bounded) so that none is left with an
                // evaluation pending behind this error
                c->err = m->err;
                for (int q = 0; q < k; ++q) {
                    gprf_ctx *mq = c->kids[q];
                    (void)hipSetDevice(mq->device);
                    (void)finish_eval(mq, mq->stream, nullptr, nullptr);
                }
                (void)hipSetDevice(c->device);
                return rc;
            }
            HIP_TRY(c, hipEventRecord(c->ev_kid[k], m->stream));
        }
        HIP_TRY(c, hipSetDevice(c->device));
        for (int k = 0; k < N; ++k) HIP_TRY(c, hipStreamWaitEvent(c->red_stream, c->ev_kid[k], 0));
        launch_sum_parts(c->slots, N, c->slot_stride, nout, c->h_out.d, c->red_stream);
        c->done_seq = c->done_seq >= 0x3fffffff ? 1 : c->done_seq + 1;
        launch_done(c->h_done.d, c->done_seq, c->red_stream);
        HIP_TRY(c, hipGetLastError());
        {   // the front's completion flag (bounded wait, like finish_eval)
            volatile int32_t *flag = c->h_done.p;
            auto t0 = std::chrono::steady_clock::now();
            long spins = 0;
            while (*flag != c->done_seq) {
                __builtin_ia32_pause();
                if ((++spins & 0xfff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.05) {
                    int rc = bounded_stream_wait(c, c->red_stream);
                    if (rc != GPRF_OK) {
                        for (gprf_ctx *m : c->kids) { m->poisoned = c->poisoned; m->eval_pending = false; m->poll_pending = false; }
                        return rc;
                    }
                    break;
                }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        // every member's own status (its stream is complete: the reduction waited for it)
        int worst = GPRF_OK, bad = -1;
        for (int k = 0; k < N; ++k) {
            gprf_ctx *m = c->kids[k];
            HIP_TRY(c, hipSetDevice(m->device));
            int32_t b = -1, rb = 0;
            int rc = finish_eval(m, m->stream, &b, &rb);
            any_reblocked |= rb;
            if (rc < 0) { c->err = m->err; return rc; }
            if (rc == GPRF_RETRY) worst = GPRF_RETRY;
            if (rc == GPRF_NOT_PD) {
                if (worst != GPRF_RETRY) worst = GPRF_NOT_PD;
                if (bad < 0 || b < bad) { bad = b; c->err = m->err; }
            }
        }
        if (worst == GPRF_RETRY) { reblock = false; continue; }      // the outgrown members have grown: everybody once more
        if (reblocked) *reblocked = any_reblocked;
        c->last_reblocked = any_reblocked != 0;
        if (worst == GPRF_NOT_PD) {
            if (first_bad_unit) *first_bad_unit = bad;
            return GPRF_NOT_PD;
        }
        *ll_out = c->h_out.p[0];
        if (want_gx) memcpy(gradX_out, c->h_out.p + 1, nx * sizeof(double));
        if (want_gc) memcpy(gradC_out, c->h_out.p + 1 + nx, c->ncov * sizeof(double));
        return GPRF_OK;
    }
    return fail(c, GPRF_ERR_STATE, "the unit tables did not fit the workspace after growing it twice");
}

// host X in -> host result out, optionally re-partitioning first; repeats when the workspace had to grow
int run_checked(gprf_ctx *c, const double *X, int want_gx, int want_gc, double *ll_out, double *gradX_out,
                double *gradC_out, int32_t *first_bad_unit, bool reblock, int32_t *reblocked, bool objective = false) {
    if (!c->kids.empty())
        return group_run(c, X, want_gx, want_gc, ll_out, gradX_out, gradC_out, first_bad_unit, reblock, reblocked, objective);
    hipStream_t s = c->stream;
    size_t nx = (size_t)c->n * c->dx;
    size_t nout = 1 + nx + c->ncov + 2;
    // zero-copy: the kernels read X from, and write the result to, pinned host memory directly (160 KB each way at
    // n = 10000: two fabric round trips instead of three copy commands with their launch latencies)
    auto tp0 = std::chrono::steady_clock::now();
    memcpy(c->h_X.p, X, nx * sizeof(double));
    auto tp1 = std::chrono::steady_clock::now();
    (void)nout;
    int any_reblocked = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        c->objective_call = objective;
        int rc;
        if (c->io_mode == 0) {
            rc = enqueue_eval(c, c->h_X.d, want_gx, want_gc, c->h_out.d, s, 6, reblock, 1);
        } else {
            // A/B forms (GPRF_IO_MODE): the result through HBM and a D2H copy; X by a copy command (1) or zero-copy (2)
            const double *xin = c->h_X.d;
            if (c->io_mode == 1) {
                HIP_TRY(c, hipMemcpyAsync(c->d_X.p, c->h_X.p, nx * sizeof(double), hipMemcpyHostToDevice, s));
                xin = c->d_X.p;
            }
            rc = enqueue_eval(c, xin, want_gx, want_gc, c->d_out.p, s, 6, reblock, 2);
        }
        c->objective_call = false;
        if (rc != GPRF_OK) return rc;
        auto tp2 = std::chrono::steady_clock::now();
        int32_t rb = 0;
        rc = finish_eval(c, s, first_bad_unit, &rb);
        auto tp3 = std::chrono::steady_clock::now();
        c->host_us[0] += std::chrono::duration<double, std::micro>(tp1 - tp0).count();
        c->host_us[1] += std::chrono::duration<double, std::micro>(tp2 - tp1).count();
        c->host_us[2] += std::chrono::duration<double, std::micro>(tp3 - tp2).count();
        tp0 = tp1 = tp3;
        any_reblocked |= rb;
        if (rc == GPRF_RETRY) { reblock = false; continue; }      // the new partition is in d_assign; tables: need_build
        if (reblocked) *reblocked = any_reblocked;
        if (rc != GPRF_OK) return rc;
        *ll_out = c->h_out.p[0];
        if (want_gx) memcpy(gradX_out, c->h_out.p + 1, nx * sizeof(double));
        if (want_gc) memcpy(gradC_out, c->h_out.p + 1 + nx, c->ncov * sizeof(double));
        c->host_us[3] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tp3).count();
        c->host_n++;
        return GPRF_OK;
    }
    return fail(c, GPRF_ERR_STATE, "the unit tables did not fit the workspace after growing it twice");
}

// the partition as block-of-point, whoever made it
int fetch_assignment(gprf_ctx *c, int32_t *out) {
    if (c->host_blocks_dirty && (int)c->h_assign_host.size() == c->n) {
        memcpy(out, c->h_assign_host.data(), (size_t)c->n * sizeof(int32_t));
        return GPRF_OK;
    }
    if (!c->assign_valid) return fail(c, GPRF_ERR_STATE, "no partition on the device yet");
    if (c->ev_last) HIP_TRY(c, hipEventSynchronize(c->ev_last));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->n > 0) HIP_TRY(c, hipMemcpy(out, c->d_assign.p, (size_t)c->n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return GPRF_OK;
}

}  // namespace

extern "C" {

int gprf_create(gprf_ctx **out, int32_t n, int32_t dx, int32_t dy, int32_t dist_id, int32_t kern_id,
                int32_t device) {
    if (!out) return GPRF_ERR_ARG;
    *out = nullptr;
    g_create_err.clear();
    char why[200];
    if (n < 0 || dy < 1 || dy > YPAD) {
        snprintf(why, sizeof why, "n = %d, dy = %d: need n >= 0 and 1 <= dy <= %d", n, dy, YPAD);
        return create_fail(GPRF_ERR_ARG, why);
    }
    bool se = (dist_id == GPRF_DIST_EUCLIDEAN && kern_id == GPRF_KERN_SE);
    bool mt = (dist_id == GPRF_DIST_LLD && kern_id == GPRF_KERN_MATERN32);
    // the two combinations the reference's callers use
    if (!se && !mt) return create_fail(GPRF_ERR_ARG, "covariance must be (euclidean, se) or (lld, matern32)");
    if (se && (dx < 1 || dx > 3)) return create_fail(GPRF_ERR_ARG, "(euclidean, se): 1 <= dx <= 3");
    if (mt && dx != 3) return create_fail(GPRF_ERR_ARG, "(lld, matern32): dx must be 3 (lon, lat, depth)");
    int ndev = 0;
    hipError_t ce = hipGetDeviceCount(&ndev);
    if (ce != hipSuccess) return create_fail(GPRF_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(ce));
    if (device < 0 || device >= ndev) {
        snprintf(why, sizeof why, "device ordinal %d: this process sees %d HIP device(s)", device, ndev);
        return create_fail(GPRF_ERR_HIP, why);
    }
    gprf_ctx *c = new gprf_ctx();
    c->n = n; c->dx = dx; c->dy = dy; c->dist_id = dist_id; c->kern_id = kern_id; c->device = device;
    c->ndfn = se ? dx : 2;
    c->ncov = 2 + c->ndfn;
    c->n_chunks = (n + CHUNK - 1) / CHUNK;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) {
        delete c;
        return create_fail(GPRF_ERR_HIP, "hipSetDevice / hipStreamCreate failed");
    }
    if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        gprf_destroy(c);
        return create_fail(GPRF_ERR_HIP, "could not create the side stream / its events");
    }
    {
        // (optional: without it the evaluation keeps its launch-wide stage boundaries)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, least) != hipSuccess) { c->stream3 = nullptr; (void)hipGetLastError(); }
    }
    {
        int can = 0;
        (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, device);
        if (diag("side_events", 0)) can = 0;      // diagnostics: force events
        if (can && c->d_side.reserve(16, 1.0) == hipSuccess && hipMemset(c->d_side.p, 0, 16 * sizeof(uint32_t)) == hipSuccess)
            c->side_values = true;
    }
    size_t nout = 1 + (size_t)n * dx + c->ncov + 2;
    if (c->d_X.reserve((size_t)n * dx + 1, 1.0) != hipSuccess || c->d_Y.reserve((size_t)n * dy + 1, 1.0) != hipSuccess ||
        c->d_out.reserve(nout + 4, 1.0) != hipSuccess || c->h_X.reserve((size_t)n * dx + 1) != hipSuccess ||
        c->h_out.reserve(nout + 4) != hipSuccess || c->h_done.reserve(16) != hipSuccess ||
        c->d_xpart.reserve((size_t)(n + 31) / 32 + 1, 1.0) != hipSuccess) {
        gprf_destroy(c);
        return create_fail(GPRF_ERR_HIP, "out of memory for the context's resident buffers");
    }
    c->h_done.p[0] = 0;
    if (const char *e = getenv("GPRF_SYNC")) c->spin = !(e[0] == 'b');
    if (const char *e = getenv("GPRF_IO_MODE")) c->io_mode = (e[0] == '1' || e[0] == '2') ? e[0] - '0' : 0;
    *out = c;
    return GPRF_OK;
}

int gprf_create_multi(gprf_ctx **out, int32_t n, int32_t dx, int32_t dy, int32_t dist_id, int32_t kern_id,
                      int32_t n_devices, const int32_t *devices) {
    if (!out) return GPRF_ERR_ARG;
    *out = nullptr;
    g_create_err.clear();
    if (n_devices < 1 || n_devices > 64 || !devices) return create_fail(GPRF_ERR_ARG, "1 <= n_devices <= 64 device ordinals");
    gprf_ctx *c = nullptr;
    int rc = gprf_create(&c, n, dx, dy, dist_id, kern_id, devices[0]);
    if (rc != GPRF_OK) return rc;
    size_t nout = 1 + (size_t)n * dx + c->ncov + 2;
    c->slot_stride = (nout + 31) & ~(size_t)31;
    std::string why;
    auto bail = [&](int code, const std::string &msg) {
        gprf_destroy(c);
        return create_fail(code, msg);
    };
    if (hipSetDevice(devices[0]) != hipSuccess || hipStreamCreateWithFlags(&c->red_stream, hipStreamNonBlocking) != hipSuccess)
        return bail(GPRF_ERR_HIP, "could not create the reduction stream on the first device");
    // Can every member's device store into the first device's memory?  If not — no peer access between two of the
    // devices (different PCIe roots, IOMMU / container restrictions) — the slots live in pinned host memory instead:
    // slower (every partial vector crosses a host link twice) but always available.  GPRF_GROUP_HOST_SLOTS=1 forces it
    // (tests on one-GPU boxes).
    bool host_slots = false;
    if (const char *e = getenv("GPRF_GROUP_HOST_SLOTS")) {
        if (e[0] == '1') { host_slots = true; why = "GPRF_GROUP_HOST_SLOTS=1"; }
    }
    for (int k = 0; k < n_devices && !host_slots; ++k) {
        if (devices[k] == devices[0]) continue;
        int can = 0;
        hipError_t e = hipDeviceCanAccessPeer(&can, devices[k], devices[0]);
        if (e != hipSuccess || !can) {
            char buf[160];
            snprintf(buf, sizeof buf, "device %d cannot access device %d's memory (hipDeviceCanAccessPeer: %s)", devices[k], devices[0],
                     e != hipSuccess ? hipGetErrorString(e) : "no");
            host_slots = true;
            why = buf;
        }
    }
    (void)hipGetLastError();
    for (int k = 0; k < n_devices; ++k) {
        gprf_ctx *m = nullptr;
        rc = gprf_create(&m, n, dx, dy, dist_id, kern_id, devices[k]);
        if (rc != GPRF_OK) {
            std::string msg = "member " + std::to_string(k) + ": " + g_create_err;
            return bail(rc, msg);
        }
        c->kids.push_back(m);
        m->rank = k;
        m->world = n_devices;
        hipEvent_t ev = nullptr;
        if (hipSetDevice(devices[k]) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
            return bail(GPRF_ERR_HIP, "member " + std::to_string(k) + ": could not create its completion event");
        c->ev_kid.push_back(ev);
        if (!host_slots && devices[k] != devices[0]) {
            hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
            (void)hipGetLastError();
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                host_slots = true;
                why = "hipDeviceEnablePeerAccess(" + std::to_string(devices[0]) + ") from device " + std::to_string(devices[k]) + ": " +
                      hipGetErrorString(e);
            }
        }
    }
    if (hipSetDevice(devices[0]) != hipSuccess) return bail(GPRF_ERR_HIP, "hipSetDevice(first device)");
    const size_t slot_bytes = (c->slot_stride * (size_t)n_devices + 32) * sizeof(double);
    if (!host_slots) {
        // fine-grained device memory: peer stores are coherent at system scope, the front device's loads never hit a
        // stale cached line (coarse-grained hipMalloc memory gives no such guarantee across devices inside one process)
        void *ptr = nullptr;
        hipError_t e = hipExtMallocWithFlags(&ptr, slot_bytes, hipDeviceMallocFinegrained);
        (void)hipGetLastError();
        if (e == hipSuccess) {
            c->slots_base = ptr;
            c->slots = (double *)ptr;
        } else {
            host_slots = true;
            why = std::string("hipExtMallocWithFlags(fine-grained): ") + hipGetErrorString(e);
        }
    }
    if (host_slots) {
        void *hp = nullptr, *dp = nullptr;
        hipError_t e = hipHostMalloc(&hp, slot_bytes, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&dp, hp, 0);
        if (e != hipSuccess) {
            if (hp) (void)hipHostFree(hp);
            return bail(GPRF_ERR_HIP, std::string("pinned host memory for the members' slots: ") + hipGetErrorString(e));
        }
        c->slots_base = hp;
        c->slots = (double *)dp;
        c->slots_on_host = true;
        c->slots_why = why;
    }
    *out = c;
    return GPRF_OK;
}

int gprf_group_info(const gprf_ctx *c, int32_t *n_members, int32_t *slots_on_host, int32_t cap, int32_t *devices_out,
                    int32_t *units_out) {
    if (!c) return GPRF_ERR_ARG;
    const int N = (int)c->kids.size();
    if (n_members) *n_members = N;
    if (slots_on_host) *slots_on_host = c->slots_on_host ? 1 : 0;
    for (int k = 0; k < N && k < cap; ++k) {
        if (devices_out) devices_out[k] = c->kids[k]->device;
        if (units_out) units_out[k] = c->kids[k]->static_dirty ? -1 : c->kids[k]->n_local;
    }
    return GPRF_OK;
}

const char *gprf_runtime_config(void) {
    static thread_local char buf[256];
    const char *io = getenv("GPRF_IO_MODE"), *sy = getenv("GPRF_SYNC"), *dg = getenv("GPRF_DIAG");
    snprintf(buf, sizeof buf, "side_mode=%d tool_env=%d class_depth=%d io_mode=%d sync=%s diag=%.120s", potrf_side_mode(), potrf_tool_env() ? 1 : 0,
             diag("solve_class", 1) != 0 ? diag("class_depth", 3) : 0,
             (io && (io[0] == '1' || io[0] == '2')) ? io[0] - '0' : 0, (sy && sy[0] == 'b') ? "block" : "poll", dg ? dg : "");
    return buf;
}

int gprf_destroy(gprf_ctx *c) {
    if (!c) return GPRF_OK;
    if (c->host_n > 0 && getenv("GPRF_HOST_TRACE"))
        fprintf(stderr, "gprf host phases over %llu evaluations (us): copy X in %.1f | enqueue %.1f | wait %.1f | copy result out %.1f\n",
                (unsigned long long)c->host_n, c->host_us[0] / c->host_n, c->host_us[1] / c->host_n, c->host_us[2] / c->host_n,
                c->host_us[3] / c->host_n);
    for (gprf_ctx *k : c->kids) {
        if (c->poisoned) k->poisoned = true;
        (void)gprf_destroy(k);
    }
    c->kids.clear();
    if (c->poisoned) {
        // an evaluation of this context never finished (bounded_stream_wait gave up): its queues may never drain, and
        // the device may still touch its buffers — synchronising would hang exactly where the timeout got out, freeing
        // would pull memory from under running kernels.  Everything is leaked on purpose; the process should exit.
        fprintf(stderr, "gprf_destroy: context poisoned by a timed-out evaluation; its streams and buffers are leaked\n");
        delete c;
        return GPRF_OK;
    }
    (void)hipSetDevice(c->device);
    if (c->red_stream) { (void)hipStreamSynchronize(c->red_stream); (void)hipStreamDestroy(c->red_stream); }
    for (hipEvent_t e : c->ev_kid) (void)hipEventDestroy(e);
    if (c->slots_base) { if (c->slots_on_host) (void)hipHostFree(c->slots_base); else (void)hipFree(c->slots_base); }
    if (c->ev_last) (void)hipEventSynchronize(c->ev_last);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->d_X.release(); c->d_Y.release(); c->d_out.release(); c->d_tab.release(); c->h_tab.release();
    c->d_m.release(); c->d_rowoff.release(); c->d_offj.release(); c->d_upt.release(); c->d_assign.release();
    c->d_posb.release(); c->d_rank.release(); c->d_big_list.release(); c->d_small_list.release(); c->d_srec.release(); c->d_big_rec.release(); c->d_small_rec.release(); c->d_pe.release(); c->d_ebase.release(); c->d_einfo.release(); c->d_cnt.release(); c->d_matoff.release(); c->d_res.release();
    c->h_res.release(); c->h_up.release();
    c->d_cs.release(); c->d_c2.release(); c->d_side.release(); c->d_Xobs.release(); c->d_xpart.release();
    c->d_tvec.release(); c->d_tcenter.release(); c->d_tsplit.release(); c->d_tleft.release(); c->d_tright.release();
    c->d_tleaf.release();
    c->d_Vb.release();
    c->d_K.release(); c->d_U.release(); c->d_W.release(); c->d_V.release(); c->d_Xu.release();
    c->d_Z.release(); c->d_At.release(); c->d_gXu.release(); c->d_logdet.release(); c->d_zzpart.release(); c->d_usum.release();
    c->d_gcpart.release(); c->d_rowpart.release(); c->d_colpart.release(); c->d_dbg.release();
    c->h_X.release(); c->h_out.release(); c->h_done.release();
    if (c->ev_valid)
        for (int r = 0; r < gprf_ctx::RING; ++r)
            for (int i = 0; i <= GPRF_N_STAGES; ++i) (void)hipEventDestroy(c->ev[r][i]);
    if (c->ev_tables) (void)hipEventDestroy(c->ev_tables);
    if (c->ev_last) (void)hipEventDestroy(c->ev_last);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    if (c->stream3) { (void)hipStreamSynchronize(c->stream3); (void)hipStreamDestroy(c->stream3); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return GPRF_OK;
}

const char *gprf_last_error(const gprf_ctx *c) {
    if (c) return c->err.c_str();
    return g_create_err.empty() ? "null context" : g_create_err.c_str();      // why this thread's last create failed
}

int gprf_set_Y(gprf_ctx *c, const double *Y) {
    if (!c || !Y) return GPRF_ERR_ARG;
    GROUP_FORWARD(c, gprf_set_Y(k, Y))
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->ev_last) HIP_TRY(c, hipEventSynchronize(c->ev_last));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->d_Y.p, Y, (size_t)c->n * c->dy * sizeof(double), hipMemcpyHostToDevice));
    c->have_Y = true;
    return GPRF_OK;
}

int gprf_set_theta(gprf_ctx *c, const double *theta, int32_t ntheta) {
    if (!c || !theta) return GPRF_ERR_ARG;
    if (ntheta != c->ncov) return fail(c, GPRF_ERR_ARG, "theta must be [noise_var, signal_var, dfn_params...]");
    for (int i = 0; i < ntheta; ++i)
        if (!std::isfinite(theta[i])) return fail(c, GPRF_ERR_ARG, "non-finite hyper-parameter");
    c->theta.assign(theta, theta + ntheta);
    c->have_theta = true;
    GROUP_FORWARD(c, gprf_set_theta(k, theta, ntheta))
    return GPRF_OK;
}

int gprf_set_blocks(gprf_ctx *c, int32_t n_blocks, const int64_t *block_ptr, const int32_t *point_idx) {
    if (!c || n_blocks < 0 || !block_ptr) return GPRF_ERR_ARG;
    GROUP_FORWARD(c, gprf_set_blocks(k, n_blocks, block_ptr, point_idx))
    if (block_ptr[0] != 0) return fail(c, GPRF_ERR_ARG, "block_ptr[0] must be 0");
    for (int b = 0; b < n_blocks; ++b)
        if (block_ptr[b + 1] < block_ptr[b]) return fail(c, GPRF_ERR_ARG, "block_ptr must be non-decreasing");
    int64_t tot = block_ptr[n_blocks];
    if (tot > 0 && !point_idx) return GPRF_ERR_ARG;
    if (n_blocks != c->n_blocks && c->n_pairs > 0)
        for (int q = 0; q < 2 * c->n_pairs; ++q)     // pairs refer to block ids
            if (c->pairs[q] >= n_blocks) return fail(c, GPRF_ERR_STATE, "existing neighbor pairs exceed the new block count");
    // a point belongs to at most one block: the scatter of the gradient rows (gprf.py:258-273) is a gather here
    std::vector<int32_t> assign((size_t)c->n, -1), posb((size_t)c->n, 0), bsize((size_t)n_blocks, 0);
    for (int b = 0; b < n_blocks; ++b) {
        bsize[b] = (int32_t)(block_ptr[b + 1] - block_ptr[b]);
        for (int64_t k = block_ptr[b]; k < block_ptr[b + 1]; ++k) {
            int32_t p = point_idx[k];
            if (p < 0 || p >= c->n) return fail(c, GPRF_ERR_ARG, "point index out of range");
            if (assign[p] >= 0) {
                char buf[160];
                snprintf(buf, sizeof buf, "point %d is listed twice (blocks %d and %d): blocks must be disjoint", p, assign[p], b);
                return fail(c, GPRF_ERR_ARG, buf);
            }
            assign[p] = b;
            posb[p] = (int32_t)(k - block_ptr[b]);
        }
    }
    c->n_blocks = n_blocks;
    c->h_assign_host.swap(assign);
    c->h_posb_host.swap(posb);
    c->h_bsize.swap(bsize);
    c->host_blocks_dirty = true;
    c->have_blocks = true;
    // launch order, shard and workspace follow the sizes: redone with every host partition
    c->static_dirty = true;
    c->owner_dirty = true;
    return GPRF_OK;
}

int gprf_nearest_center(int32_t n, int32_t dx, const double *X, int32_t nc, const double *centers,
                        int32_t *block_of) {
    if (n < 0 || dx < 1 || dx > 8 || nc < 1 || !X || !centers || !block_of) return GPRF_ERR_ARG;
    // centres as structure-of-arrays so that the radicand loop vectorises
    std::vector<double> c2(nc), cs((size_t)dx * nc);
    for (int k = 0; k < nc; ++k) {
#pragma clang fp contract(off)
        double s = 0.0;
        for (int d = 0; d < dx; ++d) {
            double v = centers[(size_t)k * dx + d];
            cs[(size_t)d * nc + k] = v;
            s += v * v;
        }
        c2[k] = s;
    }
    auto work = [&](int p0, int p1) {
#pragma clang fp contract(off)        // bit-for-bit the arithmetic of the device kernel k_assign
        std::vector<double> r(nc);
        for (int p = p0; p < p1; ++p) {
            const double *x = X + (size_t)p * dx;
            double x2 = 0.0;
            for (int d = 0; d < dx; ++d) x2 += x[d] * x[d];
            for (int k = 0; k < nc; ++k) r[k] = 0.0;
            for (int d = 0; d < dx; ++d) {
                const double xd = x[d];
                const double *cd = cs.data() + (size_t)d * nc;
                for (int k = 0; k < nc; ++k) r[k] += xd * cd[k];
            }
            // r = x2 - 2 x.c + c2, the radicand of pair_distances (block_clustering.py:4-5); numpy's argmin over
            // sqrt(r): a negative r gives NaN and the FIRST NaN wins, otherwise the first minimum
            int best = 0;
            double bestv = x2 - 2.0 * r[0] + c2[0];
            bool best_nan = bestv < 0.0;
            for (int k = 1; k < nc && !best_nan; ++k) {
                double v = x2 - 2.0 * r[k] + c2[k];
                if (v < 0.0) { best = k; best_nan = true; }
                else if (v < bestv) { best = k; bestv = v; }
            }
            block_of[p] = best;
        }
    };
    int nthreads = (int)std::min<long>(8, std::max<long>(1, (long)n * nc / 200000));
    if (nthreads <= 1) {
        work(0, n);
    } else {
        std::vector<std::thread> th;
        int chunk = (n + nthreads - 1) / nthreads;
        for (int t = 0; t < nthreads; ++t) {
            int p0 = t * chunk, p1 = std::min(n, p0 + chunk);
            if (p0 < p1) th.emplace_back(work, p0, p1);
        }
        for (auto &t : th) t.join();
    }
    return GPRF_OK;
}

int gprf_set_block_assignment(gprf_ctx *c, int32_t n_blocks, const int32_t *block_of) {
    if (!c || n_blocks < 0 || (c->n > 0 && !block_of)) return GPRF_ERR_ARG;
    std::vector<int64_t> ptr((size_t)n_blocks + 1, 0);
    for (int p = 0; p < c->n; ++p) {
        int b = block_of[p];
        if (b < 0 || b >= n_blocks) return fail(c, GPRF_ERR_ARG, "block id out of range");
        ptr[b + 1]++;
    }
    for (int b = 0; b < n_blocks; ++b) ptr[b + 1] += ptr[b];
    std::vector<int32_t> pts((size_t)c->n);
    std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
    for (int p = 0; p < c->n; ++p) pts[cur[block_of[p]]++] = p;
    return gprf_set_blocks(c, n_blocks, ptr.data(), pts.data());
}

int gprf_set_centers(gprf_ctx *c, int32_t nc, const double *centers) {
    if (!c || nc < 1 || !centers) return GPRF_ERR_ARG;
    if (!c->kids.empty()) { c->n_centers = nc; c->tree_nodes = 0; }
    GROUP_FORWARD(c, gprf_set_centers(k, nc, centers))
    HIP_TRY(c, hipSetDevice(c->device));
    int dx = c->dx;
    std::vector<double> c2(nc), cs((size_t)dx * nc);
    for (int k = 0; k < nc; ++k) {
#pragma clang fp contract(off)
        double s = 0.0;
        for (int d = 0; d < dx; ++d) {
            double v = centers[(size_t)k * dx + d];
            cs[(size_t)d * nc + k] = v;
            s += v * v;
        }
        c2[k] = s;
    }
    HIP_TRY(c, c->d_cs.reserve(cs.size() + 1));
    HIP_TRY(c, c->d_c2.reserve(c2.size() + 1));
    if (c->ev_last) HIP_TRY(c, hipEventSynchronize(c->ev_last));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->d_cs.p, cs.data(), cs.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_c2.p, c2.data(), c2.size() * sizeof(double), hipMemcpyHostToDevice));
    c->n_centers = nc;
    c->tree_nodes = 0;
    // the reference's grid_centers (gprfopt.py:519-523): centre ix * g + iy = (a[ix], b[iy]), both axes ascending and uniformly
    // spaced — recognised here so that k_assign looks at the 3 x 3 centres around a point only (GridHint; diag grid_hint=0: off)
    c->grid_hint = GridHint{0, 0.0, 0.0, 0.0, 0.0};
    int g = (int)std::lround(std::sqrt((double)nc));
    if (dx == 2 && g >= 2 && g * g == nc && diag("grid_hint", 1) != 0) {
        bool ok = true;
        for (int ix = 0; ix < g && ok; ++ix)
            for (int iy = 0; iy < g && ok; ++iy) {
                const double *ck = centers + (size_t)(ix * g + iy) * 2;
                ok = ck[0] == centers[(size_t)(ix * g) * 2] && ck[1] == centers[(size_t)iy * 2 + 1];
            }
        const double a0 = centers[0], b0 = centers[1];
        const double ha = ok ? (centers[(size_t)((g - 1) * g) * 2] - a0) / (g - 1) : 0.0;
        const double hb = ok ? (centers[(size_t)(g - 1) * 2 + 1] - b0) / (g - 1) : 0.0;
        ok = ok && ha > 0.0 && hb > 0.0 && std::isfinite(ha) && std::isfinite(hb);
        for (int i = 0; i < g && ok; ++i) {
            ok = std::fabs(centers[(size_t)(i * g) * 2] - (a0 + i * ha)) <= 1e-9 * ha &&
                 std::fabs(centers[(size_t)i * 2 + 1] - (b0 + i * hb)) <= 1e-9 * hb;
        }
        // (the margin argument of the fast path wants the grid's extent and spacing in a sane range)
        ok = ok && std::fabs(a0) <= 1e3 && std::fabs(b0) <= 1e3 && g * ha <= 1e3 && g * hb <= 1e3 && ha >= 1e-4 && hb >= 1e-4;
        if (ok) c->grid_hint = GridHint{g, a0, 1.0 / ha, b0, 1.0 / hb};
    }
    return GPRF_OK;
}

int gprf_set_split_tree(gprf_ctx *c, int32_t n_nodes, int32_t dim, int32_t lon_wrap, const double *vec,
                        const double *center, const double *split, const int32_t *left, const int32_t *right,
                        const int32_t *leaf_block) {
    if (!c || n_nodes < 1 || dim < 1 || !vec || !center || !split || !left || !right || !leaf_block) return GPRF_ERR_ARG;
    if (!c->kids.empty()) {
        GROUP_BROADCAST(c, gprf_set_split_tree(k, n_nodes, dim, lon_wrap, vec, center, split, left, right, leaf_block))
        c->n_centers = c->kids[0]->n_centers;
        c->tree_nodes = n_nodes;
        return GPRF_OK;
    }
    if (dim > c->dx || dim > 8) return fail(c, GPRF_ERR_ARG, "tree dimension exceeds the point dimension");
    // a well-formed tree: children point forward (so every descent ends), leaves carry distinct block ids 0..n_leaves-1
    int n_leaves = 0;
    for (int k = 0; k < n_nodes; ++k) {
        if (left[k] < 0) { ++n_leaves; continue; }
        if (left[k] <= k || left[k] >= n_nodes || right[k] <= k || right[k] >= n_nodes)
            return fail(c, GPRF_ERR_ARG, "tree children must have larger node ids than their parent");
    }
    std::vector<char> seen((size_t)n_leaves, 0);
    for (int k = 0; k < n_nodes; ++k) {
        if (left[k] >= 0) continue;
        int b = leaf_block[k];
        if (b < 0 || b >= n_leaves || seen[b]) return fail(c, GPRF_ERR_ARG, "leaf block ids must be a permutation of 0..n_leaves-1");
        seen[b] = 1;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    size_t nd = (size_t)n_nodes * dim;
    HIP_TRY(c, c->d_tvec.reserve(nd + 1));
    HIP_TRY(c, c->d_tcenter.reserve(nd + 1));
    HIP_TRY(c, c->d_tsplit.reserve((size_t)n_nodes + 1));
    HIP_TRY(c, c->d_tleft.reserve((size_t)n_nodes + 1));
    HIP_TRY(c, c->d_tright.reserve((size_t)n_nodes + 1));
    HIP_TRY(c, c->d_tleaf.reserve((size_t)n_nodes + 1));
    if (c->ev_last) HIP_TRY(c, hipEventSynchronize(c->ev_last));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->d_tvec.p, vec, nd * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_tcenter.p, center, nd * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_tsplit.p, split, (size_t)n_nodes * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_tleft.p, left, (size_t)n_nodes * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_tright.p, right, (size_t)n_nodes * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_tleaf.p, leaf_block, (size_t)n_nodes * sizeof(int32_t), hipMemcpyHostToDevice));
    c->tree_nodes = n_nodes; c->tree_dim = dim; c->tree_wrap = lon_wrap ? 1 : 0;
    c->n_centers = n_leaves;
    return GPRF_OK;
}

int gprf_assign_blocks(gprf_ctx *c, const double *X, int32_t *changed, int32_t *block_of_out) {
    if (!c || !X || !changed) return GPRF_ERR_ARG;
    if (!c->kids.empty()) {      // every member holds the whole partition (its shard of the units is cut from it)
        for (size_t k = 0; k < c->kids.size(); ++k) {
            int32_t ch = 0;
            int rc = gprf_assign_blocks(c->kids[k], X, &ch, k == 0 ? block_of_out : nullptr);
            if (rc != GPRF_OK) { c->err = c->kids[k]->err; return rc; }
            if (k == 0) *changed = ch;
        }
        return GPRF_OK;
    }
    if (c->n_centers < 1) return fail(c, GPRF_ERR_STATE, "gprf_set_centers or gprf_set_split_tree first");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->have_blocks || c->n_blocks != c->n_centers || (int)c->h_bsize.size() != c->n_centers) {
        // no partition yet: size everything for an even one, the overflow path corrects it
        c->n_blocks = c->n_centers;
        c->h_bsize.assign((size_t)c->n_centers, (c->n + c->n_centers - 1) / std::max(c->n_centers, 1));
        c->h_assign_host.clear();
        c->h_posb_host.clear();
        c->host_blocks_dirty = false;
        c->have_blocks = true;
        c->static_dirty = true;
        c->owner_dirty = true;
        c->assign_valid = false;
    }
    hipStream_t s = c->stream;
    size_t nx = (size_t)c->n * c->dx;
    int any_changed = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        int rc = prepare(c, s);
        if (rc != GPRF_OK) return rc;
        memcpy(c->h_X.p, X, nx * sizeof(double));
        HIP_TRY(c, hipMemcpyAsync(c->d_X.p, c->h_X.p, nx * sizeof(double), hipMemcpyHostToDevice, s));
        rc = enqueue_partition(c, c->d_X.p, s);
        if (rc != GPRF_OK) return rc;
        if (build_scatter_fits(make_build(c))) {
            launch_build_scatter(make_build(c), c->d_X.p, c->dx, c->dist_id, c->need_build ? 1 : 0, c->epoch, s);
        } else {
            launch_build_tables(make_build(c), 1, c->need_build ? 1 : 0, c->epoch, s);
            launch_scatter_x(make_build(c), c->d_X.p, c->dx, c->dist_id, 1, c->need_build ? 1 : 0, c->epoch, s);
        }
        c->need_build = false;
        c->pending_epoch = c->epoch;
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(c->h_res.p, c->d_res.p, c->res_words * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        int32_t rb = 0;
        rc = absorb_control_words(c, true, &rb);
        any_changed |= rb;
        if (rc == GPRF_RETRY) continue;
        if (rc != GPRF_OK) return rc;
        *changed = any_changed;
        if (any_changed && block_of_out) return fetch_assignment(c, block_of_out);
        return GPRF_OK;
    }
    return fail(c, GPRF_ERR_STATE, "the unit tables did not fit the workspace after growing it twice");
}

int gprf_get_block_assignment(gprf_ctx *c, int32_t *block_of_out) {
    if (!c || !block_of_out) return GPRF_ERR_ARG;
    if (!c->kids.empty()) return gprf_get_block_assignment(c->kids[0], block_of_out);
    HIP_TRY(c, hipSetDevice(c->device));
    return fetch_assignment(c, block_of_out);
}

int gprf_pair_kernel_max(gprf_ctx *c, const double *X, int32_t n_blocks, const int64_t *block_ptr, const int32_t *point_idx,
                         double threshold, int32_t n_cand, const int32_t *cand_ij, int32_t *keep_out, double *max_out) {
    if (!c || !X || n_blocks < 0 || !block_ptr || n_cand < 0 || (n_cand > 0 && (!cand_ij || !keep_out))) return GPRF_ERR_ARG;
    if (!c->kids.empty()) {      // one-time setup work: the first member's device
        int rc = gprf_pair_kernel_max(c->kids[0], X, n_blocks, block_ptr, point_idx, threshold, n_cand, cand_ij, keep_out, max_out);
        if (rc != GPRF_OK) c->err = c->kids[0]->err;
        return rc;
    }
    if (!c->have_theta) return fail(c, GPRF_ERR_STATE, "gprf_set_theta has not been called");
    if (n_cand == 0) return GPRF_OK;
    int64_t tot = block_ptr[n_blocks];
    if (tot > 0 && !point_idx) return GPRF_ERR_ARG;
    for (int64_t k = 0; k < tot; ++k)
        if (point_idx[k] < 0 || point_idx[k] >= c->n) return fail(c, GPRF_ERR_ARG, "point index out of range");
    for (int q = 0; q < 2 * n_cand; ++q)
        if (cand_ij[q] < 0 || cand_ij[q] >= n_blocks) return fail(c, GPRF_ERR_ARG, "candidate pair refers to a block out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    // one-time setup work (the reference caches its result in a file, run_seismic.py:377-404): plain allocations
    DevBuf<double> dX, dmax;
    DevBuf<int64_t> dptr;
    DevBuf<int32_t> dpts, dcand, dkeep;
    auto cleanup = [&]() { dX.release(); dmax.release(); dptr.release(); dpts.release(); dcand.release(); dkeep.release(); };
    size_t nx = (size_t)c->n * c->dx;
    hipError_t e = hipSuccess;
    if ((e = dX.reserve(nx + 1, 1.0)) != hipSuccess || (e = dptr.reserve((size_t)n_blocks + 1, 1.0)) != hipSuccess ||
        (e = dpts.reserve((size_t)tot + 1, 1.0)) != hipSuccess || (e = dcand.reserve(2 * (size_t)n_cand, 1.0)) != hipSuccess ||
        (e = dkeep.reserve((size_t)n_cand, 1.0)) != hipSuccess || (max_out && (e = dmax.reserve((size_t)n_cand, 1.0)) != hipSuccess)) {
        cleanup();
        return fail(c, GPRF_ERR_HIP, std::string("gprf_pair_kernel_max: ") + hipGetErrorString(e));
    }
    hipStream_t s = c->stream;
    bool ok = hipMemcpyAsync(dX.p, X, nx * sizeof(double), hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(dptr.p, block_ptr, ((size_t)n_blocks + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s) == hipSuccess &&
              (tot == 0 || hipMemcpyAsync(dpts.p, point_idx, (size_t)tot * sizeof(int32_t), hipMemcpyHostToDevice, s) == hipSuccess) &&
              hipMemcpyAsync(dcand.p, cand_ij, 2 * (size_t)n_cand * sizeof(int32_t), hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemsetAsync(dkeep.p, 0, (size_t)n_cand * sizeof(int32_t), s) == hipSuccess;
    if (ok) {
        launch_pair_max(c->dist_id, c->kern_id, dX.p, c->dx, dptr.p, dpts.p, dcand.p, n_cand, make_kparams(c), threshold,
                        max_out ? 1 : 0, dkeep.p, max_out ? dmax.p : nullptr, s);
        ok = hipGetLastError() == hipSuccess &&
             hipMemcpyAsync(keep_out, dkeep.p, (size_t)n_cand * sizeof(int32_t), hipMemcpyDeviceToHost, s) == hipSuccess &&
             (!max_out || hipMemcpyAsync(max_out, dmax.p, (size_t)n_cand * sizeof(double), hipMemcpyDeviceToHost, s) == hipSuccess) &&
             hipStreamSynchronize(s) == hipSuccess;
    }
    cleanup();
    return ok ? GPRF_OK : fail(c, GPRF_ERR_HIP, "gprf_pair_kernel_max: HIP error");
}

int gprf_set_neighbors(gprf_ctx *c, int32_t n_pairs, const int32_t *pairs_ij) {
    if (!c || n_pairs < 0 || (n_pairs > 0 && !pairs_ij)) return GPRF_ERR_ARG;
    if (!c->kids.empty()) c->n_pairs = n_pairs;
    GROUP_FORWARD(c, gprf_set_neighbors(k, n_pairs, pairs_ij))
    c->n_pairs = n_pairs;
    c->pairs.assign(pairs_ij, pairs_ij + 2 * (size_t)n_pairs);
    c->static_dirty = true;
    c->owner_dirty = true;
    return GPRF_OK;
}

int gprf_partition_units(int32_t n_units, const int32_t *m, int32_t dy, int32_t world, int32_t *owner_out) {
    if (n_units < 0 || world < 1 || (n_units > 0 && (!m || !owner_out))) return GPRF_ERR_ARG;
    std::vector<int> order(n_units);
    std::iota(order.begin(), order.end(), 0);
    auto cost = [&](int u) { double mm = m[u]; return mm * mm * mm + 4.0 * mm * mm * dy; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost(a) > cost(b); });
    std::vector<double> load(world, 0.0);
    for (int u : order) {
        int best = 0;
        for (int r = 1; r < world; ++r)
            if (load[r] < load[best]) best = r;
        owner_out[u] = best;
        load[best] += cost(u);
    }
    return GPRF_OK;
}

int gprf_set_shard(gprf_ctx *c, int32_t rank, int32_t world) {
    if (!c || world < 1 || rank < 0 || rank >= world) return GPRF_ERR_ARG;
    GROUP_REFUSE(c, "gprf_set_shard")
    c->rank = rank;
    c->world = world;
    c->static_dirty = true;
    c->owner_dirty = true;
    return GPRF_OK;
}

int gprf_set_unit_jitter(gprf_ctx *c, int32_t n_units, const double *jitter) {
    if (!c) return GPRF_ERR_ARG;
    GROUP_FORWARD(c, gprf_set_unit_jitter(k, n_units, jitter))
    if (!jitter) {
        // the drivers clear the jitter before every evaluation (jitchol is stateless): clearing what is already
        // clear costs nothing
        if (c->unit_jitter.empty()) return GPRF_OK;
        c->unit_jitter.clear();
    } else {
        if (n_units < 0) return GPRF_ERR_ARG;
        if ((size_t)n_units == c->unit_jitter.size() && std::equal(jitter, jitter + n_units, c->unit_jitter.begin()))
            return GPRF_OK;
        c->unit_jitter.assign(jitter, jitter + n_units);
    }
    c->jitter_dirty = true;
    return GPRF_OK;
}

int gprf_eval_device(gprf_ctx *c, const double *d_X, int32_t want_gradX, int32_t want_gradC, double *d_out,
                     void *stream) {
    GROUP_REFUSE(c, "gprf_eval_device")
    int rc = check_ready(c);
    if (rc != GPRF_OK) return rc;
    if (!d_X || !d_out) return GPRF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return enqueue_eval(c, d_X, want_gradX, want_gradC, d_out, s, 6, false);
}

int gprf_update_eval_device(gprf_ctx *c, const double *d_X, int32_t want_gradX, int32_t want_gradC, double *d_out,
                            void *stream) {
    GROUP_REFUSE(c, "gprf_update_eval_device")
    int rc = check_ready(c);
    if (rc != GPRF_OK) return rc;
    if (!d_X || !d_out) return GPRF_ERR_ARG;
    if (c->n_centers < 1) return fail(c, GPRF_ERR_STATE, "gprf_set_centers or gprf_set_split_tree first");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return enqueue_eval(c, d_X, want_gradX, want_gradC, d_out, s, 6, true);
}

int gprf_eval_status(gprf_ctx *c, int32_t *first_bad_unit) {
    if (!c) return GPRF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->eval_pending) {
        if (first_bad_unit) *first_bad_unit = -1;
        return GPRF_OK;
    }
    return finish_eval(c, c->last_stream ? c->last_stream : c->stream, first_bad_unit, nullptr);
}

int gprf_eval(gprf_ctx *c, const double *X, int32_t want_gradX, int32_t want_gradC, double *ll_out,
              double *gradX_out, double *gradC_out, int32_t *first_bad_unit) {
    int rc = check_ready(c);
    if (rc != GPRF_OK) return rc;
    if (!X || !ll_out || (want_gradX && !gradX_out) || (want_gradC && !gradC_out)) return GPRF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    return run_checked(c, X, want_gradX, want_gradC, ll_out, gradX_out, gradC_out, first_bad_unit, false, nullptr);
}

int gprf_update_eval(gprf_ctx *c, const double *X, int32_t want_gradX, int32_t want_gradC, double *ll_out,
                     double *gradX_out, double *gradC_out, int32_t *first_bad_unit, int32_t *reblocked) {
    int rc = check_ready(c);
    if (rc != GPRF_OK) return rc;
    if (!X || !ll_out || (want_gradX && !gradX_out) || (want_gradC && !gradC_out)) return GPRF_ERR_ARG;
    if (c->n_centers < 1) return fail(c, GPRF_ERR_STATE, "gprf_set_centers or gprf_set_split_tree first");
    HIP_TRY(c, hipSetDevice(c->device));
    return run_checked(c, X, want_gradX, want_gradC, ll_out, gradX_out, gradC_out, first_bad_unit, true, reblocked);
}

// ---- the optimiser-facing objective (gprfopt.py:320-417): pure host pieces first ----
int gprf_x_prior(int64_t n_elems, const double *x, const double *x_obs, double obs_std, double *ll_out, double *grad_out) {
    if (n_elems < 0 || (n_elems > 0 && (!x || !x_obs)) || !ll_out || !(obs_std > 0.0)) return GPRF_ERR_ARG;
    const double var = obs_std * obs_std;
    // blocked summation (32 elements x dx lanes per partial on the device; here blocks of 64): error growth like the
    // pairwise sum numpy uses, fixed order
    double total = 0.0;
    for (int64_t i0 = 0; i0 < n_elems; i0 += 64) {
        double part = 0.0;
        const int64_t i1 = std::min<int64_t>(n_elems, i0 + 64);
        for (int64_t i = i0; i < i1; ++i) {
            const double d = x[i] - x_obs[i];
            const double r = d / obs_std;
            part += r * r;
            if (grad_out) grad_out[i] = -d / var;
        }
        total += part;
    }
    *ll_out = -0.5 * total - 0.5 * (double)n_elems * std::log(2.0 * M_PI * var);
    return GPRF_OK;
}

static int hyper_count(int32_t mode, int32_t ncov) { return mode == GPRF_HYPER_TIED ? 1 : (mode == GPRF_HYPER_FULL ? ncov : 0); }

int gprf_hyper_unpack(int32_t mode, double cov_scale, double fixed_nv, double fixed_sv, int32_t ncov, const double *zh,
                      double *theta_out) {
    if (ncov < 3 || !theta_out || (mode != GPRF_HYPER_TIED && mode != GPRF_HYPER_FULL) || !zh || !(cov_scale > 0.0))
        return GPRF_ERR_ARG;
    if (mode == GPRF_HYPER_TIED) {
        const double ls = std::exp(zh[0] / cov_scale);
        theta_out[0] = fixed_nv;
        theta_out[1] = fixed_sv;
        for (int t = 2; t < ncov; ++t) theta_out[t] = ls;
    } else {
        for (int t = 0; t < ncov; ++t) theta_out[t] = std::exp(zh[t] / cov_scale);
    }
    return GPRF_OK;
}

int gprf_hyper_grad(int32_t mode, double cov_scale, double prior_mean, double prior_std, int32_t ncov, const double *zh,
                    const double *gradC, double *prior_ll_out, double *grad_zh_out) {
    if (ncov < 1 || (mode == GPRF_HYPER_TIED && ncov < 3) || (mode != GPRF_HYPER_TIED && mode != GPRF_HYPER_FULL) || !zh ||
        !gradC || !prior_ll_out || !grad_zh_out || !(cov_scale > 0.0) || !(prior_std > 0.0))
        return GPRF_ERR_ARG;
    const int nh = hyper_count(mode, ncov);
    const double pvar = prior_std * prior_std;
    double ss = 0.0;
    for (int t = 0; t < nh; ++t) {
        const double cl = zh[t] / cov_scale;                 // the log parameter
        const double r = (cl - prior_mean) / prior_std;
        ss += r * r;
        // chain rule through theta = exp(cl): d ll / d cl = (d ll / d theta) theta; a tied lengthscale collects every
        // lengthscale's derivative
        double dth = 0.0;
        if (mode == GPRF_HYPER_TIED)
            for (int q = 2; q < ncov; ++q) dth += gradC[q];
        else
            dth = gradC[t];
        grad_zh_out[t] = (dth * std::exp(cl) + -(cl - prior_mean) / pvar) / cov_scale;
    }
    *prior_ll_out = -0.5 * ss - 0.5 * (double)nh * std::log(2.0 * M_PI * pvar);
    return GPRF_OK;
}

int gprf_set_x_prior(gprf_ctx *c, const double *X_obs, double obs_std) {
    if (!c) return GPRF_ERR_ARG;
    if (!c->kids.empty()) {      // the data lives with the members (the first adds the prior), the layout of z with the front
        GROUP_BROADCAST(c, gprf_set_x_prior(k, X_obs, obs_std))
        c->xprior = X_obs != nullptr;
        c->obs_std = obs_std;
        c->h_Xobs_front.assign(X_obs ? X_obs : nullptr, X_obs ? X_obs + (size_t)c->n * c->dx : nullptr);
        return GPRF_OK;
    }
    if (!X_obs) { c->xprior = false; return GPRF_OK; }
    if (!(obs_std > 0.0) || !std::isfinite(obs_std)) return fail(c, GPRF_ERR_ARG, "obs_std must be positive");
    HIP_TRY(c, hipSetDevice(c->device));
    size_t nx = (size_t)c->n * c->dx;
    HIP_TRY(c, c->d_Xobs.reserve(nx + 1, 1.0));
    if (c->ev_last) HIP_TRY(c, hipEventSynchronize(c->ev_last));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (nx) HIP_TRY(c, hipMemcpy(c->d_Xobs.p, X_obs, nx * sizeof(double), hipMemcpyHostToDevice));
    c->obs_std = obs_std;
    c->xprior = true;
    return GPRF_OK;
}

int gprf_set_hyper_param(gprf_ctx *c, int32_t mode, double cov_scale, double prior_mean, double prior_std,
                         double fixed_noise_var, double fixed_signal_var) {
    if (!c) return GPRF_ERR_ARG;
    if (mode != GPRF_HYPER_NONE && mode != GPRF_HYPER_TIED && mode != GPRF_HYPER_FULL)
        return fail(c, GPRF_ERR_ARG, "unknown hyper-parameter mode");
    if (mode != GPRF_HYPER_NONE && (!(cov_scale > 0.0) || !(prior_std > 0.0)))
        return fail(c, GPRF_ERR_ARG, "cov_scale and prior_std must be positive");
    c->hyper_mode = mode; c->cov_scale = cov_scale; c->hp_mean = prior_mean; c->hp_std = prior_std;
    c->fixed_nv = fixed_noise_var; c->fixed_sv = fixed_signal_var;
    GROUP_FORWARD(c, gprf_set_hyper_param(k, mode, cov_scale, prior_mean, prior_std, fixed_noise_var, fixed_signal_var))
    return GPRF_OK;
}

int gprf_objective(gprf_ctx *c, const double *z, int32_t nz, const double *X_fixed, int32_t reblock, double *f_out,
                   double *grad_out, double *parts_out, int32_t *first_bad_unit, int32_t *reblocked) {
    if (!c || !f_out || !grad_out || (nz > 0 && !z)) return GPRF_ERR_ARG;
    const int nx = c->xprior ? c->n * c->dx : 0;
    const int nh = hyper_count(c->hyper_mode, c->ncov);
    if (nz != nx + nh) return fail(c, GPRF_ERR_ARG, "z must hold [locations (when a location prior is set) | hyper-parameters]");
    if (nx == 0 && !X_fixed) return fail(c, GPRF_ERR_ARG, "no location prior set: pass the (fixed) locations");
    if (nh > 0) {
        std::vector<double> theta((size_t)c->ncov);
        int rc = gprf_hyper_unpack(c->hyper_mode, c->cov_scale, c->fixed_nv, c->fixed_sv, c->ncov, z + nx, theta.data());
        if (rc != GPRF_OK) return fail(c, rc, "gprf_hyper_unpack");
        rc = gprf_set_theta(c, theta.data(), c->ncov);
        if (rc != GPRF_OK) return rc;
    }
    int rc = check_ready(c);
    if (rc != GPRF_OK) return rc;
    if (reblock && c->n_centers < 1) return fail(c, GPRF_ERR_STATE, "gprf_set_centers or gprf_set_split_tree first");
    HIP_TRY(c, hipSetDevice(c->device));
    const double *X = nx ? z : X_fixed;
    double f = 0.0;
    std::vector<double> gC((size_t)c->ncov, 0.0);
    // gradX goes straight into the caller's vector (already negated, prior included: k_assemble)
    rc = run_checked(c, X, nx ? 1 : 0, nh ? 1 : 0, &f, grad_out, gC.data(), first_bad_unit, reblock != 0, reblocked, true);
    if (rc != GPRF_OK) return rc;
    const size_t nout = 1 + (size_t)c->n * c->dx + c->ncov + 2;
    double hp_ll = 0.0;
    if (nh > 0) {
        rc = gprf_hyper_grad(c->hyper_mode, c->cov_scale, c->hp_mean, c->hp_std, c->ncov, z + nx, gC.data(), &hp_ll,
                             grad_out + nx);
        if (rc != GPRF_OK) return fail(c, rc, "gprf_hyper_grad");
        for (int t = 0; t < nh; ++t) grad_out[nx + t] = -grad_out[nx + t];
        f -= hp_ll;
    }
    *f_out = f;
    if (parts_out) {
        if (c->kids.empty()) {
            parts_out[0] = c->h_out.p[nout]; parts_out[1] = c->h_out.p[nout + 1];
        } else {      // (a group's partial sums carry no split: the location prior again, on the host)
            double xp = 0.0;
            if (nx) (void)gprf_x_prior(nx, z, c->h_Xobs_front.data(), c->obs_std, &xp, nullptr);
            parts_out[1] = xp;
            parts_out[0] = -(f + hp_ll) - xp;
        }
        parts_out[2] = hp_ll;
    }
    return GPRF_OK;
}

int gprf_objective_device(gprf_ctx *c, const double *d_X, int32_t want_gradX, int32_t want_gradC, double *d_out,
                          void *stream, int32_t reblock) {
    GROUP_REFUSE(c, "gprf_objective_device")
    int rc = check_ready(c);
    if (rc != GPRF_OK) return rc;
    if (!d_X || !d_out) return GPRF_ERR_ARG;
    if (reblock && c->n_centers < 1) return fail(c, GPRF_ERR_STATE, "gprf_set_centers or gprf_set_split_tree first");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    c->objective_call = true;
    rc = enqueue_eval(c, d_X, want_gradX, want_gradC, d_out, s, 6, reblock != 0);
    c->objective_call = false;
    return rc;
}

int gprf_num_units(const gprf_ctx *c, int32_t *n_total, int32_t *n_local) {
    if (!c) return GPRF_ERR_ARG;
    if (!c->kids.empty()) {
        int32_t tot = 0, loc = 0, sum = 0;
        for (const gprf_ctx *k : c->kids) {
            (void)gprf_num_units(k, &tot, &loc);
            sum = (loc < 0 || sum < 0) ? -1 : sum + loc;
        }
        if (n_total) *n_total = tot;
        if (n_local) *n_local = sum;
        return GPRF_OK;
    }
    if (n_total) *n_total = c->n_blocks + c->n_pairs;
    if (n_local) *n_local = c->static_dirty ? -1 : c->n_local;
    return GPRF_OK;
}

int gprf_work_estimate(gprf_ctx *c, double *flops, double *fill_bytes) {
    int rc = check_ready(c);
    if (rc != GPRF_OK) return rc;
    if (!c->kids.empty()) {
        double f = 0, b = 0;
        for (gprf_ctx *k : c->kids) {
            double fk = 0, bk = 0;
            rc = gprf_work_estimate(k, &fk, &bk);
            if (rc != GPRF_OK) { c->err = k->err; return rc; }
            f += fk; b += bk;
        }
        if (flops) *flops = f;
        if (fill_bytes) *fill_bytes = b;
        return GPRF_OK;
    }
    if (c->static_dirty) {
        HIP_TRY(c, hipSetDevice(c->device));
        rc = rebuild_static(c);
        if (rc != GPRF_OK) return rc;
    }
    refresh_host_units(c);
    if (flops) *flops = c->work_flops;
    if (fill_bytes) *fill_bytes = c->work_fill_bytes;
    return GPRF_OK;
}

// every diagnostic define the sources know (ablations skip work inside the timed kernels; stamps / traces cost registers)
const char *gprf_build_flags(void) {
    return ""
#ifdef GPRF_PROFILE
           " GPRF_PROFILE"
#endif
#ifdef GPRF_WGTRACE
           " GPRF_WGTRACE"
#endif
#ifdef GPRF_MGRAD_FINE
           " GPRF_MGRAD_FINE"
#endif
#ifdef GPRF_MGRAD_LOOP
           " GPRF_MGRAD_LOOP"
#endif
#ifdef GPRF_SOLVE_STAMP_PART
           " GPRF_SOLVE_STAMP_PART"
#endif
#ifdef GPRF_SOLVE_GLDS
           " GPRF_SOLVE_GLDS"
#endif
        ;
}

int gprf_last_reblocked(const gprf_ctx *c, int32_t *reblocked) {
    if (!c || !reblocked) return GPRF_ERR_ARG;
    *reblocked = c->last_reblocked ? 1 : 0;
    return GPRF_OK;
}

int gprf_table_builds(gprf_ctx *c, int32_t *builds) {
    if (!c || !builds) return GPRF_ERR_ARG;
    if (!c->kids.empty()) return gprf_table_builds(c->kids[0], builds);
    *builds = 0;
    if (!c->d_res.p) return GPRF_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->ev_last) HIP_TRY(c, hipEventSynchronize(c->ev_last));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(builds, res_ctl(c) + CTL_BUILDS, sizeof(int32_t), hipMemcpyDeviceToHost));
    return GPRF_OK;
}

int gprf_set_stream_pipelines(gprf_ctx *c, int32_t enable) {
    if (!c) return GPRF_ERR_ARG;
    c->caller_pipelines = enable != 0;
    for (gprf_ctx *k : c->kids) k->caller_pipelines = enable != 0;
    return GPRF_OK;
}

int gprf_set_timing(gprf_ctx *c, int32_t enable) {
    if (!c) return GPRF_ERR_ARG;
    if (!c->kids.empty()) return gprf_set_timing(c->kids[0], enable);      // (a group: the first member's kernels)
    c->timing = enable != 0;
    if (enable == 2) {  // reset the running totals
        if (c->ev_valid) {
            HIP_TRY(c, hipSetDevice(c->device));
            for (int r = 0; r < gprf_ctx::RING; ++r)
                if (c->slot_pending[r]) { int rc = fold_slot(c, r); if (rc != GPRF_OK) return rc; }
        }
        c->n_timed = c->n_folded = 0;
        for (int i = 0; i < GPRF_N_STAGES; ++i) c->stage_ms_sum[i] = c->stage_ms_last[i] = 0.0;
    }
    return GPRF_OK;
}

int gprf_get_timing(gprf_ctx *c, int32_t n, double *ms_out) {
    if (!c || !ms_out || n < GPRF_N_STAGES) return GPRF_ERR_ARG;
    if (!c->kids.empty()) return gprf_get_timing(c->kids[0], n, ms_out);
    if (!c->ev_valid || c->n_timed == 0) return fail(c, GPRF_ERR_STATE, "no timed evaluation yet");
    HIP_TRY(c, hipSetDevice(c->device));
    for (int r = 0; r < gprf_ctx::RING; ++r)
        if (c->slot_pending[r]) { int rc = fold_slot(c, r); if (rc != GPRF_OK) return rc; }
    for (int i = 0; i < GPRF_N_STAGES; ++i) ms_out[i] = c->stage_ms_sum[i] / (double)c->n_folded;
    if (n >= 2 * GPRF_N_STAGES + 1) {
        for (int i = 0; i < GPRF_N_STAGES; ++i) ms_out[GPRF_N_STAGES + i] = c->stage_ms_last[i];
        ms_out[2 * GPRF_N_STAGES] = (double)c->n_folded;
    }
    return GPRF_OK;
}

int gprf_debug_run(gprf_ctx *c, const double *X, int32_t stop_after) {
    GROUP_REFUSE(c, "gprf_debug_run")
    int rc = check_ready(c);
    if (rc != GPRF_OK) return rc;
    if (!X) return GPRF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    size_t nx = (size_t)c->n * c->dx;
    memcpy(c->h_X.p, X, nx * sizeof(double));
    HIP_TRY(c, hipMemcpyAsync(c->d_X.p, c->h_X.p, nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->last_stop_after = stop_after;
    c->debug_mode = true;
    rc = enqueue_eval(c, c->d_X.p, 1, 1, c->d_out.p, c->stream, stop_after, false);
    c->debug_mode = false;
    if (rc != GPRF_OK) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->eval_pending = false;
    return GPRF_OK;
}

int gprf_debug_unit_shape(gprf_ctx *c, int32_t l, int32_t *m, int32_t *mp, int32_t *global_unit) {
    if (!c || c->static_dirty || l < 0 || l >= c->n_local) return GPRF_ERR_ARG;
    refresh_host_units(c);
    if (m) *m = c->l_m[l];
    if (mp) *mp = pad16(c->l_m[l]);
    if (global_unit) *global_unit = c->l_global[l];
    return GPRF_OK;
}

int gprf_debug_fetch(gprf_ctx *c, int32_t l, int32_t what, double *out, int64_t out_len) {
    if (!c || !out || c->static_dirty || l < 0 || l >= c->n_local) return GPRF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    refresh_host_units(c);
    int64_t mp = pad16(c->l_m[l]);
    int64_t roff = c->l_rowoff[l];
    const double *src = nullptr;
    int64_t len = 0;
    switch (what) {
        case 0:   // after a fill-only debug run the K pool (upper 64x64 blocks), otherwise the factor
            src = (c->last_stop_after == 0 ? c->d_K.p : c->d_U.p) + c->l_matoff[l]; len = mp * mp; break;
        case 1: src = c->d_W.p + c->l_matoff[l]; len = mp * mp; break;
        case 2: src = c->d_Z.p + roff * YPAD; len = mp * YPAD; break;
        case 3: src = c->d_At.p + roff * YPAD; len = mp * YPAD; break;
        case 4:
            if (c->gxu_pending) {      // the evaluation folded the partials inside the assembly: the per-unit form now
                hipStream_t ls = c->last_stream ? c->last_stream : c->stream;      // (behind the evaluation, wherever it ran)
                launch_gx_finalize(make_tab(c), make_pools(c), make_kparams(c), c->gxu_want_gc, ls);
                HIP_TRY(c, hipStreamSynchronize(ls));
                c->gxu_pending = false;
            }
            src = c->d_gXu.p + roff * XPAD; len = mp * XPAD; break;
        case 6: src = c->d_dbg.p + (size_t)l * 8; len = 8; break;
#ifdef GPRF_WGTRACE
        case 11: src = c->d_dbg.p + (size_t)std::max(c->n_local, 1) * 8; len = 4 * GPRF_WGTRACE_MAX; break;
#endif
        case 7: case 8: {   // per-block partials of the gradient reduction: mp x TBm x XPAD
            int64_t tbm = std::max((c->max_T + 3) / 4, 1);
            src = (what == 7 ? c->d_colpart.p : c->d_rowpart.p) + roff * tbm * XPAD;
            len = mp * tbm * XPAD;
            break;
        }
        case 5: {
            if (out_len < 4) return GPRF_ERR_ARG;
            double zz[4];
            int32_t info = 0;
            HIP_TRY(c, hipMemcpy(out + 1, c->d_logdet.p + l, sizeof(double), hipMemcpyDeviceToHost));
            HIP_TRY(c, hipMemcpy(zz, c->d_zzpart.p + (size_t)l * 4, 4 * sizeof(double), hipMemcpyDeviceToHost));
            HIP_TRY(c, hipMemcpy(&info, res_info(c) + l, sizeof(int32_t), hipMemcpyDeviceToHost));
            out[2] = (zz[0] + zz[1]) + (zz[2] + zz[3]);
            out[3] = info;
            out[0] = -0.5 * out[2] - 0.5 * c->dy * out[1] - 0.5 * c->dy * c->l_m[l] * std::log(2.0 * M_PI);
            return GPRF_OK;
        }
        case 9: {   // the unit's own (unweighted) hyper-parameter gradient, theta order (gprf.py:577-584)
            if (out_len < c->ncov) return GPRF_ERR_ARG;
            int64_t tbm = std::max((c->max_T + 3) / 4, 1);
            int64_t tb = (mp / 16 + 3) / 4;
            int64_t npair = tb * (tb + 1) / 2, stride = tbm * (tbm + 1) / 2;
            std::vector<double> part((size_t)std::max<int64_t>(npair, 1) * GC_SLOTS, 0.0);
            if (npair > 0)
                HIP_TRY(c, hipMemcpy(part.data(), c->d_gcpart.p + (size_t)l * stride * GC_SLOTS,
                                     (size_t)npair * GC_SLOTS * sizeof(double), hipMemcpyDeviceToHost));
            double g[5] = {0, 0, 0, 0, 0};
            for (int64_t P = 0; P < npair; ++P)
                for (int t = 0; t < 5; ++t) g[t] += part[(size_t)P * GC_SLOTS + t];
            out[0] = 0.5 * g[0];
            out[1] = 0.5 * g[1] / c->theta[1];
            for (int t = 2; t < c->ncov; ++t) out[t] = 0.5 * g[t];
            return GPRF_OK;
        }
        case 10: {  // the unit row -> point table of the unit (as doubles; rows >= m read -1)
            if (out_len < mp) return GPRF_ERR_ARG;
            std::vector<int32_t> r((size_t)std::max<int64_t>(mp, 1));
            if (mp > 0) HIP_TRY(c, hipMemcpy(r.data(), c->d_upt.p + roff, (size_t)mp * sizeof(int32_t), hipMemcpyDeviceToHost));
            for (int64_t k = 0; k < mp; ++k) out[k] = k < c->l_m[l] ? (double)r[k] : -1.0;
            return GPRF_OK;
        }
        default: return GPRF_ERR_ARG;
    }
    if (out_len < len) return GPRF_ERR_ARG;
    if (len > 0) HIP_TRY(c, hipMemcpy(out, src, len * sizeof(double), hipMemcpyDeviceToHost));
    return GPRF_OK;
}

}  // extern "C"
