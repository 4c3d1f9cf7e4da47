// gprf_capi.hip — host side of libgprf_hip.so: context, unit tables, workspace pools, the C ABI of
// include/gprf_hip.h.  No torch, no Python; plain pointers and sizes.
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
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 64;
        hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

struct gprf_ctx {
    int n = 0, dx = 0, dy = 0, dist_id = 0, kern_id = 0, device = 0, ndfn = 0, ncov = 0;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;
    std::string err;

    // host-side model state (what the reference keeps on the GPRF object)
    std::vector<double> theta;
    bool have_Y = false, have_theta = false, have_blocks = false;
    int n_blocks = 0, n_pairs = 0;
    std::vector<int64_t> block_ptr;
    std::vector<int32_t> block_pts;
    std::vector<int32_t> pairs;           // (i, j) rows
    std::vector<double> unit_jitter;      // global unit ids
    bool units_dirty = true;

    // local units
    int n_local = 0, max_T = 0;
    long total_rows = 0;
    int64_t total_mat = 0;
    std::vector<int32_t> l_global, l_m, l_rowoff;
    std::vector<int64_t> l_matoff;
    double work_flops = 0, work_fill_bytes = 0;

    // device state
    DevBuf<double> d_X, d_Y, d_out;
    // tables: views into the single staged table buffer d_tab (see rebuild_units)
    template <typename T> struct View { T *p = nullptr; void release() { p = nullptr; } };
    View<int32_t> d_m, d_rowoff, d_upt, d_slot_row, d_ids;
    DevBuf<int32_t> d_row_unit;           // filled on the device (k_row_unit)
    std::vector<int32_t> w_upt, w_slot_row, w_pcnt;   // rebuild_units scratch (kept: no allocation per re-blocking)
    std::vector<int64_t> w_slot_ptr;
    View<int64_t> d_matoff, d_slot_ptr;
    View<double> d_weight, d_jitter;
    DevBuf<char> d_tab;
    PinBuf<char> h_tab;
    DevBuf<int32_t> d_info;
    // device re-blocking (gprf_set_centers / gprf_assign_blocks)
    DevBuf<double> d_cs, d_c2;            // centres as structure of arrays [dx][nc] and their squared norms
    DevBuf<int32_t> d_assign, d_changed;  // current block of every point; "somebody moved" flag
    PinBuf<int32_t> h_assign, h_changed;
    int last_stop_after = 6;              // stage the last gprf_debug_run stopped after
    int n_centers = 0;
    // ... or through a split tree (gprf_set_split_tree): node arrays, leaf -> block id
    DevBuf<double> d_tvec, d_tcenter, d_tsplit;
    DevBuf<int32_t> d_tleft, d_tright, d_tleaf;
    int tree_nodes = 0, tree_dim = 0, tree_wrap = 0;
    bool assign_valid = false;            // d_assign holds the partition the unit tables were built from
    DevBuf<double> d_K, d_U, d_W, d_V, d_Xu, d_Yu, d_Z, d_At, d_gXu, d_logdet, d_zzpart, d_gcpart, d_rowpart, d_colpart, d_dbg;
    PinBuf<double> h_X, h_out;
    PinBuf<int32_t> h_info;

    // stream groups: the local units are dealt round-robin (by descending cost) into n_groups sets; each set's
    // fill->potrf->solve->at->grad chain runs on its own stream so that the latency-bound factorisation
    // of one set overlaps the throughput-bound stages of another; joined before the assembly
    static constexpr int MAX_GROUPS = 8;
    int n_groups = 1;
    hipStream_t gstream[MAX_GROUPS] = {};
    hipEvent_t gev_start = nullptr, gev_done[MAX_GROUPS] = {};
    bool groups_ready = false;
    int group_begin[MAX_GROUPS + 1] = {};   // ranges into d_ids

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
    hipEvent_t ev_tables = nullptr;   // recorded on the context stream after the table upload + Y gather
};

namespace {

int fail(gprf_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}

#define HIP_TRY(c, expr)                                                                           \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail((c), GPRF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));    \
    } while (0)

inline int pad16(int m) { return (m + 15) & ~15; }

UnitTab make_tab(gprf_ctx *c) {
    UnitTab t;
    t.m = c->d_m.p;
    t.row_off = c->d_rowoff.p;
    t.mat_off = c->d_matoff.p;
    t.weight = c->d_weight.p;
    t.jitter = c->d_jitter.p;
    t.upt = c->d_upt.p;
    t.row_unit = c->d_row_unit.p;
    t.n_units = c->n_local;
    t.max_T = c->max_T;
    t.ids = c->d_ids.p;
    t.n_ids = c->n_local;
    return t;
}

Pools make_pools(gprf_ctx *c) {
    Pools p;
    p.K = c->d_K.p; p.U = c->d_U.p; p.W = c->d_W.p; p.V = c->d_V.p; p.Xu = c->d_Xu.p; p.Yu = c->d_Yu.p; p.Z = c->d_Z.p;
    p.At = c->d_At.p; p.gXu = c->d_gXu.p; p.logdet = c->d_logdet.p; p.zzpart = c->d_zzpart.p;
    p.gcpart = c->d_gcpart.p; p.info = c->d_info.p; p.rowpart = c->d_rowpart.p; p.colpart = c->d_colpart.p; p.dbg = c->d_dbg.p;
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

// (Re)build the local unit tables after blocks / neighbours / shard / jitter changed.
// Units: blocks 0..n_blocks-1 (gprf.py:236), then pairs in the caller's order (gprf.py:239).
int rebuild_units(gprf_ctx *c) {
    static const bool rb_timing = [] { const char *e = getenv("GPRF_REBUILD_TIMING"); return e && e[0] == '1'; }();
    auto rb_now = [] { return std::chrono::steady_clock::now(); };
    auto rb_t0 = rb_now();
    auto rb_lap = [&](const char *what) {
        if (!rb_timing) return;
        auto t = rb_now();
        fprintf(stderr, "[rebuild] %-18s %7.1f us\n", what, std::chrono::duration<double, std::micro>(t - rb_t0).count());
        rb_t0 = t;
    };
    const int nb = c->n_blocks, np = c->n_pairs;
    const int nu = nb + np;
    std::vector<int32_t> um(nu);
    std::vector<int> deg(nb, 0);
    for (int b = 0; b < nb; ++b) um[b] = (int)(c->block_ptr[b + 1] - c->block_ptr[b]);
    for (int q = 0; q < np; ++q) {
        int i = c->pairs[2 * q], j = c->pairs[2 * q + 1];
        if (i < 0 || i >= nb || j < 0 || j >= nb || i == j)
            return fail(c, GPRF_ERR_ARG, "neighbor pair refers to a block out of range");
        um[nb + q] = um[i] + um[j];
        deg[i]++;
        deg[j]++;
    }
    for (int u = 0; u < nu; ++u)
        if (um[u] > GPRF_MAX_UNIT) {
            char buf[160];
            snprintf(buf, sizeof buf, "unit %d has %d points; the kernels accept at most %d per unit", u, um[u],
                     GPRF_MAX_UNIT);
            return fail(c, GPRF_ERR_ARG, buf);
        }
    // shard: longest-processing-time-first over cost m^3 + 4 m^2 dy
    std::vector<int> owner(nu, 0);
    if (c->world > 1 && nu > 0) gprf_partition_units(nu, um.data(), c->dy, c->world, owner.data());
    c->l_global.clear(); c->l_m.clear(); c->l_rowoff.clear(); c->l_matoff.clear();
    std::vector<double> weight, jitter;
    long rows = 0;
    int64_t mat = 0;
    int maxT = 0;
    double flops = 0, fbytes = 0;
    for (int u = 0; u < nu; ++u) {
        if (owner[u] != c->rank) continue;
        int m = um[u], mp = pad16(m);
        c->l_global.push_back(u);
        c->l_m.push_back(m);
        c->l_rowoff.push_back((int32_t)rows);
        c->l_matoff.push_back(mat);
        weight.push_back(u < nb ? (double)(1 - deg[u]) : 1.0);
        jitter.push_back((size_t)u < c->unit_jitter.size() ? c->unit_jitter[u] : 0.0);
        rows += mp;
        mat += (int64_t)mp * mp;
        maxT = std::max(maxT, mp / 16);
        flops += (double)m * m * m + 4.0 * m * m * c->dy;
        fbytes += 8.0 * m * m;
    }
    const int nl = (int)c->l_global.size();
    c->n_local = nl;
    c->max_T = maxT;
    c->total_rows = rows;
    c->total_mat = mat;
    c->work_flops = flops;
    c->work_fill_bytes = fbytes;

    rb_lap("units + shard");
    // unit row -> point table and the point -> slots CSR for the deterministic gather (gprf.py:258-273)
    // Every point sits in exactly one block, so its slots are its block's local units in ascending unit order: counts
    // and ranks are kept per BLOCK, the row tables are block-wise copies, and the only per-row work left is the scatter
    // of the slot rows.  (Scratch vectors live in the context: no allocation per re-blocking.)
    std::vector<int32_t> &upt = c->w_upt;
    std::vector<int64_t> &slot_ptr = c->w_slot_ptr;
    std::vector<int32_t> &slot_row = c->w_slot_row;
    std::vector<int32_t> &pcnt = c->w_pcnt;
    upt.resize((size_t)rows);
    std::vector<int32_t> bcnt((size_t)nb, 0), brank((size_t)nb, 0);
    for (int l = 0; l < nl; ++l) {
        int u = c->l_global[l];
        if (u < nb) bcnt[u]++;
        else { bcnt[c->pairs[2 * (u - nb)]]++; bcnt[c->pairs[2 * (u - nb) + 1]]++; }
    }
    pcnt.assign((size_t)c->n, 0);
    for (int b = 0; b < nb; ++b)
        for (int64_t k = c->block_ptr[b]; k < c->block_ptr[b + 1]; ++k) pcnt[c->block_pts[k]] += bcnt[b];
    slot_ptr.resize((size_t)c->n + 1);
    slot_ptr[0] = 0;
    for (int p = 0; p < c->n; ++p) slot_ptr[p + 1] = slot_ptr[p] + pcnt[p];
    slot_row.resize((size_t)slot_ptr[c->n]);
    for (int l = 0; l < nl; ++l) {
        int u = c->l_global[l];
        int32_t *dst = upt.data() + c->l_rowoff[l];
        auto place = [&](int b, int off) {
            int64_t s = c->block_ptr[b], e = c->block_ptr[b + 1];
            const int32_t *pts = c->block_pts.data() + s;
            int cnt = (int)(e - s);
            if (cnt) memcpy(dst + off, pts, (size_t)cnt * sizeof(int32_t));
            int rk = brank[b]++;
            int32_t base = c->l_rowoff[l] + off;
            for (int k = 0; k < cnt; ++k) slot_row[slot_ptr[pts[k]] + rk] = base + k;
            return cnt;
        };
        int mfill;
        if (u < nb) {
            mfill = place(u, 0);
        } else {
            int ni = place(c->pairs[2 * (u - nb)], 0);
            mfill = ni + place(c->pairs[2 * (u - nb) + 1], ni);
        }
        int mp = pad16(c->l_m[l]);
        for (int r = mfill; r < mp; ++r) dst[r] = -1;
    }
    rb_lap("upt + slots");
    // unit id lists: group g = every n_groups-th unit in descending-cost order (group 0 when n_groups == 1
    // is simply all units, largest first, so that the long factorizations start first)
    std::vector<int32_t> ids(nl);
    {
        std::vector<int> order(nl);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return c->l_m[a] > c->l_m[b]; });
        int G = std::max(1, std::min(c->n_groups, gprf_ctx::MAX_GROUPS));
        // groups are CONTIGUOUS slices of the descending-size order, so that a group of smaller units leaves
        // its (latency-bound, step-count-proportional) factorisation early and its later stages overlap the
        // factorisation tail of the big units; group g gets an equal share of the summed cost
        for (int k = 0; k < nl; ++k) ids[k] = order[k];
        double total = 0.0, run = 0.0;
        auto cost = [&](int l) { double mm = c->l_m[l]; return mm * mm * mm + 4.0 * mm * mm * c->dy; };
        for (int k = 0; k < nl; ++k) total += cost(order[k]);
        int g = 0;
        c->group_begin[0] = 0;
        for (int k = 0; k < nl; ++k) {
            run += cost(order[k]);
            if (g + 1 < G && run >= total * (g + 1) / G) c->group_begin[++g] = k + 1;
        }
        for (int gg = g + 1; gg <= gprf_ctx::MAX_GROUPS; ++gg) c->group_begin[gg] = nl;
    }

    rb_lap("order + groups");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    size_t nl1 = (size_t)std::max(nl, 1);
    HIP_TRY(c, c->d_logdet.reserve(nl1));
    HIP_TRY(c, c->d_zzpart.reserve(nl1 * 4));
    HIP_TRY(c, c->d_info.reserve(nl1));
    HIP_TRY(c, c->d_dbg.reserve(nl1 * 8));
    HIP_TRY(c, c->h_info.reserve(nl1));
    size_t tbm = (size_t)std::max((maxT + 3) / 4, 1);      // 64-point blocks per edge of the largest local unit
    HIP_TRY(c, c->d_gcpart.reserve(nl1 * (tbm * (tbm + 1) / 2) * GC_SLOTS));
    HIP_TRY(c, c->d_rowpart.reserve((size_t)rows * tbm * XPAD + 1));
    HIP_TRY(c, c->d_colpart.reserve((size_t)rows * tbm * XPAD + 1));
    HIP_TRY(c, c->d_K.reserve((size_t)mat + GPRF_POOL_SLACK));
    HIP_TRY(c, c->d_U.reserve((size_t)mat + GPRF_POOL_SLACK));
    HIP_TRY(c, c->d_W.reserve((size_t)mat + GPRF_POOL_SLACK));
    HIP_TRY(c, c->d_V.reserve((size_t)rows * 16 + 1));
    HIP_TRY(c, c->d_Xu.reserve((size_t)rows * 8 + 1));      // XPAD, or 8 for the lld record
    HIP_TRY(c, c->d_Yu.reserve((size_t)rows * YPAD + 1));
    HIP_TRY(c, c->d_Z.reserve((size_t)rows * YPAD + 1));
    HIP_TRY(c, c->d_At.reserve((size_t)rows * YPAD + 1));
    HIP_TRY(c, c->d_gXu.reserve((size_t)rows * XPAD + 1));
    HIP_TRY(c, c->d_row_unit.reserve((size_t)rows + 1));

    // ONE staged upload: every table is packed (256-byte aligned) into one pinned buffer and copied with a
    // single asynchronous H2D on the context stream (ten small synchronous copies cost ~0.25 ms per re-blocking)
    rb_lap("reserve");
    HIP_TRY(c, hipStreamSynchronize(s));     // the previous staging buffer / tables may still be in use
    rb_lap("stream sync");
    {
        struct Seg { const void *src; size_t bytes; void **dst; };
        Seg segs[] = {
            {c->l_m.data(), (size_t)nl * sizeof(int32_t), (void **)&c->d_m.p},
            {ids.data(), (size_t)nl * sizeof(int32_t), (void **)&c->d_ids.p},
            {c->l_rowoff.data(), (size_t)nl * sizeof(int32_t), (void **)&c->d_rowoff.p},
            {c->l_matoff.data(), (size_t)nl * sizeof(int64_t), (void **)&c->d_matoff.p},
            {weight.data(), (size_t)nl * sizeof(double), (void **)&c->d_weight.p},
            {jitter.data(), (size_t)nl * sizeof(double), (void **)&c->d_jitter.p},
            {upt.data(), (size_t)rows * sizeof(int32_t), (void **)&c->d_upt.p},
            {slot_ptr.data(), slot_ptr.size() * sizeof(int64_t), (void **)&c->d_slot_ptr.p},
            {slot_row.data(), slot_row.size() * sizeof(int32_t), (void **)&c->d_slot_row.p},
        };
        size_t total = 0;
        for (auto &sg : segs) total += (sg.bytes + 255) & ~(size_t)255;
        total += 256;
        HIP_TRY(c, c->d_tab.reserve(total));
        HIP_TRY(c, c->h_tab.reserve(total));
        size_t off = 0;
        for (auto &sg : segs) {
            if (sg.bytes) memcpy(c->h_tab.p + off, sg.src, sg.bytes);
            *sg.dst = (void *)(c->d_tab.p + off);
            off += (sg.bytes + 255) & ~(size_t)255;
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_tab.p, c->h_tab.p, off, hipMemcpyHostToDevice, s));
    }
    rb_lap("pack + H2D enqueue");
    // Y rows of every unit (Y never changes; membership does)
    UnitTab ut = make_tab(c);
    Pools pl = make_pools(c);
    launch_row_unit(ut, c->d_row_unit.p, s);
    launch_gather_y(ut, pl, c->d_Y.p, c->dy, (int)rows, s);
    HIP_TRY(c, hipGetLastError());
    // an evaluation enqueued on a caller's stream waits for this event instead of a host sync (enqueue_eval)
    if (!c->ev_tables) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_tables, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->ev_tables, s));
    rb_lap("launches");
    c->units_dirty = false;
    return GPRF_OK;
}

int check_ready(gprf_ctx *c) {
    if (!c) return GPRF_ERR_ARG;
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

// enqueue one evaluation on stream s reading d_X, writing d_out; stop_after < 6 truncates (debug)
int enqueue_eval(gprf_ctx *c, const double *d_X, int want_gx, int want_gc, double *d_out, hipStream_t s,
                 int stop_after) {
    if (c->units_dirty) {
        int rc = rebuild_units(c);
        if (rc != GPRF_OK) return rc;
    }
    if (s != c->stream && c->ev_tables) HIP_TRY(c, hipStreamWaitEvent(s, c->ev_tables, 0));
    UnitTab ut = make_tab(c);
    Pools pl = make_pools(c);
    KParams kp = make_kparams(c);
    AssembleTab at{c->d_slot_ptr.p, c->d_slot_row.p};
    bool tm = c->timing;
    if (tm && !c->ev_valid) {
        for (int r = 0; r < gprf_ctx::RING; ++r)
            for (int i = 0; i <= GPRF_N_STAGES; ++i) HIP_TRY(c, hipEventCreate(&c->ev[r][i]));
        c->ev_valid = true;
    }
    int slot = (int)(c->n_timed % gprf_ctx::RING);
    if (tm) {
        if (c->slot_pending[slot]) {
            int rc = fold_slot(c, slot);
            if (rc != GPRF_OK) return rc;
        }
        c->slot_pending[slot] = true;
        c->n_timed++;
    }
    int stage = 0;
    auto mark = [&]() { if (tm) (void)hipEventRecord(c->ev[slot][stage], s); ++stage; };
    int G = std::max(1, std::min(c->n_groups, gprf_ctx::MAX_GROUPS));
    bool do_grad = stop_after >= 4 && (want_gx || want_gc);
    if (G == 1 || tm) {
        mark();
        launch_gather_x(c->dist_id, ut, pl, d_X, c->dx, (int)c->total_rows, s);
        mark();
        // (the K pool exists only when somebody reads it: a fill-only debug run, the generic Cholesky, big units)
        bool gen = stop_after >= 1 && potrf_generates_K(c->dist_id, c->kern_id, ut);
        if (!gen) launch_fill(c->dist_id, c->kern_id, ut, pl, kp, s);
        mark();
        if (stop_after >= 1) launch_potrf(ut, pl, kp, gen, s);
        mark();
        if (stop_after >= 2) launch_solve(ut, pl, s);
        mark();
        if (stop_after >= 3) launch_at(ut, pl, s);
        mark();
        if (do_grad) {
            launch_grad(c->dist_id, c->kern_id, ut, pl, kp, want_gc, (int)c->total_rows, !gen, s);
            launch_gx_finalize(ut, pl, (int)c->total_rows, s);
        }
        mark();
        if (stop_after >= 5) launch_assemble(ut, pl, at, kp, c->n, want_gx, want_gc, d_out, s);
        mark();
    } else {
        if (!c->groups_ready) {
            HIP_TRY(c, hipEventCreateWithFlags(&c->gev_start, hipEventDisableTiming));
            for (int g = 0; g < gprf_ctx::MAX_GROUPS; ++g) {
                HIP_TRY(c, hipStreamCreateWithFlags(&c->gstream[g], hipStreamNonBlocking));
                HIP_TRY(c, hipEventCreateWithFlags(&c->gev_done[g], hipEventDisableTiming));
            }
            c->groups_ready = true;
        }
        launch_gather_x(c->dist_id, ut, pl, d_X, c->dx, (int)c->total_rows, s);
        HIP_TRY(c, hipEventRecord(c->gev_start, s));
        for (int g = 0; g < G; ++g) {
            UnitTab ug = ut;
            ug.ids = c->d_ids.p + c->group_begin[g];
            ug.n_ids = c->group_begin[g + 1] - c->group_begin[g];
            hipStream_t gs = c->gstream[g];
            HIP_TRY(c, hipStreamWaitEvent(gs, c->gev_start, 0));
            bool gen = stop_after >= 1 && potrf_generates_K(c->dist_id, c->kern_id, ug);
            if (!gen) launch_fill(c->dist_id, c->kern_id, ug, pl, kp, gs);
            if (stop_after >= 1) launch_potrf(ug, pl, kp, gen, gs);
            if (stop_after >= 2) launch_solve(ug, pl, gs);
            if (stop_after >= 3) launch_at(ug, pl, gs);
            if (do_grad) launch_grad(c->dist_id, c->kern_id, ug, pl, kp, want_gc, (int)c->total_rows, !gen, gs);
            HIP_TRY(c, hipEventRecord(c->gev_done[g], gs));
            HIP_TRY(c, hipStreamWaitEvent(s, c->gev_done[g], 0));
        }
        if (do_grad) launch_gx_finalize(ut, pl, (int)c->total_rows, s);
        if (stop_after >= 5) launch_assemble(ut, pl, at, kp, c->n, want_gx, want_gc, d_out, s);
    }
    HIP_TRY(c, hipGetLastError());
    // unit status -> pinned host
    if (c->n_local > 0)
        HIP_TRY(c, hipMemcpyAsync(c->h_info.p, c->d_info.p, c->n_local * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    c->eval_pending = true;
    return GPRF_OK;
}

int finish_eval(gprf_ctx *c, hipStream_t s, int32_t *first_bad_unit) {
    HIP_TRY(c, hipStreamSynchronize(s));
    c->eval_pending = false;
    int bad = -1;
    for (int l = 0; l < c->n_local; ++l)
        if (c->h_info.p[l] != 0) { bad = c->l_global[l]; break; }
    if (first_bad_unit) *first_bad_unit = bad;
    if (bad >= 0) {
        char buf[128];
        snprintf(buf, sizeof buf, "unit %d: kernel matrix not positive definite", bad);
        c->err = buf;
        return GPRF_NOT_PD;
    }
    return GPRF_OK;
}

}  // namespace

extern "C" {

int gprf_create(gprf_ctx **out, int32_t n, int32_t dx, int32_t dy, int32_t dist_id, int32_t kern_id,
                int32_t device) {
    if (!out) return GPRF_ERR_ARG;
    *out = nullptr;
    if (n < 0 || dy < 1 || dy > YPAD) return GPRF_ERR_ARG;
    bool se = (dist_id == GPRF_DIST_EUCLIDEAN && kern_id == GPRF_KERN_SE);
    bool mt = (dist_id == GPRF_DIST_LLD && kern_id == GPRF_KERN_MATERN32);
    if (!se && !mt) return GPRF_ERR_ARG;  // the two combinations the reference's callers use
    if (se && (dx < 1 || dx > 3)) return GPRF_ERR_ARG;
    if (mt && dx != 3) return GPRF_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return GPRF_ERR_HIP;
    gprf_ctx *c = new gprf_ctx();
    c->n = n; c->dx = dx; c->dy = dy; c->dist_id = dist_id; c->kern_id = kern_id; c->device = device;
    c->ndfn = se ? dx : 2;
    c->ncov = 2 + c->ndfn;
    if (const char *g = getenv("GPRF_GROUPS")) c->n_groups = std::max(1, std::min(atoi(g), (int)gprf_ctx::MAX_GROUPS));
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) {
        delete c;
        return GPRF_ERR_HIP;
    }
    size_t nout = 1 + (size_t)n * dx + c->ncov;
    if (c->d_X.reserve((size_t)n * dx + 1, 1.0) != hipSuccess || c->d_Y.reserve((size_t)n * dy + 1, 1.0) != hipSuccess ||
        c->d_out.reserve(nout, 1.0) != hipSuccess || c->h_X.reserve((size_t)n * dx + 1) != hipSuccess ||
        c->h_out.reserve(nout) != hipSuccess) {
        gprf_destroy(c);
        return GPRF_ERR_HIP;
    }
    *out = c;
    return GPRF_OK;
}

int gprf_destroy(gprf_ctx *c) {
    if (!c) return GPRF_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->d_X.release(); c->d_Y.release(); c->d_out.release(); c->d_m.release(); c->d_rowoff.release();
    c->d_upt.release(); c->d_slot_row.release(); c->d_info.release(); c->d_matoff.release();
    c->d_slot_ptr.release(); c->d_weight.release(); c->d_jitter.release();
    c->d_K.release(); c->d_U.release(); c->d_W.release(); c->d_V.release(); c->d_Xu.release(); c->d_Yu.release();
    c->d_Z.release(); c->d_At.release(); c->d_gXu.release(); c->d_logdet.release(); c->d_zzpart.release();
    c->d_gcpart.release(); c->d_rowpart.release(); c->d_colpart.release(); c->d_dbg.release(); c->d_row_unit.release(); c->h_X.release(); c->h_out.release(); c->h_info.release();
    c->d_cs.release(); c->d_c2.release(); c->d_assign.release(); c->d_changed.release(); c->h_assign.release(); c->h_changed.release();
    if (c->ev_valid)
        for (int r = 0; r < gprf_ctx::RING; ++r)
            for (int i = 0; i <= GPRF_N_STAGES; ++i) (void)hipEventDestroy(c->ev[r][i]);
    if (c->groups_ready) {
        (void)hipEventDestroy(c->gev_start);
        for (int g = 0; g < gprf_ctx::MAX_GROUPS; ++g) {
            (void)hipStreamSynchronize(c->gstream[g]);
            (void)hipStreamDestroy(c->gstream[g]);
            (void)hipEventDestroy(c->gev_done[g]);
        }
    }
    c->d_tab.release(); c->h_tab.release();
    if (c->ev_tables) (void)hipEventDestroy(c->ev_tables);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return GPRF_OK;
}

const char *gprf_last_error(const gprf_ctx *c) { return c ? c->err.c_str() : "null context"; }

int gprf_set_Y(gprf_ctx *c, const double *Y) {
    if (!c || !Y) return GPRF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->d_Y.p, Y, (size_t)c->n * c->dy * sizeof(double), hipMemcpyHostToDevice));
    c->have_Y = true;
    c->units_dirty = true;  // Yu must be re-gathered
    return GPRF_OK;
}

int gprf_set_theta(gprf_ctx *c, const double *theta, int32_t ntheta) {
    if (!c || !theta) return GPRF_ERR_ARG;
    if (ntheta != c->ncov) return fail(c, GPRF_ERR_ARG, "theta must be [noise_var, signal_var, dfn_params...]");
    for (int i = 0; i < ntheta; ++i)
        if (!std::isfinite(theta[i])) return fail(c, GPRF_ERR_ARG, "non-finite hyper-parameter");
    c->theta.assign(theta, theta + ntheta);
    c->have_theta = true;
    return GPRF_OK;
}

int gprf_set_blocks(gprf_ctx *c, int32_t n_blocks, const int64_t *block_ptr, const int32_t *point_idx) {
    if (!c || n_blocks < 0 || !block_ptr) return GPRF_ERR_ARG;
    if (block_ptr[0] != 0) return fail(c, GPRF_ERR_ARG, "block_ptr[0] must be 0");
    for (int b = 0; b < n_blocks; ++b)
        if (block_ptr[b + 1] < block_ptr[b]) return fail(c, GPRF_ERR_ARG, "block_ptr must be non-decreasing");
    int64_t tot = block_ptr[n_blocks];
    if (tot > 0 && !point_idx) return GPRF_ERR_ARG;
    c->assign_valid = false;     // (gprf_assign_blocks sets it again after installing its own partition)
    for (int64_t k = 0; k < tot; ++k)
        if (point_idx[k] < 0 || point_idx[k] >= c->n) return fail(c, GPRF_ERR_ARG, "point index out of range");
    if (n_blocks != c->n_blocks) {
        // pairs refer to block ids; a different block count invalidates them unless re-set
        if (c->n_pairs > 0) {
            for (int q = 0; q < 2 * c->n_pairs; ++q)
                if (c->pairs[q] >= n_blocks) return fail(c, GPRF_ERR_STATE, "existing neighbor pairs exceed the new block count");
        }
    }
    c->n_blocks = n_blocks;
    c->block_ptr.assign(block_ptr, block_ptr + n_blocks + 1);
    c->block_pts.assign(point_idx, point_idx + tot);
    c->have_blocks = true;
    c->units_dirty = true;
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
    HIP_TRY(c, c->d_assign.reserve((size_t)c->n + 1));
    HIP_TRY(c, c->d_changed.reserve(1));
    HIP_TRY(c, c->h_assign.reserve((size_t)c->n + 1));
    HIP_TRY(c, c->h_changed.reserve(1));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->d_cs.p, cs.data(), cs.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_c2.p, c2.data(), c2.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemset(c->d_assign.p, 0xff, (size_t)c->n * sizeof(int32_t)));   // -1: everybody "moves" first time
    c->n_centers = nc;
    c->tree_nodes = 0;
    c->assign_valid = false;
    return GPRF_OK;
}

int gprf_set_split_tree(gprf_ctx *c, int32_t n_nodes, int32_t dim, int32_t lon_wrap, const double *vec,
                        const double *center, const double *split, const int32_t *left, const int32_t *right,
                        const int32_t *leaf_block) {
    if (!c || n_nodes < 1 || dim < 1 || !vec || !center || !split || !left || !right || !leaf_block) return GPRF_ERR_ARG;
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
    HIP_TRY(c, c->d_assign.reserve((size_t)c->n + 1));
    HIP_TRY(c, c->d_changed.reserve(1));
    HIP_TRY(c, c->h_assign.reserve((size_t)c->n + 1));
    HIP_TRY(c, c->h_changed.reserve(1));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->d_tvec.p, vec, nd * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_tcenter.p, center, nd * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_tsplit.p, split, (size_t)n_nodes * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_tleft.p, left, (size_t)n_nodes * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_tright.p, right, (size_t)n_nodes * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_tleaf.p, leaf_block, (size_t)n_nodes * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemset(c->d_assign.p, 0xff, (size_t)c->n * sizeof(int32_t)));
    c->tree_nodes = n_nodes; c->tree_dim = dim; c->tree_wrap = lon_wrap ? 1 : 0;
    c->n_centers = n_leaves;
    c->assign_valid = false;
    return GPRF_OK;
}

int gprf_assign_blocks(gprf_ctx *c, const double *X, int32_t *changed, int32_t *block_of_out) {
    if (!c || !X || !changed) return GPRF_ERR_ARG;
    if (c->n_centers < 1) return fail(c, GPRF_ERR_STATE, "gprf_set_centers or gprf_set_split_tree first");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    size_t nx = (size_t)c->n * c->dx;
    if (!c->assign_valid) HIP_TRY(c, hipMemsetAsync(c->d_assign.p, 0xff, (size_t)c->n * sizeof(int32_t), s));
    HIP_TRY(c, hipMemsetAsync(c->d_changed.p, 0, sizeof(int32_t), s));
    memcpy(c->h_X.p, X, nx * sizeof(double));
    HIP_TRY(c, hipMemcpyAsync(c->d_X.p, c->h_X.p, nx * sizeof(double), hipMemcpyHostToDevice, s));
    if (c->tree_nodes > 0)
        launch_route(c->d_X.p, c->n, c->dx, c->tree_dim, c->tree_wrap, c->d_tvec.p, c->d_tcenter.p, c->d_tsplit.p,
                     c->d_tleft.p, c->d_tright.p, c->d_tleaf.p, c->d_assign.p, c->d_changed.p, s);
    else
        launch_assign(c->d_X.p, c->n, c->dx, c->d_cs.p, c->d_c2.p, c->n_centers, c->d_assign.p, c->d_changed.p, s);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->h_changed.p, c->d_changed.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    *changed = c->h_changed.p[0] ? 1 : 0;
    if (!*changed) return GPRF_OK;
    // somebody moved: bring the partition back and rebuild the unit tables from it, like a host re-blocking
    HIP_TRY(c, hipMemcpy(c->h_assign.p, c->d_assign.p, (size_t)c->n * sizeof(int32_t), hipMemcpyDeviceToHost));
    int rc = gprf_set_block_assignment(c, c->n_centers, c->h_assign.p);
    if (rc != GPRF_OK) { c->assign_valid = false; return rc; }
    c->assign_valid = true;
    if (block_of_out) memcpy(block_of_out, c->h_assign.p, (size_t)c->n * sizeof(int32_t));
    return GPRF_OK;
}

int gprf_set_neighbors(gprf_ctx *c, int32_t n_pairs, const int32_t *pairs_ij) {
    if (!c || n_pairs < 0 || (n_pairs > 0 && !pairs_ij)) return GPRF_ERR_ARG;
    c->n_pairs = n_pairs;
    c->pairs.assign(pairs_ij, pairs_ij + 2 * (size_t)n_pairs);
    c->units_dirty = true;
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
    c->rank = rank;
    c->world = world;
    c->units_dirty = true;
    return GPRF_OK;
}

int gprf_set_unit_jitter(gprf_ctx *c, int32_t n_units, const double *jitter) {
    if (!c) return GPRF_ERR_ARG;
    if (!jitter) {
        // the drivers clear the jitter before every evaluation (jitchol is stateless): clearing what is already
        // clear must not cost a rebuild of the unit tables
        if (c->unit_jitter.empty()) return GPRF_OK;
        c->unit_jitter.clear();
    } else {
        if (n_units < 0) return GPRF_ERR_ARG;
        if ((size_t)n_units == c->unit_jitter.size() && std::equal(jitter, jitter + n_units, c->unit_jitter.begin()))
            return GPRF_OK;
        c->unit_jitter.assign(jitter, jitter + n_units);
    }
    c->units_dirty = true;
    return GPRF_OK;
}

int gprf_eval_device(gprf_ctx *c, const double *d_X, int32_t want_gradX, int32_t want_gradC, double *d_out,
                     void *stream) {
    int rc = check_ready(c);
    if (rc != GPRF_OK) return rc;
    if (!d_X || !d_out) return GPRF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    // table uploads in rebuild_units run on the context stream synchronously, so any stream may follow
    return enqueue_eval(c, d_X, want_gradX, want_gradC, d_out, s, 6);
}

int gprf_eval_status(gprf_ctx *c, int32_t *first_bad_unit) {
    if (!c) return GPRF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    return finish_eval(c, c->stream, first_bad_unit);
}

int gprf_eval(gprf_ctx *c, const double *X, int32_t want_gradX, int32_t want_gradC, double *ll_out,
              double *gradX_out, double *gradC_out, int32_t *first_bad_unit) {
    int rc = check_ready(c);
    if (rc != GPRF_OK) return rc;
    if (!X || !ll_out || (want_gradX && !gradX_out) || (want_gradC && !gradC_out)) return GPRF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    size_t nx = (size_t)c->n * c->dx;
    size_t nout = 1 + nx + c->ncov;
    memcpy(c->h_X.p, X, nx * sizeof(double));
    HIP_TRY(c, hipMemcpyAsync(c->d_X.p, c->h_X.p, nx * sizeof(double), hipMemcpyHostToDevice, s));
    rc = enqueue_eval(c, c->d_X.p, want_gradX, want_gradC, c->d_out.p, s, 6);
    if (rc != GPRF_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->h_out.p, c->d_out.p, nout * sizeof(double), hipMemcpyDeviceToHost, s));
    rc = finish_eval(c, s, first_bad_unit);
    if (rc != GPRF_OK) return rc;
    *ll_out = c->h_out.p[0];
    if (want_gradX) memcpy(gradX_out, c->h_out.p + 1, nx * sizeof(double));
    if (want_gradC) memcpy(gradC_out, c->h_out.p + 1 + nx, c->ncov * sizeof(double));
    return GPRF_OK;
}

int gprf_num_units(const gprf_ctx *c, int32_t *n_total, int32_t *n_local) {
    if (!c) return GPRF_ERR_ARG;
    if (n_total) *n_total = c->n_blocks + c->n_pairs;
    if (n_local) *n_local = c->units_dirty ? -1 : c->n_local;
    return GPRF_OK;
}

int gprf_work_estimate(gprf_ctx *c, double *flops, double *fill_bytes) {
    int rc = check_ready(c);
    if (rc != GPRF_OK) return rc;
    if (c->units_dirty) {
        HIP_TRY(c, hipSetDevice(c->device));
        rc = rebuild_units(c);
        if (rc != GPRF_OK) return rc;
    }
    if (flops) *flops = c->work_flops;
    if (fill_bytes) *fill_bytes = c->work_fill_bytes;
    return GPRF_OK;
}

int gprf_set_timing(gprf_ctx *c, int32_t enable) {
    if (!c) return GPRF_ERR_ARG;
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
    int rc = check_ready(c);
    if (rc != GPRF_OK) return rc;
    if (!X) return GPRF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    size_t nx = (size_t)c->n * c->dx;
    memcpy(c->h_X.p, X, nx * sizeof(double));
    HIP_TRY(c, hipMemcpyAsync(c->d_X.p, c->h_X.p, nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->last_stop_after = stop_after;
    rc = enqueue_eval(c, c->d_X.p, 1, 1, c->d_out.p, c->stream, stop_after);
    if (rc != GPRF_OK) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->eval_pending = false;
    return GPRF_OK;
}

int gprf_debug_unit_shape(gprf_ctx *c, int32_t l, int32_t *m, int32_t *mp, int32_t *global_unit) {
    if (!c || c->units_dirty || l < 0 || l >= c->n_local) return GPRF_ERR_ARG;
    if (m) *m = c->l_m[l];
    if (mp) *mp = pad16(c->l_m[l]);
    if (global_unit) *global_unit = c->l_global[l];
    return GPRF_OK;
}

int gprf_debug_fetch(gprf_ctx *c, int32_t l, int32_t what, double *out, int64_t out_len) {
    if (!c || !out || c->units_dirty || l < 0 || l >= c->n_local) return GPRF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
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
        case 4: src = c->d_gXu.p + roff * XPAD; len = mp * XPAD; break;
        case 6: src = c->d_dbg.p + (size_t)l * 8; len = 8; break;
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
            HIP_TRY(c, hipMemcpy(&info, c->d_info.p + l, sizeof(int32_t), hipMemcpyDeviceToHost));
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
        default: return GPRF_ERR_ARG;
    }
    if (out_len < len) return GPRF_ERR_ARG;
    if (len > 0) HIP_TRY(c, hipMemcpy(out, src, len * sizeof(double), hipMemcpyDeviceToHost));
    return GPRF_OK;
}

}  // extern "C"
