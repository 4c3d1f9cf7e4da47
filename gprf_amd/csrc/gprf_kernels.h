// gprf_kernels.h — shared declarations between the HIP kernels (gprf_kernels.hip) and the C-ABI host
// layer (gprf_capi.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gprf {

constexpr int TILE = 16;          // MFMA f64 16x16x4 tile edge
constexpr int MAX_MP = 1024;       // largest padded unit (GPRF_MAX_UNIT)
constexpr int MAX_T = MAX_MP / TILE;
constexpr int YPAD = 64;          // dy padded to 4 column tiles
constexpr int XPAD = 4;           // dx padded (dx <= 3)
constexpr int GC_SLOTS = 8;       // per-(unit, column-tile) hyper-gradient partials
constexpr int MAX_TB = MAX_T / 4;   // 64-point blocks per unit edge (k_grad2's row-sum slab)

// Kernel hyper-parameters, passed by value.  theta = [nv, sv, ls...] (gprf.py:160-164).
struct KParams {
    double nv, sv;
    double ls[3];
    double inv_ls[3];   // 1 / ls[d], rounded once on the host
    int dx, ndfn, dy;
};

// Per-unit tables (device pointers), one entry per LOCAL unit.
struct UnitTab {
    const int32_t *m;        // points in the unit
    const int32_t *row_off;  // padded-row offset: sum of mp over previous units
    const int64_t *mat_off;  // element offset of the unit's mp x mp matrices in the U / W pools
    const double *weight;    // Bethe weight: 1 - deg(i) for unaries, 1 for pairs
    const double *jitter;    // extra diagonal (jitchol retry)
    const int32_t *upt;      // [total padded rows] global point index of each unit row, -1 for padding
    const int32_t *row_unit; // [total padded rows] local unit id of each padded row (filled on the device: k_row_unit)
    const int32_t *ids;      // the local unit ids this launch covers (one stream group) ...
    int n_ids;               // ... and how many
    int n_units;
    int max_T;               // max over units of mp/16
};

// spare doubles behind the last unit's matrix in the U / W / K pools (row-panel loads address whole 64-column
// chunks; the lanes beyond the unit's edge are masked off, the slack keeps even an unmasked variant in bounds)
constexpr size_t GPRF_POOL_SLACK = 512;

struct Pools {
    double *K;     // kernel matrices, row-major mp x mp per unit: k_fill writes the 64x64 blocks ti <= tj only
                   // (diagonal blocks whole); read by the Cholesky once and by k_mgrad
    double *U;     // upper Cholesky factors (same layout; the strictly-lower part is never written)
    double *W;     // U^-T (lower triangular, row-major)
    double *V;     // inverses of U's 16x16 diagonal tiles: T tiles per unit at 16*row_off
    double *Xu;    // gathered unit coordinates per padded row: XPAD doubles (euclidean) or the 8-double half-angle
                   // record of the lld distance (k_gather_x)
    double *Yu;    // gathered unit outputs, YPAD per padded row (zero padded)
    double *Z;     // U^-T Yu, YPAD per padded row
    double *At;    // (K^-1 Yu)^T : per unit YPAD x mp at YPAD*row_off
    double *gXu;   // per-unit-row gradient slab, XPAD per padded row
    double *rowpart;  // k_mgrad: per padded row, TBm x XPAD partial row sums (one per column block JB <= its block)
    double *colpart;  // k_mgrad: per padded row, TBm x XPAD partial column sums (one per row block IB >= its block)
    double *logdet;   // per unit
    double *zzpart;   // per unit x 4 : partial sums of ||Z||_F^2 per Y column block
    double *gcpart;   // per unit x TBm (TBm + 1) / 2 block pairs x GC_SLOTS,  TBm = ceil(max_T / 4)
    int32_t *info;    // per unit: 0 ok, k>0 = non-positive pivot at row k-1
    double *dbg;      // per unit x 8: in-kernel cycle stamps of diagnostic builds (GPRF_POTRF_ABLATE & 16)
};

struct AssembleTab {
    const int64_t *slot_ptr;   // [n+1]
    const int32_t *slot_row;   // padded-row index into gXu (the slot's weight is its unit's: weight[row_unit[row]])
};

void launch_route(const double *X, int n, int dx, int dim, int lon_wrap, const double *vec, const double *center,
                  const double *split, const int32_t *left, const int32_t *right, const int32_t *leaf_block,
                  int32_t *block_of, int32_t *changed, hipStream_t s);
void launch_row_unit(const UnitTab &ut, int32_t *row_unit, hipStream_t s);
void launch_gather_y(const UnitTab &ut, const Pools &p, const double *Y, int dy, int total_rows, hipStream_t s);
void launch_gather_x(int dist_id, const UnitTab &ut, const Pools &p, const double *X, int dx, int total_rows, hipStream_t s);
void launch_fill(int dist_id, int kern_id, const UnitTab &ut, const Pools &p, const KParams &kp, hipStream_t s);
bool potrf_generates_K(int dist_id, int kern_id, const UnitTab &ut);
void launch_potrf(const UnitTab &ut, const Pools &p, const KParams &kp, bool gen, hipStream_t s);
void launch_solve(const UnitTab &ut, const Pools &p, hipStream_t s);
void launch_at(const UnitTab &ut, const Pools &p, hipStream_t s);
void launch_grad(int dist_id, int kern_id, const UnitTab &ut, const Pools &p, const KParams &kp, int want_gc,
                 int total_rows, bool have_K, hipStream_t s);
void launch_assemble(const UnitTab &ut, const Pools &p, const AssembleTab &at, const KParams &kp, int n,
                     int want_gx, int want_gc, double *out, hipStream_t s);
void launch_gx_finalize(const UnitTab &ut, const Pools &p, int total_rows, hipStream_t s);
void launch_assign(const double *X, int n, int dx, const double *cs, const double *c2, int nc, int32_t *block_of,
                   int32_t *changed, hipStream_t s);

}  // namespace gprf
