/*
 * gprf_hip.h — C ABI of libgprf_hip.so: the MI355X (gfx950) implementation of the GPRF block-local
 * log-likelihood + gradient path of davmre/gprf.
 *
 * Boundary: everything from "blocks + neighbour pairs + X + Y + hypers" down to
 * "(ll, gradX, gradCov)" — i.e. what the reference computes in
 *     GPRF.llgrad            /root/reference/gprf.py:206-296
 *     GPRF.llgrad_unary      gprf.py:299-308
 *     GPRF.llgrad_joint      gprf.py:310-330
 *     GPRF.gaussian_llgrad   gprf.py:496-591
 *     GPRF.kernel/dKdx/dKdi  gprf.py:333-375   (treegp VectorTree.kernel_matrix /
 *                                                kernel_deriv_wrt_xi_row / kernel_deriv_wrt_i)
 *     pdinv / jitchol / dpotrs   /root/reference/gpy_linalg.py:77-97, 139-148, 219-240
 * is one call, gprf_eval().  The Python object above it (gprf_amd/gprf.py) keeps the reference's
 * GPRF surface; INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions: all matrices are row-major (C order) float64, exactly the numpy arrays the reference
 * holds; index arrays are int32/int64 as stated; every function returns an int status:
 *     0  GPRF_OK
 *     1  GPRF_NOT_PD     some unit's kernel matrix is not positive definite (jitchol would retry;
 *                        gpy_linalg.py:77-97) — *first_bad_unit says which; outputs are undefined
 *     2  GPRF_RETRY      (asynchronous form only, see gprf_eval_status)
 *    <0  a HIP / argument error; gprf_last_error() has the text
 * No callbacks, no global state; host pointers are borrowed for the duration of the call only.
 * One context is bound to one device and one stream; calls on one context must not overlap.
 */
#ifndef GPRF_HIP_H
#define GPRF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gprf_ctx gprf_ctx;

#define GPRF_OK 0
#define GPRF_NOT_PD 1
#define GPRF_RETRY 2      /* gprf_eval_status only: a re-partition outgrew the workspace; it has been grown — enqueue again */
#define GPRF_ERR_ARG (-1)
#define GPRF_ERR_HIP (-2)
#define GPRF_ERR_STATE (-3)

/* dfn_str / wfn_str of treegp's GPCov (gprf.py:109,163; run_seismic.py:299-301) */
#define GPRF_DIST_EUCLIDEAN 0 /* "euclidean": r^2 = sum_d ((x_d - x'_d)/l_d)^2, one l per input dim */
#define GPRF_DIST_LLD 1       /* "lld": (lon deg, lat deg, depth km), haversine km / l_0, ddepth / l_1 */
#define GPRF_KERN_SE 0        /* "se":       k = sv * exp(-r^2)   (no 1/2 factor) */
#define GPRF_KERN_MATERN32 1  /* "matern32": k = sv * (1 + sqrt3 r) exp(-sqrt3 r) */

/* Largest unit (block, or concatenated block pair) the kernels accept, in points.  Units of up to 1024 points run one
 * workgroup per unit and stage; larger ones (the reference's 9-block / 1-block / 16-block runs, gprfopt_analyze.py:195,
 * 237-238) go through a blocked multi-launch Cholesky / substitution (64 x 64 blocks) and the ordinary tiled kernels. */
#define GPRF_MAX_UNIT 16384

/* Replaces GPRF.__init__ (gprf.py:85-117) + the VectorTree construction (gprf.py:109).
 * n points, dx input dims (2 or 3), dy output columns; device = HIP device ordinal. */
int gprf_create(gprf_ctx **out, int32_t n, int32_t dx, int32_t dy, int32_t dist_id, int32_t kern_id,
                int32_t device);
/* The same object over SEVERAL devices of one node, driven from ONE host thread — the reference's drivers are one Python
 * process around scipy with the fan-out over units inside llgrad (gprfopt.py:377-422, gprf.py:218-233).  devices: n_devices
 * HIP device ordinals (an ordinal may repeat: logical members on one GPU, tests).  Member k evaluates shard (k, n_devices)
 * of the units (gprf_partition_units) on its own device and stream; X goes once into pinned host memory that every member
 * reads; every member's assembly kernel stores its partial [ll | gradX | gradC | s0 s1] into its slot in device
 * devices[0]'s memory (peer stores over xGMI), one kernel there adds the slots in member order into pinned host memory:
 * one download, one synchronisation, no inter-process collective.  Every setter below applies to all members; gprf_eval /
 * gprf_update_eval / gprf_objective evaluate over the group (the location prior is added by member 0 alone); the
 * *_device forms, gprf_set_shard and the debug hooks are refused (GPRF_ERR_STATE); timing and table-build counters
 * report member 0.
 * The slots are fine-grained (system-scope coherent) memory of devices[0].  When some member's device cannot store into
 * devices[0]'s memory (hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess refuse) the group is still created: the slots
 * then live in pinned host memory (every member stores over its own host link, the summing kernel reads them zero-copy);
 * gprf_group_info says which form is in use.  On failure *out = NULL and gprf_last_error(NULL) has the reason. */
int gprf_create_multi(gprf_ctx **out, int32_t n, int32_t dx, int32_t dy, int32_t dist_id, int32_t kern_id,
                      int32_t n_devices, const int32_t *devices);
/* A multi-device group's shape (0 members for a plain context): *slots_on_host = 1 when the partial vectors are staged in
 * pinned host memory (no peer access); devices_out / units_out (each `cap` entries, may be NULL): every member's HIP
 * device ordinal and how many units its shard holds (-1 before the first evaluation has dealt them). */
int gprf_group_info(const gprf_ctx *ctx, int32_t *n_members, int32_t *slots_on_host, int32_t cap, int32_t *devices_out,
                    int32_t *units_out);
int gprf_destroy(gprf_ctx *ctx);
/* The text of the context's last error; ctx = NULL: why this thread's last gprf_create / gprf_create_multi failed. */
const char *gprf_last_error(const gprf_ctx *ctx);

/* self.Y (gprf.py:97): n x dy, uploaded once and kept resident in HBM. */
int gprf_set_Y(gprf_ctx *ctx, const double *Y);

/* update_covs (gprf.py:160-167): theta = [noise_var, signal_var, dfn_params...], ntheta = 2 + #dfn_params
 * (euclidean: dx lengthscales; lld: 2). */
int gprf_set_theta(gprf_ctx *ctx, const double *theta, int32_t ntheta);

/* self.block_idxs (gprf.py:100,172): CSR form of the list of index arrays; block b owns
 * point_idx[block_ptr[b] .. block_ptr[b+1]).  Order inside a block is kept (it fixes the row order of
 * the unit matrices, gprf.py:301,317-326).  Blocks may be empty (gprf.py:507-513) and points may be left out
 * (their gradient rows are zero), but blocks must be DISJOINT: a point listed twice is refused with
 * GPRF_ERR_ARG (every partition the reference's callers produce — Blocker.block_clusters, pdtree reblock — is
 * one; the gradient scatter of gprf.py:258-273 is evaluated as a per-point gather here). */
int gprf_set_blocks(gprf_ctx *ctx, int32_t n_blocks, const int64_t *block_ptr, const int32_t *point_idx);

/* Fast host path for the reference's grid Blocker (block_clustering.py:17-26, called from update_X on EVERY
 * evaluation, gprf.py:171-172): block_of[p] = argmin_c ||x_p - c|| with the reference's
 * a^2 - 2ab + b^2 radicand and first-minimum tie rule.  Pure host code, no context needed. */
int gprf_nearest_center(int32_t n, int32_t dx, const double *X, int32_t n_centers, const double *centers,
                        int32_t *block_of);
/* Same effect as gprf_set_blocks for the partition block_of[p] in [0, n_blocks): points keep ascending index
 * order inside each block, exactly like `all_idxs[blocks == i]` (block_clustering.py:21-24). */
int gprf_set_block_assignment(gprf_ctx *ctx, int32_t n_blocks, const int32_t *block_of);

/* Device re-blocking for grid Blockers (gprfopt.py re-runs block_fn = cluster_rpc(...) on every
 * objective call through GPRF.update_X, gprf.py:169-174).  gprf_set_centers keeps the cluster centres
 * (n_centers x dx, row-major) in the context.  gprf_assign_blocks uploads X (n x dx), assigns every point to its
 * nearest centre ON THE DEVICE with exactly the arithmetic and tie rule of gprf_nearest_center, and compares
 * with the partition the unit tables were built from: *changed = 0 -> nothing else happens (the common case
 * between L-BFGS iterates); *changed = 1 -> the unit tables (sizes, offsets, unit row -> point, point -> rows)
 * have been rebuilt ON THE DEVICE for the new partition (points keep ascending index order inside a block, as
 * gprf_set_block_assignment would give) and, if block_of_out != NULL, the partition is copied there (n int32).
 * gprf_get_block_assignment copies the current partition (whoever made it) at any time. */
int gprf_set_centers(gprf_ctx *ctx, int32_t n_centers, const double *centers);
int gprf_assign_blocks(gprf_ctx *ctx, const double *X, int32_t *changed, int32_t *block_of_out);
int gprf_get_block_assignment(gprf_ctx *ctx, int32_t *block_of_out);

/* The same for the seismic driver's partition (run_seismic.py:375 passes pdtree_cluster's reblock as
 * block_fn, pdtree_clustering.py:65-94): a binary split tree of n_nodes nodes, node 0 the root, children with larger
 * ids than their parent; node k is a leaf when left[k] < 0 and then stands for block leaf_block[k] (a permutation of
 * 0..n_leaves-1), otherwise a point goes left when (x[0:dim] - center[k]) . vec[k] < split[k] (projection accumulated
 * column by column, multiplies and adds rounded separately) and right otherwise.  lon_wrap != 0 first maps x[0] to
 * (x[0] + 22) % 360 - 22 (pdtree_clustering.py:82).  vec/center are n_nodes x dim row-major.  After this call
 * gprf_assign_blocks routes through the tree (n_blocks = n_leaves) until gprf_set_centers is called again. */
int gprf_set_split_tree(gprf_ctx *ctx, int32_t n_nodes, int32_t dim, int32_t lon_wrap, const double *vec,
                        const double *center, const double *split, const int32_t *left, const int32_t *right,
                        const int32_t *leaf_block);

/* compute_neighbors (gprf.py:119-150): for every candidate block pair (cand_ij: n_cand rows (i, j)), decides whether the
 * largest |k(x_p, x_q)| / signal_var over p in block i, q in block j exceeds `threshold` (keep_out[c] = 1) — the
 * reference's `np.max(np.abs(self.kernel(X1, X2=X2) / wfn_var)) > threshold` with the kernel of gprf_set_theta.  Blocks
 * as in gprf_set_blocks (CSR; they need not be installed).  The caller supplies the candidates — all pairs j < i like
 * the reference, or a pruned superset (gprf_amd/neighbors.py prunes by bounding boxes / spherical caps).  max_out
 * (may be NULL): the exact maxima, without the early exit.  One-time setup work: allocates and frees its buffers. */
int gprf_pair_kernel_max(gprf_ctx *ctx, const double *X, int32_t n_blocks, const int64_t *block_ptr,
                         const int32_t *point_idx, double threshold, int32_t n_cand, const int32_t *cand_ij,
                         int32_t *keep_out, double *max_out);

/* self.neighbors (gprf.py:112,212): n_pairs rows (i, j); each becomes one joint unit with block i's rows
 * first (gprf.py:322).  Bethe weights 1 - deg(i) for the unaries are derived here
 * (compute_neighbor_count gprf.py:152-157; llgrad gprf.py:253-254, 264, 287).  For local=False pass all
 * pairs (gprf.py:214-216). */
int gprf_set_neighbors(gprf_ctx *ctx, int32_t n_pairs, const int32_t *pairs_ij);

/* Unit-sharding for one-process-per-GPU runs: this context evaluates only the units that the
 * longest-processing-time partition over `world` ranks gives to `rank`; gprf_eval then returns that
 * rank's partial sums (to be all-reduced by the caller).  Default (0, 1) = everything. */
int gprf_set_shard(gprf_ctx *ctx, int32_t rank, int32_t world);

/* The partition gprf_set_shard uses, exposed so that callers / tests can inspect it without a GPU:
 * unit u has m[u] points; owner_out[u] = rank that evaluates it.  Greedy longest-processing-time-first
 * on cost m^3 + 4 m^2 dy, ties to the lowest rank, stable in u.  Pure host code. */
int gprf_partition_units(int32_t n_units, const int32_t *m, int32_t dy, int32_t world, int32_t *owner_out);

/* jitchol's retry (gpy_linalg.py:86-96): extra diagonal added to unit `u`'s kernel matrix.
 * Units are numbered: blocks 0..n_blocks-1, then pairs in the order given.  NULL clears all. */
int gprf_set_unit_jitter(gprf_ctx *ctx, int32_t n_units, const double *jitter);

/* One objective+gradient evaluation = GPRF.llgrad(local=True, grad_X=, grad_cov=) at self.X = X
 * (gprf.py:206-296).  X: n x dx (host).  ll_out: 1 double.  gradX_out: n x dx, fully overwritten (NULL if
 * !want_gradX).  gradC_out: ntheta doubles in theta order (NULL if !want_gradC).
 * first_bad_unit: set to the lowest failing unit id when GPRF_NOT_PD is returned, else -1. */
int gprf_eval(gprf_ctx *ctx, const double *X, int32_t want_gradX, int32_t want_gradC, double *ll_out,
              double *gradX_out, double *gradC_out, int32_t *first_bad_unit);

/* update_X + llgrad in ONE call — what the reference's objective callbacks do per L-BFGS-B function evaluation
 * (gprfopt.py:377-417: gprf.update_X(xx) re-runs block_fn, gprf.py:169-174; then gprf.llgrad, gprf.py:206-296).
 * Needs gprf_set_centers / gprf_set_split_tree.  X goes up once; the points are re-partitioned on the device;
 * if any point changed block the unit tables are rebuilt there; the evaluation follows on the same stream; the
 * result comes down; ONE synchronisation.  *reblocked (may be NULL) = 1 when the partition changed.  Other
 * arguments and return values as gprf_eval. */
int gprf_update_eval(gprf_ctx *ctx, const double *X, int32_t want_gradX, int32_t want_gradC, double *ll_out,
                     double *gradX_out, double *gradC_out, int32_t *first_bad_unit, int32_t *reblocked);

/* Same evaluation with device-resident input and output (the timed form; also what a multi-GPU caller
 * all-reduces).  d_X: n*dx doubles in HBM.  d_out: 1 + n*dx + ntheta + 2 doubles in HBM laid out
 * [ll | gradX row-major | gradC | s0 | s1]; gradX / gradC parts are zero-filled when not requested.  s0 = 1 if this
 * context's re-partition outgrew its workspace (GPRF_RETRY will be reported), s1 = number of this context's units
 * that were not positive definite: after a SUM all-reduce of the vector every rank knows whether ANY rank has to
 * repeat or to jitter, in the same collective that sums the objective.
 * stream: a hipStream_t (NULL = the context's own stream); the call only enqueues work.
 * Follow with gprf_eval_status() after synchronising the stream. */
int gprf_eval_device(gprf_ctx *ctx, const double *d_X, int32_t want_gradX, int32_t want_gradC,
                     double *d_out, void *stream);
/* gprf_update_eval's device-resident form: re-partition + (if needed) table rebuild + evaluation, all enqueued. */
int gprf_update_eval_device(gprf_ctx *ctx, const double *d_X, int32_t want_gradX, int32_t want_gradC,
                            double *d_out, void *stream);
/* Blocks until the context's last evaluation has finished; returns GPRF_OK / GPRF_NOT_PD as gprf_eval, or
 * GPRF_RETRY after a gprf_update_eval_device whose new partition did not fit the workspace (a unit grew past the
 * launch-wide tile bound, or the matrices past the pools): d_out is undefined, the workspace has been grown and the
 * new partition is installed — enqueue the evaluation again with gprf_eval_device.  (The host forms gprf_eval /
 * gprf_update_eval do that themselves.) */
int gprf_eval_status(gprf_ctx *ctx, int32_t *first_bad_unit);
/* Whether the evaluation gprf_eval_status last finished changed the partition (host-side flag, no device access). */
int gprf_last_reblocked(const gprf_ctx *ctx, int32_t *reblocked);

/* ---- The optimiser-facing objective: what the reference's drivers wrap around llgrad before handing it to scipy's
 * L-BFGS-B (gprfopt.py:320-417, do_optimization.lgpllgrad): the vector z = [X.flatten() | hyper-parameters in log space
 * times cov_scale], the Gaussian location prior around the observed positions (x_prior, gprfopt.py:172-182), the
 * near-uniform prior on the log hyper-parameters (cov_prior, gprfopt.py:324-331), the chain rule through exp
 * (gprfopt.py:403-407) and the sign flip (gprfopt.py:417).  The location prior's terms are added by the assembly kernel
 * and the result comes down already in the optimiser's layout, in the evaluation's one download. ---- */
#define GPRF_HYPER_NONE 0 /* hyper-parameters fixed (task x) */
#define GPRF_HYPER_TIED 1 /* one free parameter: the common lengthscale; noise / signal variance fixed
                             (full_cov / collapse_cov_grad with a 1-column C, gprfopt.py:333-355) */
#define GPRF_HYPER_FULL 2 /* every entry of theta free (the 4-column C of gprfopt.py:343,352) */

/* N(X_obs, obs_std^2 I) prior on the locations (gprfopt.py:172-182); X_obs: n x dx, copied.  With a prior set, z of
 * gprf_objective starts with the n*dx locations.  NULL removes it. */
int gprf_set_x_prior(gprf_ctx *ctx, const double *X_obs, double obs_std);
/* Hyper-parameter part of z: zh = cov_scale * log(free parameters) (gprfopt.py:364-368,383), each log parameter with
 * a N(prior_mean, prior_std^2) prior (gprfopt.py:324-331: mean -1, std 10).  TIED: theta = [fixed_noise_var,
 * fixed_signal_var, l, l, ...] (gprfopt.py:336-341 pins them to the data's noise variance and 1.0). */
int gprf_set_hyper_param(gprf_ctx *ctx, int32_t mode, double cov_scale, double prior_mean, double prior_std,
                         double fixed_noise_var, double fixed_signal_var);
/* One call of the optimiser's callback (gprfopt.py:377-417): z (nz = [n*dx if a location prior is set] + [1 | ntheta | 0
 * hyper-parameters]) -> *f_out = -(ll + priors), grad_out (nz) = -d(ll + priors)/dz.  X_fixed: the locations when they are
 * not part of z (no location prior set), else ignored.  reblock != 0: re-partition first, like gprf_update_eval
 * (gprf.update_X inside the callback, gprfopt.py:385).  parts_out (may be NULL) <- [ll of the GPRF terms, location prior,
 * hyper-parameter prior].  Return values and first_bad_unit / reblocked as gprf_update_eval. */
int gprf_objective(gprf_ctx *ctx, const double *z, int32_t nz, const double *X_fixed, int32_t reblock, double *f_out,
                   double *grad_out, double *parts_out, int32_t *first_bad_unit, int32_t *reblocked);
/* The device-resident form (sharded runs all-reduce it): as gprf_eval_device / gprf_update_eval_device, but d_out =
 * [-(ll + location prior) | -(gradX + prior gradient) | gradC (unchanged: the hyper-parameter chain rule is host work on
 * ntheta numbers, gprf_hyper_grad) | s0 | s1].  In a sharded job only rank 0's context adds the location prior, so a
 * SUM all-reduce counts it once. */
int gprf_objective_device(gprf_ctx *ctx, const double *d_X, int32_t want_gradX, int32_t want_gradC, double *d_out,
                          void *stream, int32_t reblock);
/* The host pieces on their own (no context, no GPU): the location prior (ll_out; grad_out may be NULL), theta from the
 * optimiser's hyper variables, and d(ll + prior)/d zh from gradC = d ll / d theta (not negated; *prior_ll_out = the prior). */
int gprf_x_prior(int64_t n_elems, const double *x, const double *x_obs, double obs_std, double *ll_out, double *grad_out);
int gprf_hyper_unpack(int32_t mode, double cov_scale, double fixed_noise_var, double fixed_signal_var, int32_t ntheta,
                      const double *zh, double *theta_out);
int gprf_hyper_grad(int32_t mode, double cov_scale, double prior_mean, double prior_std, int32_t ntheta, const double *zh,
                    const double *gradC, double *prior_ll_out, double *grad_zh_out);

/* Bookkeeping a caller may want. */
int gprf_num_units(const gprf_ctx *ctx, int32_t *n_units_total, int32_t *n_units_local);
/* sum over local units of the algorithmic work of SURVEY.md §8d: flops = m^3 + 4 m^2 dy,
 * fill bytes = 8 m^2. */
int gprf_work_estimate(gprf_ctx *ctx, double *flops, double *fill_bytes);
/* How many times the unit tables have been (re)built on the device since the context was created (tests: an
 * evaluation whose re-partition moved nobody must not rebuild). */
int gprf_table_builds(gprf_ctx *ctx, int32_t *builds);

/* The diagnostic defines this library was compiled with (in-kernel cycle stamps, workgroup traces, ablations that skip
 * work inside the kernels): "" for the product build — tests assert that, so that no measured number comes from a
 * diagnostic variant.  Space-separated names otherwise. */
const char *gprf_build_flags(void);
/* Launch structure the library runs with in THIS process environment, as "key=value ..." : side_mode (how the two
 * Cholesky queues fork / join: 4 = kernel-written word + stream memory operation, the product path; 0 = events, chosen
 * whenever a profiler or a serialising launch mode is in the environment — a trace taken under such a tool shows THAT
 * structure), tool_env, potrf_dual, io_mode (GPRF_IO_MODE: 0 zero-copy host I/O + polled completion word, 1 copy
 * commands + stream synchronisation, 2 zero-copy in / copy out + polled word), sync (GPRF_SYNC). */
const char *gprf_runtime_config(void);

/* HIP-event timing of the kernels, recorded on the stream the evaluation is enqueued on.
 * Stages: "gather","fill","potrf","solve","at","grad","assemble".  gprf_set_timing(ctx, 1) turns recording
 * on (events between every kernel of every evaluation, kept in a ring so back-to-back evaluations need no
 * host sync), 0 off, 2 = on + reset the running totals.  gprf_get_timing waits for outstanding
 * evaluations and returns ms_out[0..6] = average duration per stage over the evaluations recorded since
 * the reset; if n >= 15 also ms_out[7..13] = the last evaluation's durations and ms_out[14] = the count. */
int gprf_set_timing(gprf_ctx *ctx, int32_t enable);
/* With these timers ON the library runs every stage as ONE launch over all units, one after the other (what the stage
 * durations mean).  With them off an evaluation of the kind the reference's drivers issue — gprf_eval / gprf_update_eval /
 * gprf_objective: host in, host out, one at a time (gprfopt.py:377-417) — pipelines the two size classes of the Cholesky
 * (gprf.py:520: pdinv; units of more than / at most 13 tiles on two queues) through substitution, At and gradient
 * (gprf.py:521-584) on those two queues and joins them in front of the assembly (gprf.py:253-288); the results are the
 * same bits.  gprf_set_stream_pipelines(ctx, 1) asks for the same on a CALLER's stream (gprf_eval_device,
 * gprf_update_eval_device, gprf_objective_device): for a caller that enqueues one evaluation and waits for it — one
 * process per GPU, the all-reduce behind every evaluation; default 0: many contexts enqueued back to back on one stream
 * share the hardware queues of their side streams and lose more than they gain. */
int gprf_set_stream_pipelines(gprf_ctx *ctx, int32_t enable);
int gprf_get_timing(gprf_ctx *ctx, int32_t n, double *ms_out);
#define GPRF_N_STAGES 7

/* Per-stage parity hooks (tests only): after an evaluation, copy one local unit's intermediates to the
 * host.  what: 0 the factor pool (mp x mp, upper triangle = Cholesky factor U, K = U^T U; the strictly-lower
 * part is unspecified) — or, after a fill-only gprf_debug_run (stop_after = 0), the K pool (64x64 blocks
 * ti <= tj, the rest unspecified),
 * 1 W = U^-T (mp x mp, lower), 2 Z = U^-T Y (mp x 64), 3 At = (K^-1 Y)^T (64 x mp),
 * 4 per-row gradient slab (mp x 4), 5 [ll_u, logdet_u, zz_u, info_u], 6 eight in-kernel cycle
 * counters of diagnostic builds, 7 / 8 the gradient reduction's per-block column / row partials
 * (mp x TBm x 4, TBm = ceil(max local unit rows / 64)), 9 the unit's own (unweighted) gradient with respect to
 * theta (ntheta doubles; needs an evaluation run with want_gradC), 10 the unit row -> point table of the unit (mp
 * values, -1 in the padding).  mp = m rounded up to 16.
 * `stop_after` for gprf_debug_run: run the pipeline only up to a stage (0 = fill only ... 6 = all). */
int gprf_debug_run(gprf_ctx *ctx, const double *X, int32_t stop_after);
int gprf_debug_fetch(gprf_ctx *ctx, int32_t local_unit, int32_t what, double *out, int64_t out_len);
int gprf_debug_unit_shape(gprf_ctx *ctx, int32_t local_unit, int32_t *m, int32_t *mp, int32_t *global_unit);

#ifdef __cplusplus
}
#endif
#endif /* GPRF_HIP_H */
