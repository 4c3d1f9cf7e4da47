"""``GPRF`` — the reference's model object (``/root/reference/gprf.py:83-296``) over the MI355X HIP
library.  Same constructor, same ``update_X / update_covs / llgrad`` surface and return shapes, so the
reference's objective callbacks (gprfopt.py:377-417, run_seismic.py:121-199) drive it unchanged; the
per-unit dense GP algebra (gprf.py:299-375, 496-591; gpy_linalg.py:77-253) runs in
``libgprf_hip.so`` through the C ABI in ``include/gprf_hip.h``.

There is no CPU path here: constructing a GPRF without the HIP library or a GPU raises.
"""
from collections import defaultdict

import numpy as np

from . import _capi
from .cov import GPCov, SUPPORTED


def symmetrize_neighbors(neighbors):
    """gprf.py:76-81"""
    nd = defaultdict(set)
    for (i, j) in neighbors:
        nd[i].add(j)
        nd[j].add(i)
    return nd


def _csr_from_block_idxs(block_idxs):
    lens = np.fromiter((len(b) for b in block_idxs), dtype=np.int64, count=len(block_idxs))
    ptr = np.zeros(len(block_idxs) + 1, dtype=np.int64)
    np.cumsum(lens, out=ptr[1:])
    if ptr[-1] > 0:
        pts = np.concatenate([np.asarray(b, dtype=np.int32).ravel() for b in block_idxs]).astype(np.int32, copy=False)
    else:
        pts = np.zeros(0, dtype=np.int32)
    return ptr, pts


class GPRF(object):

    def __init__(self, X, Y, block_fn, cov, noise_var, kernelized=False, dy=None,
                 neighbor_threshold=1e-3, nonstationary=False, nonstationary_prec=False,
                 block_idxs=None, neighbors=None, device=0, shard=None, group=None, reduce=True, devices=None):
        """Arguments as gprf.py:85-87.  ``kernelized`` / ``nonstationary`` are dead or broken branches in
        the reference (Appendix A.9/11) and are refused.  ``device`` = HIP device
        ordinal.  ``shard`` = (rank, world): this object evaluates only its rank's share of the units and ``llgrad``
        all-reduces the partial sums over the torch.distributed ``group`` (one process per GPU; the counterpart of
        the reference's process-pool fan-out inside llgrad, gprf.py:218-233), so every rank returns the full result
        and an optimiser can run unchanged on all ranks.  ``reduce=False`` returns the rank's partial sums instead
        (tests, callers that reduce themselves).  ``devices`` = a list of HIP device ordinals (or a count N = devices
        0..N-1): ONE process drives them all (gprf_create_multi): the units are sharded over the devices inside the
        library, partial sums meet on the first device — the reference's single-process driver needs no torchrun."""
        if kernelized or nonstationary or nonstationary_prec:
            raise NotImplementedError("kernelized / nonstationary GPRF variants are unreachable in the "
                                      "reference (gprf.py:90-97,302) and are not provided")
        if (cov.dfn_str, cov.wfn_str) not in SUPPORTED:
            raise ValueError("unsupported covariance (%s, %s)" % (cov.dfn_str, cov.wfn_str))
        if len(cov.wfn_params) != 1:
            raise ValueError('gradient computation currently assumes just a single scaling parameter for '
                             'weight function, but currently wfn_params=%s' % (cov.wfn_params,))  # gprf.py:369-370
        self.X = X
        self.kernelized = False
        self.Y = Y
        self._block_idxs = None
        self._block_of = None
        self._reblock_pending = False
        if block_idxs is None:
            block_idxs = block_fn(X)
        self.block_idxs = block_idxs
        self.block_fn = block_fn
        self.n_blocks = len(block_idxs)
        self.nonstationary = False
        self.cov = cov
        self.noise_var = noise_var
        self.neighbor_threshold = neighbor_threshold

        n, dx = X.shape
        if devices is not None:
            if shard is not None:
                raise ValueError("devices= (one process, several GPUs) and shard= (one process per GPU) exclude each other")
            devices = list(range(devices)) if isinstance(devices, int) else [int(d) for d in devices]
        self._ctx = _capi.Context(n, dx, Y.shape[1], _capi.DIST_IDS[cov.dfn_str], _capi.KERN_IDS[cov.wfn_str],
                                  device=device, devices=devices)
        self._shard = (int(shard[0]), int(shard[1])) if shard is not None else (0, 1)
        self._group, self._reduce, self._dist_eval = group, bool(reduce), None
        if shard is not None:
            self._ctx.set_shard(*self._shard)
        self._ctx.set_Y(Y)
        self._push_theta()
        self._blocks_pushed = None
        self._nbrs_pushed = None
        self._jitter = None
        self._push_blocks()

        if neighbors is not None:
            self.neighbors = neighbors
        else:
            self.compute_neighbors(threshold=neighbor_threshold)
        self.compute_neighbor_count()
        self.neighbor_dict = symmetrize_neighbors(self.neighbors)

    # ------------------------------------------------------------------ state -> device
    def _theta(self):
        return np.concatenate([[self.noise_var, self.cov.wfn_params[0]],
                               np.asarray(self.cov.dfn_params, dtype=np.float64).ravel()])

    def _push_theta(self):
        self._ctx.set_theta(self._theta())

    @property
    def block_idxs(self):
        """list of index arrays (gprf.py:100); after a re-blocking on the device it is fetched (and the re-blocking
        itself run, if update_X has only announced it) when somebody reads it"""
        if self._reblock_pending:
            # update_X was called but no evaluation has happened yet: partition now (what block_fn(new_X) would return)
            changed, block_of = self._ctx.assign_blocks(np.ascontiguousarray(self.X, dtype=np.float64))
            self._reblock_pending = False
            if changed:
                self._block_of, self._block_idxs = block_of, None
                self._blocks_pushed = "device"       # the library holds exactly this partition: nothing to upload (an upload
                                                     # would also re-deal the shard ownership on this rank alone)
        if self._block_idxs is None:
            if self._block_of is None:
                self._block_of = self._ctx.get_block_assignment()
            order = np.argsort(self._block_of, kind="stable")
            counts = np.bincount(self._block_of, minlength=self.n_blocks)
            self._block_idxs = np.split(order, np.cumsum(counts)[:-1])
            if self._blocks_pushed == "device":
                self._blocks_pushed = self._block_idxs      # the library already holds exactly this partition
        return self._block_idxs

    @block_idxs.setter
    def block_idxs(self, v):
        self._block_idxs = v
        self._block_of = None
        self._reblock_pending = False

    def _push_blocks(self):
        if self._blocks_pushed == "device" and self._block_idxs is None:
            return
        if self._blocks_pushed is self._block_idxs:
            return
        ptr, pts = _csr_from_block_idxs(self.block_idxs)
        if len(ptr) - 1 != self.n_blocks:
            raise ValueError("block_fn returned %d blocks, expected %d" % (len(ptr) - 1, self.n_blocks))
        self._ctx.set_blocks(ptr, pts)
        self._blocks_pushed = self._block_idxs
        self._jitter = None
        self._ctx.set_unit_jitter(None)

    def _push_neighbors(self, neighbors):
        if self._nbrs_pushed is neighbors:
            return
        self._ctx.set_neighbors(neighbors)
        self._nbrs_pushed = neighbors
        self._jitter = None
        self._ctx.set_unit_jitter(None)

    # ------------------------------------------------------------------ reference surface
    def compute_neighbors(self, threshold=1e-3):
        """gprf.py:119-150: connect blocks whose largest cross-covariance (relative to the signal
        variance) exceeds ``threshold``; 1.0 means no pairs.  The host prunes the block pairs by geometry
        (neighbors.candidate_block_pairs: no kernel evaluation), the device decides the candidates
        (gprf_pair_kernel_max): the same list, in the same order, as the reference's exhaustive double loop."""
        from .neighbors import candidate_block_pairs
        cand = candidate_block_pairs(self.X, self.block_idxs, self.cov, threshold)
        if not cand:
            self.neighbors = []
            return
        ptr, pts = _csr_from_block_idxs(self.block_idxs)
        keep = self._ctx.pair_kernel_max(self.X, ptr, pts, threshold, cand)
        self.neighbors = [c for c, k in zip(cand, keep) if k]

    def compute_neighbor_count(self):
        """gprf.py:152-157"""
        neighbor_count = defaultdict(int)
        for (i, j) in self.neighbors:
            neighbor_count[i] += 1
            neighbor_count[j] += 1
        self.neighbor_count = neighbor_count

    def update_covs(self, covs):
        """gprf.py:160-167: covs[0] = [noise_var, signal_var, lengthscales...]"""
        nv, sv = covs[0, :2]
        lscales = covs[0, 2:]
        self.cov = GPCov(wfn_params=[sv, ], dfn_params=lscales, dfn_str=self.cov.dfn_str, wfn_str=self.cov.wfn_str)
        self.noise_var = nv
        self._push_theta()

    def _device_router(self):
        """The object whose partition rule the library can evaluate itself: a grid ``Blocker`` (block_fn is its
        ``block_clusters``) or the split tree behind ``seismic.pdtree_cluster``'s ``reblock``; None otherwise."""
        fn = self.block_fn
        if fn is None:
            return None
        cached = getattr(self, "_router_cache", None)
        if cached is not None and cached[0] is fn and cached[1] == self.n_blocks:
            return cached[2]                     # (asked on every update_X: once per optimiser step)
        router = self._find_device_router(fn)
        self._router_cache = (fn, self.n_blocks, router)
        return router

    def _find_device_router(self, fn):
        from .blocking import Blocker
        blocker = getattr(fn, "__self__", None)
        tree = getattr(fn, "tree", None)
        if isinstance(blocker, Blocker) and getattr(fn, "__name__", "") == "block_clusters" \
                and blocker.n_blocks == self.n_blocks:
            if getattr(self, "_centers_of", None) is not blocker:
                self._ctx.set_centers(blocker.block_centers)
                self._centers_of = blocker
            return blocker
        if tree is not None and hasattr(tree, "leaf_block") and len(tree.leaf_order) == self.n_blocks:
            if getattr(self, "_centers_of", None) is not tree:
                self._ctx.set_split_tree(tree.vec, tree.center, tree.split, tree.left, tree.right, tree.leaf_block,
                                         getattr(fn, "lon_wrap", False))
                self._centers_of = tree
            return tree
        return None

    def update_X(self, new_X, update_blocks=True, recompute_neighbors=False):
        """gprf.py:169-174: rebinding X re-runs block_fn on every call; the neighbour list stays.

        When ``block_fn`` is the ``block_clusters`` method of a grid ``Blocker`` (or the ``reblock`` of
        ``seismic.pdtree_cluster``) the library evaluates the partition rule itself, on the device, as the first step of
        the NEXT evaluation (``gprf_update_eval``: one upload of X, re-partition, unit tables rebuilt on the device if
        anybody changed block, evaluation, one download); ``block_idxs`` is fetched only if somebody reads it."""
        self.X = new_X
        if self.block_fn is not None:
            if self._device_router() is not None:
                self._push_blocks()              # (a partition given by the host is installed first: nothing is lost)
                self._reblock_pending = True
            else:
                self.block_idxs = self.block_fn(new_X)
                self._push_blocks()
        if recompute_neighbors:
            self.compute_neighbors(threshold=self.neighbor_threshold)
            self.compute_neighbor_count()
            self.neighbor_dict = symmetrize_neighbors(self.neighbors)

    def update_X_block(self, i, new_X):
        """gprf.py:176-179"""
        self.X[self.block_idxs[i]] = new_X

    def subset_llgrad(self, blocks):
        """gprf.py:182-204: the objective over a SUBSET of the blocks — their unaries and only the pairs between members
        of the subset, Bethe weights from the neighbour counts INSIDE the subset.  Returns the log-likelihood alone, as the
        reference does.  One library evaluation: the blocks outside the subset are installed empty (an empty unit
        contributes exactly 0, gprf.py:507-513) and the in-set pairs are the neighbour list, from which the library
        derives the local counts; the object's own partition and neighbour list are re-installed by the next call that
        needs them.  A block listed k times counts k times, as in the reference's list comprehension."""
        if self._shard[1] > 1 and self._reduce:
            raise NotImplementedError("subset_llgrad on a sharded GPRF: evaluate it on one rank (the reference never calls it)")
        blocks = [int(b) for b in blocks]
        block_set = set(blocks)
        neighbors_in_set = [(i, j) for (i, j) in self.neighbors if i in block_set and j in block_set]
        full = self.block_idxs                   # (runs a pending re-blocking, like reading the attribute anywhere else)
        empty = np.zeros(0, dtype=np.int64)
        X = np.ascontiguousarray(self.X, dtype=np.float64)

        def evaluate(members, nbrs):
            ptr, pts = _csr_from_block_idxs([full[b] if b in members else empty for b in range(self.n_blocks)])
            self._ctx.set_blocks(ptr, pts)
            self._ctx.set_neighbors(nbrs)
            self._ctx.set_unit_jitter(None)
            self._blocks_pushed, self._nbrs_pushed, self._jitter = None, nbrs, None
            rc, ll, _, _, bad = self._ctx.eval(X, False, False)
            if rc == _capi.GPRF_NOT_PD:
                rc, ll, _, _ = self._retry_with_jitter(X, False, False, bad)
            return ll

        try:
            ll = evaluate(block_set, neighbors_in_set)
            counts = defaultdict(int)
            for (i, j) in neighbors_in_set:
                counts[i] += 1
                counts[j] += 1
            seen = set()
            for b in blocks:                     # a repeated block: its weighted unary once more per repetition
                if b in seen:
                    ll += (1 - counts[b]) * evaluate({b}, [])
                seen.add(b)
        finally:
            # the library holds the subset now: the next llgrad / objective call re-installs the object's own state
            self._blocks_pushed, self._nbrs_pushed, self._jitter = None, None, None
            self._ctx.set_unit_jitter(None)
        return ll

    def llgrad(self, parallel=False, local=True, grad_X=False, grad_cov=False, **kwargs):
        """gprf.py:206-296 -> (ll, gradX (n,dx) or (0,0), gradCov (1,ncov) or (0,0)).

        ``parallel`` (a multiprocessing pool over units in the reference, gprf.py:218-233) is accepted and
        ignored: every unit already runs concurrently on the GPU.  ``local=False`` = all block pairs
        (gprf.py:214-216)."""
        if kwargs:
            raise TypeError("unsupported llgrad arguments: %s" % sorted(kwargs))
        self._push_blocks()
        if self._jitter is not None:
            # every reference call starts un-jittered (jitchol is stateless, gpy_linalg.py:77-80)
            self._jitter = None
            self._ctx.set_unit_jitter(None)
        if local:
            neighbors = self.neighbors
        else:
            if getattr(self, "_all_pairs_nb", None) != self.n_blocks:
                self._all_pairs = [(i, j) for i in range(self.n_blocks) for j in range(i)]
                self._all_pairs_nb = self.n_blocks
            neighbors = self._all_pairs
        self._push_neighbors(neighbors)

        X = np.ascontiguousarray(self.X, dtype=np.float64)
        if self._shard[1] > 1 and self._reduce:
            return self._llgrad_sharded(X, grad_X, grad_cov)
        if self._reblock_pending:
            # update_X's re-blocking and the evaluation in one library call
            rc, ll, gX, gC, bad, reblocked = self._ctx.update_eval(X, grad_X, grad_cov)
            self._reblock_pending = False
            if reblocked:
                self._block_of, self._block_idxs = None, None
                self._blocks_pushed = "device"
        else:
            rc, ll, gX, gC, bad = self._ctx.eval(X, grad_X, grad_cov)
        if rc == _capi.GPRF_NOT_PD:
            rc, ll, gX, gC = self._retry_with_jitter(X, grad_X, grad_cov, bad)

        gradX = gX if grad_X else np.zeros((0, 0))
        gradCov = gC.reshape((1, -1)) if grad_cov else np.zeros((0, 0))
        return ll, gradX, gradCov

    # ------------------------------------------------------------------ the optimiser's callback, in the library
    def objective_setup(self, X_obs, obs_std, hyper_mode, cov_scale, hyper_prior, fixed_nv, fixed_sv):
        """Install what ``gprf_objective`` needs (include/gprf_hip.h): the location prior N(X_obs, obs_std^2) — None when
        the locations are not optimised — and the log-space parametrisation of the hyper-parameters."""
        self._ctx.set_x_prior(X_obs, obs_std)
        self._ctx.set_hyper_param(hyper_mode, cov_scale, hyper_prior[0], hyper_prior[1], fixed_nv, fixed_sv)
        self._objective_par = (hyper_mode, cov_scale, hyper_prior)

    def objective_call(self, z, layout):
        """One callback of the optimiser (gprfopt.py:377-417) as ONE library call: unpack z, re-block (update_X), evaluate,
        add the priors, chain rule, signs -> (f, grad, (GPRF terms, location prior, hyper prior)).  The object's ``X`` /
        ``cov`` / ``noise_var`` follow z like ``update_X`` / ``update_covs`` would leave them."""
        if layout.nx:
            self.update_X(layout.locations(z))
        if layout.nh:
            theta = layout.theta_row(z)
            self.cov = GPCov(wfn_params=[theta[0, 1]], dfn_params=theta[0, 2:], dfn_str=self.cov.dfn_str, wfn_str=self.cov.wfn_str)
            self.noise_var = theta[0, 0]
        self._push_blocks()
        if self._jitter is not None:
            self._jitter = None
            self._ctx.set_unit_jitter(None)
        self._push_neighbors(self.neighbors)
        X_fixed = None if layout.nx else np.ascontiguousarray(self.X, dtype=np.float64)
        if self._shard[1] > 1 and self._reduce:
            return self._objective_sharded(z, layout, X_fixed)
        reblock, self._reblock_pending = self._reblock_pending, False
        rc, f, grad, parts, bad, reblocked = self._ctx.objective(z, X_fixed, reblock=reblock)
        if reblocked:
            self._block_of, self._block_idxs = None, None
            self._blocks_pushed = "device"
        if rc == _capi.GPRF_NOT_PD:
            from .dist import jitter_schedule
            n_units = self.n_blocks + len(self._nbrs_pushed)

            def ev(jitter):
                self._ctx.set_unit_jitter(jitter)
                self._jitter = jitter
                rc2, f2, g2, p2, b2, _ = self._ctx.objective(z, X_fixed, reblock=False)
                return (f2, g2, p2), (b2 if rc2 == _capi.GPRF_NOT_PD else -1)

            (f, grad, parts), _ = jitter_schedule(ev, bad, n_units, self.cov.wfn_params[0] + self.noise_var, self._jitter)
        return f, grad, tuple(parts)

    def _objective_sharded(self, z, layout, X_fixed):
        """the same callback over a sharded job: rank 0's context adds the location prior, ONE all-reduce, then the
        hyper-parameter chain rule on the host (ntheta numbers)"""
        from . import dist as gdist
        if self._dist_eval is None:
            self._dist_eval = gdist.DeviceEvaluator(self, self._group)
        X = layout.locations(z) if layout.nx else X_fixed
        if layout.nh:
            self._push_theta()
        reblock, self._reblock_pending = self._reblock_pending, False
        f, gX, gC, reblocked = self._dist_eval.evaluate(X, layout.nx > 0, layout.nh > 0, reblock=reblock, objective=True)
        if reblocked:
            self._block_of, self._block_idxs = None, None
            self._blocks_pushed = "device"
        grad = np.empty(layout.nx + layout.nh)
        hp = 0.0
        if layout.nx:
            grad[:layout.nx] = gX.ravel()
        if layout.nh:
            mode, cov_scale, prior = self._objective_par
            hp, hg = _capi.hyper_grad(mode, cov_scale, prior[0], prior[1], z[layout.nx:], gC)
            grad[layout.nx:] = -hg
            f -= hp
        return f, grad, (None, None, hp)

    def _llgrad_sharded(self, X, grad_X, grad_cov):
        """This rank's units on this GPU, ONE all-reduce (sum) over the ranks, the same result on every rank."""
        from . import dist as gdist
        import torch.distributed as tdist
        if not (tdist.is_available() and tdist.is_initialized()):
            raise RuntimeError("GPRF(shard=(%d, %d)).llgrad needs torch.distributed initialised (one process per GPU); "
                               "pass reduce=False for this rank's partial sums" % self._shard)
        if tdist.get_world_size(self._group) != self._shard[1]:
            raise RuntimeError("shard world %d != process group size %d" % (self._shard[1], tdist.get_world_size(self._group)))
        if self._dist_eval is None:
            self._dist_eval = gdist.DeviceEvaluator(self, self._group)
        ll, gX, gC, reblocked = self._dist_eval.evaluate(X, grad_X, grad_cov, reblock=self._reblock_pending)
        self._reblock_pending = False
        if reblocked:
            self._block_of, self._block_idxs = None, None
            self._blocks_pushed = "device"
        return ll, gX, gC

    def _retry_with_jitter(self, X, grad_X, grad_cov, bad):
        """jitchol's policy (gpy_linalg.py:81-97) applied per failing unit: require a positive diagonal,
        then retry on K + j I with j = mean(diag K) * 1e-6 * 10^k, k = 0..4; the factor of the jittered
        matrix is then used for everything.  diag K = sv + nv for both kernels."""
        from .dist import jitter_schedule
        n_units = self.n_blocks + len(self._nbrs_pushed)
        diag_mean = self.cov.wfn_params[0] + self.noise_var

        def ev(jitter):
            self._ctx.set_unit_jitter(jitter)
            self._jitter = jitter
            rc, ll, gX, gC, b = self._ctx.eval(X, grad_X, grad_cov)
            return (rc, ll, gX, gC), (b if rc == _capi.GPRF_NOT_PD else -1)

        (rc, ll, gX, gC), _ = jitter_schedule(ev, bad, n_units, diag_mean, self._jitter)
        return rc, ll, gX, gC

    # reference attribute names some callers read
    @property
    def n_units(self):
        return self._ctx.num_units()[0]

    def close(self):
        self._ctx.close()

    def __getstate__(self):
        raise TypeError("GPRF holds device state and is not picklable (the reference pickles it only for "
                        "its multiprocessing pool, gprf.py:738-746, which this build does not use)")
