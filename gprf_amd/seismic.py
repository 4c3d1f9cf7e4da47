"""The seismic harness around the GPRF path: what ``run_seismic.py`` puts on either side of
``GPRF.llgrad`` for the great-circle / Matern-3/2 configuration —

* the principal-direction tree partition with its longitude wrap (``pdtree_clustering.py:4-94``) as a flat,
  array-based tree (``PDTree``, ``pdtree_cluster``): blocks of fewer than ``blocksize`` events, and a ``reblock``
  callable that routes moved events through the stored splits (the ``block_fn`` of the seismic GPRF);
* the location prior built from the observed locations (``run_seismic.py:353-365``) and the log-space covariance
  prior with its large-lengthscale penalty (``run_seismic.py:68-87``);
* the L-BFGS-B callback ``lgpllgrad`` (``run_seismic.py:90-199``): depth rescaling, parameter clamps, gradient
  clipping, the random-gradient answer to a failed evaluation.

The reference's data file (``sorted_isc.npy``) is not distributed: ``synthetic_events`` draws a clearly labelled
stand-in catalogue with the same columns.  Host-side setup and glue only; every evaluation goes through the HIP path.
"""
import os
import time

import numpy as np
import scipy.optimize

from .objective import OutOfTimeError


def wrap_longitude(lon):
    """pdtree_clustering.py:82: cut the circle at -22 degrees (mid-Atlantic) instead of the date line."""
    return (np.asarray(lon, dtype=np.float64) + 22) % 360 - 22


def _project(D, V):
    """Row-wise D . V accumulated column by column with separate multiplies and adds, so that a point's coordinate
    along a split direction has the same bits whether it is computed while building (all rows against one vector) or
    while routing (each row against its node's vector): the median point then routes to the side it was built on."""
    V = np.broadcast_to(V, D.shape)
    a = D[:, 0] * V[:, 0]
    for j in range(1, D.shape[1]):
        a = a + D[:, j] * V[:, j]
    return a


class PDTree(object):
    """Principal-direction tree (pdtree_clustering.py:4-77), stored as parallel arrays: node k is a leaf when
    ``left[k] < 0``; an inner node splits on ``(x - center[k]) . vec[k] < split[k]`` (left) / ``>=`` (right).
    ``leaf_order`` lists the leaves left to right — the order of the reference's ``leaf_idx()`` / ``recluster()``."""

    def __init__(self, X, minsize):
        X = np.asarray(X, dtype=np.float64)
        self.dim = X.shape[1]
        vec, center, split, left, right, leaf_pts = [], [], [], [], [], {}
        # depth-first, left child first: (point indices, parent node, is_right_child)
        stack = [(np.arange(len(X)), -1, False)]
        while stack:
            idx, parent, is_right = stack.pop()
            k = len(left)
            if parent >= 0:
                (right if is_right else left)[parent] = k
            vec.append(np.zeros(self.dim)); center.append(np.zeros(self.dim)); split.append(0.0)
            left.append(-1); right.append(-1)
            if len(idx) < minsize:                      # pdtree_clustering.py:31-32
                leaf_pts[k] = idx
                continue
            data = X[idx]
            mean = data.mean(axis=0)
            data = data - mean
            # np.linalg.eig as in the reference (pdtree_clustering.py:39): the sign it gives the eigenvector decides
            # on which side the median point falls, so the same routine must be used
            ev, evec = np.linalg.eig(data.T.dot(data))
            pvec = evec[:, np.argmax(ev)]
            a = _project(data, pvec)
            med = np.median(a)
            if not np.any(a < med):
                # every point on one side (coincident events, or more than half of them on the median): the reference
                # recursion would never end here; the points stay together as one oversized leaf
                leaf_pts[k] = idx
                continue
            vec[k], center[k], split[k] = pvec, mean, med
            stack.append((idx[a >= med], k, True))      # popped second
            stack.append((idx[a < med], k, False))      # popped first: left subtree numbered before the right one
        self.vec, self.center = np.array(vec).reshape(-1, self.dim), np.array(center).reshape(-1, self.dim)
        self.split, self.left, self.right = np.array(split), np.array(left), np.array(right)
        # with left-first depth-first numbering the leaves appear left to right in increasing node id
        self.leaf_order = [k for k in range(len(self.left)) if self.left[k] < 0]
        self.leaf_block = np.full(len(self.left), -1, dtype=np.int32)      # node id -> block index (leaves only)
        self.leaf_block[self.leaf_order] = np.arange(len(self.leaf_order), dtype=np.int32)
        self._build_leaves = [leaf_pts[k] for k in self.leaf_order]

    def leaf_idx(self):
        """pdtree_clustering.py:53-63"""
        return [np.array(i) for i in self._build_leaves]

    def recluster(self, X):
        """pdtree_clustering.py:65-77: every point descends from the root; one vectorised pass per tree level."""
        X = np.asarray(X, dtype=np.float64)
        node = np.zeros(len(X), dtype=np.int64)
        active = np.nonzero(self.left[node] >= 0)[0]
        while len(active):
            k = node[active]
            a = _project(X[active] - self.center[k], self.vec[k])
            node[active] = np.where(a < self.split[k], self.left[k], self.right[k])
            active = active[self.left[node[active]] >= 0]
        pos = {k: i for i, k in enumerate(self.leaf_order)}
        which = np.array([pos[k] for k in node], dtype=np.int64) if len(node) else np.zeros(0, dtype=np.int64)
        order = np.argsort(which, kind="stable")        # ascending point index inside a leaf, like idx[a < split]
        counts = np.bincount(which, minlength=len(self.leaf_order))
        return np.split(order, np.cumsum(counts)[:-1])


def pdtree_cluster(X, blocksize=300):
    """pdtree_clustering.py:79-94 -> (block index arrays, reblock).  ``reblock(XX)`` does not touch XX (the reference
    wraps its longitude column in place and restores it)."""
    X2 = np.array(np.asarray(X)[:, :2], dtype=np.float64)
    X2[:, 0] = wrap_longitude(X2[:, 0])
    tree = PDTree(X2, minsize=blocksize)

    def reblock(XX):
        Z = np.array(np.asarray(XX)[:, :2], dtype=np.float64)
        Z[:, 0] = wrap_longitude(Z[:, 0])
        return tree.recluster(Z)

    # GPRF.update_X recognises these two attributes and routes on the device (gprf_set_split_tree); calling reblock
    # itself stays the host path
    reblock.tree = tree
    reblock.lon_wrap = True
    return tree.leaf_idx(), reblock


def seismic_cov_prior(c):
    """run_seismic.py:68-87: N(means, 1.5^2) on the log parameters, plus an exponential wall against horizontal
    lengthscales beyond e^5 km (edges are not recomputed when the lengthscale grows)."""
    means = np.array((-2.3, 0.0, 3.6, 3.6))
    std = 1.5
    c = np.asarray(c, dtype=np.float64).reshape(-1)
    ll = -.5 * np.sum(((c - means) / std) ** 2) - .5 * len(c) * np.log(2 * np.pi * std ** 2)
    lderiv = -(c - means) / std ** 2
    if c[2] > 5:
        wall = np.exp(70 * (c[2] - 5))
        ll -= wall
        lderiv[2] -= 70 * wall
    return ll, lderiv


def make_x_prior(means, obs_std):
    """run_seismic.py:353-365: independent Gaussians around the observed (lon, lat, depth) with standard deviations
    obs_std * (0.01, 0.01, 1)."""
    means = np.array(means, dtype=np.float64)
    prior_std = obs_std * np.array([.01, .01, 1.])
    const = 3 * np.log(2 * np.pi) + np.sum(np.log(prior_std ** 2))

    def x_prior(X):
        r = (np.asarray(X, dtype=np.float64) - means) / prior_std
        return -.5 * np.sum(r ** 2) - .5 * X.shape[0] * const, -(r / prior_std)

    return x_prior


class SeismicObjective(object):
    """``obj(x) -> (-ll, -grad)`` for ``scipy.optimize.minimize(..., jac=True)`` — run_seismic.py:90-199.

    The optimiser sees depth / 100 (``depth_scale``) and the logarithms of [noise, signal, horizontal, depth]
    covariance parameters; the signal variance is pinned to 1, the noise variance capped at 10 and both lengthscales
    kept in [1, 999] before they reach the GPRF.  X0 is copied (the reference rescales the caller's array in place).
    The reference also indexes the depth column of the (0, 0) placeholder it gets back when no location gradient was
    requested, so its task 'cov' cannot run; here the factor is applied to a real gradient only."""

    depth_scale = 100.

    def __init__(self, gprf, X0, C0, cov_prior=seismic_cov_prior, x_prior=None, maxsec=None, log_dir=None):
        self.gprf, self.cov_prior, self.x_prior = gprf, cov_prior, x_prior
        self.gradX, self.gradC = (X0 is not None), (C0 is not None)
        if self.gradX and x_prior is None:
            raise ValueError("a location prior is needed when the locations are optimised")
        self.X0 = None
        if self.gradX:
            self.X0 = np.array(X0, dtype=np.float64)
            self.X0[:, 2] /= self.depth_scale
        self.C0 = None if C0 is None else np.array(C0, dtype=np.float64)
        x0 = self.X0.ravel() if self.gradX else np.zeros(0)
        c0 = np.log(self.C0.ravel()) if self.gradC else np.zeros(0)
        self.nx = len(x0)
        self.full0 = np.concatenate([x0, c0])
        self.maxsec, self.t0, self.step = maxsec, time.time(), 0
        self.f_log = open(os.path.join(log_dir, "log.txt"), "w") if log_dir else None
        self.last_ll, self.last_cov = None, None

    @staticmethod
    def clamp_cov(FC):
        """run_seismic.py:136-150"""
        FC = np.array(FC, dtype=np.float64)
        FC[0, 1] = 1.0
        FC[0, 0] = min(FC[0, 0], 10.0)
        FC[0, 2:4] = np.clip(FC[0, 2:4], 1.0, 999.0)
        return FC

    def __call__(self, x):
        x = np.asarray(x, dtype=np.float64)
        xx, xc = x[:self.nx], x[self.nx:]
        XX = FC = None
        if self.gradX:
            XX = xx.reshape(self.X0.shape).copy()
            XX[:, 2] *= self.depth_scale
            self.gprf.update_X(XX)
        if self.gradC:
            FC = self.clamp_cov(np.exp(xc.reshape(self.C0.shape)))
            self.gprf.update_covs(FC)
        try:
            ll, gX, gC = self.gprf.llgrad(local=True, grad_X=self.gradX, grad_cov=self.gradC)
        except np.linalg.LinAlgError:
            # run_seismic.py:155-159: a failed evaluation (a block that is not positive definite even with jitter)
            # is answered with a huge objective and a random direction.  Only THAT failure: a library / HIP error, or a
            # re-partition that grows a unit past GPRF_MAX_UNIT points (GprfHipError), propagates to the caller
            return 1e10, np.random.randn(*x.shape)
        pieces = []
        if self.gradX:
            gX = np.array(gX, dtype=np.float64)
            gX[:, 2] *= self.depth_scale
            prior_ll, prior_grad = self.x_prior(XX)
            prior_grad = np.array(prior_grad, dtype=np.float64)
            prior_grad[:, 2] *= self.depth_scale
            ll += prior_ll
            pieces.append(gX.ravel() + prior_grad.ravel())
        if self.gradC:
            prior_ll, prior_grad = self.cov_prior(xc)
            ll += prior_ll
            gC = (np.asarray(gC, dtype=np.float64) * FC).ravel() + prior_grad
            gC[1] = 0.0                                  # the signal variance is not learned
            biggest = np.max(np.abs(gC[2:]))
            if biggest > 10:                             # run_seismic.py:178-180: damp huge lengthscale gradients
                gC[2:] *= 2. / (1 + biggest / 10.)
            pieces.append(gC)
        self.last_ll, self.last_cov = ll, FC
        if self.f_log:
            self.f_log.write("%d %.2f %.2f\n" % (self.step, time.time() - self.t0, ll))   # run_seismic.py:187
            self.f_log.flush()
        self.step += 1
        if self.maxsec is not None and time.time() - self.t0 > self.maxsec:
            raise OutOfTimeError
        return -ll, -np.concatenate(pieces) if pieces else np.zeros(0)

    def close(self):
        if self.f_log:
            self.f_log.close()
            self.f_log = None


def do_seismic_optimization(gprf, X0, C0, x_prior, cov_prior=seismic_cov_prior, maxsec=3600, maxiter=None, log_dir=None):
    """run_seismic.py:90-214 -> (scipy result or None when out of time, objective)."""
    obj = SeismicObjective(gprf, X0, C0, cov_prior=cov_prior, x_prior=x_prior, maxsec=maxsec, log_dir=log_dir)
    opts = {} if maxiter is None else {"maxiter": maxiter}
    try:
        r = scipy.optimize.minimize(obj, obj.full0, jac=True, method="l-bfgs-b", bounds=None, options=opts)
    except OutOfTimeError:
        r = None
    obj.close()
    return r, obj


def synthetic_events(n, seed=0):
    """STAND-IN for the ISC catalogue columns (lon, lat, depth) the reference loads from sorted_isc.npy
    (run_seismic.py:288-292), which is not distributed: events scattered around a few great-circle arcs ("plate
    boundaries", some crossing the date line) with exponential depths (mean 30 km, clipped to 700)."""
    rng = np.random.RandomState(seed)
    n_arcs = 6
    lon0 = rng.uniform(-180, 180, n_arcs)
    lat0 = rng.uniform(-50, 50, n_arcs)
    heading = rng.uniform(0, 2 * np.pi, n_arcs)
    length = rng.uniform(20, 60, n_arcs)              # degrees along the arc
    arc = rng.randint(0, n_arcs, n)
    t = rng.uniform(-0.5, 0.5, n) * length[arc]
    lon = lon0[arc] + t * np.cos(heading[arc]) / np.maximum(np.cos(np.radians(lat0[arc])), 0.3) + rng.randn(n) * 1.0
    lat = np.clip(lat0[arc] + t * np.sin(heading[arc]) + rng.randn(n) * 1.0, -85, 85)
    lon = (lon + 180) % 360 - 180
    depth = np.minimum(rng.exponential(30.0, n), 700.0)
    return np.stack([lon, lat, depth], axis=1)


def sample_y(X, cov, noise_var, yd, seed=0):
    """Dense draw Y = chol(k(X, X) + noise_var I) Z (the small-n branch of run_seismic.sample_y / synthetic.py:103-114)."""
    from .synthetic import prior_kernel_matrix as kernel_matrix
    rng = np.random.RandomState(seed)
    K = kernel_matrix(X, X, cov) + noise_var * np.eye(len(X))
    return np.linalg.cholesky(K).dot(rng.randn(len(X), yd))
