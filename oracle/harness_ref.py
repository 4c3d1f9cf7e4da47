"""ORACLE — restatement of the callers and the input recipe around the GPRF path, enough to regenerate
the reference's published traces from seeds.  Test infrastructure only.

Follows:
  block_clustering.py:4-45  pair_distances, Blocker.block_clusters, Blocker.neighbors
  gprfopt.py:519-523        grid_centers
  synthetic.py:103-114,139-153   sample_y (dense branch, n < 40000), sample_synthetic
  gprfopt.py:21-74          SampledData (__init__, set_centers, build_gprf)
  gprfopt.py:172-182        x_prior
  gprfopt.py:320-417        do_optimization: cov_prior, full_cov, collapse_cov_grad, lgpllgrad
"""
import numpy as np

from .gprf_ref import GPRFRef
from .linalg_ref import jitchol
from .vector_tree import GPCov, VectorTree


def pair_distances(Xi, Xj):
    """block_clustering.py:4-5 (a^2 - 2ab + b^2 form, kept literally: it decides argmin ties)."""
    return np.sqrt(np.outer(np.sum(Xi ** 2, axis=1), np.ones((Xj.shape[0]),)) - 2 * np.dot(Xi, Xj.T)
                   + np.outer((np.ones(Xi.shape[0]),), np.sum(Xj ** 2, axis=1)))


class BlockerRef(object):
    def __init__(self, block_centers):
        self.block_centers = np.asarray(block_centers)
        self.n_blocks = len(block_centers)

    def block_clusters(self, X):
        """block_clustering.py:17-26"""
        blocks = np.argmin(pair_distances(X, self.block_centers), axis=1)
        all_idxs = np.arange(len(X))
        return [all_idxs[blocks == i] for i in range(self.n_blocks)]

    def neighbors_literal(self, diag_connections=True):
        """block_clustering.py:28-45 exactly as written.  FRAGILE: ``cc[cc > 0]`` keeps
        self-distances that the a^2-2ab+b^2 formula returns as ~1e-9 instead of 0, so with today's numpy
        the 'second smallest distance' can collapse onto the axis distance and the diagonal edges are
        lost.  The published objectives are reproduced only with the intended 8-neighbourhood below."""
        if len(self.block_centers) <= 1:
            return []
        cd = pair_distances(self.block_centers, self.block_centers)
        with np.errstate(invalid="ignore"):
            cc = cd.flatten()
            cc = cc[cc > 0]
            min_dist = np.min(cc) + 1e-6
            diag_dist = np.min(cc[cc > min_dist]) + 1e-6
        connect = diag_dist if diag_connections else min_dist
        return [(i, j) for i in range(self.n_blocks) for j in range(i) if cd[i, j] < connect]

    def neighbors(self, diag_connections=True):
        """The behaviour the reference intends and its published traces exhibit: connect centres closer
        than the 2nd-smallest distinct centre distance (+1e-6).  Same rule as the literal version with
        the self-distances removed by index instead of by ``> 0`` and distances from exact differences."""
        C = self.block_centers
        if len(C) <= 1:
            return []
        diff = C[:, None, :] - C[None, :, :]
        cd = np.sqrt(np.sum(diff * diff, axis=2))
        off = cd[~np.eye(len(C), dtype=bool)]
        min_dist = np.min(off) + 1e-6
        bigger = off[off > min_dist]
        diag_dist = (np.min(bigger) + 1e-6) if len(bigger) else min_dist
        connect = diag_dist if diag_connections else min_dist
        return [(i, j) for i in range(self.n_blocks) for j in range(i) if cd[i, j] < connect]


def grid_centers(nblocks):
    """gprfopt.py:519-523 (np.linspace needs an int count in py3)."""
    pmax = int(np.ceil(np.sqrt(nblocks)) * 2 + 1)
    pts = np.linspace(0, 1, pmax)[1::2]
    return [np.array((xx, yy)) for xx in pts for yy in pts]


def mcov(X, cov, noise_var):
    """treegp.gp.mcov [recollection]: kernel matrix + noise_var * I."""
    t = VectorTree(X[:1], 1, cov.dfn_str, cov.dfn_params, cov.wfn_str, cov.wfn_params)
    K = t.kernel_matrix(X, X, False)
    K += np.eye(X.shape[0]) * noise_var
    return K


def sample_y(X, cov, noise_var, yd):
    """synthetic.py:103-114 (dense branch)."""
    assert X.shape[0] < 40000
    L = jitchol(mcov(X, cov, noise_var))
    Z = np.random.randn(X.shape[0], yd)
    return np.dot(L, Z)


def sample_synthetic(seed=1, n=400, xd=2, yd=10, lscale=0.1, noise_var=0.01):
    """synthetic.py:139-153 (seed < 1000 branch)."""
    np.random.seed(seed)
    X = np.random.rand(n, xd)
    cov = GPCov(wfn_params=[1.0], dfn_params=[lscale, lscale], dfn_str="euclidean", wfn_str="se")
    y = sample_y(X, cov, noise_var, yd)
    return X, y, cov


class SampledDataRef(object):
    """gprfopt.py:19-74, 172-182"""

    def __init__(self, noise_var=0.01, n=30, ntrain=20, lscale=0.5, obs_std=0.05, yd=10, seed=1):
        self.noise_var, self.n, self.ntrain, self.lscale = noise_var, n, ntrain, lscale
        Xfull, Yfull, cov = sample_synthetic(n=n, noise_var=noise_var, yd=yd, lscale=lscale, seed=seed)
        self.cov = cov
        self.SX, self.SY = Xfull[:ntrain, :], Yfull[:ntrain, :]
        self.Xtest, self.Ytest = Xfull[ntrain:, :], Yfull[ntrain:, :]
        self.block_idxs = None
        self.obs_std = obs_std
        np.random.seed(seed)
        self.X_obs = self.SX + np.random.randn(*self.SX.shape) * obs_std

    def set_centers(self, centers):
        self.centers = np.asarray(centers)
        b = BlockerRef(self.centers)
        self.block_idxs = b.block_clusters(self.X_obs)
        self.reblock = lambda X: b.block_clusters(X)
        self.neighbors = b.neighbors(diag_connections=True)

    def build_gprf(self, X=None, cov=None, local_dist=1e-4, mode="matrix"):
        if X is None:
            X = self.X_obs
        if cov is None:
            cov, noise_var = self.cov, self.noise_var
        else:
            noise_var = cov[0, 0]
            cov = GPCov(wfn_params=[cov[0, 1]], dfn_params=cov[0, 2:], dfn_str="euclidean", wfn_str="se")
        return GPRFRef(X, Y=self.SY, block_fn=self.reblock, block_idxs=self.block_idxs, cov=cov,
                       noise_var=noise_var, neighbor_threshold=local_dist,
                       neighbors=self.neighbors if local_dist < 1.0 else [], mode=mode)

    def x_prior(self, xx):
        flatobs = self.X_obs.flatten()
        n = len(xx)
        r = (xx - flatobs) / self.obs_std
        ll = -.5 * np.sum(r ** 2) - .5 * n * np.log(2 * np.pi * self.obs_std ** 2)
        lderiv = -(xx - flatobs) / (self.obs_std ** 2)
        return ll, lderiv


def cov_prior(c):
    """gprfopt.py:324-331"""
    mean, std = -1, 10
    r = (c - mean) / std
    ll = -.5 * np.sum(r ** 2) - .5 * len(c) * np.log(2 * np.pi * std ** 2)
    return ll, -(c - mean) / (std ** 2)


class ObjectiveRef(object):
    """gprfopt.py:320-417: the L-BFGS-B callback ``lgpllgrad`` (without the np.save checkpoints and the
    log file), as a callable object.  ``__call__(x) -> (-ll, -grad)``."""

    cov_scale = 5.

    def __init__(self, gprf, X0, C0, sdata):
        self.gprf, self.X0, self.C0, self.sdata = gprf, X0, C0, sdata
        self.gradX, self.gradC = (X0 is not None), (C0 is not None)
        x0 = X0.flatten() if self.gradX else np.array(())
        c0 = np.log(C0.flatten()) * self.cov_scale if self.gradC else np.array(())
        self.nx = len(x0)
        self.full0 = np.concatenate([x0, c0])
        self.last_ll = None

    def full_cov(self, C):
        if C.shape[1] == 1:
            FC = np.empty((self.C0.shape[0], 2 + self.sdata.X_obs.shape[1]))
            FC[:, 0] = self.sdata.noise_var
            FC[:, 1] = 1.0
            FC[:, 2:3] = C
            FC[:, 3:4] = C
            return FC
        if C.shape[1] == 4:
            return C
        raise Exception("unrecognized cov param shape")

    def collapse_cov_grad(self, grad_FC):
        if self.C0.shape[1] == 1:
            return grad_FC[:, 2:3] + grad_FC[:, 3:4]
        if self.C0.shape[1] == 4:
            return grad_FC
        raise Exception("unrecognized cov param shape")

    def __call__(self, x):
        xx = x[:self.nx]
        xc = x[self.nx:] / self.cov_scale
        if self.gradX:
            self.gprf.update_X(xx.reshape(self.X0.shape))
        if self.gradC:
            C = np.exp(xc.reshape(self.C0.shape))
            self.gprf.update_covs(self.full_cov(C))
        ll, gX, gC = self.gprf.llgrad(local=True, grad_X=self.gradX, grad_cov=self.gradC)
        if self.gradX:
            prior_ll, prior_grad = self.sdata.x_prior(xx)
            ll += prior_ll
            gX = gX.flatten() + prior_grad
        if self.gradC:
            prior_ll, prior_grad = cov_prior(xc)
            ll += prior_ll
            gC = (np.array(self.collapse_cov_grad(gC)) * C).flatten() + prior_grad
            gC /= self.cov_scale
        self.last_ll = ll
        return -ll, -np.concatenate([gX.flatten(), gC.flatten()])
