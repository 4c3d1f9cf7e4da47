"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the reference's seismic harness around the GPRF path, function by function:
principal-direction tree partition with the longitude wrap (pdtree_clustering.py:4-94), the location and
covariance priors and the L-BFGS-B callback of run_seismic.py (run_seismic.py:68-199, 353-365), without the
file output.  Parity unpinned beyond these sources: the reference ships no fixture for this path (its data file
sorted_isc.npy is absent), so the restatement is checked by construction against the cited lines only.
"""
import numpy as np


class _Leaf(object):
    def __init__(self, idx):
        self.idx = idx


class _Inner(object):
    def __init__(self, split_vec, center, split, left, right):
        self.split_vec, self.center, self.split, self.left, self.right = split_vec, center, split, left, right


class PDTreeRef(object):
    """pdtree_clustering.py:4-77"""

    def __init__(self, X, minsize):
        self.X = X
        self.tree = self._build(np.arange(len(X)), minsize)

    def _build(self, idx, minsize):
        # pdtree_clustering.py:29-51
        if len(idx) < minsize:
            return _Leaf(idx)
        data = self.X[idx]
        dmean = np.mean(data, axis=0)
        data = data - dmean
        ev, evec = np.linalg.eig(np.dot(data.T, data))
        pvec = evec[:, np.argmax(ev)]
        a = np.dot(data, pvec)
        split = np.median(a)
        return _Inner(pvec, dmean, split, self._build(idx[a < split], minsize), self._build(idx[a >= split], minsize))

    def leaf_idx(self):
        # pdtree_clustering.py:53-63
        def rec(node):
            return [node.idx] if isinstance(node, _Leaf) else rec(node.left) + rec(node.right)
        return rec(self.tree)

    def recluster(self, X):
        # pdtree_clustering.py:65-77
        def rec(node, idx):
            if isinstance(node, _Leaf):
                return [idx]
            a = np.dot(X[idx] - node.center, node.split_vec)
            return rec(node.left, idx[a < node.split]) + rec(node.right, idx[a >= node.split])
        return rec(self.tree, np.arange(len(X)))


def pdtree_cluster_ref(X, blocksize=300):
    """pdtree_clustering.py:79-94 (the reference's reblock() wraps XX's longitudes in place and restores them;
    here on a copy: same return value)."""
    X2 = X[:, :2].copy()
    X2[:, 0] = (X2[:, 0] + 22) % 360 - 22
    t = PDTreeRef(X2, minsize=blocksize)

    def reblock(XX):
        Z = np.array(XX[:, :2], dtype=np.float64)
        Z[:, 0] = (Z[:, 0] + 22) % 360 - 22
        return t.recluster(Z)

    return t.leaf_idx(), reblock


def seismic_cov_prior_ref(c):
    """run_seismic.py:68-87"""
    means = np.array((-2.3, 0.0, 3.6, 3.6))
    std = 1.5
    r = (c - means) / std
    ll = -.5 * np.sum(r ** 2) - .5 * len(c) * np.log(2 * np.pi * std ** 2)
    lderiv = (-(c - means) / (std ** 2)).reshape((-1,))
    c = c.reshape((-1,))
    if c[2] > 5:
        ll -= np.exp(70 * (c[2] - 5))
        lderiv[2] -= 70 * np.exp(70 * (c[2] - 5))
    return ll, lderiv


def make_x_prior_ref(means, obs_std):
    """run_seismic.py:353-365"""
    prior_std = obs_std * np.array([.01, .01, 1.])

    def x_prior(X):
        r = (X - means) / prior_std
        r2 = r / prior_std
        n = X.shape[0]
        ll = -.5 * np.sum(r.flatten() ** 2) - .5 * n * (3 * np.log(2 * np.pi) + np.sum(np.log(prior_std ** 2)))
        return ll, -r2.reshape(X.shape)

    return x_prior


class SeismicObjectiveRef(object):
    """run_seismic.py:90-199: ``lgpllgrad`` as a callable (no step files, no log).  X0 is copied before its depth
    column is rescaled (the reference rescales the caller's array in place, run_seismic.py:94-95).  The reference
    indexes gX[:, 2] even when no location gradient was requested, which raises on its (0, 0) placeholder: task
    'cov' is unreachable there; here the depth factor is applied only to a real gradient."""

    depth_scale = 100

    def __init__(self, gprf, X0, C0, cov_prior, x_prior):
        self.gprf, self.cov_prior, self.x_prior = gprf, cov_prior, x_prior
        self.gradX, self.gradC = (X0 is not None), (C0 is not None)
        self.X0 = None
        if self.gradX:
            self.X0 = np.array(X0, dtype=np.float64)
            self.X0[:, 2] /= self.depth_scale
        self.C0 = C0
        x0 = self.X0.flatten() if self.gradX else np.array(())
        c0 = np.log(C0.flatten()) if self.gradC else np.array(())
        self.nx = len(x0)
        self.full0 = np.concatenate([x0, c0])

    def __call__(self, x):
        xx, xc = x[:self.nx], x[self.nx:]
        if self.gradX:
            XX = xx.reshape(self.X0.shape).copy()
            XX[:, 2] *= self.depth_scale
            self.gprf.update_X(XX)
        if self.gradC:
            FC = np.exp(xc.reshape(self.C0.shape))
            FC[0, 1] = 1.0
            if FC[0, 0] > 10.0:
                FC[0, 0] = 10.0
            for k in (2, 3):
                if FC[0, k] > 999:
                    FC[0, k] = 999
                elif FC[0, k] < 1.0:
                    FC[0, k] = 1.0
            self.gprf.update_covs(FC)
        try:
            ll, gX, gC = self.gprf.llgrad(local=True, grad_X=self.gradX, grad_cov=self.gradC)
        except Exception:
            return 1e10, np.random.randn(*x.shape)
        if self.gradX:
            gX = np.array(gX)
            gX[:, 2] *= self.depth_scale
            prior_ll, prior_grad = self.x_prior(XX)
            prior_grad = np.array(prior_grad)
            prior_grad[:, 2] *= self.depth_scale
            ll += prior_ll
            gX = gX.flatten() + prior_grad.flatten()
        if self.gradC:
            prior_ll, prior_grad = self.cov_prior(xc)
            ll += prior_ll
            gC = (np.array(gC) * FC).flatten() + prior_grad
            gC[1] = 0.0
            max_grad = np.max(np.abs(gC[2:]))
            if max_grad > 10:
                gC[2:] *= 2. / (1 + max_grad / 10.)
        return -ll, -np.concatenate([np.asarray(gX).flatten(), np.asarray(gC).flatten()])
