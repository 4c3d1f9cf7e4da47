"""ORACLE — CPU restatement of the reference's GPRF objective/gradient path.  Test infrastructure only;
``gprf_amd`` never imports this.

Follows ``/root/reference/gprf.py``:
  GPRF.__init__ :85-117, compute_neighbors :119-150, compute_neighbor_count :152-157,
  update_covs :160-167, update_X :169-174, llgrad :206-296, llgrad_unary :299-308,
  llgrad_joint :310-330, kernel :333-343, dKdx :345-360, dKdi :362-375, gaussian_llgrad :496-591.

``mode="rows"`` keeps the reference's shape exactly: one Python->C call per (point, coordinate)
(gprf.py:556-561).  ``mode="matrix"`` evaluates the same per-entry formulas for all rows inside one
C call; the arithmetic per entry is identical, only the interpreter crossings are removed.
"""
from collections import defaultdict

import numpy as np

from .linalg_ref import dpotrs, pdinv
from .vector_tree import GPCov, VectorTree


def symmetrize_neighbors(neighbors):
    """gprf.py:76-81"""
    nd = defaultdict(set)
    for (i, j) in neighbors:
        nd[i].add(j)
        nd[j].add(i)
    return nd


class GPRFRef(object):

    def __init__(self, X, Y, block_fn, cov, noise_var, neighbor_threshold=1e-3, block_idxs=None,
                 neighbors=None, mode="matrix"):
        self.X = X
        self.Y = Y
        if block_idxs is None:
            block_idxs = block_fn(X)
        self.block_idxs = block_idxs
        self.block_fn = block_fn
        self.n_blocks = len(block_idxs)
        self.cov = cov
        self.noise_var = noise_var
        self.mode = mode
        self._make_tree()
        if neighbors is not None:
            self.neighbors = neighbors
        else:
            self.compute_neighbors(threshold=neighbor_threshold)
        self.compute_neighbor_count()
        self.neighbor_dict = symmetrize_neighbors(self.neighbors)
        self.neighbor_threshold = neighbor_threshold

    def _make_tree(self):
        dummy = np.zeros((1, self.X.shape[1]))
        c = self.cov
        self.predict_tree = VectorTree(dummy, 1, c.dfn_str, c.dfn_params, c.wfn_str, c.wfn_params)

    # -- gprf.py:119-150
    def compute_neighbors(self, threshold=1e-3):
        nbrs = []
        if threshold == 1.0:
            self.neighbors = nbrs
            return
        wfn_var = self.cov.wfn_params[0]
        for i in range(self.n_blocks):
            X1 = self.X[self.block_idxs[i]]
            for j in range(i):
                X2 = self.X[self.block_idxs[j]]
                if X1.shape[0] == 0 or X2.shape[0] == 0:
                    continue
                maxk = np.max(np.abs(self.kernel(X1, X2=X2) / wfn_var))
                if maxk > threshold:
                    nbrs.append((i, j))
        self.neighbors = nbrs

    # -- gprf.py:152-157
    def compute_neighbor_count(self):
        nc = defaultdict(int)
        for (i, j) in self.neighbors:
            nc[i] += 1
            nc[j] += 1
        self.neighbor_count = nc

    # -- gprf.py:160-167
    def update_covs(self, covs):
        nv, sv = covs[0, :2]
        lscales = covs[0, 2:]
        self.cov = GPCov(wfn_params=[sv, ], dfn_params=lscales, dfn_str=self.cov.dfn_str,
                         wfn_str=self.cov.wfn_str)
        self.noise_var = nv
        self._make_tree()

    # -- gprf.py:169-174
    def update_X(self, new_X, update_blocks=True, recompute_neighbors=False):
        self.X = new_X
        if self.block_fn is not None:
            self.block_idxs = self.block_fn(new_X)
        if recompute_neighbors:
            self.compute_neighbors(threshold=self.neighbor_threshold)

    # -- gprf.py:182-204
    def subset_llgrad(self, blocks):
        block_set = set(blocks)
        neighbors_in_set = [(i, j) for (i, j) in self.neighbors if i in block_set and j in block_set]
        local_neighbor_counts = defaultdict(int)
        for (i, j) in neighbors_in_set:
            local_neighbor_counts[i] += 1
            local_neighbor_counts[j] += 1
        unary_lls = [self.llgrad_unary(i, grad_X=False, grad_cov=False)[0] for i in blocks]
        pair_lls = [self.llgrad_joint(i, j, grad_X=False, grad_cov=False)[0] for (i, j) in neighbors_in_set]
        ll = np.sum(pair_lls)
        ll += np.sum([(1 - local_neighbor_counts[blocks[i]]) * ull for (i, ull) in enumerate(unary_lls)])
        return ll

    # -- gprf.py:206-296 (serial branch)
    def llgrad(self, parallel=False, local=True, **kwargs):
        if local:
            neighbors = self.neighbors
            neighbor_count = self.neighbor_count
        else:
            neighbors = [(i, j) for i in range(self.n_blocks) for j in range(i)]
            neighbor_count = dict([(i, self.n_blocks - 1) for i in range(self.n_blocks)])

        unaries = [self.llgrad_unary(i, **kwargs) for i in range(self.n_blocks)]
        pairs = [self.llgrad_joint(i, j, **kwargs) for (i, j) in neighbors]

        unary_lls, unary_gX, unary_gC = zip(*unaries)
        if len(pairs) > 0:
            pair_lls, pair_gX, pair_gC = zip(*pairs)
        else:
            pair_lls, pair_gX, pair_gC = [], [], []

        ll = np.sum(pair_lls)
        ll += np.sum([(1 - neighbor_count[i]) * ull for (i, ull) in enumerate(unary_lls)])

        if kwargs.get("grad_X", False):
            gradX = np.zeros(self.X.shape)
            for i in range(self.n_blocks):
                gradX[self.block_idxs[i], :] -= (neighbor_count[i] - 1) * unary_gX[i]
            for pair_idx, (i, j) in enumerate(neighbors):
                idxs, jdxs = self.block_idxs[i], self.block_idxs[j]
                ni = len(idxs)
                gradX[idxs] += pair_gX[pair_idx][:ni]
                gradX[jdxs] += pair_gX[pair_idx][ni:]
        else:
            gradX = np.zeros((0, 0))

        if kwargs.get("grad_cov", False):
            ncov = 2 + len(self.cov.dfn_params)
            gradCov = np.sum(pair_gC, axis=0) if len(pair_gC) > 0 else np.zeros(ncov)
            gradCov = gradCov - np.sum([(neighbor_count[i] - 1) * unary_gC[i]
                                        for i in range(self.n_blocks)], axis=0)
            gradCov = gradCov.reshape((1, -1))
        else:
            gradCov = np.zeros((0, 0))
        return ll, gradX, gradCov

    # -- gprf.py:299-308
    def llgrad_unary(self, i, **kwargs):
        idxs = self.block_idxs[i]
        return self.gaussian_llgrad(self.X[idxs], self.Y[idxs], **kwargs)

    # -- gprf.py:310-330
    def llgrad_joint(self, i, j, **kwargs):
        idxs, jdxs = self.block_idxs[i], self.block_idxs[j]
        X = np.vstack([self.X[idxs], self.X[jdxs]])
        Y = np.vstack([self.Y[idxs], self.Y[jdxs]])
        return self.gaussian_llgrad(X, Y, **kwargs)

    # -- gprf.py:333-343: noise added only when X2 is None
    def kernel(self, X, X2=None):
        if X2 is None:
            K = self.predict_tree.kernel_matrix(X, X, False)
            K += np.eye(X.shape[0]) * self.noise_var
        else:
            K = self.predict_tree.kernel_matrix(X, X2, False)
        return K

    # -- gprf.py:345-355 (return_vec branch)
    def dKdx(self, X, p, i, dKv):
        self.predict_tree.kernel_deriv_wrt_xi_row(X, p, i, dKv)
        dKv[p] = 0
        return dKv

    # -- gprf.py:362-375
    def dKdi(self, X1, i):
        cov = self.cov
        if i == 0:
            return np.eye(X1.shape[0])
        if i == 1:
            if len(cov.wfn_params) != 1:
                raise ValueError("gradient computation currently assumes just a single scaling "
                                 "parameter for weight function, but currently wfn_params=%s" % (cov.wfn_params,))
            return self.kernel(X1, X1) / cov.wfn_params[0]
        dc = self.predict_tree.kernel_matrix(X1, X1, True)
        return self.predict_tree.kernel_deriv_wrt_i(X1, X1, i - 2, 1, dc)

    def _dK_all_rows(self, X, i):
        """mode='matrix': all rows of d k(x_p, x_q)/d x_p[i] with the diagonal zeroed; entry-wise the same
        C routine as dKdx, looped over p inside C."""
        return self.predict_tree.kernel_deriv_wrt_xi_allrows(X, i)

    # -- gprf.py:496-591
    def gaussian_llgrad(self, X, Y, grad_X=False, grad_cov=False, return_parts=False):
        n, dx = X.shape
        dy = Y.shape[1]
        gradX = np.zeros(())
        gradC = np.zeros(())
        if n == 0:
            if grad_X:
                gradX = np.zeros(X.shape)
            if grad_cov:
                gradC = np.zeros((2 + len(self.cov.dfn_params),))
            return 0.0, gradX, gradC

        K = self.kernel(X)
        prec, L, Lprec, logdet = pdinv(K)
        Alpha, _ = dpotrs(L, Y, lower=1)

        ll = -.5 * np.sum(Y * Alpha)
        ll += -.5 * dy * logdet
        ll += -.5 * dy * n * np.log(2 * np.pi)

        if grad_X:
            gradX = np.zeros((n, dx))
            if self.mode == "rows":
                dcv = np.zeros((n,), dtype=np.float64)
                dK = [np.zeros(K.shape) for _ in range(dx)]
                for p in range(n):
                    for i in range(dx):
                        self.dKdx(X, p, i, dKv=dcv)
                        dK[i][p, :] = dcv
            else:
                dK = [self._dK_all_rows(X, i) for i in range(dx)]
            for i in range(dx):
                dKi = dK[i]
                d_logdet = -dy * np.sum(np.multiply(prec, dKi), axis=1)
                gradX[:, i] = d_logdet
                dK_alpha = np.dot(dKi, Alpha)
                gradX[:, i] += np.sum(dK_alpha * Alpha, axis=1)

        if grad_cov:
            ncov = 2 + len(self.cov.dfn_params)
            gradC = np.zeros((ncov,))
            for i in range(ncov):
                dKdi = self.dKdi(X, i)
                dlldi = .5 * np.sum(np.multiply(Alpha, np.dot(dKdi, Alpha)))
                dlldi -= .5 * dy * np.sum(np.sum(np.multiply(prec, dKdi)))
                gradC[i] = dlldi

        if return_parts:
            return ll, gradX, gradC, dict(K=K, L=L, prec=prec, Alpha=Alpha, logdet=logdet)
        return ll, gradX, gradC
