"""Synthetic benchmark inputs with the reference's recipe (seeds, shapes, RNG stream):
``synthetic.sample_synthetic`` / ``sample_y`` (synthetic.py:103-114, 139-153) and
``gprfopt.SampledData`` (gprfopt.py:19-74, 172-182).  Not part of the accelerated path; it only
defines what the path is fed.  The N x N prior Cholesky may run on the GPU through torch (plumbing).
"""
import os

import numpy as np

from .blocking import Blocker
from .cov import GPCov
from .neighbors import great_circle_km


def prior_kernel_matrix(X1, X2, cov):
    """The covariance the synthetic outputs are DRAWN from (treegp's ``mcov``, synthetic.py:108): noise-free k(X1, X2)
    for ("euclidean" | "lld") x ("se" | "matern32"), plain numpy.  Part of the input recipe (what the path is fed) — the
    GPRF objective itself is never evaluated with it."""
    X1, X2 = np.asarray(X1, dtype=np.float64), np.asarray(X2, dtype=np.float64)
    ls = np.asarray(cov.dfn_params, dtype=np.float64)
    if cov.dfn_str == "euclidean":
        diff = (X1[:, None, :] - X2[None, :, :]) / ls[None, None, :]
        d = np.sqrt(np.sum(diff * diff, axis=2))
    elif cov.dfn_str == "lld":
        g = great_circle_km(X1[:, None, 0], X1[:, None, 1], X2[None, :, 0], X2[None, :, 1]) / ls[0]
        dz = (X1[:, None, 2] - X2[None, :, 2]) / ls[1]
        d = np.sqrt(g * g + dz * dz)
    else:
        raise ValueError(cov.dfn_str)
    sv = cov.wfn_params[0]
    if cov.wfn_str == "se":
        return sv * np.exp(-1.0 * d * d)
    if cov.wfn_str == "matern32":
        s3d = np.sqrt(3.0) * d
        return sv * (1.0 + s3d) * np.exp(-s3d)
    raise ValueError(cov.wfn_str)


def _prior_cholesky_times_z(X, cov, noise_var, Z, use_gpu):
    n = X.shape[0]
    if use_gpu:
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        Xd = torch.as_tensor(X, dtype=torch.float64, device=dev)
        ls = torch.as_tensor(np.asarray(cov.dfn_params, dtype=np.float64), device=dev)
        K = torch.empty((n, n), dtype=torch.float64, device=dev)
        step = 4096
        for s in range(0, n, step):
            diff = (Xd[s:s + step, None, :] - Xd[None, :, :]) / ls
            d = torch.sqrt((diff * diff).sum(dim=2))
            K[s:s + step] = cov.wfn_params[0] * torch.exp(-1.0 * d * d)
            del diff, d
        K.diagonal().add_(noise_var)
        if n > 32768:
            # BASELINE config 4 (N = 80500: 52 GB of fp64): factor in place, panel by panel, out of GEMMs whose
            # operands stay below 2^31 elements (a second N x N buffer and the library routine's index range are
            # both avoided); only the lower triangle is ever read
            L = _cholesky_lower_inplace(K)
        else:
            L = torch.linalg.cholesky(K)
            del K
        Zd = torch.as_tensor(Z, dtype=torch.float64, device=dev)
        Y = torch.empty((n, Zd.shape[1]), dtype=torch.float64, device=dev)
        for s in range(0, n, 8192):                       # row panels of the lower triangle: no N x N temporary
            Y[s:s + 8192] = torch.tril(L[s:s + 8192, :s + 8192], diagonal=s) @ Zd[:s + 8192]
        Y = Y.cpu().numpy()
        del L
        torch.cuda.empty_cache()
        return Y
    from scipy.linalg import lapack
    K = np.empty((n, n))
    step = 2048
    for s in range(0, n, step):
        K[s:s + step] = prior_kernel_matrix(X[s:s + step], X, cov)
    K[np.diag_indices(n)] += noise_var
    L, info = lapack.dpotrf(K, lower=1, overwrite_a=1)
    if info != 0:
        raise np.linalg.LinAlgError("prior covariance not positive definite")
    return np.dot(L, Z)


def _cholesky_lower_inplace(A, nb=2048, wide=16384):
    """Right-looking blocked Cholesky of the symmetric positive definite torch matrix ``A`` (fp64, on the GPU),
    overwriting its lower triangle with L (A = L L^T); the strict upper triangle is left as it was."""
    import torch
    n = A.shape[0]
    for k in range(0, n, nb):
        e = min(k + nb, n)
        Lkk = torch.linalg.cholesky(A[k:e, k:e])
        A[k:e, k:e] = Lkk
        if e == n:
            break
        # panel: P = A[e:, k:e] Lkk^-T
        P = torch.linalg.solve_triangular(Lkk, A[e:, k:e].mT, upper=False).mT.contiguous()
        A[e:, k:e] = P
        # trailing update of the lower triangle, one wide block column at a time (diagonal block downwards)
        for j in range(e, n, wide):
            je = min(j + wide, n)
            A[j:, j:je].addmm_(P[j - e:], P[j - e:je - e].mT, alpha=-1.0)
        del P
    return A


def sample_synthetic(seed=1, n=400, xd=2, yd=10, lscale=0.1, noise_var=0.01, use_gpu=False):
    """synthetic.py:139-153 (seed < 1000): X ~ U[0,1]^xd, Y = chol(K + nv I) Z with Z drawn from the same
    RNG stream right after X (no reseed).  Dense only (the reference switches to a sparse CHOLMOD
    approximation at n >= 40000, synthetic.py:106, which is not reproduced)."""
    np.random.seed(seed)
    X = np.random.rand(n, xd)
    cov = GPCov(wfn_params=[1.0], dfn_params=[lscale] * xd, dfn_str="euclidean", wfn_str="se")
    Z = np.random.randn(n, yd)
    Y = _prior_cholesky_times_z(X, cov, noise_var, Z, use_gpu)
    return X, Y, cov


class SampledData(object):
    """gprfopt.py:19-74, 172-182."""

    def __init__(self, noise_var=0.01, n=30, ntrain=20, lscale=0.5, obs_std=0.05, yd=10, seed=1, use_gpu=False,
                 cache_dir=None):
        self.noise_var, self.n, self.ntrain, self.lscale = noise_var, n, ntrain, lscale
        Xfull = Yfull = None
        cache = None
        if cache_dir is not None:
            os.makedirs(cache_dir, exist_ok=True)
            cache = os.path.join(cache_dir, "%d_%d_%.6f_%d_%d_%.4f.npz" % (n, ntrain, lscale, yd, seed, noise_var))
            if os.path.exists(cache):
                z = np.load(cache)
                Xfull, Yfull = z["X"], z["Y"]
                cov = GPCov(wfn_params=[1.0], dfn_params=[lscale, lscale], dfn_str="euclidean", wfn_str="se")
        if Xfull is None:
            Xfull, Yfull, cov = sample_synthetic(n=n, noise_var=noise_var, yd=yd, lscale=lscale, seed=seed,
                                                 use_gpu=use_gpu)
            if cache is not None:
                np.savez(cache, X=Xfull, Y=Yfull)
        self.cov = cov
        self.SX, self.SY = Xfull[:ntrain, :], np.ascontiguousarray(Yfull[:ntrain, :])
        self.Xtest, self.Ytest = Xfull[ntrain:, :], Yfull[ntrain:, :]
        self.block_idxs = None
        self.obs_std = obs_std
        np.random.seed(seed)
        self.X_obs = self.SX + np.random.randn(*self.SX.shape) * obs_std

    def set_centers(self, centers):
        """gprfopt.py:41-46"""
        self.centers = np.asarray(centers)
        b = Blocker(self.centers)
        self.blocker = b
        self.block_idxs = b.block_clusters(self.X_obs)
        self.reblock = b.block_clusters      # bound method: GPRF.update_X recognises it and re-blocks in C
        self.neighbors = b.neighbors(diag_connections=True)

    def model_hypers(self, theta_row=None):
        """(GPCov, noise variance) of a model over this data: the sampling hyper-parameters, or the row
        [noise_var, signal_var, lengthscales...] a driver passes around (gprf.py:160-164)."""
        if theta_row is None:
            return self.cov, self.noise_var
        theta = np.asarray(theta_row, dtype=np.float64)
        if theta.ndim != 2 or theta.shape[0] != 1:
            raise Exception("invalid cov params %s" % (theta_row,))
        return GPCov(wfn_params=[theta[0, 1]], dfn_params=theta[0, 2:], dfn_str="euclidean", wfn_str="se"), theta[0, 0]

    def build_gprf(self, X=None, cov=None, local_dist=1e-4, **kw):
        """The model the reference's driver optimises (gprfopt.py:55-74): grid blocks re-evaluated at every ``update_X``,
        the blocker's neighbour pairs when ``local_dist`` < 1 and none (independent local GPs) otherwise."""
        from .gprf import GPRF
        gpcov, noise_var = self.model_hypers(cov)
        pairs = self.neighbors if local_dist < 1.0 else []
        return GPRF(self.X_obs if X is None else X, Y=self.SY, block_fn=self.reblock, block_idxs=self.block_idxs,
                    cov=gpcov, noise_var=noise_var, kernelized=False, neighbor_threshold=local_dist, neighbors=pairs, **kw)

    def x_prior(self, xx):
        """log density and gradient of the location prior N(X_obs, obs_std^2 I) at the flat vector ``xx``
        (gprfopt.py:172-182), evaluated by the library's host function (``gprf_x_prior``)."""
        from . import _capi
        return _capi.x_prior(xx, self.X_obs, self.obs_std)
