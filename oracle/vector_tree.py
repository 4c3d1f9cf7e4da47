"""ORACLE — ctypes face of oracle/treegp_kernels.c with the method names the reference calls on
``treegp.cover_tree.VectorTree`` (gprf.py:109, 339, 342, 353, 373-374) and a ``GPCov`` record
(gprf.py:163; treegp.gp.GPCov).  Test infrastructure only."""
import ctypes
from collections import namedtuple

import numpy as np

from . import build as _build

GPCov = namedtuple("GPCov", ["wfn_params", "dfn_params", "dfn_str", "wfn_str"])

DIST_IDS = {"euclidean": 0, "lld": 1}
KERN_IDS = {"se": 0, "matern32": 1}

_lib = None
_dp = ctypes.POINTER(ctypes.c_double)


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(_build.build())
        i, d = ctypes.c_int, _dp
        L.tg_kernel_matrix.argtypes = [d, i, d, i, i, i, d, i, d, i, d]
        L.tg_kernel_matrix.restype = None
        L.tg_kernel_deriv_wrt_xi_row.argtypes = [d, i, i, i, i, i, d, i, d, d]
        L.tg_kernel_deriv_wrt_xi_row.restype = None
        L.tg_kernel_deriv_wrt_xi_allrows.argtypes = [d, i, i, i, i, d, i, d, d]
        L.tg_kernel_deriv_wrt_xi_allrows.restype = None
        L.tg_kernel_deriv_wrt_i.argtypes = [d, i, d, i, i, i, i, d, i, d, d, d]
        L.tg_kernel_deriv_wrt_i.restype = None
        L.tg_dist_km_pub.argtypes = [ctypes.c_double] * 4
        L.tg_dist_km_pub.restype = ctypes.c_double
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class VectorTree(object):
    """Only the dense-matrix entry points the GPRF path uses; the cover tree itself is not needed."""

    def __init__(self, dummy_X, _ignored, dfn_str, dfn_params, wfn_str, wfn_params):
        self.dist_id = DIST_IDS[dfn_str]
        self.kern_id = KERN_IDS[wfn_str]
        self.dfn_params = _c(np.asarray(dfn_params, dtype=np.float64).ravel())
        self.wfn_params = _c(np.asarray(wfn_params, dtype=np.float64).ravel())

    def kernel_matrix(self, X1, X2, distance_only):
        X1, X2 = _c(X1), _c(X2)
        out = np.empty((X1.shape[0], X2.shape[0]))
        lib().tg_kernel_matrix(_p(X1), X1.shape[0], _p(X2), X2.shape[0], X1.shape[1], self.dist_id,
                               _p(self.dfn_params), self.kern_id, _p(self.wfn_params),
                               1 if distance_only else 0, _p(out))
        return out

    def kernel_deriv_wrt_xi_row(self, X, p, i, out):
        X = _c(X)
        assert out.flags.c_contiguous and out.dtype == np.float64
        lib().tg_kernel_deriv_wrt_xi_row(_p(X), X.shape[0], X.shape[1], int(p), int(i), self.dist_id,
                                         _p(self.dfn_params), self.kern_id, _p(self.wfn_params), _p(out))

    def kernel_deriv_wrt_xi_allrows(self, X, i):
        X = _c(X)
        out = np.empty((X.shape[0], X.shape[0]))
        lib().tg_kernel_deriv_wrt_xi_allrows(_p(X), X.shape[0], X.shape[1], int(i), self.dist_id,
                                             _p(self.dfn_params), self.kern_id, _p(self.wfn_params), _p(out))
        return out

    def kernel_deriv_wrt_i(self, X1, X2, i, _one, dists):
        X1, X2, dists = _c(X1), _c(X2), _c(dists)
        out = np.empty((X1.shape[0], X2.shape[0]))
        lib().tg_kernel_deriv_wrt_i(_p(X1), X1.shape[0], _p(X2), X2.shape[0], X1.shape[1], int(i),
                                    self.dist_id, _p(self.dfn_params), self.kern_id, _p(self.wfn_params),
                                    _p(dists), _p(out))
        return out


def dist_km(loc1, loc2):
    """run_seismic.py:53-63"""
    return lib().tg_dist_km_pub(loc1[0], loc1[1], loc2[0], loc2[1])
