"""Host (numpy) evaluation of the two covariance functions, used only for one-time setup work that the
reference also does once on the host: threshold neighbour discovery (gprf.py:119-150) and synthetic data
sampling (synthetic.py:103-114).  The per-evaluation path never comes here — it is HIP only."""
import numpy as np

EARTH_R_KM = 6371.0  # run_seismic.py:52


def great_circle_km(lon1, lat1, lon2, lat2):
    """run_seismic.py:19-63 (haversine), broadcasting."""
    rlon1, rlat1, rlon2, rlat2 = map(np.radians, (lon1, lat1, lon2, lat2))
    a = np.sin((rlat1 - rlat2) / 2.0) ** 2 + np.cos(rlat1) * np.cos(rlat2) * np.sin((rlon1 - rlon2) / 2.0) ** 2
    return np.radians(np.degrees(2 * np.arcsin(np.sqrt(np.minimum(a, 1.0))))) * EARTH_R_KM


def scaled_distance(X1, X2, cov):
    ls = np.asarray(cov.dfn_params, dtype=np.float64)
    if cov.dfn_str == "euclidean":
        diff = (X1[:, None, :] - X2[None, :, :]) / ls[None, None, :]
        return np.sqrt(np.sum(diff * diff, axis=2))
    if cov.dfn_str == "lld":
        g = great_circle_km(X1[:, None, 0], X1[:, None, 1], X2[None, :, 0], X2[None, :, 1]) / ls[0]
        dz = (X1[:, None, 2] - X2[None, :, 2]) / ls[1]
        return np.sqrt(g * g + dz * dz)
    raise ValueError(cov.dfn_str)


def kernel_matrix(X1, X2, cov):
    """Noise-free k(X1, X2) (VectorTree.kernel_matrix(X1, X2, False), gprf.py:342)."""
    d = scaled_distance(np.asarray(X1, dtype=np.float64), np.asarray(X2, dtype=np.float64), cov)
    sv = cov.wfn_params[0]
    if cov.wfn_str == "se":
        return sv * np.exp(-1.0 * d * d)
    if cov.wfn_str == "matern32":
        s3d = np.sqrt(3.0) * d
        return sv * (1.0 + s3d) * np.exp(-s3d)
    raise ValueError(cov.wfn_str)


def cross_kernel_max(X1, X2, cov):
    """max |k(X1, X2)| / sv  (gprf.py:141-142)"""
    return float(np.max(np.abs(kernel_matrix(X1, X2, cov) / cov.wfn_params[0])))
