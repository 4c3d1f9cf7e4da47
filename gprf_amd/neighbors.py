"""Threshold neighbour discovery (``GPRF.compute_neighbors``, gprf.py:119-150), host half: which block pairs CAN exceed the
threshold.  The reference evaluates the cross-kernel matrix of every pair j < i; here a pair is first pruned by geometry —
both covariance functions decrease with the scaled distance, so a pair whose bounding volumes are already too far apart
cannot qualify — and the surviving candidates (a superset of the answer, in the reference's (i, j < i) order) are decided
on the device by ``gprf_pair_kernel_max`` (kernel k_pair_max): the C4 configuration's 353 k block pairs shrink to a few
thousand candidates.  No covariance matrix is evaluated on the host."""
import numpy as np

EARTH_R_KM = 6371.0  # run_seismic.py:52


def great_circle_km(lon1, lat1, lon2, lat2):
    """run_seismic.py:19-63 (haversine), broadcasting — used for the blocks' bounding caps only."""
    rlon1, rlat1, rlon2, rlat2 = map(np.radians, (lon1, lat1, lon2, lat2))
    a = np.sin((rlat1 - rlat2) / 2.0) ** 2 + np.cos(rlat1) * np.cos(rlat2) * np.sin((rlon1 - rlon2) / 2.0) ** 2
    return np.radians(np.degrees(2 * np.arcsin(np.sqrt(np.minimum(a, 1.0))))) * EARTH_R_KM


def _unit_kernel_of_distance(d, cov):
    """k / sv as a function of the scaled distance (both kernels decrease monotonically in it)."""
    if cov.wfn_str == "se":
        return np.exp(-1.0 * d * d)
    if cov.wfn_str == "matern32":
        s3d = np.sqrt(3.0) * np.where(np.isfinite(d), d, 1e300)     # (an empty block's box is infinitely far)
        return (1.0 + s3d) * np.exp(-s3d)
    raise ValueError(cov.wfn_str)


def candidate_block_pairs(X, block_idxs, cov, threshold):
    """Block pairs (i, j < i), in the reference's loop order, whose largest cross-covariance / sv MAY exceed
    ``threshold``: every pair the exhaustive loop of gprf.py:134-148 would keep is among them.  For the Euclidean
    distance the scaled distance between two blocks' bounding boxes bounds every point pair from below; for "lld" the
    bound is the great-circle distance between two blocks' spherical caps (centre distance minus both angular radii: the
    great-circle distance is a metric) combined with the gap between their depth ranges, less a margin that covers the
    haversine's rounding.  The bound is relaxed by a relative 1e-9 so that rounding differences between this host
    arithmetic and the device's kernel evaluation can never prune a pair the device would keep."""
    X = np.asarray(X, dtype=np.float64)
    nb = len(block_idxs)
    pairs = []
    if threshold == 1.0:
        return pairs
    thr = threshold * (1.0 - 1e-9)
    nonempty = [len(b) > 0 for b in block_idxs]
    if cov.dfn_str == "euclidean":
        ls = np.asarray(cov.dfn_params, dtype=np.float64)[None, :]
        lo = np.full((nb, X.shape[1]), np.inf)
        hi = np.full((nb, X.shape[1]), -np.inf)
        for i, b in enumerate(block_idxs):
            if nonempty[i]:
                Z = X[b] / ls
                lo[i], hi[i] = Z.min(axis=0), Z.max(axis=0)
        for i in range(nb):
            if not nonempty[i] or i == 0:
                continue
            # gap between box i and every box j < i along each axis (0 where they overlap)
            gap = np.maximum(0.0, np.maximum(lo[i][None, :] - hi[:i], lo[:i] - hi[i][None, :]))
            dmin = np.sqrt(np.sum(gap * gap, axis=1)) * (1.0 - 1e-9)
            cand = np.nonzero(_unit_kernel_of_distance(dmin, cov) > thr)[0]
            pairs.extend((i, int(j)) for j in cand if nonempty[j])
        return pairs
    if cov.dfn_str == "lld":
        ls = np.asarray(cov.dfn_params, dtype=np.float64)
        clon, clat, rad = np.zeros(nb), np.zeros(nb), np.full(nb, np.inf)
        zlo, zhi = np.full(nb, np.inf), np.full(nb, -np.inf)
        for i, b in enumerate(block_idxs):
            if nonempty[i]:
                lon, lat = np.radians(X[b, 0]), np.radians(X[b, 1])
                c = np.array([np.mean(np.cos(lat) * np.cos(lon)), np.mean(np.cos(lat) * np.sin(lon)), np.mean(np.sin(lat))])
                if np.linalg.norm(c) > 1e-6:            # (a block spread over the whole globe keeps an infinite cap)
                    c /= np.linalg.norm(c)
                    clon[i], clat[i] = np.degrees(np.arctan2(c[1], c[0])), np.degrees(np.arcsin(np.clip(c[2], -1, 1)))
                    rad[i] = np.max(great_circle_km(clon[i], clat[i], X[b, 0], X[b, 1]))
                zlo[i], zhi[i] = X[b, 2].min(), X[b, 2].max()
        for i in range(1, nb):
            if not nonempty[i]:
                continue
            gc = great_circle_km(clon[i], clat[i], clon[:i], clat[:i]) - rad[i] - rad[:i]
            gc = np.where(np.isfinite(gc), np.maximum(0.0, gc * (1 - 1e-9) - 1e-6), 0.0)
            zgap = np.maximum(0.0, np.maximum(zlo[i] - zhi[:i], zlo[:i] - zhi[i]))
            zgap = np.where(np.isfinite(zgap), zgap * (1 - 1e-9), 0.0)
            dmin = np.sqrt((gc / ls[0]) ** 2 + (zgap / ls[1]) ** 2)
            cand = np.nonzero(_unit_kernel_of_distance(dmin, cov) > thr)[0]
            pairs.extend((i, int(j)) for j in cand if nonempty[j])
        return pairs
    return [(i, j) for i in range(nb) for j in range(i) if nonempty[i] and nonempty[j]]
