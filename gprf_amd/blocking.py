"""Block partition helpers with the reference's semantics (host side; they decide the unit list).

Mirrors ``/root/reference/block_clustering.py``: pair_distances :4-5, Blocker :7-45 (duplicated at
gprf.py:33-74) and ``gprfopt.grid_centers`` (gprfopt.py:519-523).
"""
import numpy as np


def pair_distances(Xi, Xj):
    """block_clustering.py:4-5.  The a^2 - 2ab + b^2 form is kept because nearest-centre ties and
    near-ties must resolve exactly as in the reference."""
    return np.sqrt(np.outer(np.sum(Xi ** 2, axis=1), np.ones((Xj.shape[0]),)) - 2 * np.dot(Xi, Xj.T)
                   + np.outer((np.ones(Xi.shape[0]),), np.sum(Xj ** 2, axis=1)))


class Blocker(object):
    """Nearest-centre blocks and the grid neighbour graph."""

    def __init__(self, block_centers):
        self.block_centers = np.asarray(block_centers, dtype=np.float64)
        self.n_blocks = len(block_centers)

    def get_block(self, X_new):
        return int(np.argmin([np.linalg.norm(X_new - c) for c in self.block_centers]))

    def block_assignment(self, X):
        """argmin_c ||x - c|| per point (block_clustering.py:18-19)."""
        with np.errstate(invalid="ignore"):
            return np.argmin(pair_distances(X, self.block_centers), axis=1)

    def block_assignment_fast(self, X):
        """Same assignment through the C library's host helper (gprf_nearest_center): the reference's radicand
        and tie rule in plain double arithmetic instead of numpy temporaries + BLAS.  Can differ from
        ``block_assignment`` only for a point whose two nearest centres are equidistant to the last bit."""
        from . import _capi
        return _capi.nearest_center(X, self.block_centers)

    def block_clusters(self, X):
        """block_clustering.py:17-26 -> list of index arrays, ascending inside each block."""
        blocks = self.block_assignment(X)
        order = np.argsort(blocks, kind="stable")
        counts = np.bincount(blocks, minlength=self.n_blocks)
        return np.split(order, np.cumsum(counts)[:-1])

    def neighbors(self, diag_connections=True):
        """block_clustering.py:28-45: connect centres closer than the smallest (axis) or second smallest
        (diagonal) distinct centre distance, + 1e-6.  The reference removes self-distances with
        ``cc[cc > 0]``, which is fragile (the a^2-2ab+b^2 self-distance can come out ~1e-9 and then
        masquerades as the minimum) — its published runs have the full
        8-neighbourhood, which is what this returns: self-distances are dropped by index and the
        distances are formed from exact coordinate differences."""
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
    """gprfopt.py:519-523: a g x g grid on the unit square, g = ceil(sqrt(nblocks)) (800 -> 841)."""
    pmax = int(np.ceil(np.sqrt(nblocks)) * 2 + 1)
    pts = np.linspace(0, 1, pmax)[1::2]
    return [np.array((xx, yy)) for xx in pts for yy in pts]
