"""The device's re-partition kernels against the partitions THE REFERENCE ITSELF produces (tests/golden/ref_partitions.npz,
written in the build container by tests/golden/make_reference_fixtures.py from /root/reference/pdtree_clustering.py:4-94 and
block_clustering.py:7-45): `k_route` (split-tree descent with the longitude wrap) and `k_assign` (nearest centre).
Index work: bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_partitions.npz")
MOVES = ("same", "near", "far", "emptied", "cut")


@pytest.fixture(scope="module")
def ref():
    return np.load(FIX)


def unpack(ref, key):
    ptr, idx = ref[key + "_ptr"], ref[key + "_idx"]
    return [idx[ptr[i]:ptr[i + 1]] for i in range(len(ptr) - 1)]


@pytest.mark.parametrize("bs", [120, 210])
def test_device_tree_routing_equals_the_references_reblock(ref, bs):
    """GPRF.update_X with pdtree_cluster's reblock routes on the device (gprf_set_split_tree + k_route): the partition the
    next evaluation uses is the reference's `reblock` output, list by list — build points, moved points, a leaf left empty,
    events pushed across the -22 degree cut — and the evaluation on it is finite."""
    from gprf_amd import GPCov, seismic
    from gprf_amd.gprf import GPRF
    X = ref["pd_X"]
    Y = np.random.RandomState(0).randn(len(X), 3)
    cov = GPCov([1.0], [150.0, 150.0], "lld", "matern32")
    blocks, reblock = seismic.pdtree_cluster(X, blocksize=bs)
    want = unpack(ref, "pd%d_leaf" % bs)
    assert len(blocks) == len(want) and all(np.array_equal(a, b) for a, b in zip(blocks, want))
    g = GPRF(X, Y, reblock, cov, 0.1, neighbor_threshold=0.6)
    for mv in MOVES:
        g.update_X(ref["pd%d_%s_X" % (bs, mv)].copy())
        ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
        assert np.isfinite(ll) and np.all(np.isfinite(gX)) and np.all(np.isfinite(gC))
        got, exp = g.block_idxs, unpack(ref, "pd%d_%s" % (bs, mv))
        assert len(got) == len(exp)
        assert all(np.array_equal(a, b) for a, b in zip(got, exp)), mv
    assert g._centers_of is reblock.tree                     # the device path was the one taken
    g.close()


@pytest.mark.parametrize("nb", [4, 100, 841])
def test_device_nearest_centre_equals_the_references_block_clusters(ref, nb):
    """k_assign on the fixture's points (on centres, outside the unit square, exact ties): the reference's
    `Blocker.block_clusters` wherever the nearest centre is unique beyond the last bits; on the three constructed ties one
    of the tied centres (the reference's own choice there hangs on BLAS's rounding of a^2 - 2ab + b^2)."""
    from gprf_amd import Blocker, GPCov
    from gprf_amd.gprf import GPRF
    C, P = ref["bc%d_centers" % nb], ref["bc%d_X" % nb]
    lists = unpack(ref, "bc%d" % nb)
    want = np.empty(len(P), dtype=np.int64)
    for i, idx in enumerate(lists):
        want[idx] = i
    Y = np.random.RandomState(1).randn(len(P), 2)
    b = Blocker(C)
    g = GPRF(P, Y, None, GPCov([1.0], [0.2, 0.2], "euclidean", "se"), 0.01, neighbors=[], block_idxs=lists)
    g._ctx.set_centers(C)
    changed, dev = g._ctx.assign_blocks(P)
    dev = g._ctx.get_block_assignment()
    d = np.sort(np.linalg.norm(P[:, None, :] - C[None], axis=2), axis=1)
    clear = (d[:, 1] - d[:, 0]) > 1e-12
    assert clear.sum() >= len(P) - 3
    assert np.array_equal(dev[clear], want[clear])
    assert np.all(np.linalg.norm(P - C[dev], axis=1) <= d[:, 0] + 1e-12)
    if np.array_equal(dev, want):
        assert not changed                                    # the uploaded partition WAS the reference's
    # ... and through the object: update_X with the Blocker's callable re-partitions on the device
    g2 = GPRF(P, Y, b.block_clusters, GPCov([1.0], [0.2, 0.2], "euclidean", "se"), 0.01, neighbors=b.neighbors())
    P2 = P.copy()
    P2[300:400] = P[400:500]
    g2.update_X(P2)
    ll = g2.llgrad(grad_X=True)[0]
    assert np.isfinite(ll)
    with np.errstate(invalid="ignore"):
        exp = b.block_clusters(P2)
    got = g2.block_idxs
    diff = [i for i in range(len(exp)) if not np.array_equal(got[i], exp[i])]
    moved = set(np.concatenate([got[i] for i in diff] + [exp[i] for i in diff]).tolist()) if diff else set()
    assert moved <= {200, 201, 202}                          # only the constructed ties may sit elsewhere
    g.close(); g2.close()
