"""Threshold neighbour discovery on the device (gprf.py:119-150): ``GPRF.compute_neighbors`` = host
pruning by geometry (gprf_amd/neighbors.py) + ``gprf_pair_kernel_max`` (kernel k_pair_max: cross-kernel maximum of a
candidate block pair, early exit).  The final list must be the oracle's exhaustive double loop's, pair for pair, in the
same order — on the C4 grid (841 blocks, SE kernel, threshold 1e-3 and 0.6) and on the seismic stand-in (great-circle /
Matern-3/2, threshold 0.6 and 1e-3); the exact maxima agree with the oracle's cross-kernel matrices to 1e-13."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_list(X, blocks, dfn, wfn, sv, ls, thr, only=None):
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov as OC
    ref = GPRFRef(X, np.zeros((len(X), 1)), None, OC([sv], ls, dfn, wfn), 0.1, block_idxs=blocks, neighbors=[])
    if only is None:
        ref.compute_neighbors(threshold=thr)
        return [(int(i), int(j)) for (i, j) in ref.neighbors]
    return [float(np.max(np.abs(ref.kernel(X[blocks[i]], X2=X[blocks[j]]) / sv))) for (i, j) in only]


@pytest.mark.parametrize("thr", [1e-3, 0.6])
def test_c4_grid_neighbours_equal_oracle(thr):
    """n = 80000 points on BASELINE configs[3]'s 29 x 29 grid (lscale 0.02): 353 k block pairs, a few thousand candidates;
    the oracle's exhaustive loop is restricted to the candidates + a random sample of pruned pairs (all 353 k cross
    matrices would take the CPU minutes), whose maxima must all be at or below the threshold."""
    from gprf_amd import GPCov, Blocker, grid_centers
    from gprf_amd.gprf import GPRF
    from gprf_amd.neighbors import candidate_block_pairs
    rng = np.random.RandomState(0)
    n = 80000
    X = rng.rand(n, 2)
    Y = np.zeros((n, 1))
    b = Blocker(grid_centers(800))
    blocks = [np.asarray(v) for v in b.block_clusters(X)]
    cov = GPCov([1.0], [0.02, 0.02], "euclidean", "se")
    g = GPRF(X, Y, None, cov, 0.01, block_idxs=blocks, neighbor_threshold=thr)
    cand = candidate_block_pairs(X, blocks, cov, thr)
    assert len(cand) < 12000
    mx = _oracle_list(X, blocks, "euclidean", "se", 1.0, [0.02, 0.02], thr, only=cand)
    want = [c for c, m in zip(cand, mx) if m > thr]
    assert g.neighbors == want and len(want) > 2000
    # the device's exact maxima for the candidates
    from gprf_amd.gprf import _csr_from_block_idxs
    ptr, pts = _csr_from_block_idxs(blocks)
    keep, dmx = g._ctx.pair_kernel_max(X, ptr, pts, thr, cand, want_max=True)
    assert np.allclose(dmx, mx, rtol=1e-13, atol=1e-300)
    assert [c for c, k in zip(cand, keep) if k] == want
    # pruned pairs really are below the threshold
    cs = set(cand)
    pruned = []
    while len(pruned) < 300:
        i, j = sorted(rng.randint(0, len(blocks), 2), reverse=True)
        if i != j and (i, j) not in cs:
            pruned.append((int(i), int(j)))
    assert max(_oracle_list(X, blocks, "euclidean", "se", 1.0, [0.02, 0.02], thr, only=pruned)) <= thr
    g.close()


@pytest.mark.parametrize("thr,ls", [(0.6, [40.0, 40.0]), (1e-3, [40.0, 40.0]), (0.3, [150.0, 20.0])])
def test_seismic_stand_in_neighbours_equal_oracle(thr, ls):
    """("lld", "matern32") on the stand-in catalogue with principal-direction-tree blocks (run_seismic.py:299-301,375):
    the oracle's exhaustive list, including an empty block and two interleaved blocks."""
    from gprf_amd import GPCov, seismic
    from gprf_amd.gprf import GPRF
    X = seismic.synthetic_events(4000, seed=2)
    Y = np.zeros((len(X), 1))
    blocks, _ = seismic.pdtree_cluster(X, blocksize=210)
    blocks = [np.asarray(b) for b in blocks]
    blocks[5] = np.zeros(0, dtype=np.int64)
    blocks[2], blocks[9] = np.concatenate([blocks[2][::2], blocks[9][::2]]), np.concatenate([blocks[2][1::2], blocks[9][1::2]])
    g = GPRF(X, Y, None, GPCov([0.7], ls, "lld", "matern32"), 0.1, block_idxs=blocks, neighbor_threshold=thr)
    want = _oracle_list(X, blocks, "lld", "matern32", 0.7, ls, thr)
    assert g.neighbors == want and len(want) > 0
    g.close()
