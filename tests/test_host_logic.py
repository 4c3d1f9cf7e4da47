"""Host-side logic of the product (no GPU): blocking, neighbour graph, CSR packing, objective transforms,
synthetic recipe — each against the oracle's literal restatement of the reference."""
import numpy as np
import pytest

from gprf_amd import Blocker, grid_centers, pair_distances
from gprf_amd.gprf import _csr_from_block_idxs
from gprf_amd import objective as pobj
from gprf_amd.synthetic import prior_kernel_matrix
from gprf_amd.cov import GPCov
from oracle import harness_ref as H
from oracle.vector_tree import VectorTree


def test_grid_centers_round_up():
    assert len(grid_centers(100)) == 100
    assert len(grid_centers(800)) == 841
    assert len(grid_centers(4)) == 4
    c = np.array(grid_centers(4))
    assert np.allclose(sorted(set(c[:, 0])), [0.25, 0.75])
    assert np.allclose(np.array(grid_centers(9)), np.array(H.grid_centers(9)))


@pytest.mark.parametrize("nb,expected", [(4, 6), (9, 20), (100, 342), (841, 3192)])
def test_neighbors_are_the_8_neighbourhood(nb, expected):
    nbrs = Blocker(grid_centers(nb)).neighbors()
    assert len(nbrs) == expected                  # 342 pairs at 100 blocks, 3192 at 841
    assert all(j < i for (i, j) in nbrs)
    assert nbrs == H.BlockerRef(H.grid_centers(nb)).neighbors()
    g = int(round(np.sqrt(nb)))
    assert len(Blocker(grid_centers(nb)).neighbors(diag_connections=False)) == 2 * g * (g - 1)


def test_block_clusters_match_reference_semantics():
    rng = np.random.RandomState(5)
    X = rng.rand(3000, 2) * 1.2 - 0.1             # some points outside the unit square
    for nb in (4, 9, 100):
        a = Blocker(grid_centers(nb)).block_clusters(X)
        b = H.BlockerRef(H.grid_centers(nb)).block_clusters(X)
        assert len(a) == len(b) == nb
        for u, v in zip(a, b):
            assert np.array_equal(u, v)           # same members, same (ascending) order
    with np.errstate(invalid="ignore"):   # the a^2-2ab+b^2 self-distance can be sqrt(-1e-17) = nan
        assert np.array_equal(pair_distances(X[:5], X[:7]), H.pair_distances(X[:5], X[:7]), equal_nan=True)


def test_fast_nearest_center_equals_numpy_assignment():
    """gprf_nearest_center (C, host) against the numpy restatement, incl. points exactly on centres (where the
    reference's radicand goes negative -> NaN -> argmin picks it) and points outside the unit square."""
    rng = np.random.RandomState(6)
    for nb in (4, 9, 100, 841):
        b = Blocker(grid_centers(nb))
        X = rng.rand(5000, 2) * 1.3 - 0.15
        X[:4] = b.block_centers[:4]
        assert np.array_equal(b.block_assignment_fast(X), b.block_assignment(X))
        assert np.array_equal(b.block_assignment_fast(X), H.BlockerRef(H.grid_centers(nb)).block_clusters and
                              np.argmin(H.pair_distances(X, b.block_centers), axis=1))


def test_csr_packing():
    blocks = [np.array([4, 1]), np.array([], dtype=int), np.array([0, 2, 3])]
    ptr, pts = _csr_from_block_idxs(blocks)
    assert ptr.tolist() == [0, 2, 2, 5] and pts.tolist() == [4, 1, 0, 2, 3]
    assert pts.dtype == np.int32 and ptr.dtype == np.int64
    ptr, pts = _csr_from_block_idxs([np.array([], dtype=int)])
    assert ptr.tolist() == [0, 0] and len(pts) == 0


class _FakeGPRF(object):
    """Stands in for the device object so the callback's algebra can be checked on CPU."""

    def __init__(self, n, dx):
        self.n, self.dx = n, dx
        self.rng = np.random.RandomState(0)

    def update_X(self, X):
        self.X = X

    def update_covs(self, FC):
        self.FC = FC.copy()

    def llgrad(self, local=True, grad_X=False, grad_cov=False, parallel=False):
        ll = -float(np.sum(self.X ** 2)) if grad_X else -1.0
        gX = -2 * self.X if grad_X else np.zeros((0, 0))
        gC = np.arange(1, 5, dtype=float).reshape(1, 4) * (self.FC[0, 2] if grad_cov else 1) if grad_cov else np.zeros((0, 0))
        return ll, gX, gC


class _SD(object):
    def __init__(self, X_obs, obs_std):
        self.X_obs, self.obs_std, self.noise_var = X_obs, obs_std, 0.01

    x_prior = H.SampledDataRef.x_prior


@pytest.mark.parametrize("task", ["x", "xcov", "cov4"])
def test_objective_transforms_match_reference_callback(task):
    rng = np.random.RandomState(1)
    X0 = rng.rand(20, 2)
    sd = _SD(X0 + 0.01 * rng.randn(20, 2), 0.05)
    C0 = {"x": None, "xcov": np.array([[0.3]]), "cov4": np.array([[0.01, 1.0, 0.05, 0.05]])}[task]
    Xarg = None if task == "cov4" else X0
    a = pobj.Objective(_FakeGPRF(20, 2), Xarg, C0, sd)
    b = H.ObjectiveRef(_FakeGPRF(20, 2), Xarg, C0, sd)
    if task == "cov4":
        a.gprf.X = b.gprf.X = X0
    x = a.full0 + 0.01 * rng.randn(len(a.full0))
    fa, ga = a(x)
    fb, gb = b(x)
    # (the library's host functions sum the prior in blocks, numpy pairwise: equal to rounding, not to the bit)
    assert np.isclose(fa, fb, rtol=1e-14, atol=0) and np.allclose(ga, gb, rtol=1e-14, atol=0)
    assert np.allclose(a.gprf.X, b.gprf.X) and (task == "x" or np.allclose(a.gprf.FC, b.gprf.FC, rtol=1e-15))
    assert np.array_equal(a.full0, b.full0)
    assert a.trace[0][0] == 0 and a.step == 1
    assert np.isclose(sum(a.parts), -fa, rtol=1e-14)
    assert np.isclose(pobj.cov_prior(np.array([0.3]))[0], H.cov_prior(np.array([0.3]))[0], rtol=1e-15)
    assert np.allclose(pobj.cov_prior(np.array([0.3, -2.0, 7.0]))[1], H.cov_prior(np.array([0.3, -2.0, 7.0]))[1], rtol=1e-15)


def test_library_host_prior_functions():
    """gprf_x_prior / gprf_hyper_unpack / gprf_hyper_grad (pure host code of the C ABI) against the oracle's restatement of
    gprfopt.py:172-182, 324-355."""
    from gprf_amd import _capi
    rng = np.random.RandomState(2)
    x, xo = rng.rand(10007), rng.rand(10007)
    class _S(object):
        X_obs, obs_std = xo, 0.03
    rll, rg = H.SampledDataRef.x_prior(_S, x)
    ll, g = _capi.x_prior(x, xo, 0.03)
    assert np.isclose(ll, rll, rtol=1e-14) and np.array_equal(g, rg)
    assert _capi.x_prior(x, xo, 0.03, want_grad=False)[1] is None
    th = _capi.hyper_unpack(_capi.HYPER_TIED, 5.0, 0.01, 1.0, 4, np.array([5 * np.log(0.3)]))
    assert np.allclose(th, [0.01, 1.0, 0.3, 0.3], rtol=1e-15)
    th = _capi.hyper_unpack(_capi.HYPER_FULL, 5.0, 0.0, 0.0, 4, 5 * np.log([0.02, 1.5, 0.3, 0.4]))
    assert np.allclose(th, [0.02, 1.5, 0.3, 0.4], rtol=1e-15)
    gC = np.array([3.0, -2.0, 0.7, 1.1])
    z = np.array([5 * np.log(0.3)])
    pll, gz = _capi.hyper_grad(_capi.HYPER_TIED, 5.0, -1.0, 10.0, z, gC)
    rpl, rpg = H.cov_prior(z / 5.0)
    assert np.isclose(pll, rpl, rtol=1e-15) and np.allclose(gz, ((0.7 + 1.1) * 0.3 + rpg) / 5.0, rtol=1e-15)
    with pytest.raises(_capi.GprfHipError):
        _capi.hyper_unpack(7, 5.0, 0.01, 1.0, 4, z)


def test_log_line_format(tmp_path):
    rng = np.random.RandomState(1)
    X0 = rng.rand(6, 2)
    o = pobj.Objective(_FakeGPRF(6, 2), X0, None, _SD(X0, 0.05), log_dir=str(tmp_path))
    o(o.full0)
    o.close()
    lines = open(tmp_path / "log.txt").read().strip().split("\n")
    f = lines[0].split()
    assert f[0] == "0" and len(f) == 3 and "." in f[2] and len(f[2].split(".")[1]) == 2   # "%d %.2f %.2f"
    assert lines[-1].startswith("optimization finished after")


def test_input_recipe_kernels_match_oracle_c():
    """The covariance the synthetic outputs are drawn from (gprf_amd.synthetic.prior_kernel_matrix) is the oracle's."""
    rng = np.random.RandomState(3)
    X = rng.rand(30, 2)
    c = GPCov([1.4], [0.2, 0.3], "euclidean", "se")
    assert np.allclose(prior_kernel_matrix(X, X, c), VectorTree(None, 1, "euclidean", [0.2, 0.3], "se", [1.4]).kernel_matrix(X, X, False), rtol=1e-14)
    X3 = np.stack([130 + rng.randn(20), -2 + rng.randn(20), np.abs(rng.randn(20)) * 30], axis=1)
    c = GPCov([0.9], [40.0, 20.0], "lld", "matern32")
    assert np.allclose(prior_kernel_matrix(X3, X3, c), VectorTree(None, 1, "lld", [40.0, 20.0], "matern32", [0.9]).kernel_matrix(X3, X3, False), rtol=1e-12)


def test_synthetic_recipe_matches_oracle_recipe():
    from gprf_amd.synthetic import SampledData
    a = SampledData(n=700, ntrain=500, lscale=0.4, obs_std=0.04, yd=10, seed=0)
    b = H.SampledDataRef(n=700, ntrain=500, lscale=0.4, obs_std=0.04, yd=10, seed=0)
    assert np.array_equal(a.X_obs, b.X_obs) and np.array_equal(a.SX, b.SX)
    assert np.allclose(a.SY, b.SY, rtol=0, atol=1e-11)
    a.set_centers(grid_centers(4))
    b.set_centers(H.grid_centers(4))
    assert a.neighbors == b.neighbors
    assert all(np.array_equal(u, v) for u, v in zip(a.block_idxs, b.block_idxs))
    assert np.isclose(a.x_prior(a.X_obs.flatten() + 0.01)[0], b.x_prior(b.X_obs.flatten() + 0.01)[0], rtol=1e-14)
    th = np.array([[0.02, 1.3, 0.4, 0.5]])
    cov, nv = a.model_hypers(th)
    assert nv == 0.02 and cov.wfn_params == [1.3] and list(cov.dfn_params) == [0.4, 0.5]
    with pytest.raises(Exception):
        a.model_hypers(np.zeros((2, 4)))


def _cases_euclid():
    from gprf_amd import Blocker, grid_centers
    rng = np.random.RandomState(4)
    X = rng.rand(1200, 2)
    b = Blocker(grid_centers(36))
    blocks = [np.asarray(i) for i in b.block_clusters(X)]
    blocks[7] = np.zeros(0, dtype=np.int64)                         # an empty block
    blocks[3], blocks[20] = np.concatenate([blocks[3][::2], blocks[20][::2]]), np.concatenate([blocks[3][1::2], blocks[20][1::2]])
    cases = (("se", [0.05, 0.08], 1e-3), ("se", [0.12, 0.03], 0.3), ("matern32", [0.04, 0.04], 1e-2),
             ("se", [0.05, 0.05], 1.0), ("se", [0.5, 0.5], 1e-6))
    return X, blocks, cases


def _cases_lld():
    from gprf_amd import seismic
    X = seismic.synthetic_events(1500, seed=2)
    rng = np.random.RandomState(0)
    polar = np.stack([rng.uniform(-180, 180, 60), rng.uniform(86, 89.9, 60), rng.exponential(30, 60)], axis=1)
    X = np.concatenate([X, polar])
    blocks, _ = seismic.pdtree_cluster(X, blocksize=70)
    blocks = [np.asarray(b) for b in blocks]
    blocks[5] = np.zeros(0, dtype=np.int64)
    blocks[2], blocks[9] = np.concatenate([blocks[2][::2], blocks[9][::2]]), np.concatenate([blocks[2][1::2], blocks[9][1::2]])
    cases = (("matern32", [40.0, 40.0], 0.6), ("matern32", [150.0, 20.0], 0.3), ("se", [300.0, 50.0], 1e-3),
             ("matern32", [40.0, 40.0], 1.0), ("matern32", [2000.0, 500.0], 1e-4))
    return X, blocks, cases


@pytest.mark.parametrize("dfn", ["euclidean", "lld"])
def test_candidate_block_pairs_cover_the_exhaustive_oracle_list(dfn):
    """GPRF.compute_neighbors' host half (geometric pruning, no kernel evaluation) never loses a pair: the oracle's
    exhaustive list (gprf.py:119-150) is a subsequence of the candidates, which come in the reference's loop order —
    both kernels, several thresholds, anisotropic lengthscales, empty blocks, interleaved blocks (boxes overlap), blocks
    across the date line and near a pole.  The pruning is worth something: far fewer candidates than block pairs.  (The
    device half, and the equality of the final list, is tests/test_gpu_neighbors.py.)"""
    from gprf_amd import GPCov
    from gprf_amd.neighbors import candidate_block_pairs
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov as OC
    X, blocks, cases = _cases_euclid() if dfn == "euclidean" else _cases_lld()
    Y = np.zeros((len(X), 1))
    sv = 1.3 if dfn == "euclidean" else 0.7
    total = 0
    for wfn, ls, thr in cases:
        if dfn == "euclidean" and wfn == "matern32":
            continue                                     # (the library pairs matern32 with the lld distance only)
        cand = candidate_block_pairs(X, blocks, GPCov([sv], ls, dfn, wfn), thr)
        ref = GPRFRef(X, Y, None, OC([sv], ls, dfn, wfn), 0.1, block_idxs=blocks, neighbors=[])
        ref.compute_neighbors(threshold=thr)
        want = [(int(i), int(j)) for (i, j) in ref.neighbors]
        assert cand == sorted(cand, key=lambda p: (p[0], p[1])) and len(set(cand)) == len(cand)
        it = iter(cand)
        assert all(p in it for p in want), (wfn, ls, thr)        # subsequence, same order
        assert all(len(blocks[i]) and len(blocks[j]) and j < i for (i, j) in cand)
        if thr == 1.0:
            assert cand == []
        total += len(want)
        nb = len(blocks)
        if thr >= 1e-3 and max(ls) < 1000:
            assert len(cand) < 0.5 * nb * (nb - 1) / 2
    assert total > 50


def test_numa_pin_is_a_no_op_without_a_gpu_and_parses_cpu_lists(tmp_path, monkeypatch):
    """gprf_amd.numa: no GPU (or no NUMA information) -> nothing is changed; GPRF_NUMA_PIN=0 -> nothing is changed"""
    import os
    from gprf_amd import numa
    before = os.sched_getaffinity(0)
    import torch
    if not torch.cuda.is_available():
        assert numa.gpu_numa_node(0) == -1
        assert numa.pin_to_gpu_node(0) == (-1, 0)
    monkeypatch.setenv("GPRF_NUMA_PIN", "0")
    assert numa.pin_to_gpu_node(0) == (-1, 0)
    assert os.sched_getaffinity(0) == before
    if os.path.exists("/sys/devices/system/node/node0/cpulist"):
        assert len(numa._node_cpus(0)) >= 1
