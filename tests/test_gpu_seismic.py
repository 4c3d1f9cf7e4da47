"""Seismic configuration (great-circle/depth distance, Matern-3/2, principal-direction-tree blocks,
L-BFGS-B callback of run_seismic.py) through the HIP path, against the oracle's GPRF restatement driven by the oracle's
restatement of the same callback.  fp64; tolerances per assertion.  Parity unpinned beyond the restatement: the
reference's catalogue file is not distributed (gprf_amd.seismic.synthetic_events is a stand-in)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(n=420, blocksize=60, yd=4, threshold=0.6, seed=0):
    from gprf_amd import GPCov, seismic
    from gprf_amd.gprf import GPRF
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov as OC
    from oracle import seismic_ref
    Xtrue = seismic.synthetic_events(n, seed=seed)
    # run_seismic.py's noise and signal variance; 150 km instead of its 40 km lengthscales so that a catalogue this
    # small still has block pairs above the edge threshold
    theta = (0.1, 1.0, 150.0, 150.0)
    cov = GPCov([theta[1]], list(theta[2:]), "lld", "matern32")
    Y = seismic.sample_y(Xtrue, cov, theta[0], yd, seed=seed)
    rng = np.random.RandomState(seed + 1)
    obs_std = 2.0
    Xobs = Xtrue + rng.randn(n, 3) * obs_std * np.array([.01, .01, 1.0])
    Xobs[:, 2] = np.abs(Xobs[:, 2])
    blocks, reblock = seismic.pdtree_cluster(Xobs, blocksize=blocksize)
    rblocks, rreblock = seismic_ref.pdtree_cluster_ref(Xobs, blocksize=blocksize)
    g = GPRF(Xobs, Y, reblock, cov, theta[0], neighbor_threshold=threshold)
    r = GPRFRef(Xobs, Y, rreblock, OC([theta[1]], list(theta[2:]), "lld", "matern32"), theta[0],
                neighbor_threshold=threshold)
    return g, r, Xobs, np.array([theta]), obs_std


def test_partition_and_neighbours_match():
    g, r, Xobs, C0, _ = _setup()
    assert len(g.block_idxs) == len(r.block_idxs) >= 8
    assert all(np.array_equal(a, b) for a, b in zip(g.block_idxs, r.block_idxs))
    assert sorted(map(tuple, g.neighbors)) == sorted(map(tuple, r.neighbors)) and len(g.neighbors) > 0
    g.close()


@pytest.mark.parametrize("task", ["x", "xcov"])
def test_objective_matches_oracle_through_the_callback(task):
    from gprf_amd import seismic
    from oracle import seismic_ref
    g, r, Xobs, C0, obs_std = _setup()
    C = C0 if task == "xcov" else None
    o = seismic.SeismicObjective(g, Xobs, C, seismic.seismic_cov_prior, seismic.make_x_prior(Xobs, obs_std))
    q = seismic_ref.SeismicObjectiveRef(r, Xobs, C, seismic_ref.seismic_cov_prior_ref,
                                        seismic_ref.make_x_prior_ref(Xobs, obs_std))
    rng = np.random.RandomState(3)
    x = o.full0.copy()
    for it in range(3):
        (f, gr), (rf, rgr) = o(x), q(x.copy())
        assert np.isclose(f, rf, rtol=1e-11)
        assert np.max(np.abs(gr - rgr)) <= 1e-9 * np.max(np.abs(rgr))
        # move the events (they change blocks: the tree re-routes them) and, for xcov, the log-hyperparameters
        step = rng.randn(len(x)) * 0.02
        step[2:o.nx:3] *= 0.05                                          # depth is in units of 100 km here
        x = x + step
    assert any(len(a) != len(b) for a, b in zip(g.block_idxs, seismic.pdtree_cluster(Xobs, 60)[0]))
    assert all(np.array_equal(a, b) for a, b in zip(g.block_idxs, r.block_idxs))
    g.close()


def test_short_optimisation_improves_objective_and_locations():
    """A few L-BFGS-B iterations through do_seismic_optimization: the objective falls and stays finite (no oracle)."""
    from gprf_amd import seismic
    g, _, Xobs, C0, obs_std = _setup()
    xp = seismic.make_x_prior(Xobs, obs_std)
    res, obj = seismic.do_seismic_optimization(g, Xobs, None, xp, maxsec=300, maxiter=5)
    assert res is not None and np.isfinite(res.fun)
    f0, _ = seismic.SeismicObjective(g, Xobs, None, x_prior=xp)(obj.full0)
    assert res.fun < f0
    g.close()


def test_device_tree_routing_equals_host_routing():
    """GPRF.update_X with pdtree_cluster's reblock routes on the device (gprf_set_split_tree + gprf_assign_blocks,
    kernel k_route): the partition is bit-for-bit the host routine's — for the build points themselves (every split's
    median point sits exactly on its threshold), for moved points, across the date line, and with a leaf left empty —
    and the evaluation equals one driven through the plain-callable (host) path."""
    from gprf_amd import GPCov, seismic
    from gprf_amd.gprf import GPRF
    n = 3000
    X = seismic.synthetic_events(n, seed=5)
    X[:40, 0] = np.where(np.arange(40) % 2 == 0, 179.9, -179.9) + np.linspace(-0.05, 0.05, 40)   # date-line cluster
    Y = np.random.RandomState(0).randn(n, 3)
    cov = GPCov([1.0], [150.0, 150.0], "lld", "matern32")
    blocks, reblock = seismic.pdtree_cluster(X, blocksize=120)
    g = GPRF(X, Y, reblock, cov, 0.1, neighbor_threshold=0.6)
    slow = GPRF(X, Y, lambda Z: reblock(Z), cov, 0.1, neighbors=g.neighbors, block_idxs=blocks)
    slow.block_fn = lambda Z: reblock(Z)
    rng = np.random.RandomState(1)
    moves = [X, X + rng.randn(n, 3) * [0.3, 0.3, 3.0], X + rng.randn(n, 3) * [5.0, 5.0, 20.0]]
    far = X.copy(); far[np.asarray(blocks[3])] += [40.0, 10.0, 0.0]                               # block 3 empties
    moves.append(far)
    for k, X2 in enumerate(moves):
        g.update_X(X2)
        host = reblock(X2)
        assert len(g.block_idxs) == len(host)
        assert all(np.array_equal(a, b) for a, b in zip(g.block_idxs, host)), k
        slow.update_X(X2)
        a, b = g.llgrad(grad_X=True, grad_cov=True), slow.llgrad(grad_X=True, grad_cov=True)
        # (reading block_idxs above installed the device's partition as it is: its launch-wide tile bound follows a shrinking
        # partition with hysteresis, the host-partition object's is exact — another kernel instantiation can change the last bit)
        assert np.isclose(a[0], b[0], rtol=1e-13, atol=0)
        assert np.allclose(a[1], b[1], rtol=0, atol=1e-13 * np.abs(b[1]).max()) and np.allclose(a[2], b[2], rtol=1e-12)
    assert g._centers_of is reblock.tree                     # the device path was the one taken
    assert any(len(b) == 0 for b in g.block_idxs)
    # unchanged points: no partition comes back
    changed, _ = g._ctx.assign_blocks(np.ascontiguousarray(far))
    assert not changed
    g.close(); slow.close()


def test_config5_block_size_lld_matern32_units_above_256_points():
    """BASELINE configs[4]'s own shape (run_seismic.py:299-301, 375): ("lld","matern32") with 40 km lengthscales,
    principal-direction-tree blocks of fewer than 210 events, edges above 0.6 — block pairs of 257..418 points, which
    take the K-pool fill (k_fill<lld, matern32>), the generic Cholesky (k_potrf) and the 28-tile forward substitution
    instead of the register-resident kernels the small stand-in above exercises.  Against the oracle's GPRFRef on the
    same partition: ll relative 1e-11, gradX 1e-9 of its largest entry, hyper-gradient relative 1e-8."""
    from gprf_amd import GPCov, seismic
    from gprf_amd.gprf import GPRF
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov as OC
    n, yd = 3000, 6
    X = seismic.synthetic_events(n, seed=0)
    Y = np.random.RandomState(1).randn(n, yd)
    blocks, reblock = seismic.pdtree_cluster(X, 210)
    ls = [40.0, 40.0]
    g = GPRF(X, Y, reblock, GPCov([1.0], ls, "lld", "matern32"), 0.1, block_idxs=blocks, neighbor_threshold=0.6)
    sz = [len(b) for b in blocks]
    assert max(sz) < 210 and len(g.neighbors) > 0
    pair_sizes = [sz[i] + sz[j] for (i, j) in g.neighbors]
    assert max(pair_sizes) > 256 and max(pair_sizes) <= 418
    r = GPRFRef(X, Y, None, OC([1.0], ls, "lld", "matern32"), 0.1, block_idxs=g.block_idxs, neighbors=g.neighbors)
    for task_cov in (False, True):
        a = g.llgrad(grad_X=True, grad_cov=task_cov)
        b = r.llgrad(grad_X=True, grad_cov=task_cov)
        assert np.isclose(a[0], b[0], rtol=1e-11)
        assert np.max(np.abs(a[1] - b[1])) <= 1e-9 * np.max(np.abs(b[1]))
        if task_cov:
            assert np.allclose(a[2], b[2], rtol=1e-8)
    # the events move (some change leaf), the hypers move: still the same numbers as the oracle
    rng = np.random.RandomState(2)
    X2 = X + rng.randn(n, 3) * [0.05, 0.05, 2.0]
    X2[:, 2] = np.abs(X2[:, 2])
    g.update_X(X2)
    g.update_covs(np.array([[0.12, 0.9, 35.0, 45.0]]))
    r.block_fn = seismic.pdtree_cluster(X, 210)[1]
    r.update_X(X2)
    r.update_covs(np.array([[0.12, 0.9, 35.0, 45.0]]))
    assert all(np.array_equal(p, q) for p, q in zip(g.block_idxs, r.block_idxs))
    a, b = g.llgrad(grad_X=True, grad_cov=True), r.llgrad(grad_X=True, grad_cov=True)
    assert np.isclose(a[0], b[0], rtol=1e-11)
    assert np.max(np.abs(a[1] - b[1])) <= 1e-9 * np.max(np.abs(b[1]))
    assert np.allclose(a[2], b[2], rtol=1e-8)
    g.close()


def test_far_apart_events_take_the_closed_form_great_circle_branch():
    """The lld kernel evaluates the great-circle terms by polynomials in the haversine for pairs closer than 2560 km and in
    closed form (asin, square roots) beyond: events scattered over the whole globe in two blocks and their pair — most pairs
    are far apart, some close, a few nearly antipodal — against the oracle (numpy's arcsin), values and both gradients."""
    from gprf_amd import GPCov
    from gprf_amd.gprf import GPRF
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov as OC
    rng = np.random.RandomState(12)
    n = 140
    X = np.column_stack([rng.uniform(-180.0, 180.0, n), rng.uniform(-75.0, 75.0, n), rng.uniform(0.0, 120.0, n)])
    X[5] = [X[4, 0] + 0.3, X[4, 1] - 0.2, X[4, 2] + 3.0]                 # a close pair
    # a nearly antipodal one, half a degree off (AT the antipode the haversine itself is ill-conditioned — dg/da = R / sqrt(a (1 - a))
    # — and two correct evaluations differ by 1e-8 in k: measured, 6e-11 in ll)
    X[7] = [((X[6, 0] + 360.0) % 360.0) - 180.0 + 0.5, -X[6, 1] + 0.4, X[6, 2]]
    Y = rng.randn(n, 6)
    blocks = [np.arange(0, 70), np.arange(70, n)]
    nbrs = [(0, 1)]
    ls = [9000.0, 150.0]
    g = GPRF(X, Y, None, GPCov([1.3], ls, "lld", "matern32"), 0.2, block_idxs=blocks, neighbors=nbrs)
    r = GPRFRef(X, Y, None, OC([1.3], ls, "lld", "matern32"), 0.2, block_idxs=blocks, neighbors=nbrs)
    a = g.llgrad(grad_X=True, grad_cov=True)
    b = r.llgrad(grad_X=True, grad_cov=True)
    g.close()
    assert np.isclose(a[0], b[0], rtol=1e-11)
    assert np.max(np.abs(a[1] - b[1])) <= 1e-9 * np.max(np.abs(b[1]))
    assert np.allclose(a[2], b[2], rtol=1e-8, atol=1e-9 * np.abs(b[2]).max())
