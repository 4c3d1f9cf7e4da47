"""Host side of the seismic configuration against the oracle's restatement of pdtree_clustering.py and
run_seismic.py: partition tree, priors, and the optimiser callback's transforms (with a scripted stand-in for the
GPRF object, so no GPU is involved)."""
import numpy as np
import pytest

from gprf_amd import seismic
from oracle import seismic_ref


def _same_partition(a, b):
    return len(a) == len(b) and all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a, b))


@pytest.mark.parametrize("n,blocksize,seed", [(0, 10, 0), (7, 10, 0), (500, 40, 1), (2000, 210, 2), (1023, 64, 3)])
def test_pdtree_partition_equals_reference_restatement(n, blocksize, seed):
    X = seismic.synthetic_events(n, seed=seed) if n else np.zeros((0, 3))
    blocks, reblock = seismic.pdtree_cluster(X, blocksize=blocksize)
    rblocks, rreblock = seismic_ref.pdtree_cluster_ref(X, blocksize=blocksize)
    assert _same_partition(blocks, rblocks)
    assert all(len(b) < blocksize for b in blocks)                        # pdtree_clustering.py:31
    assert sorted(np.concatenate(blocks).tolist() if n else []) == list(range(n))
    # routing the same points through the stored splits reproduces the build partition ...
    assert _same_partition(reblock(X), blocks)
    # ... and moved points go where the reference's recluster sends them; the input is left untouched
    rng = np.random.RandomState(seed)
    X2 = X + rng.randn(*X.shape) * np.array([2.0, 2.0, 5.0])
    keep = X2.copy()
    assert _same_partition(reblock(X2), rreblock(X2.copy()))
    assert np.array_equal(X2, keep)
    assert len(reblock(X2)) == len(blocks)                                # empty leaves keep their slot


def test_longitude_wrap_keeps_date_line_neighbours_together():
    # two tight clusters, one of them straddling the date line: without the wrap the principal direction would see
    # it as 360 degrees wide
    rng = np.random.RandomState(0)
    a = np.stack([np.where(rng.rand(200) < 0.5, 179.5, -179.5) + rng.randn(200) * 0.1, rng.randn(200) * 0.1], 1)
    b = np.stack([20 + rng.randn(200) * 0.1, rng.randn(200) * 0.1], 1)
    X = np.concatenate([a, b])
    X = np.concatenate([X, np.zeros((400, 1))], axis=1)
    blocks, _ = seismic.pdtree_cluster(X, blocksize=250)
    assert len(blocks) == 2
    assert {frozenset((np.asarray(bl) < 200).tolist()) for bl in blocks} == {frozenset([True]), frozenset([False])}
    assert np.allclose(seismic.wrap_longitude([-179.5, 179.5, -22.0, -22.5]), [180.5, 179.5, -22.0, 337.5])


def test_priors_equal_reference_restatement():
    rng = np.random.RandomState(5)
    for c in (np.array([-2.0, 0.1, 3.0, 4.0]), np.array([0.3, -0.2, 5.04, 3.6]), np.array([-2.3, 0.0, 3.6, 3.6])):
        ll, g = seismic.seismic_cov_prior(c)
        rll, rg = seismic_ref.seismic_cov_prior_ref(c.copy())
        assert np.isclose(ll, rll, rtol=1e-15) and np.allclose(g, rg, rtol=1e-15)
    means = rng.randn(50, 3) * [50, 30, 40]
    xp, rxp = seismic.make_x_prior(means, 2.0), seismic_ref.make_x_prior_ref(means, 2.0)
    X = means + rng.randn(50, 3) * [.02, .02, 2.0]
    (ll, g), (rll, rg) = xp(X), rxp(X)
    assert np.isclose(ll, rll, rtol=1e-14) and np.allclose(g, rg, rtol=1e-14)
    # finite-difference check of the location prior's gradient
    e = np.zeros_like(X); e[3, 2] = 1e-5
    assert np.isclose((xp(X + e)[0] - xp(X - e)[0]) / 2e-5, g[3, 2], rtol=1e-6)


class _ScriptedGPRF(object):
    """Answers llgrad with a smooth function of what update_X / update_covs last received."""

    def __init__(self, n, fail_at=None, exc=None):
        self.n, self.calls, self.fail_at = n, 0, fail_at
        self.exc = exc if exc is not None else np.linalg.LinAlgError("not positive definite")
        self.X, self.FC = None, None

    def update_X(self, X):
        self.X = np.array(X)

    def update_covs(self, FC):
        self.FC = np.array(FC)

    def llgrad(self, local=True, grad_X=False, grad_cov=False, **kw):
        self.calls += 1
        if self.fail_at == self.calls:
            raise self.exc
        ll = -3.0
        gX, gC = np.zeros((0, 0)), np.zeros((0, 0))
        if self.X is not None:
            ll += -0.5 * np.sum(np.sin(self.X) ** 2)
            if grad_X:
                gX = -np.sin(self.X) * np.cos(self.X)
        if self.FC is not None:
            ll += np.sum(np.log(self.FC)) * 40.0
            if grad_cov:
                gC = 40.0 / self.FC
        return ll, gX, gC


@pytest.mark.parametrize("task", ["x", "xcov"])
def test_objective_transforms_equal_reference_restatement(task):
    rng = np.random.RandomState(11)
    n = 40
    Xtrue = seismic.synthetic_events(n, seed=4)
    means = Xtrue + rng.randn(n, 3) * [.02, .02, 2.0]
    C0 = np.array([[0.1, 1.0, 40.0, 40.0]]) if task == "xcov" else None
    objs = []
    for mod_obj, xprior, cprior in ((seismic.SeismicObjective, seismic.make_x_prior(means, 2.0), seismic.seismic_cov_prior),
                                    (seismic_ref.SeismicObjectiveRef, seismic_ref.make_x_prior_ref(means, 2.0),
                                     seismic_ref.seismic_cov_prior_ref)):
        g = _ScriptedGPRF(n)
        X0 = means.copy()
        o = mod_obj(g, X0, C0, cprior, xprior)
        assert np.array_equal(X0, means)                                  # the caller's array is not rescaled
        objs.append((o, g))
    (o, g), (r, rg) = objs
    assert np.array_equal(o.full0, r.full0) and o.full0[2] == means[0, 2] / 100.0
    xs = [o.full0.copy()]
    xs.append(o.full0 + rng.randn(len(o.full0)) * 0.01)
    if task == "xcov":
        big = o.full0.copy(); big[-4:] = np.log([50.0, 3.0, 5000.0, 0.2]); xs.append(big)     # every clamp bites
        steep = o.full0.copy(); steep[-2:] = np.log([1.5, 1.2]); xs.append(steep)              # gradient clipping
    for x in xs:
        (f, gr), (rf, rgr) = o(x), r(x.copy())
        assert np.isclose(f, rf, rtol=1e-14) and np.allclose(gr, rgr, rtol=1e-13, atol=1e-13)
        if task == "xcov":
            assert g.FC[0, 1] == 1.0 and g.FC[0, 0] <= 10.0 and np.all((g.FC[0, 2:] >= 1.0) & (g.FC[0, 2:] <= 999.0))
            assert gr[-3] == 0.0                                          # the signal variance is not learned
        assert np.array_equal(g.X, rg.X) and g.X[0, 2] == pytest.approx(x[2] * 100.0)
    if task == "xcov":
        assert np.array_equal(g.FC, rg.FC)


def test_failed_evaluation_is_answered_like_the_reference():
    n = 10
    means = seismic.synthetic_events(n, seed=1)
    g = _ScriptedGPRF(n, fail_at=1)
    o = seismic.SeismicObjective(g, means.copy(), None, x_prior=seismic.make_x_prior(means, 2.0))
    f, gr = o(o.full0)
    assert f == 1e10 and gr.shape == o.full0.shape and np.all(np.isfinite(gr))     # run_seismic.py:155-159
    f2, _ = o(o.full0)
    assert f2 < 1e9
    # ... but only THAT failure: a library error (HIP error, a unit grown past GPRF_MAX_UNIT) is not swallowed
    from gprf_amd import _capi
    g = _ScriptedGPRF(n, fail_at=1, exc=_capi.GprfHipError("gprf_update_eval failed (-1): unit 3 has 1100 points"))
    o = seismic.SeismicObjective(g, means.copy(), None, x_prior=seismic.make_x_prior(means, 2.0))
    with pytest.raises(_capi.GprfHipError):
        o(o.full0)


def test_coincident_events_end_in_one_leaf_instead_of_splitting_forever():
    X = np.concatenate([np.tile([[10.0, 20.0, 5.0]], (50, 1)), seismic.synthetic_events(100, seed=3)])
    blocks, reblock = seismic.pdtree_cluster(X, blocksize=30)
    assert sorted(np.concatenate(blocks).tolist()) == list(range(150))
    big = [b for b in blocks if len(b) >= 30]
    assert len(big) == 1 and set(range(50)) <= set(big[0].tolist())     # the 50 duplicates could not be split
    assert all(np.array_equal(a, b) for a, b in zip(reblock(X), blocks))
