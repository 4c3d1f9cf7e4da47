"""BASELINE.json configs[3] at full size through the C ABI: n = 80000, "800" -> 841 grid blocks + 3192 neighbouring
pairs (4033 units), yd = 50, lscale = 0.02, local_dist = 0.5, task xcov (gradient w.r.t. X and the kernel hypers).

Inputs by the reference's recipe (gprfopt.py:21-39, 525-546; synthetic.py:103-114, 139-153), seed 0: X ~ U[0,1]^2
for N = 80500 points, Y = chol(K + 0.01 I) Z from the same RNG stream, the first 80000 rows, X_obs = SX + N(0,
obs_std = lscale / 10).  N = 80500 is past the reference's own dense sampler (it switches to a CHOLMOD approximation
at 40000, synthetic.py:106, unavailable here): the dense fp64 Cholesky runs on the GPU box (52 GB, blocked, in place).

The oracle cannot evaluate 4033 units in test time, so parity is checked per unit on a seeded sample of 48 units
(12 blocks, 36 pairs, the largest pair among them) — ll, the unit's gradient rows and its hyper-parameter gradient
against ``GPRFRef.gaussian_llgrad`` on the same rows — plus what the domain offers at full size: the Bethe assembly
of the full (ll, gradX, gradC) from the device's own per-unit pieces, a directional finite difference in X and one
in the tied log-lengthscale of task xcov (gprfopt.py:333-355).  fp64 tolerances are written at each assertion.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LSCALE, NV, YD, NTRAIN = 0.02, 0.01, 50, 80000


@pytest.fixture(scope="module")
def c4():
    from gprf_amd.synthetic import SampledData
    from gprf_amd import grid_centers
    sd = SampledData(n=NTRAIN + 500, ntrain=NTRAIN, lscale=LSCALE, obs_std=LSCALE / 10, yd=YD, seed=0, use_gpu=True)
    sd.set_centers(grid_centers(800))
    g = sd.build_gprf(local_dist=0.5)
    yield sd, g
    g.close()


def _unit_rows(g, u):
    nb = g.n_blocks
    if u < nb:
        return np.asarray(g.block_idxs[u])
    i, j = g.neighbors[u - nb]
    return np.concatenate([g.block_idxs[i], g.block_idxs[j]])


def test_shape_of_the_configuration(c4):
    sd, g = c4
    assert g.n_blocks == 841 and len(g.neighbors) == 3192          # 29 x 29 grid, 8-neighbourhood
    g._push_neighbors(g.neighbors)                                  # (the pair list reaches the library with the first llgrad)
    assert g._ctx.num_units()[0] == 4033
    sizes = np.array([len(b) for b in g.block_idxs])
    assert sizes.sum() == NTRAIN and sizes.min() > 0
    # the prior draw itself (blocked in-place Cholesky of the 80500 x 80500 covariance on the GPU): at the TRUE
    # locations every unit's whitened outputs U^-T Y are i.i.d. N(0,1), so ||U^-T Y||^2 / (m dy) = 1 +- sqrt(2 / (m dy))
    ctx = g._ctx
    g.update_X(sd.SX)
    ctx.debug_run(np.ascontiguousarray(sd.SX), 2)
    for u in list(range(0, 841, 60)) + list(range(841, 4033, 400)):
        m = ctx.debug_unit_shape(u)[0]
        zz = ctx.debug_fetch(u, 5)[2]
        assert abs(zz / (m * YD) - 1.0) < 5.0 * np.sqrt(2.0 / (m * YD)), (u, m, zz / (m * YD))
    g.update_X(sd.X_obs)


def test_sampled_units_against_oracle_and_bethe_assembly(c4):
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov as OC
    sd, g = c4
    nb, nbrs = g.n_blocks, g.neighbors
    ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
    assert np.isfinite(ll) and np.all(np.isfinite(gX)) and np.all(np.isfinite(gC))
    ctx = g._ctx
    ctx.debug_run(np.ascontiguousarray(sd.X_obs), 6)               # leaves every unit's pieces in the pools
    nl = ctx.num_units()[1]
    assert nl == 4033
    # ---- the full result is the Bethe-weighted sum of the device's per-unit pieces (gprf.py:253-288)
    deg = np.zeros(nb, dtype=int)
    for (i, j) in nbrs:
        deg[i] += 1
        deg[j] += 1
    sizes = np.array([ctx.debug_unit_shape(l)[0] for l in range(nl)])
    rng = np.random.RandomState(4)
    big = int(np.argmax(sizes))
    sample = sorted(set(rng.choice(nb, 12, replace=False).tolist() + (nb + rng.choice(len(nbrs), 35, replace=False)).tolist()
                        + [big]))
    assert len(sample) >= 40
    ref = GPRFRef(sd.X_obs, sd.SY, None, OC([1.0], [LSCALE, LSCALE], "euclidean", "se"), NV,
                  block_idxs=g.block_idxs, neighbors=nbrs)
    gmax = float(np.max(np.abs(gX)))
    worst = dict(ll=0.0, gx=0.0, gc=0.0)
    part_ll, part_gX, part_gC = 0.0, np.zeros_like(gX), np.zeros(4)
    o_ll, o_gX, o_gC = 0.0, np.zeros_like(gX), np.zeros(4)
    for u in sample:
        m, mp, gu = ctx.debug_unit_shape(u)
        assert gu == u                                              # unsharded: local id = global id
        idx = _unit_rows(g, u)
        assert len(idx) == m
        w = (1 - deg[u]) if u < nb else 1
        d_gx = ctx.debug_fetch(u, 4)[:m, :2]
        d_ll = ctx.debug_fetch(u, 5)[0]
        d_gc = ctx.debug_fetch(u, 9)
        r_ll, r_gx, r_gc = ref.gaussian_llgrad(sd.X_obs[idx], sd.SY[idx], grad_X=True, grad_cov=True)
        worst["ll"] = max(worst["ll"], abs(d_ll - r_ll) / abs(r_ll))
        worst["gx"] = max(worst["gx"], float(np.max(np.abs(d_gx - r_gx))))
        worst["gc"] = max(worst["gc"], float(np.max(np.abs(d_gc - r_gc) / np.abs(r_gc))))
        # per unit: ll relative 1e-12; gradient rows at the two paths' common rounding floor (DESIGN.md Numerics:
        # 2.5e-13 of the largest gradient entry, unit weight 1); hyper-gradient relative 1e-9
        assert abs(d_ll - r_ll) <= 1e-12 * abs(r_ll), (u, d_ll, r_ll)
        assert np.max(np.abs(d_gx - r_gx)) <= 2.5e-13 * max(gmax, float(np.max(np.abs(r_gx)))), u
        assert np.allclose(d_gc, r_gc, rtol=1e-9, atol=1e-9 * float(np.max(np.abs(r_gc)))), (u, d_gc, r_gc)
        part_ll += w * d_ll; part_gX[idx] += w * d_gx; part_gC += w * d_gc
        o_ll += w * r_ll; o_gX[idx] += w * r_gx; o_gC += w * r_gc
    print("C4 sampled units: %d (largest m=%d)  worst ll rel %.2e  gX abs %.2e (max|g| %.3g)  gC rel %.2e"
          % (len(sample), sizes[big], worst["ll"], worst["gx"], gmax, worst["gc"]))
    # the Bethe-weighted sum restricted to the sample: device pieces vs oracle pieces
    assert abs(part_ll - o_ll) <= 1e-12 * abs(o_ll)
    assert np.max(np.abs(part_gX - o_gX)) <= 2.5e-13 * gmax * 9
    assert np.allclose(part_gC, o_gC, rtol=1e-9)
    # ---- and over ALL units it is exactly what gprf_eval returned (same weights, the device's fixed-order sums)
    tot_ll, tot_gX, tot_gC = 0.0, np.zeros_like(gX), np.zeros(4)
    for u in range(nl):
        w = (1 - deg[u]) if u < nb else 1
        idx = _unit_rows(g, u)
        tot_ll += w * ctx.debug_fetch(u, 5)[0]
        tot_gX[idx] += w * ctx.debug_fetch(u, 4)[:len(idx), :2]
        tot_gC += w * ctx.debug_fetch(u, 9)
    assert abs(tot_ll - ll) <= 1e-12 * abs(ll)
    assert np.max(np.abs(tot_gX - gX)) <= 1e-12 * gmax
    assert np.allclose(tot_gC, gC.ravel(), rtol=1e-10)


def test_directional_finite_differences_x_and_lengthscale(c4):
    sd, g = c4
    fn = g.block_fn
    g.block_fn = None                                   # FD must not straddle a re-blocking
    try:
        X0 = sd.X_obs.copy()
        g.update_X(X0)
        ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
        rng = np.random.RandomState(0)
        d = rng.randn(*X0.shape)
        d /= np.linalg.norm(d)
        h = 2e-7                                        # lengthscale 0.02: third-order term ~ (h / lscale)^2
        g.update_X(X0 + h * d); fp = g.llgrad()[0]
        g.update_X(X0 - h * d); fm = g.llgrad()[0]
        assert np.isclose((fp - fm) / (2 * h), np.sum(gX * d), rtol=2e-6)
        g.update_X(X0)
        # task xcov ties l0 = l1 = lscale (full_cov / collapse_cov_grad, gprfopt.py:333-355): d ll / d lscale = gC[2] + gC[3]
        eps = 1e-7 * LSCALE
        vals = []
        for s in (+1, -1):
            g.update_covs(np.array([[NV, 1.0, LSCALE + s * eps, LSCALE + s * eps]]))
            vals.append(g.llgrad()[0])
        g.update_covs(np.array([[NV, 1.0, LSCALE, LSCALE]]))
        assert np.isclose((vals[0] - vals[1]) / (2 * eps), gC[0, 2] + gC[0, 3], rtol=2e-6)
        # and the noise variance (dK/dnv = I)
        eps = 1e-7 * NV
        vals = []
        for s in (+1, -1):
            g.update_covs(np.array([[NV + s * eps, 1.0, LSCALE, LSCALE]]))
            vals.append(g.llgrad()[0])
        g.update_covs(np.array([[NV, 1.0, LSCALE, LSCALE]]))
        assert np.isclose((vals[0] - vals[1]) / (2 * eps), gC[0, 0], rtol=2e-6)
    finally:
        g.block_fn = fn


def test_reblocking_on_the_device_at_this_size(c4):
    """update_X through the grid Blocker re-partitions 80000 points against 841 centres on the device
    (gprf.py:169-174): the partition equals the host Blocker's, and the evaluation equals one on host-built blocks."""
    sd, g = c4
    rng = np.random.RandomState(7)
    X2 = sd.X_obs + rng.randn(*sd.X_obs.shape) * 2e-3
    g.update_X(X2)
    host = sd.blocker.block_clusters(X2)
    assert all(np.array_equal(a, b) for a, b in zip(g.block_idxs, host))
    a = g.llgrad(grad_X=True, grad_cov=True)
    from gprf_amd.gprf import GPRF
    h = GPRF(X2, sd.SY, None, sd.cov, sd.noise_var, block_idxs=host, neighbors=sd.neighbors)
    b = h.llgrad(grad_X=True, grad_cov=True)
    h.close()
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    g.update_X(sd.X_obs)
