"""BASELINE configs[0] (the plumbing case: ntrain=500, 4 blocks, yd=10, lscale=0.4, task x) end to end under
scipy L-BFGS-B: the HIP-backed GPRF driven by the objective callback follows the oracle-driven trace."""
import numpy as np
import pytest
import scipy.optimize

pytestmark = pytest.mark.gpu


def test_lbfgs_trace_follows_oracle():
    from gprf_amd.synthetic import SampledData
    from gprf_amd import grid_centers
    from gprf_amd.objective import do_optimization
    from oracle.harness_ref import ObjectiveRef, SampledDataRef, grid_centers as ogc
    sd = SampledData(n=1000, ntrain=500, lscale=0.4, obs_std=0.04, yd=10, seed=0)
    sd.set_centers(grid_centers(4))
    g = sd.build_gprf(local_dist=0.5)
    rx, obj = do_optimization(g, sd.X_obs, None, sd, maxiter=8)
    so = SampledDataRef(n=1000, ntrain=500, lscale=0.4, obs_std=0.04, yd=10, seed=0)
    so.set_centers(ogc(4))
    oref = ObjectiveRef(so.build_gprf(local_dist=0.5), so.X_obs, None, so)
    trace = []

    def f(x):
        v = oref(x)
        trace.append(-v[0])
        return v
    scipy.optimize.minimize(f, oref.full0, jac=True, method="l-bfgs-b", options={"ftol": 1e-6, "maxiter": 8})
    gpu = [t[2] for t in obj.trace]
    assert len(gpu) == len(trace) >= 8
    assert np.allclose(gpu, trace, rtol=1e-9)
    assert gpu[-1] > gpu[0]                         # the objective went up
    err0 = np.mean(np.linalg.norm(sd.X_obs - sd.SX, axis=1))
    err1 = np.mean(np.linalg.norm(rx.reshape(-1, 2) - sd.SX, axis=1))
    assert err1 < err0
    g.close()


@pytest.mark.parametrize("local_dist,nblocks", [(1.0, 4), (0.1, 9)])
def test_lbfgs_reproduces_published_trace(published, local_dist, nblocks):
    """The reference's published optimisation trace (gprf_results.tgz, n=2000): the first L-BFGS-B objective values
    driven by the HIP path equal the published lines to the printed 2 decimals, and so do the mean location errors
    (8 decimals) — end-to-end parity of objective, gradient, priors and the optimiser interface."""
    from gprf_amd.synthetic import SampledData
    from gprf_amd import grid_centers
    from gprf_amd.objective import Objective
    ntrain = 2000
    sd = SampledData(n=ntrain + 500, ntrain=ntrain, lscale=6 / np.sqrt(ntrain), obs_std=2 / np.sqrt(ntrain), yd=50, seed=0)
    sd.set_centers(grid_centers(nblocks))
    rec = published["2000_2500_%d_0.134164_0.044721_%s_50_l-bfgs-b_x_-1_0.0100_s0_gprf0" % (nblocks, "%.4f" % local_dist)]
    steps = rec["steps"][:6]          # (the whole trace, run to convergence: tests/test_gpu_trace_full.py)
    g = sd.build_gprf(local_dist=local_dist)
    obj = Objective(g, sd.X_obs, None, sd)
    xs = []

    class _Stop(Exception):
        pass

    def f(x):
        xs.append(x.copy())
        v = obj(x)
        if len(xs) >= len(steps):
            raise _Stop
        return v
    try:
        scipy.optimize.minimize(f, obj.full0, jac=True, method="l-bfgs-b", options={"ftol": 1e-6, "maxiter": 200})
    except _Stop:
        pass
    assert len(xs) == len(steps)
    for k, step in enumerate(steps):
        assert "%.2f" % obj.trace[k][2] == step["objective"], (k, obj.trace[k][2], step["objective"])
        err = np.mean(np.sqrt(np.sum((xs[k].reshape(-1, 2) - sd.SX) ** 2, axis=1)))
        assert "%.8f" % err == step["mean_loc_err"]
        assert "%.8f" % sd.x_prior(xs[k])[0] == step["x_prior"]
    g.close()


@pytest.mark.parametrize("task", ["x", "xcov", "cov4"])
def test_library_objective_equals_reference_callback(task):
    """gprf_objective (priors added by the assembly kernel, result in the optimiser's layout) against the oracle's literal
    callback (gprfopt.py:377-417) and against the same library driven through update_X / update_covs / llgrad."""
    from gprf_amd.synthetic import SampledData
    from gprf_amd import grid_centers
    from gprf_amd.objective import Objective
    from oracle.harness_ref import ObjectiveRef, SampledDataRef, grid_centers as ogc
    kw = dict(n=1000, ntrain=500, lscale=0.4, obs_std=0.04, yd=10, seed=0)
    sd, so = SampledData(**kw), SampledDataRef(**kw)
    sd.set_centers(grid_centers(4))
    so.set_centers(ogc(4))
    C0 = {"x": None, "xcov": np.array([[0.35]]), "cov4": np.array([[0.012, 1.1, 0.35, 0.45]])}[task]
    X0 = None if task == "cov4" else sd.X_obs
    g = sd.build_gprf(local_dist=0.5)
    obj = Objective(g, X0, C0, sd)
    assert obj._native
    oref = ObjectiveRef(so.build_gprf(local_dist=0.5), None if task == "cov4" else so.X_obs, C0, so)
    rng = np.random.RandomState(4)
    z = obj.full0 + 0.01 * rng.randn(len(obj.full0))
    for zz in (obj.full0, z):
        f, gr = obj(zz)
        fr, grr = oref(zz)
        assert np.isclose(f, fr, rtol=1e-12)
        assert np.allclose(gr, grr, rtol=1e-9, atol=1e-9 * np.abs(grr).max())
        assert np.isclose(sum(obj.parts), -f, rtol=1e-13)
        # the same through the reference-shaped surface of the same object
        class _Surface(object):
            update_X, update_covs, llgrad = g.update_X, g.update_covs, g.llgrad
        via = Objective(_Surface(), X0, C0, sd)
        if task == "cov4":
            g.update_X(sd.X_obs)
        f2, gr2 = via(zz)
        assert np.isclose(f, f2, rtol=1e-13) and np.allclose(gr, gr2, rtol=1e-11, atol=1e-11 * np.abs(gr2).max())
    if task != "cov4":
        assert np.array_equal(g.X, obj.layout.locations(z))
    if task != "x":
        assert np.isclose(g.cov.dfn_params[0], np.exp(z[obj.nx:][-1] / 5.0 if task == "xcov" else z[obj.nx + 2] / 5.0), rtol=1e-14)
    g.close()
