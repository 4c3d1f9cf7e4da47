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
