"""The reference's PUBLISHED optimisation traces, whole runs ("a full optimisation run compared line by line").

`gprf_results.tgz` holds one results.txt per run, one line per function evaluation of scipy's L-BFGS-B
(gprfopt.py:377-432: ftol 1e-6, maxiter 200; line = step, seconds, objective, lengthscale ratio, mean location error,
x_prior: gprfopt.py:486).  tests/golden/extract_published.py keeps every line of the n = 2000 runs and of the two
n = 10000 / 100-block runs (BASELINE configs[1] and [2]).  Here `do_optimization` drives the HIP path to convergence and
the test states how far the published trace is reproduced:

* `lead` = the number of leading evaluations whose objective (2 decimals) AND mean location error (8 decimals) equal the
  published line to the printed digits — asserted to be at least what was measured on MI355X, rounded down (measured, rounds
  4-5: 66 of 87, 108 of 126, 47 of 89 lines; the fp64 LAPACK oracle under this scipy: 47 of the 87, then every line again
  behind ONE wild line-search point);
* wild line-search points (|objective| > 1e9, where two decimals are 14 significant digits) are compared to 1e-9 relative
  instead and do not end the run of matches: the oracle itself prints 1510520005650.69 there against the published
  1510520005745.52;
* behind the leading run the two trajectories part in the last bits of a gradient and meet again at the end.  The objective is
  only PIECEWISE smooth — update_X re-partitions the points at every evaluation (gprf.py:169-174) — and near convergence the
  line search samples both sides of a block boundary: the published north-star trace ends alternating between 409043.2 and
  409688.3, and so does this one.  Asserted: the best objective of the run equals the published best to 1e-5 relative; the
  LAST evaluation equals one of the published run's last 20 lines to 1e-5 relative; the final mean location error equals the
  published one to 0.5 %; the number of evaluations is within 35 % of the published run's.
Two fp64 evaluations of a gradient 1e-8 apart (max |g| ~ 2e5: relative 1e-13) steer L-BFGS-B identically for dozens of
iterations."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (ntrain, nblocks, local_dist, leading evaluations that must equal the published line to the printed digits)
CASES = [(2000, 9, 1.0, 60), (2000, 4, 0.1, 100), (10000, 100, 0.1, 40), (10000, 100, 1.0, 40)]


def _same_line(obj_val, err, step):
    pub = float(step["objective"])
    if abs(pub) > 1e9:
        return abs(obj_val - pub) <= 1e-9 * abs(pub)
    return "%.2f" % obj_val == step["objective"] and "%.8f" % err == step["mean_loc_err"]


@pytest.mark.parametrize("ntrain,nblocks,local_dist,min_lead", CASES)
def test_whole_published_trace(published, ntrain, nblocks, local_dist, min_lead):
    from gprf_amd import grid_centers
    from gprf_amd.objective import do_optimization
    from gprf_amd.synthetic import SampledData
    lscale, obs_std = 6 / np.sqrt(ntrain), 2 / np.sqrt(ntrain)
    run = "%d_%d_%d_%.6f_%.6f_%.4f_50_l-bfgs-b_x_-1_0.0100_s0_gprf0" % (ntrain, ntrain + 500, nblocks, lscale, obs_std, local_dist)
    rec = published[run]
    steps = rec["steps"]
    assert len(steps) == rec["n_lines"] > 50
    sd = SampledData(n=ntrain + 500, ntrain=ntrain, lscale=lscale, obs_std=obs_std, yd=50, seed=0, use_gpu=ntrain > 5000)
    sd.set_centers(grid_centers(nblocks))
    g = sd.build_gprf(local_dist=local_dist)
    xs = []
    t0 = time.time()
    z, obj = do_optimization(g, sd.X_obs, None, sd, maxsec=None, maxiter=200,
                             )
    wall = time.time() - t0
    assert z is not None
    # the objective keeps the vectors it was called with only as a trace of values: re-derive the errors from a second,
    # recording run of the same (deterministic) optimisation
    g.close()
    g = sd.build_gprf(local_dist=local_dist)
    from gprf_amd.objective import Objective
    import scipy.optimize
    obj2 = Objective(g, sd.X_obs, None, sd)

    def f(x):
        xs.append(x.copy())
        return obj2(x)
    r = scipy.optimize.minimize(f, obj2.full0, jac=True, method="l-bfgs-b", options={"ftol": 1e-6, "maxiter": 200})
    vals = [t[2] for t in obj2.trace]
    assert vals == [t[2] for t in obj.trace]                  # bit-reproducible from run to run
    errs = [float(np.mean(np.sqrt(np.sum((x.reshape(-1, 2) - sd.SX) ** 2, axis=1)))) for x in xs]
    n = min(len(vals), len(steps))
    same = [_same_line(vals[k], errs[k], steps[k]) for k in range(n)]
    lead = n if all(same) else same.index(False)
    rel_final = abs(max(vals) - max(float(st["objective"]) for st in steps)) / abs(float(steps[-1]["objective"]))
    print("%s: %d evaluations (published %d), %d equal to the printed digits, leading %d; last %.2f (published %s), best rel %.1e, "
          "mean location error %.8f (published %s); wall %.2f s = %.2f ms per evaluation (published %.0f s)"
          % (run, len(vals), len(steps), sum(same), lead, vals[-1], steps[-1]["objective"], rel_final, errs[-1],
             steps[-1]["mean_loc_err"], wall, 1e3 * wall / len(vals), rec["total_secs"]))
    for k in range(max(0, n - 6), max(len(vals), len(steps))):
        print("   line %3d  %16s %12s   published %16s %12s" % (
            k, "%.2f" % vals[k] if k < len(vals) else "-", "%.8f" % errs[k] if k < len(vals) else "-",
            steps[k]["objective"] if k < len(steps) else "-", steps[k]["mean_loc_err"] if k < len(steps) else "-"))
    assert lead >= min_lead
    assert abs(len(vals) - len(steps)) <= 0.35 * len(steps)
    pub = np.array([float(st["objective"]) for st in steps])
    assert abs(max(vals) - pub.max()) <= 1e-5 * abs(pub.max())
    assert np.min(np.abs(pub[-20:] - vals[-1])) <= 1e-5 * abs(pub.max())
    pub_err = float(steps[-1]["mean_loc_err"])
    assert abs(errs[-1] - pub_err) <= 5e-3 * pub_err
    assert -r.fun in vals                                  # what scipy returns is one of the evaluated points
    g.close()
