"""Pin the oracle: it must reproduce the reference's published optimisation traces (known-answer tests
carried by /root/reference/gprf_results.tgz, extracted to tests/golden/published_traces.json) from seeds
alone, to every printed digit."""
import numpy as np
import pytest

from oracle.harness_ref import ObjectiveRef, grid_centers

RUN = "2000_2500_%d_0.134164_0.044721_%s_50_l-bfgs-b_%s_-1_0.0100_s0_gprf0"


def _mean_err(x, sd):
    return np.mean(np.sqrt(np.sum((x.reshape(-1, 2) - sd.SX) ** 2, axis=1)))


@pytest.mark.parametrize("nblocks,local_dist,npairs", [(4, 1.0, 0), (4, 0.1, 6), (9, 1.0, 0), (9, 0.1, 20)])
def test_task_x_step0_step1_truex(sdata2000, published, nblocks, local_dist, npairs):
    sd = sdata2000
    rec = published[RUN % (nblocks, "%.4f" % local_dist, "x")]
    sd.set_centers(grid_centers(nblocks))
    g = sd.build_gprf(local_dist=local_dist)
    assert len(g.neighbors) == npairs
    obj = ObjectiveRef(g, sd.X_obs, None, sd)
    f0, g0 = obj(obj.full0)
    s0, s1 = rec["steps"][0], rec["steps"][1]
    assert "%.2f" % (-f0) == s0["objective"]
    assert "%.8f" % _mean_err(obj.full0, sd) == s0["mean_loc_err"]
    assert "%.8f" % sd.x_prior(obj.full0)[0] == s0["x_prior"]
    # L-BFGS-B's first trial point is x0 - g/||g||: pins the gradient's direction
    x1 = obj.full0 - g0 / np.linalg.norm(g0)
    f1, _ = obj(x1)
    assert "%.2f" % (-f1) == s1["objective"]
    assert "%.8f" % _mean_err(x1, sd) == s1["mean_loc_err"]
    assert "%.8f" % sd.x_prior(x1)[0] == s1["x_prior"]
    gt = sd.build_gprf(X=sd.SX, local_dist=local_dist)
    assert "%.2f" % gt.llgrad()[0] == rec["trueX_objective"]


@pytest.mark.parametrize("local_dist", [1.0, 0.1])
def test_task_xcov_step0_step1(sdata2000, published, local_dist):
    """xcov runs optimise a single tied lengthscale in log space x5 (gprfopt.py:333-345,365-368): step 0 pins
    cov_prior, step 1 pins the hyper-parameter gradient through collapse_cov_grad."""
    sd = sdata2000
    rec = published[RUN % (4, "%.4f" % local_dist, "xcov")]
    sd.set_centers(grid_centers(4))
    g = sd.build_gprf(local_dist=local_dist)
    C0 = np.array(g.cov.dfn_params[0]).reshape(1, 1)
    obj = ObjectiveRef(g, sd.X_obs, C0, sd)
    f0, g0 = obj(obj.full0)
    assert "%.2f" % (-f0) == rec["steps"][0]["objective"]
    x1 = obj.full0 - g0 / np.linalg.norm(g0)
    f1, _ = obj(x1)
    s1 = rec["steps"][1]
    assert "%.2f" % (-f1) == s1["objective"]
    lscale1 = np.exp(x1[obj.nx:] / obj.cov_scale)[0]
    assert "%.8f" % (lscale1 / sd.lscale) == s1["lscale_ratio"]
    assert "%.8f" % sd.x_prior(x1[:obj.nx])[0] == s1["x_prior"]


def test_rows_mode_equals_matrix_mode(sdata2000):
    """The reference-shaped per-row derivative loop (gprf.py:556-561) and the hoisted loop are the same
    arithmetic."""
    sd = sdata2000
    sd.set_centers(grid_centers(9))
    a = sd.build_gprf(local_dist=0.1, mode="rows")
    b = sd.build_gprf(local_dist=0.1, mode="matrix")
    i, j = a.neighbors[0]
    ra = a.llgrad_joint(i, j, grad_X=True, grad_cov=True)
    rb = b.llgrad_joint(i, j, grad_X=True, grad_cov=True)
    assert ra[0] == rb[0]
    assert np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2])


def test_subset_llgrad_of_every_block_is_the_published_objective(sdata2000, published):
    """gprf.py:182-204 restated: over ALL blocks the subset objective is the full objective, i.e. the published step-0 value
    once the location prior is added; over one block it is that block's unary term (no pairs inside a singleton)."""
    sd = sdata2000
    sd.set_centers(grid_centers(9))
    g = sd.build_gprf(local_dist=0.1)
    assert len(g.neighbors) == 20
    full = g.llgrad()[0]
    assert np.isclose(g.subset_llgrad(list(range(9))), full, rtol=1e-13)
    rec = published["2000_2500_9_0.134164_0.044721_0.1000_50_l-bfgs-b_x_-1_0.0100_s0_gprf0"]
    assert "%.2f" % (g.subset_llgrad(list(range(9))) + sd.x_prior(sd.X_obs.flatten())[0]) == rec["steps"][0]["objective"]
    assert g.subset_llgrad([4]) == g.llgrad_unary(4)[0]
    # two neighbouring blocks: their pair minus nothing (each has ONE neighbour inside the subset: weight 1 - 1 = 0)
    i, j = g.neighbors[0]
    assert np.isclose(g.subset_llgrad([i, j]), g.llgrad_joint(i, j)[0], rtol=1e-13)
