"""North-star configuration at full size (BASELINE.json configs[1], [2]): n=10000, 100 blocks, yd=50,
lscale=0.06, obs_std=0.02, with and without the 342 neighbour pairs.  Checks
  * the objective against the reference's PUBLISHED values (step 0 incl. x_prior; true-X) to the printed
    2 decimals — golden numbers from gprf_results.tgz;
  * objective and gradient against the oracle on identical inputs.  BASELINE.json's "gradient max-abs error < 1e-8" is
    BELOW the reference path's own rounding on this configuration (the fp64 LAPACK oracle is 1.0-2.1e-8 per unit, 1.05e-7
    on the assembled 342-pair gradient, away from an 80-bit evaluation; max |gradX| ~ 2e5), so what is asserted is (i)
    closeness to the 80-BIT TRUTH relative to the oracle's own (|GPU - true| <= 0.85 |oracle - true| assembled, 1.15 local; per pair
    unit: pooled maximum, mean ratio and worst ratio, 40 units) and (ii) a bound on |GPU - oracle|, two fp64 paths:
    3.1e-8 (local GP) / 1.55e-7 (342 pairs) = measured + 15 %; ll relative 1e-12; gC relative 1e-9;
  * size-independent properties: directional finite difference, Bethe sum rule.
Inputs are regenerated from seeds on the GPU box (N=10500 prior Cholesky through torch on the GPU)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RUN = "10000_10500_100_0.060000_0.020000_%s_50_l-bfgs-b_x_-1_0.0100_s0_gprf0"


@pytest.fixture(scope="module")
def sdata():
    from gprf_amd.synthetic import SampledData
    from gprf_amd import grid_centers
    sd = SampledData(n=10500, ntrain=10000, lscale=0.06, obs_std=0.02, yd=50, seed=0, use_gpu=True)
    sd.set_centers(grid_centers(100))
    return sd


@pytest.mark.parametrize("local_dist", [1.0, 0.1])
def test_published_objectives(sdata, published, local_dist):
    rec = published[RUN % ("%.4f" % local_dist)]
    g = sdata.build_gprf(local_dist=local_dist)
    assert len(g.neighbors) == (0 if local_dist == 1.0 else 342)
    ll = g.llgrad()[0] + sdata.x_prior(sdata.X_obs.flatten())[0]
    assert "%.2f" % ll == rec["steps"][0]["objective"]          # -5760728.82 / -6563678.10
    g.close()
    gt = sdata.build_gprf(X=sdata.SX, local_dist=local_dist)
    assert "%.2f" % gt.llgrad()[0] == rec["trueX_objective"]    # 206594.70 / 414491.46
    gt.close()


def _oracle(sdata, local_dist):
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov
    return GPRFRef(sdata.X_obs, sdata.SY, sdata.reblock, GPCov([1.0], [0.06, 0.06], "euclidean", "se"), 0.01,
                   block_idxs=sdata.block_idxs, neighbors=sdata.neighbors if local_dist < 1.0 else [])


# Tolerance.  BASELINE.json asks for "gradient max-abs error < 1e-8" at this configuration, where
# max|gradX| ~ 1.7e5-2e5 and cond(K_unit) ~ 5e3.  Measured against an 80-bit evaluation (tests/ld_truth.py) the
# reference CPU path (fp64 LAPACK, the oracle) is ITSELF 1-2e-8 away from the true per-unit gradient (its
# Cholesky's rounding dominates; a one-ulp change in 2% of K's entries moves the gradient by 1.2e-8), and the
# Bethe weights (1 - deg_i up to -7, plus 8 pair terms per point) scale that to ~1e-7 on the GPRF objective
# (Numerics).  Two independent fp64 evaluations therefore cannot agree to 1e-8 here.  What is
# required instead: (a) the GPU is as close to the TRUE gradient as the reference path is (asserted below on
# the full gradient and, unit by unit, on pair units: <= 1.5x the oracle's own error), and (b) GPU and oracle agree
# to the difference measured in round 5 plus 15 % (3.1e-8 local, 1.55e-7 with the 8-neighbourhood), i.e. to within their
# common rounding floor with no room for a regression.


def _truth(sdata, local_dist):
    """Full gradient in 80-bit arithmetic, assembled as gprf.py:253-273."""
    from ld_truth import unit_llgrad_ld, LD
    blocks = sdata.block_idxs
    nbrs = sdata.neighbors if local_dist < 1.0 else []
    deg = np.zeros(len(blocks), dtype=int)
    for (i, j) in nbrs:
        deg[i] += 1
        deg[j] += 1
    gX = np.zeros(sdata.X_obs.shape, dtype=LD)
    ll = LD(0)
    for b, idx in enumerate(blocks):
        l, g = unit_llgrad_ld(sdata.X_obs[idx], sdata.SY[idx], 0.01, 1.0, [0.06, 0.06])
        ll += (1 - deg[b]) * l
        gX[idx] += (1 - deg[b]) * g
    for (i, j) in nbrs:
        idx = np.concatenate([blocks[i], blocks[j]])
        l, g = unit_llgrad_ld(sdata.X_obs[idx], sdata.SY[idx], 0.01, 1.0, [0.06, 0.06])
        ll += l
        gX[idx] += g
    return ll, gX


@pytest.mark.parametrize("local_dist", [1.0, 0.1])
def test_gradient_against_oracle_and_extended_precision(sdata, local_dist):
    g = sdata.build_gprf(local_dist=local_dist)
    ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
    g.close()
    o_ll, o_gX, o_gC = _oracle(sdata, local_dist).llgrad(grad_X=True, grad_cov=True)
    t_ll, t_gX = _truth(sdata, local_dist)
    gmax = np.max(np.abs(o_gX))
    e_go = np.max(np.abs(gX - o_gX))
    e_gt = float(np.max(np.abs(gX - t_gX)))
    e_ot = float(np.max(np.abs(o_gX - t_gX)))
    print("local_dist=%g max|gX|=%.4g | gpu-oracle %.3g | gpu-true %.3g | oracle-true %.3g | ll rel: gpu-oracle %.2g gpu-true %.2g oracle-true %.2g"
          % (local_dist, gmax, e_go, e_gt, e_ot, abs(ll - o_ll) / abs(o_ll), abs(float(ll - t_ll)) / abs(float(t_ll)),
             abs(float(o_ll - t_ll)) / abs(float(t_ll))))
    # (a) as accurate as the reference CPU path.  A maximum over 20000 entries moves by several per cent with any change of a
    # rounding anywhere; measured: r03 0.98x (local) / 0.74x (342 pairs); since r04, row panel as V_jj^T C_jk on the matrix
    # pipe: 1.07-1.11x / 0.74-0.78x (profiles/r04_numerics*.log, r05_numerics.log).  Round 5 tried to buy the local-only case
    # back and could not without paying elsewhere (a Newton-Schulz step on V_jj changes nothing — the
    # CORRECTLY ROUNDED inverse gives the same 0.97-0.98 in a numpy emulation; one step of refinement on every panel restores
    # 1.00x here but costs 12 us of the Cholesky stage and 1.07x on the assembled gradient; on the small units only it costs
    # nothing and takes the assembled gradient to 1.34x): the bounds are the measured values plus a margin of a few per cent,
    # and the ABSOLUTE errors are pinned too, so that the trade made in round 4 stays the trade it was
    assert e_gt <= (1.15 if local_dist == 1.0 else 0.85) * e_ot
    assert e_gt <= (2.2e-8 if local_dist == 1.0 else 9.5e-8)
    # (b) agreement at the common rounding floor: the values measured on MI355X in round 5 (2.63e-8 without / 1.33e-7 with the
    # 342 pair units: profiles/r05_numerics.log) plus 15 % — rounds 1-5 allowed 4e-8 / 3e-7, 1.5-2.3x the measurement
    assert e_go <= (3.1e-8 if local_dist == 1.0 else 1.55e-7)
    assert np.isclose(ll, o_ll, rtol=1e-12)
    assert abs(float(ll - t_ll)) <= 1e-12 * abs(float(t_ll))
    assert np.allclose(gC, o_gC, rtol=1e-9)


def test_step1_trace_value(sdata, published):
    """L-BFGS-B's first trial point x0 - g/||g|| reproduces the published step-1 objective: the GPU gradient
    has the published direction."""
    rec = published[RUN % "0.1000"]
    from gprf_amd.objective import Objective
    g = sdata.build_gprf(local_dist=0.1)
    obj = Objective(g, sdata.X_obs, None, sdata)
    f0, g0 = obj(obj.full0)
    assert "%.2f" % (-f0) == rec["steps"][0]["objective"]
    x1 = obj.full0 - g0 / np.linalg.norm(g0)
    f1, _ = obj(x1)
    assert "%.2f" % (-f1) == rec["steps"][1]["objective"]       # -3492341.61
    err = np.mean(np.sqrt(np.sum((x1.reshape(-1, 2) - sdata.SX) ** 2, axis=1)))
    assert "%.8f" % err == rec["steps"][1]["mean_loc_err"]
    g.close()


def test_directional_fd_and_sum_rule(sdata):
    g = sdata.build_gprf(local_dist=0.1)
    g.block_fn = None                                   # FD must not straddle a re-blocking
    X0 = sdata.X_obs.copy()
    ll, gX, _ = g.llgrad(grad_X=True)
    rng = np.random.RandomState(0)
    d = rng.randn(*X0.shape)
    d /= np.linalg.norm(d)
    h = 1e-6
    g.update_X(X0 + h * d); fp = g.llgrad()[0]
    g.update_X(X0 - h * d); fm = g.llgrad()[0]
    assert np.isclose((fp - fm) / (2 * h), np.sum(gX * d), rtol=1e-6)
    g.close()
    # Bethe sum rule: with no pairs the objective is the plain sum of unary terms
    gl = sdata.build_gprf(local_dist=1.0)
    ll_local = gl.llgrad()[0]
    from gprf_amd import _capi
    ctx = gl._ctx
    tot = sum(ctx.debug_fetch(l, 5)[0] for l in range(ctx.num_units()[1]))
    assert np.isclose(ll_local, tot, rtol=1e-13)
    gl.close()


def test_pair_units_against_extended_precision_one_by_one(sdata):
    """Per unit, not only on the assembled gradient: for pair units of the north-star configuration (the largest, the
    smallest and a seeded sample) the device's gradient rows are compared with the 80-bit evaluation next to the oracle's
    (fp64 LAPACK) rows.  Both are rounding noise of the same order whose per-unit maxima scatter by a factor of two either
    way.  Round 2 measured the device 1.5x as far from the truth as LAPACK unit by unit; the cause was the order of
    accumulation in the factorisation (every product of a trailing update rounding at the running entry's magnitude:
    tests/diag/gpu_stage_error.py located it in U, tests/diag/cpu_accumulation_order.py reproduces the 1.2x of the factor in
    numpy and the cure): since round 3 a step's 16 products are summed from zero and enter the running tile with one
    addition, in the Cholesky and in the triangular solve.  Round 4 samples 42 units (the largest, the smallest, 40 seeded)
    instead of 12 and measures, on MI355X (profiles/r04_numerics_*.log):
        row panel by forward substitution (rounds 1-3):  |gpu - true| max 2.30e-8 mean 1.25e-8, ratio mean 0.85, 0.36 .. 1.67
        row panel U_jk = V_jj^T C_jk on the matrix pipe:  |gpu - true| max 2.66e-8 mean 1.32e-8, ratio mean 0.89, 0.41 .. 2.02
        |oracle - true| (fp64 LAPACK):                     max 2.51e-8 mean 1.53e-8
    (the explicit tile inverse costs ~5 % of accuracy and 8 % of the Cholesky's time less; the device stays closer to the
    truth than LAPACK on average).  The per-unit ratio divides by the oracle's error of THAT unit, which scatters by 3x, so
    its maximum over 42 units is a tail statistic: 1.67 / 2.02 measured.  Asserted (round 5: the measured values plus a small
    margin, and the absolute errors, so that a further 10-20 % of rounding cannot slip in): pooled maximum at most 1.15x the
    oracle's and 3.0e-8, mean ratio at most 0.95 and mean error 1.45e-8, no single unit more than 2.2x further from the truth
    than the oracle is."""
    from ld_truth import unit_llgrad_ld
    g = sdata.build_gprf(local_dist=0.1)
    g.llgrad(grad_X=True)
    ctx = g._ctx
    ctx.debug_run(np.ascontiguousarray(sdata.X_obs), 6)
    ref = _oracle(sdata, 0.1)
    nb = g.n_blocks
    sizes = np.array([ctx.debug_unit_shape(l)[0] for l in range(nb, nb + len(g.neighbors))])
    rng = np.random.RandomState(11)
    pick = sorted(set([int(np.argmax(sizes)), int(np.argmin(sizes))] + rng.choice(len(sizes), 40, replace=False).tolist()))
    assert len(pick) >= 40
    e_gpu, e_orc = [], []
    for q in pick:
        i, j = g.neighbors[q]
        idx = np.concatenate([g.block_idxs[i], g.block_idxs[j]])
        m = len(idx)
        d_gx = ctx.debug_fetch(nb + q, 4)[:m, :2]
        _, o_gx, _ = ref.gaussian_llgrad(sdata.X_obs[idx], sdata.SY[idx], grad_X=True)
        _, t_gx = unit_llgrad_ld(sdata.X_obs[idx], sdata.SY[idx], 0.01, 1.0, [0.06, 0.06])
        e_gpu.append(float(np.max(np.abs(d_gx - t_gx))))
        e_orc.append(float(np.max(np.abs(o_gx - t_gx))))
    e_gpu, e_orc = np.array(e_gpu), np.array(e_orc)
    ratio = e_gpu / e_orc
    print("pair units vs 80-bit: %d units (m %d..%d)  |gpu-true| max %.3g mean %.3g   |oracle-true| max %.3g mean %.3g   "
          "ratio mean %.2f min %.2f max %.2f" % (len(pick), sizes[pick].min(), sizes[pick].max(), e_gpu.max(), e_gpu.mean(),
                                                 e_orc.max(), e_orc.mean(), ratio.mean(), ratio.min(), ratio.max()))
    assert e_gpu.max() <= 1.15 * e_orc.max() and e_gpu.max() <= 3.0e-8          # measured 1.06x, 2.66e-8
    assert ratio.mean() <= 0.95 and e_gpu.mean() <= 1.45e-8                       # measured 0.89, 1.32e-8
    assert ratio.max() <= 2.2                                                      # measured 2.02
    g.close()
