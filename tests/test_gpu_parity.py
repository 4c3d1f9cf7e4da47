"""Parity of the HIP path (through the C ABI / gprf_amd.GPRF) against the oracle and the committed golden
vectors.  Floating point (fp64) throughout; tolerances are stated per test.  The north-star configuration's contract
(n=10000: closeness to an 80-bit evaluation relative to the oracle's, and |GPU - oracle| <= 3.1e-8 / 1.55e-7 — BASELINE.json's
"< 1e-8" is below the reference path's own rounding there) is stated and asserted in tests/test_gpu_northstar.py."""
import os

import numpy as np
import pytest

from conftest import load_golden, blocks_from_csr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _gprf_from_golden(z, suffix="", dfn="euclidean", wfn="se", Xkey="X", Ykey="Y", nbrs=None, **kw):
    from gprf_amd.gprf import GPRF
    from gprf_amd import GPCov
    th = z["theta"]
    blocks = blocks_from_csr(z["block_ptr"], z["block_pts"])
    if nbrs is None:
        nbrs = [tuple(int(v) for v in r) for r in z["neighbors"]]
    return GPRF(z[Xkey], z[Ykey], None, GPCov([th[1]], th[2:], dfn, wfn), th[0], block_idxs=blocks, neighbors=nbrs, **kw)


def _close(gpu, ref, rtol):
    """max-abs error relative to the largest reference magnitude"""
    scale = max(np.max(np.abs(ref)), 1e-300)
    return np.max(np.abs(gpu - ref)) <= rtol * scale


def test_library_loaded_is_hip():
    """The .so this process uses is the in-tree HIP library and it sees the GPU."""
    from gprf_amd import _capi
    lib = _capi.load()
    assert "libgprf_hip.so" in _capi.library_path()
    ctx = _capi.Context(8, 2, 3, 0, 0)
    ctx.close()


@pytest.mark.parametrize("tag", ["local", "gprf"])
def test_c1_against_golden(tag):
    """BASELINE config 1 (ntrain=500, 4 blocks, yd=10, lscale=0.4): ll rel 1e-12, gradients rel 1e-10 of max."""
    z = load_golden("c1_small.npz")
    nbrs = [] if tag == "local" else None
    g = _gprf_from_golden(z, Xkey="X_obs", Ykey="SY", nbrs=nbrs)
    ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
    assert np.isclose(ll, z["ll_" + tag], rtol=1e-12)
    assert gX.shape == (500, 2) and gC.shape == (1, 4)
    assert _close(gX, z["gX_" + tag], 1e-10)
    assert np.allclose(gC, z["gC_" + tag], rtol=1e-9)
    # no-gradient call returns the reference's empty arrays (gprf.py:275,291)
    ll2, e1, e2 = g.llgrad()
    assert ll2 == ll and e1.shape == (0, 0) and e2.shape == (0, 0)
    g.close()


def test_c1_all_pairs_local_false():
    """local=False = every block pair, Bethe weight 1-(nb-1) (gprf.py:214-216)."""
    z = load_golden("c1_small.npz")
    g = _gprf_from_golden(z, Xkey="X_obs", Ykey="SY")
    ll, gX, _ = g.llgrad(local=False, grad_X=True)
    assert np.isclose(ll, z["ll_allpairs"], rtol=1e-12)
    assert _close(gX, z["gX_allpairs"], 1e-10)
    ll3, _, _ = g.llgrad(local=True)       # and back
    assert np.isclose(ll3, z["ll_gprf"], rtol=1e-12)
    g.close()


def test_subset_llgrad_against_oracle():
    """GPRF.subset_llgrad (gprf.py:182-204): unaries of a subset of the blocks + the pairs INSIDE the subset, Bethe weights
    from the local neighbour counts; ll relative 1e-12 against the oracle's restatement.  A repeated block counts twice (the
    reference's list comprehension); the object's own partition / neighbour list are back for the next llgrad."""
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov as OCov
    z = load_golden("c1_small.npz")
    g = _gprf_from_golden(z, Xkey="X_obs", Ykey="SY")
    th = z["theta"]
    ref = GPRFRef(z["X_obs"], z["SY"], None, OCov([th[1]], th[2:], "euclidean", "se"), th[0],
                  block_idxs=blocks_from_csr(z["block_ptr"], z["block_pts"]), neighbors=list(g.neighbors))
    ll_full = g.llgrad()[0]
    for sub in ([0, 2, 3], [1], [3, 1], [0, 1, 2, 3], [3, 1, 1]):
        assert np.isclose(g.subset_llgrad(sub), ref.subset_llgrad(sub), rtol=1e-12), sub
    assert np.isclose(g.subset_llgrad([0, 1, 2, 3]), ll_full, rtol=1e-13)       # the whole set = the objective itself
    ll, gX, _ = g.llgrad(grad_X=True)                                            # and the object is as it was
    assert ll == ll_full and _close(gX, z["gX_gprf"], 1e-10)
    g.close()


def test_tiny_per_stage_matrices():
    """Stage-level parity on the pair unit of the tiny case: Cholesky factor, inverse (as W^T W), Alpha."""
    z = load_golden("tiny_parts.npz")
    g = _gprf_from_golden(z)
    ctx = g._ctx
    g._push_neighbors(g.neighbors)
    ctx.debug_run(z["X"], 6)
    l = 2                                   # units: block 0, block 1, pair (1,0)
    m, mp, gu = ctx.debug_unit_shape(l)
    assert (m, mp, gu) == (60, 64, 2)
    U = np.triu(ctx.debug_fetch(l, 0)[:m, :m])
    assert np.allclose(U.T, z["pair_L"], rtol=0, atol=1e-13)            # K = U^T U, U^T = LAPACK's L
    Ufull = ctx.debug_fetch(l, 0)
    assert np.array_equal(np.triu(Ufull[m:, m:]), np.eye(mp - m))       # identity padding
    W = np.tril(ctx.debug_fetch(l, 1)[:m, :m])
    assert _close(W.T @ W, z["pair_prec"], 1e-12)
    At = ctx.debug_fetch(l, 3)
    assert _close(At[:7, :m].T, z["pair_Alpha"], 1e-12)
    assert np.all(At[7:, :] == 0) and np.all(At[:, m:] == 0)            # zero padding stays zero
    sc = ctx.debug_fetch(l, 5)
    assert np.isclose(sc[0], z["pair_ll"], rtol=1e-13) and np.isclose(sc[1], z["pair_logdet"], rtol=1e-13)
    gXu = ctx.debug_fetch(l, 4)[:m, :2]
    assert _close(gXu, z["pair_gX"], 1e-11)
    ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
    assert np.isclose(ll, z["ll"], rtol=1e-13) and _close(gX, z["gX"], 1e-11) and np.allclose(gC, z["gC"], rtol=1e-10)
    g.close()


def test_degenerate_blocks():
    """empty block, 1-point block, exact tile multiples (16, 32), pairs with an empty side."""
    z = load_golden("degenerate.npz")
    g = _gprf_from_golden(z)
    ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
    assert np.isclose(ll, z["ll"], rtol=1e-12)
    assert _close(gX, z["gX"], 1e-10) and np.allclose(gC, z["gC"], rtol=1e-9)
    g.close()


def test_jitter_path_matches_jitchol():
    """Duplicated points, zero noise: Cholesky hits an exactly zero pivot -> GPRF_NOT_PD -> the wrapper
    applies jitchol's schedule (gpy_linalg.py:81-97): K + 1e-6*mean(diag) I succeeds.  cond ~ 1e6, so the
    comparison is to rtol 1e-6."""
    from gprf_amd.gprf import GPRF
    from gprf_amd import GPCov, _capi
    z = load_golden("degenerate.npz")
    X, Y, th = z["dup_X"], z["dup_Y"], z["dup_theta"]
    g = GPRF(X, Y, None, GPCov([th[1]], th[2:], "euclidean", "se"), th[0], block_idxs=[np.arange(24)], neighbors=[])
    rc, _, _, _, bad = g._ctx.eval(X, True, True)
    assert rc == _capi.GPRF_NOT_PD and bad == 0
    ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
    assert np.isclose(ll, z["dup_ll"], rtol=1e-6)
    assert _close(gX, z["dup_gX"], 1e-5)
    assert g._jitter[0] == pytest.approx(1e-6)
    g.close()


@pytest.mark.parametrize("m", [300, 400])
def test_not_pd_is_reported_by_the_wide_register_cholesky(m):
    """A unit of 19 / 25 tiles per edge with a duplicated point and zero noise: the eight-wave register kernel (its overflow
    tiles waiting in LDS / in the U pool) reports the failed pivot like every other Cholesky kernel; with the jitter schedule
    the evaluation goes through and matches the oracle's jitchol path (cond ~ 1e6: rtol 1e-6)."""
    from gprf_amd.gprf import GPRF
    from gprf_amd import GPCov, _capi
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov as OC
    rng = np.random.RandomState(m)
    X = rng.rand(m, 2)
    X[m - 7] = X[m - 40]                           # the duplicate sits in the last tiles: the failure comes late in the chain
    Y = rng.randn(m, 4)
    g = GPRF(X, Y, None, GPCov([1.0], [0.3, 0.3], "euclidean", "se"), 0.0, block_idxs=[np.arange(m)], neighbors=[])
    rc, _, _, _, bad = g._ctx.eval(X, True, True)
    assert rc == _capi.GPRF_NOT_PD and bad == 0
    ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
    r = GPRFRef(X, Y, None, OC([1.0], [0.3, 0.3], "euclidean", "se"), 0.0, block_idxs=[np.arange(m)], neighbors=[])
    o = r.llgrad(grad_X=True, grad_cov=True)
    assert g._jitter[0] > 0.0
    assert np.isclose(ll, o[0], rtol=1e-5)
    g.close()


def test_not_pd_even_with_jitter_raises():
    from gprf_amd.gprf import GPRF
    from gprf_amd import GPCov
    X = np.zeros((20, 2))
    Y = np.ones((20, 2))
    g = GPRF(X, Y, None, GPCov([1.0], [0.5, 0.5], "euclidean", "se"), -2.0, block_idxs=[np.arange(20)], neighbors=[])
    with pytest.raises(np.linalg.LinAlgError):
        g.llgrad(grad_X=True)
    g.close()


def test_unit_too_large_is_refused():
    from gprf_amd.gprf import GPRF
    from gprf_amd import GPCov, _capi
    # refused before anything is allocated or launched (GPRF_MAX_UNIT = 16384 points; tests/test_gpu_big_units.py runs units
    # of up to 10000)
    n = _capi.MAX_UNIT + 1
    X = np.random.RandomState(0).rand(n, 2)
    Y = np.zeros((n, 2))
    g = GPRF(X, Y, None, GPCov([1.0], [0.5, 0.5], "euclidean", "se"), 0.01, block_idxs=[np.arange(n)], neighbors=[])
    with pytest.raises(_capi.GprfHipError, match="at most %d" % _capi.MAX_UNIT):
        g.llgrad()
    g.close()


def test_large_units_512_and_1000():
    """Large units: a 256+256 pair (32 tiles), then a 500+500 pair (63 tiles; the reference's n=2000 / 4-block
    runs have such units) — the generic k_solve path and the widest k_potrf panel."""
    from gprf_amd.gprf import GPRF
    from gprf_amd import GPCov
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov as OC
    rng = np.random.RandomState(4)
    X = rng.rand(512, 2)
    Y = rng.randn(512, 6)
    blocks = [np.arange(0, 256), np.arange(256, 512)]
    g = GPRF(X, Y, None, GPCov([1.0], [0.1, 0.1], "euclidean", "se"), 0.05, block_idxs=blocks, neighbors=[(1, 0)])
    r = GPRFRef(X, Y, None, OC([1.0], [0.1, 0.1], "euclidean", "se"), 0.05, block_idxs=blocks, neighbors=[(1, 0)])
    a = g.llgrad(grad_X=True, grad_cov=True)
    b = r.llgrad(grad_X=True, grad_cov=True)
    assert np.isclose(a[0], b[0], rtol=1e-12) and _close(a[1], b[1], 1e-10) and np.allclose(a[2], b[2], rtol=1e-9)
    g.close()
    X = rng.rand(1000, 2)
    Y = rng.randn(1000, 5)
    blocks = [np.arange(0, 500), np.arange(500, 1000)]
    g = GPRF(X, Y, None, GPCov([1.0], [0.08, 0.08], "euclidean", "se"), 0.05, block_idxs=blocks, neighbors=[(1, 0)])
    r = GPRFRef(X, Y, None, OC([1.0], [0.08, 0.08], "euclidean", "se"), 0.05, block_idxs=blocks, neighbors=[(1, 0)])
    a = g.llgrad(grad_X=True, grad_cov=True)
    b = r.llgrad(grad_X=True, grad_cov=True)
    assert np.isclose(a[0], b[0], rtol=1e-12) and _close(a[1], b[1], 1e-10) and np.allclose(a[2], b[2], rtol=1e-9)
    g.close()


def test_lld_matern32_toy():
    """("lld","matern32") — parity-unpinned kernel (no treegp, no dataset): oracle restatement only."""
    z = load_golden("lld_toy.npz")
    g = _gprf_from_golden(z, dfn="lld", wfn="matern32")
    ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
    assert gC.shape == (1, 4)                       # ncov = 2 + len(dfn_params) (gprf.py:578)
    assert np.isclose(ll, z["ll"], rtol=1e-12)
    assert _close(gX, z["gX"], 1e-9) and np.allclose(gC, z["gC"], rtol=1e-8)
    g.close()


def test_update_X_reblocks_and_update_covs():
    """update_X re-runs block_fn (gprf.py:169-174); update_covs swaps hypers (gprf.py:160-167)."""
    from gprf_amd import Blocker, grid_centers, GPCov
    from gprf_amd.gprf import GPRF
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov as OC
    rng = np.random.RandomState(8)
    X = rng.rand(400, 2)
    Y = rng.randn(400, 5)
    b = Blocker(grid_centers(9))
    g = GPRF(X, Y, b.block_clusters, GPCov([1.0], [0.2, 0.2], "euclidean", "se"), 0.01, neighbors=b.neighbors())
    r = GPRFRef(X, Y, b.block_clusters, OC([1.0], [0.2, 0.2], "euclidean", "se"), 0.01, neighbors=b.neighbors())
    X2 = X + 0.05 * rng.randn(400, 2)              # points cross block borders
    g.update_X(X2)
    r.update_X(X2)
    assert any(len(u) != len(v) for u, v in zip(b.block_clusters(X), g.block_idxs))
    a, c = g.llgrad(grad_X=True), r.llgrad(grad_X=True)
    assert np.isclose(a[0], c[0], rtol=1e-12) and _close(a[1], c[1], 1e-10)
    FC = np.array([[0.02, 1.5, 0.25, 0.18]])
    g.update_covs(FC)
    r.update_covs(FC)
    a, c = g.llgrad(grad_X=True, grad_cov=True), r.llgrad(grad_X=True, grad_cov=True)
    assert np.isclose(a[0], c[0], rtol=1e-12) and _close(a[1], c[1], 1e-10) and np.allclose(a[2], c[2], rtol=1e-9)
    assert g.noise_var == 0.02 and list(g.cov.dfn_params) == [0.25, 0.18]
    g.close()


def test_update_X_fast_reblocking_path_equals_callable_path():
    """block_fn = Blocker.block_clusters (bound method) is re-blocked inside the C library; a lambda around the
    same blocker goes through Python.  Same blocks, same numbers; block_idxs is materialised on demand."""
    from gprf_amd import Blocker, grid_centers, GPCov
    from gprf_amd.gprf import GPRF
    rng = np.random.RandomState(12)
    X = rng.rand(600, 2)
    Y = rng.randn(600, 6)
    b = Blocker(grid_centers(9))
    cov = GPCov([1.0], [0.2, 0.2], "euclidean", "se")
    fast = GPRF(X, Y, b.block_clusters, cov, 0.01, neighbors=b.neighbors())
    slow = GPRF(X, Y, lambda Z: b.block_clusters(Z), cov, 0.01, neighbors=b.neighbors())
    X2 = X + 0.04 * rng.randn(600, 2)
    X2[:3] = b.block_centers[:3]                     # points exactly on centres
    fast.update_X(X2)
    slow.update_X(X2)
    assert fast._reblock_pending                     # nothing has happened yet: the re-blocking rides on the evaluation
    a, c = fast.llgrad(grad_X=True, grad_cov=True), slow.llgrad(grad_X=True, grad_cov=True)
    assert fast._blocks_pushed == "device" and fast._block_idxs is None and not fast._reblock_pending
    assert a[0] == c[0] and np.array_equal(a[1], c[1]) and np.array_equal(a[2], c[2])
    assert all(np.array_equal(u, v) for u, v in zip(fast.block_idxs, slow.block_idxs))
    fast.close(); slow.close()


def test_device_reblocking_matches_host_and_oracle():
    """gprf_assign_blocks (nearest centre on the device) == gprf_nearest_center (host helper) == the oracle's
    BlockerRef on the same points, incl. points exactly on centres, equidistant points (first minimum wins) and
    repeated calls: `changed` is raised only when a point really changes block."""
    from gprf_amd import Blocker, grid_centers, GPCov, _capi
    from gprf_amd.gprf import GPRF
    from oracle.harness_ref import BlockerRef
    rng = np.random.RandomState(21)
    n = 3000
    X = rng.rand(n, 2)
    Y = rng.randn(n, 3)
    C = np.asarray(grid_centers(25), dtype=np.float64)
    b, bref = Blocker(C), BlockerRef(C)
    X[:25] = C                                        # on the centres
    X[25] = 0.5 * (C[0] + C[1])                       # equidistant from two centres
    X[26] = 0.25 * (C[0] + C[1] + C[5] + C[6])        # ... from four
    g = GPRF(X, Y, b.block_clusters, GPCov([1.0], [0.2, 0.2], "euclidean", "se"), 0.01, neighbors=b.neighbors())
    g._ctx.set_centers(b.block_centers)
    changed, dev = g._ctx.assign_blocks(X)
    assert not changed and dev is None                # the constructor's (host) partition of X is this very one
    dev = g._ctx.get_block_assignment()
    host = _capi.nearest_center(X, b.block_centers)
    ref = np.empty(n, dtype=np.int64)
    for i, idx in enumerate(bref.block_clusters(X)):
        ref[idx] = i
    assert np.array_equal(dev, host)
    far = np.abs(np.sort(np.linalg.norm(X[:, None, :] - C[None], axis=2), axis=1)[:, :2] @ [1, -1]) > 1e-12
    assert np.array_equal(dev[far], ref[far])         # away from exact ties the oracle agrees too
    changed2, none = g._ctx.assign_blocks(X)
    assert not changed2 and none is None              # same points: nothing moves, nothing is rebuilt
    X3 = X.copy()
    X3[100] = C[(dev[100] + 7) % 25]                  # one point jumps to another block
    changed3, dev3 = g._ctx.assign_blocks(X3)
    assert changed3 and dev3[100] == (dev[100] + 7) % 25 and np.array_equal(np.delete(dev3, 100), np.delete(dev, 100))
    # and the evaluation after a device re-blocking equals the one after a host re-blocking
    slow = GPRF(X, Y, lambda Z: b.block_clusters(Z), GPCov([1.0], [0.2, 0.2], "euclidean", "se"), 0.01,
                neighbors=b.neighbors())
    g.update_X(X3); slow.update_X(X3)
    a, c = g.llgrad(grad_X=True), slow.llgrad(grad_X=True)
    assert a[0] == c[0] and np.array_equal(a[1], c[1])
    g.close(); slow.close()


@pytest.mark.parametrize("dx,dy,sizes", [(1, 1, [5, 17, 40]), (3, 64, [33, 64, 16, 1, 90]), (2, 13, [100, 3, 129, 31]),
                                         (3, 7, [200, 150]), (2, 50, [257, 255]),
                                         # pair units of exactly 14 / 15 / 16 tiles: the register-resident Cholesky's
                                         # three tile-dealing regimes (wave 0 free / overflow only / a full share)
                                         (2, 9, [112, 112, 120, 120, 128, 128, 127]),
                                         # the seismic configuration's largest pairs (blocks of up to 209 events):
                                         # 27 and 28 tiles, the one-workgroup-per-CU instantiation of k_solve_panel
                                         (3, 5, [209, 209, 30, 224, 224]),
                                         # the paper-scale catalogue's shape: leaves of 195 points, pairs of 25 tiles — the
                                         # eight-wave Cholesky with 140 tiles waiting in the U pool, the single-buffer
                                         # two-per-CU k_solve_panel (largest unit 26 tiles), dy beyond one 16-column block
                                         (3, 50, [195, 195, 196, 60, 208])])
def test_random_shapes_against_oracle(dx, dy, sizes):
    """SE kernel with 1-3 input dimensions, 1..64 output columns (64 = the padded width), ragged block sizes incl.
    tile-boundary cases (16, 64, 255/257 -> pair of 512), chain of pairs + one long-range pair."""
    from gprf_amd import GPCov
    from gprf_amd.gprf import GPRF
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov as OC
    rng = np.random.RandomState(100 * dx + dy)
    n = sum(sizes)
    X = rng.rand(n, dx)
    Y = rng.randn(n, dy)
    perm = rng.permutation(n)
    cuts = np.cumsum([0] + sizes)
    blocks = [np.sort(perm[cuts[i]:cuts[i + 1]]) for i in range(len(sizes))]
    nbrs = [(i, i - 1) for i in range(1, len(sizes))]
    if len(sizes) > 2:
        nbrs.append((len(sizes) - 1, 0))
    ls = list(0.2 + 0.3 * rng.rand(dx))
    g = GPRF(X, Y, None, GPCov([1.7], ls, "euclidean", "se"), 0.05, block_idxs=blocks, neighbors=nbrs)
    r = GPRFRef(X, Y, None, OC([1.7], ls, "euclidean", "se"), 0.05, block_idxs=blocks, neighbors=nbrs)
    a = g.llgrad(grad_X=True, grad_cov=True)
    b = r.llgrad(grad_X=True, grad_cov=True)
    assert a[1].shape == (n, dx) and a[2].shape == (1, 2 + dx)
    assert np.isclose(a[0], b[0], rtol=1e-11)
    assert _close(a[1], b[1], 1e-9) and np.allclose(a[2], b[2], rtol=1e-8, atol=1e-8 * np.abs(b[2]).max())
    g.close()


def test_two_shards_on_one_gpu_sum_to_full():
    """The multi-GPU decomposition, exercised on one device: shard (0,2) + shard (1,2) partials add up to
    the unsharded result; device-resident evaluation path (gprf_eval_device)."""
    import torch
    from gprf_amd import dist as gdist
    z = load_golden("c1_small.npz")
    full = _gprf_from_golden(z, Xkey="X_obs", Ykey="SY")
    ref = full.llgrad(grad_X=True, grad_cov=True)
    parts = []
    for rank in range(2):
        g = _gprf_from_golden(z, Xkey="X_obs", Ykey="SY", shard=(rank, 2))
        ev = gdist.DeviceEvaluator(g)
        g._push_neighbors(g.neighbors)
        ev.set_X(g.X)
        ev.enqueue(True, True)
        parts.append(ev.result(True, True))
        assert 0 < g._ctx.num_units()[1] < 10
        g.close()
    assert np.isclose(parts[0][0] + parts[1][0], ref[0], rtol=1e-13)
    assert _close(parts[0][1] + parts[1][1], ref[1], 1e-12)
    assert np.allclose(parts[0][2] + parts[1][2], ref[2], rtol=1e-11)
    full.close()


def test_device_evaluator_async_allreduce_world1(monkeypatch):
    """The RCCL leg of the multi-GPU path on one GPU: a one-rank NCCL group with the all-reduce forced on.  Several
    evaluations are enqueued back to back (each all-reduce asynchronous behind its kernels, the next evaluation
    not waiting for it); the result must equal the plain evaluation bit for bit (SUM over one rank)."""
    import socket
    import torch
    import torch.distributed as dist
    from gprf_amd import dist as gdist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(port))
    monkeypatch.setenv("GPRF_FORCE_ALLREDUCE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        z = load_golden("c1_small.npz")
        g = _gprf_from_golden(z, Xkey="X_obs", Ykey="SY")
        ref = g.llgrad(grad_X=True, grad_cov=True)
        ev = gdist.DeviceEvaluator(g)
        g._push_neighbors(g.neighbors)
        ev.set_X(g.X)
        for _ in range(4):
            ev.enqueue(True, True)
        assert ev._work is not None                       # the collective really was issued
        out = ev.result(True, True)
        assert out[0] == ref[0] and np.array_equal(out[1], ref[1]) and np.array_equal(out[2], ref[2])
        g.close()
    finally:
        dist.destroy_process_group()


def test_permutation_invariance_and_determinism():
    """Size-independent properties: re-ordering points inside blocks leaves ll unchanged (to rounding) and
    permutes gradX; repeated evaluation is bit-identical (fixed-order reductions, no float atomics)."""
    from gprf_amd import GPCov
    from gprf_amd.gprf import GPRF
    rng = np.random.RandomState(9)
    X = rng.rand(300, 2)
    Y = rng.randn(300, 4)
    blocks = [np.arange(0, 140), np.arange(140, 300)]
    cov = GPCov([1.0], [0.3, 0.3], "euclidean", "se")
    g = GPRF(X, Y, None, cov, 0.01, block_idxs=blocks, neighbors=[(1, 0)])
    a = g.llgrad(grad_X=True)
    b = g.llgrad(grad_X=True)
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    blocks2 = [rng.permutation(blocks[0]), rng.permutation(blocks[1])]
    g2 = GPRF(X, Y, None, cov, 0.01, block_idxs=blocks2, neighbors=[(1, 0)])
    c = g2.llgrad(grad_X=True)
    assert np.isclose(a[0], c[0], rtol=1e-12) and _close(c[1], a[1], 1e-10)
    g.close(); g2.close()


def test_kernel_values_to_a_few_ulp_over_the_whole_exponent_range():
    """The kernels evaluate exp() with their own routine (range reduction + degree-13 polynomial, see exp_fast in
    gprf_dev.h; summed as 1 + (r + r^2 q(r)): under 1 ulp): the filled K = sv * exp(-r^2) agrees with numpy's to 2 ulp
    from r^2 = 0 down to the subnormals, and is exactly 0 past the underflow threshold."""
    from gprf_amd import GPCov
    from gprf_amd.gprf import GPRF
    m = 64
    x = np.concatenate([np.linspace(0.0, 1.0, 24), np.linspace(1.0, 27.5, 40)[1:], [31.0]])[:, None]
    x = x + 0.013 * np.sin(7.0 * np.arange(m))[:, None]            # no special spacing
    g = GPRF(x, np.zeros((m, 1)), None, GPCov([1.3], [1.0], "euclidean", "se"), 0.25, block_idxs=[np.arange(m)], neighbors=[])
    g._push_neighbors([])
    g._ctx.debug_run(np.ascontiguousarray(x), 0)                   # fill only: the K pool
    K = g._ctx.debug_fetch(0, 0)[:m, :m]
    d2 = (x - x.T) ** 2
    ref = 1.3 * np.exp(-d2) + 0.25 * np.eye(m)
    iu = np.triu_indices(m)
    assert d2.max() > 800 and np.sum((d2 > 600) & (d2 < 740)) > 5  # the range really is covered
    normal = ref[iu] > 1e-290
    assert np.max(np.abs(K[iu][normal] / ref[iu][normal] - 1.0)) < 2 * 2.3e-16
    assert np.allclose(K[iu][~normal], ref[iu][~normal], rtol=1e-10, atol=1e-322)
    assert np.all(K[iu][d2[iu] > 746] == 0.0)
    g.close()


def test_generated_and_filled_kernel_matrices_give_the_same_evaluation(monkeypatch):
    """SE units of up to 256 points: k_potrf_reg generates K itself and k_fill does not run (GPRF_DIAG fused_fill=0 forces
    the K pool back).  Both routes against the golden vectors, and against each other to rounding."""
    z = load_golden("c1_small.npz")
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("GPRF_DIAG", "fused_fill=" + mode)
        g = _gprf_from_golden(z, Xkey="X_obs", Ykey="SY")
        out[mode] = g.llgrad(grad_X=True, grad_cov=True)
        assert np.isclose(out[mode][0], z["ll_gprf"], rtol=1e-12) and _close(out[mode][1], z["gX_gprf"], 1e-10)
        # the K pool is written in one mode only
        g._ctx.debug_run(z["X_obs"], 6)
        g.close()
    assert np.isclose(out["1"][0], out["0"][0], rtol=1e-14)
    assert _close(out["1"][1], out["0"][1], 1e-12) and np.allclose(out["1"][2], out["0"][2], rtol=1e-11)


def test_one_pair_over_256_points_leaves_the_other_units_generated(monkeypatch):
    """The kernel-matrix source is decided per unit (round 2 decided per launch: one pair growing past 256 points sent every
    unit through the K pool).  A north-star-shaped partition in which ONE block is crowded so that its pairs exceed 256
    points: units of up to 320 points are generated inside the register-resident Cholesky whatever their neighbours' sizes
    (round 4: up to 20 tiles; a launch with larger units goes through the pool as a whole); the result equals the
    all-through-the-pool evaluation to rounding and the oracle's."""
    from gprf_amd import Blocker, grid_centers, GPCov
    from gprf_amd.gprf import GPRF
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov as OCov
    rng = np.random.RandomState(21)
    n = 2500
    X = rng.rand(n, 2)
    c = np.array(grid_centers(25))
    X[:110] = c[12] + 0.05 * (rng.rand(110, 2) - 0.5)        # the centre block gets ~210 points: its pairs 300+
    Y = rng.randn(n, 6)
    b = Blocker(c)
    blocks, nbrs = b.block_clusters(X), b.neighbors()
    sizes = np.array([len(v) for v in blocks])
    pair_sizes = np.array([sizes[i] + sizes[j] for (i, j) in nbrs])
    assert (pair_sizes > 256).sum() >= 4 and (pair_sizes <= 256).sum() > 40 and pair_sizes.max() <= 1024
    cov = GPCov([1.0], [0.12, 0.12], "euclidean", "se")
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("GPRF_DIAG", "fused_fill=" + mode)
        g = GPRF(X, Y, None, cov, 0.01, block_idxs=blocks, neighbors=nbrs)
        out[mode] = g.llgrad(grad_X=True, grad_cov=True)
        g.close()
    monkeypatch.delenv("GPRF_DIAG")
    r = GPRFRef(X, Y, None, OCov([1.0], [0.12, 0.12], "euclidean", "se"), 0.01, block_idxs=blocks, neighbors=nbrs, mode="matrix")
    o = r.llgrad(grad_X=True, grad_cov=True)
    for mode in ("1", "0"):
        assert np.isclose(out[mode][0], o[0], rtol=1e-12)
        assert _close(out[mode][1], o[1], 1e-9) and np.allclose(out[mode][2], o[2], rtol=1e-8)
    assert np.isclose(out["1"][0], out["0"][0], rtol=1e-13) and _close(out["1"][1], out["0"][1], 1e-11)


def _unit_tables(g):
    ctx = g._ctx
    return [ctx.debug_fetch(l, 10) for l in range(ctx.num_units()[1])]


def test_device_built_unit_tables_equal_host_partition_tables():
    """after update_X the unit tables (sizes, offsets, unit row -> point, a point's rows) are built on the
    device from the device's own partition (k_assign -> k_build -> k_scatter_x).  They must be the tables an uploaded
    host partition gives (gprf_set_blocks with Blocker.block_clusters' lists): identical row -> point tables, bit-
    identical (ll, gradX, gradC) on every iterate of a walk that re-partitions each time; an iterate that moves nobody
    across a border rebuilds nothing; block_idxs read back equals the host Blocker's."""
    from gprf_amd import Blocker, grid_centers, GPCov
    from gprf_amd.gprf import GPRF
    rng = np.random.RandomState(33)
    n = 5000
    X = rng.rand(n, 2)
    Y = rng.randn(n, 7)
    b = Blocker(grid_centers(36))
    cov = GPCov([1.0], [0.1, 0.1], "euclidean", "se")
    fast = GPRF(X, Y, b.block_clusters, cov, 0.01, neighbors=b.neighbors())
    slow = GPRF(X, Y, lambda Z: b.block_clusters(Z), cov, 0.01, neighbors=b.neighbors())
    Xk = X
    for k in range(5):
        Xk = Xk + 0.01 * rng.randn(n, 2)
        fast.update_X(Xk); slow.update_X(Xk)
        a, c = fast.llgrad(grad_X=True, grad_cov=True), slow.llgrad(grad_X=True, grad_cov=True)
        assert a[0] == c[0] and np.array_equal(a[1], c[1]) and np.array_equal(a[2], c[2]), k
        ta, tc = _unit_tables(fast), _unit_tables(slow)
        assert len(ta) == len(tc) and all(np.array_equal(u, v) for u, v in zip(ta, tc)), k
    assert fast._block_idxs is None                    # nothing came back to the host in between
    builds = fast._ctx.table_builds()
    fast.update_X(Xk + 1e-9)                           # nobody crosses a border
    again = fast.llgrad(grad_X=True, grad_cov=True)
    assert fast._ctx.table_builds() == builds          # ... so the tables were not rebuilt
    slow.update_X(Xk + 1e-9)
    c = slow.llgrad(grad_X=True, grad_cov=True)
    assert again[0] == c[0] and np.array_equal(again[1], c[1])
    assert all(np.array_equal(u, v) for u, v in zip(fast.block_idxs, b.block_clusters(Xk + 1e-9)))
    fast.close(); slow.close()


def test_reblocking_that_outgrows_the_workspace_is_repeated():
    """A re-partition that makes a unit larger than anything the launch was sized for (more tiles per edge, bigger
    matrix pools) is detected on the device (k_build), the workspace grows and the evaluation is repeated inside
    the same gprf_update_eval call: same numbers as a fresh context on the new partition."""
    from gprf_amd import Blocker, grid_centers, GPCov
    from gprf_amd.gprf import GPRF
    rng = np.random.RandomState(34)
    n = 1800
    X = rng.rand(n, 2)
    Y = rng.randn(n, 4)
    b = Blocker(grid_centers(16))
    cov = GPCov([1.0], [0.15, 0.15], "euclidean", "se")
    g = GPRF(X, Y, b.block_clusters, cov, 0.01, neighbors=b.neighbors())
    g.llgrad(grad_X=True)
    big0 = max(len(v) for v in g.block_idxs)
    X2 = X.copy()
    movers = rng.choice(n, 500, replace=False)
    X2[movers] = b.block_centers[5] + 0.01 * rng.randn(500, 2)          # 500 points pile into block 5
    g.update_X(X2)
    a = g.llgrad(grad_X=True, grad_cov=True)
    assert max(len(v) for v in g.block_idxs) > big0 + 300
    fresh = GPRF(X2, Y, None, cov, 0.01, block_idxs=b.block_clusters(X2), neighbors=b.neighbors())
    c = fresh.llgrad(grad_X=True, grad_cov=True)
    assert a[0] == c[0] and np.array_equal(a[1], c[1]) and np.array_equal(a[2], c[2])
    # and back again: the evaluation still runs under the large launch bound (other kernel instantiations than a fresh
    # context picks: equal to rounding, not to the bit); the bound follows down after a run of smaller partitions
    g.update_X(X)
    a = g.llgrad(grad_X=True)
    fresh.close()
    fresh = GPRF(X, Y, None, cov, 0.01, block_idxs=b.block_clusters(X), neighbors=b.neighbors())
    c = fresh.llgrad(grad_X=True)
    assert np.isclose(a[0], c[0], rtol=1e-13) and _close(a[1], c[1], 1e-12)
    g.close(); fresh.close()


def test_set_blocks_refuses_a_point_listed_twice():
    """blocks must be disjoint (include/gprf_hip.h); a duplicate is an argument error, not a silent wrong
    gradient.  Points left out of every block are allowed and get a zero gradient row."""
    from gprf_amd import GPCov, _capi
    from gprf_amd.gprf import GPRF
    rng = np.random.RandomState(35)
    X, Y = rng.rand(60, 2), rng.randn(60, 3)
    cov = GPCov([1.0], [0.3, 0.3], "euclidean", "se")
    blocks = [np.arange(0, 30), np.arange(29, 60)]                       # point 29 twice
    with pytest.raises(_capi.GprfHipError, match="listed twice"):
        GPRF(X, Y, None, cov, 0.01, block_idxs=blocks, neighbors=[(1, 0)])
    blocks = [np.arange(0, 25), np.arange(30, 60)]                       # points 25..29 in no block
    g = GPRF(X, Y, None, cov, 0.01, block_idxs=blocks, neighbors=[(1, 0)])
    ll, gX, _ = g.llgrad(grad_X=True)
    assert np.all(gX[25:30] == 0.0) and np.all(np.any(gX[:25] != 0.0, axis=1))
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov as OC
    r = GPRFRef(X, Y, None, OC([1.0], [0.3, 0.3], "euclidean", "se"), 0.01, block_idxs=blocks, neighbors=[(1, 0)])
    o = r.llgrad(grad_X=True)
    assert np.isclose(ll, o[0], rtol=1e-12) and _close(gX, o[1], 1e-10)
    g.close()


def test_cholesky_launch_lists_follow_the_partition():
    """The register-resident Cholesky runs as two instantiations side by side: units of up to 13 tiles per edge (208
    points) two to a CU, larger ones one to a CU, each over its own list, which k_build re-derives ON THE DEVICE from the
    sizes of every new partition (the host only sizes the two launches, from the last synchronised lengths + slack; a
    list that outgrows its launch makes the evaluation repeat with larger ones).  Partitions that move units between
    the classes — a few, then many at once — give the numbers of a fresh context on that partition, bit for bit."""
    from gprf_amd import Blocker, grid_centers, GPCov
    from gprf_amd.gprf import GPRF
    rng = np.random.RandomState(41)
    n = 6000
    X = rng.rand(n, 2)
    Y = rng.randn(n, 5)
    b = Blocker(grid_centers(64))
    cov = GPCov([1.0], [0.06, 0.06], "euclidean", "se")
    nb = b.neighbors()
    g = GPRF(X, Y, b.block_clusters, cov, 0.01, neighbors=nb)
    g.llgrad(grad_X=True)

    def classes(Xk):
        sz = np.array([len(v) for v in b.block_clusters(Xk)])
        ps = np.array([sz[i] + sz[j] for (i, j) in nb])
        assert ps.max() <= 256
        return int((ps > 208).sum()), int((ps <= 208).sum())

    big0, small0 = classes(X)
    assert 0 < big0 < 30 and small0 > 0
    # (1) a jitter that moves a handful of units across the 208-point line; (2) 500 points re-drawn from the border
    # blocks into the interior: dozens of pairs enter the large class at once (more than the launch slack)
    X1 = X + 0.004 * rng.randn(n, 2)
    r2 = np.random.RandomState(5)
    mv = r2.choice(np.where(np.abs(X - 0.5).max(axis=1) > 0.375)[0], 500, replace=False)
    X2 = X.copy()
    X2[mv] = 0.125 + 0.75 * r2.rand(500, 2)
    big2, _ = classes(X2)
    assert big2 > big0 + 40
    for Xk in (X1, X2, X):
        g.update_X(Xk)
        a = g.llgrad(grad_X=True, grad_cov=True)
        fresh = GPRF(Xk, Y, None, cov, 0.01, block_idxs=b.block_clusters(Xk), neighbors=nb)
        c = fresh.llgrad(grad_X=True, grad_cov=True)
        assert a[0] == c[0] and np.array_equal(a[1], c[1]) and np.array_equal(a[2], c[2])
        fresh.close()
    g.close()


def test_more_units_than_the_table_build_keeps_in_lds():
    """k_build keeps the unit sizes / offsets of the first 8192 units in LDS for its second pass (launch-slot records,
    Cholesky lists) and reads the rest back from global memory.  A partition of tiny blocks with more than 8192 units
    takes that second path; its result must be the sum of two shards of the same problem, each of which stays below
    8192 local units (first path), and the device-built tables must equal an uploaded host partition's."""
    from gprf_amd import Blocker, grid_centers, GPCov
    from gprf_amd.gprf import GPRF
    rng = np.random.RandomState(77)
    n = 9000
    X = rng.rand(n, 2)
    Y = rng.randn(n, 3)
    b = Blocker(grid_centers(2800))
    nb = b.neighbors()
    cov = GPCov([1.0], [0.02, 0.02], "euclidean", "se")
    full = GPRF(X, Y, b.block_clusters, cov, 0.05, neighbors=nb)
    ref = full.llgrad(grad_X=True, grad_cov=True)
    nt, nl = full._ctx.num_units()
    assert nl == nt and nt > 8192 + 1000, nt
    acc = [0.0, np.zeros_like(ref[1]), np.zeros_like(ref[2])]
    for rank in range(2):
        g = GPRF(X, Y, b.block_clusters, cov, 0.05, neighbors=nb, shard=(rank, 2), reduce=False)
        r = g.llgrad(grad_X=True, grad_cov=True)
        assert 0 < g._ctx.num_units()[1] < 8192
        acc[0] += r[0]; acc[1] += r[1]; acc[2] += r[2]
        g.close()
    assert np.isclose(acc[0], ref[0], rtol=1e-12)
    assert np.allclose(acc[1], ref[1], rtol=0, atol=1e-11 * np.abs(ref[1]).max())
    assert np.allclose(acc[2], ref[2], rtol=1e-10)
    # ... and an uploaded host partition (tables forced through the same kernel, from block sizes) gives the same bits
    host = GPRF(X, Y, None, cov, 0.05, block_idxs=b.block_clusters(X), neighbors=nb)
    c = host.llgrad(grad_X=True, grad_cov=True)
    assert ref[0] == c[0] and np.array_equal(ref[1], c[1]) and np.array_equal(ref[2], c[2])
    full.close(); host.close()


def test_one_queue_cholesky_under_counter_collection_gives_the_same_bits():
    """rocprofv3 --pmc serialises the dispatches of all queues; the stream-memory-operation join of the two Cholesky
    queues would then never complete, so the library puts both instantiations on ONE queue when it sees
    ROCPROF_COUNTER_COLLECTION (set by --pmc) or GPRF_DIAG one_queue=1.  Same kernels, same units: the result must be bit
    for bit the two-queue result (fresh processes: the switch is read once per process)."""
    import subprocess, sys
    code = r'''
import numpy as np, hashlib
from gprf_amd import Blocker, grid_centers, GPCov
from gprf_amd.gprf import GPRF
rng = np.random.RandomState(12)
n = 6000
X = rng.rand(n, 2); Y = rng.randn(n, 5)
b = Blocker(grid_centers(64))
g = GPRF(X, Y, b.block_clusters, GPCov([1.0], [0.06, 0.06], "euclidean", "se"), 0.01, neighbors=b.neighbors())
ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
sz = np.array([len(v) for v in g.block_idxs]); nb = b.neighbors()
big = sum(1 for (i, j) in nb if sz[i] + sz[j] > 208)
print("RESULT", repr(float(ll)), hashlib.sha1(np.ascontiguousarray(gX).tobytes()).hexdigest(), hashlib.sha1(np.ascontiguousarray(gC).tobytes()).hexdigest(), big)
g.close()
'''
    outs = []
    for extra in ({}, {"ROCPROF_COUNTER_COLLECTION": "1"}, {"GPRF_DIAG": "one_queue=1"}):
        env = dict(os.environ)
        env.pop("ROCPROF_COUNTER_COLLECTION", None)
        env.update(extra)
        env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()
        outs.append(line[1:])
    assert int(outs[0][3]) > 0                      # the partition really has units of both size classes
    assert outs[0] == outs[1] == outs[2]
