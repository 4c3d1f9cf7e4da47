"""Oracle C kernels: closed forms, finite differences, the reference's haversine doctests."""
import numpy as np
import pytest

from oracle.gprf_ref import GPRFRef
from oracle.vector_tree import GPCov, VectorTree, dist_km

from conftest import load_golden, blocks_from_csr


def test_haversine_doctests():
    """run_seismic.py:24-33"""
    deg = lambda a, b: np.degrees(dist_km(a, b) / 6371.0)
    assert int(deg((10, 0), (20, 0))) == 10
    assert int(deg((10, 0), (10, 45))) == 45
    assert int(deg((-78, -12), (-10.25, 52))) == 86
    assert deg((132.86521, -0.45606493), (132.86521, -0.45606493)) < 1e-4
    assert deg((127.20443, 2.8123965), (127.20443, 2.8123965)) < 1e-4


def test_se_closed_form():
    rng = np.random.RandomState(0)
    X = rng.rand(40, 2)
    ls = np.array([0.3, 0.45])
    t = VectorTree(None, 1, "euclidean", ls, "se", [1.7])
    K = t.kernel_matrix(X, X, False)
    d2 = (((X[:, None, :] - X[None, :, :]) / ls) ** 2).sum(-1)
    assert np.allclose(K, 1.7 * np.exp(-d2), rtol=1e-14, atol=0)       # no 1/2 factor (gprfopt.py:238-239)
    D = t.kernel_matrix(X, X, True)
    assert np.allclose(D, np.sqrt(d2), rtol=1e-15, atol=0)
    row = np.empty(40)
    t.kernel_deriv_wrt_xi_row(X, 5, 1, row)
    assert np.allclose(row, -2 * (X[5, 1] - X[:, 1]) / ls[1] ** 2 * K[5], rtol=1e-13, atol=1e-300)
    dl = t.kernel_deriv_wrt_i(X, X, 0, 1, D)
    assert np.allclose(dl, 2 * (X[:, None, 0] - X[None, :, 0]) ** 2 / ls[0] ** 3 * K, rtol=1e-13, atol=1e-300)


@pytest.mark.parametrize("dfn,wfn,params", [("euclidean", "se", [0.3, 0.4]), ("lld", "matern32", [40.0, 20.0])])
def test_kernel_derivatives_fd(dfn, wfn, params):
    rng = np.random.RandomState(2)
    if dfn == "lld":
        X = np.stack([130 + rng.randn(12) * 0.2, -2 + rng.randn(12) * 0.2, np.abs(rng.randn(12)) * 20], axis=1)
        h = 1e-5
    else:
        X = rng.rand(12, 2)
        h = 1e-6
    dx = X.shape[1]
    t = VectorTree(None, 1, dfn, params, wfn, [1.2])
    for p in (0, 7):
        for i in range(dx):
            row = np.empty(12)
            t.kernel_deriv_wrt_xi_row(X, p, i, row)
            Xp, Xm = X.copy(), X.copy()
            Xp[p, i] += h
            Xm[p, i] -= h
            fd = (t.kernel_matrix(Xp[p:p + 1], X, False) - t.kernel_matrix(Xm[p:p + 1], X, False))[0] / (2 * h)
            mask = np.arange(12) != p
            assert np.allclose(row[mask], fd[mask], rtol=2e-6, atol=1e-9)
    D = t.kernel_matrix(X, X, True)
    for i in range(2):
        dl = t.kernel_deriv_wrt_i(X, X, i, 1, D)
        pp, pm = list(params), list(params)
        hh = params[i] * 1e-6
        pp[i] += hh
        pm[i] -= hh
        fd = (VectorTree(None, 1, dfn, pp, wfn, [1.2]).kernel_matrix(X, X, False)
              - VectorTree(None, 1, dfn, pm, wfn, [1.2]).kernel_matrix(X, X, False)) / (2 * hh)
        assert np.allclose(dl, fd, rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("name", ["tiny_parts.npz", "lld_toy.npz", "degenerate.npz", "c1_small.npz"])
def test_oracle_reproduces_golden(name):
    """The committed vectors were produced by this oracle; on any machine (GPU box included) it must give
    them back, so the GPU parity tests compare against the same numbers everywhere."""
    z = load_golden(name)
    if name == "c1_small.npz":
        X, Y, suffix = z["X_obs"], z["SY"], "_gprf"
    else:
        X, Y, suffix = z["X"], z["Y"], ""
    th = z["theta"]
    dfn, wfn = ("lld", "matern32") if name == "lld_toy.npz" else ("euclidean", "se")
    cov = GPCov([th[1]], th[2:], dfn, wfn)
    blocks = blocks_from_csr(z["block_ptr"], z["block_pts"])
    nbrs = [tuple(r) for r in z["neighbors"]]
    g = GPRFRef(X, Y, None, cov, th[0], block_idxs=blocks, neighbors=nbrs)
    ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
    assert np.isclose(ll, z["ll" + suffix], rtol=1e-12)
    assert np.allclose(gX, z["gX" + suffix], rtol=1e-9, atol=1e-9 * np.abs(gX).max())
    assert np.allclose(gC, z["gC" + suffix], rtol=1e-9)


def test_gradient_fd_small():
    z = load_golden("tiny_parts.npz")
    th = z["theta"]
    cov = GPCov([th[1]], th[2:], "euclidean", "se")
    blocks = blocks_from_csr(z["block_ptr"], z["block_pts"])
    g = GPRFRef(z["X"].copy(), z["Y"], None, cov, th[0], block_idxs=blocks, neighbors=[(1, 0)])
    _, gX, _ = g.llgrad(grad_X=True)
    X0 = z["X"]
    for (p, i) in [(0, 0), (30, 1), (59, 0)]:
        Xp, Xm = X0.copy(), X0.copy()
        Xp[p, i] += 1e-6
        Xm[p, i] -= 1e-6
        g.X = Xp
        fp = g.llgrad()[0]
        g.X = Xm
        fm = g.llgrad()[0]
        assert np.isclose((fp - fm) / 2e-6, gX[p, i], rtol=1e-5)
