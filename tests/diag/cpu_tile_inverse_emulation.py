"""Does the device's blocked forward substitution with EXPLICIT 16 x 16 diagonal-tile inverses cost accuracy?  A numpy fp64
emulation of that algorithm (LAPACK Cholesky in front) against the 80-bit evaluation, next to exact tile solves, one step
of refinement per tile, and the LAPACK path (potri / potrs): all four within 4 % of each other on five pair-sized units
— the tile inverses are not where the device's per-unit 1.5x against LAPACK comes from.  CPU only:
    python tests/diag/cpu_tile_inverse_emulation.py"""
import sys, numpy as np, scipy.linalg as sl
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ld_truth import unit_llgrad_ld
rng = np.random.RandomState(5)
def unit(m=200):
    X = rng.rand(m, 2) * [0.2, 0.1]
    ls = np.array([0.06, 0.06]); nv = 0.01
    d = (X[:, None, :] - X[None, :, :]) / ls
    Knf = np.exp(-np.sum(d * d, axis=2)); K = Knf + nv * np.eye(m)
    Y = np.linalg.cholesky(K) @ rng.randn(m, 50)
    return X, Y, K, Knf, ls, nv
def grad_from(M, X, Knf, ls):
    Kz = Knf.copy(); np.fill_diagonal(Kz, 0)
    g = np.zeros_like(X)
    for dd in range(2):
        D = -2 * (X[:, None, dd] - X[None, :, dd]) / (ls[dd] ** 2) * Kz
        g[:, dd] = np.sum(M * D, axis=1)
    return g
def lapack(K, Y, dy):
    L = sl.cholesky(K, lower=True)
    P = sl.cho_solve((L, True), np.eye(len(K)))
    A = sl.cho_solve((L, True), Y)
    return A @ A.T - dy * P
def blocked(K, Y, dy, mode):
    m = len(K); mp = (m + 15) // 16 * 16; T = mp // 16
    Kp = np.eye(mp); Kp[:m, :m] = K
    Yp = np.zeros((mp, Y.shape[1])); Yp[:m] = Y
    U = sl.cholesky(Kp, lower=False)
    R = np.hstack([np.eye(mp), Yp])            # U^T [W | Z] = [I | Y]
    out = np.zeros_like(R)
    acc = R.copy()
    for r in range(T):
        sl_ = slice(16 * r, 16 * r + 16)
        Urr = U[sl_, sl_]
        if mode == "exact":
            w = sl.solve_triangular(Urr.T, acc[sl_], lower=True)
        else:
            V = sl.solve_triangular(Urr, np.eye(16), lower=False)      # explicit inverse, as the device keeps it
            w = V.T @ acc[sl_]
            if mode == "refine":
                res = acc[sl_] - Urr.T @ w
                w = w + V.T @ res
        out[sl_] = w
        acc[16 * r + 16:] -= U[sl_, 16 * r + 16:].T @ w
    W = out[:, :mp]; Z = out[:, mp:]
    A = W.T @ Z
    M = A @ A.T - dy * (W.T @ W)
    return M[:m, :m]
res = {k: [] for k in ("lapack", "explicit", "exact", "refine")}
for it in range(5):
    X, Y, K, Knf, ls, nv = unit(150 + 20 * it)
    _, gt = unit_llgrad_ld(X, Y, nv, 1.0, ls)
    gt = gt.astype(np.float64)
    res["lapack"].append(np.max(np.abs(grad_from(lapack(K, Y, 50), X, Knf, ls) - gt)))
    for mode in ("explicit", "exact", "refine"):
        res[mode].append(np.max(np.abs(grad_from(blocked(K, Y, 50, mode), X, Knf, ls) - gt)))
    print(it, {k: "%.2e" % v[-1] for k, v in res.items()}, "max|g| %.2e" % np.abs(gt).max())
for k, v in res.items(): print(k, "mean %.3e" % np.mean(v), "ratio to lapack %.2f" % (np.mean(v) / np.mean(res["lapack"])))
