"""Where does the device's per-unit gradient pick up its (small) excess error over LAPACK?  For a few pair units of the
north-star configuration: the device's U, W = U^-T, A = K^-1 Y and gradient rows against an 80-bit evaluation, next to
the same quantities from fp64 LAPACK.
    python tests/diag/gpu_stage_error.py"""
import os, sys
import numpy as np
import scipy.linalg as sl
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ld_truth import _chol, _solve_lower, _solve_upper, LD
from gprf_amd.synthetic import SampledData
from gprf_amd import grid_centers

sd = SampledData(n=10500, ntrain=10000, lscale=0.06, obs_std=0.02, yd=50, seed=0, use_gpu=True)
sd.set_centers(grid_centers(100))
g = sd.build_gprf(local_dist=0.1)
g.llgrad(grad_X=True)
ctx = g._ctx
X = np.ascontiguousarray(sd.X_obs)
ctx.debug_run(X, 6)
nb = g.n_blocks
rng = np.random.RandomState(3)
rows = []
stage_rows = []
keep = []
for q in rng.choice(len(g.neighbors), int(os.environ.get("NUNITS", "4")), replace=False):
    i, j = g.neighbors[q]
    idx = np.concatenate([g.block_idxs[i], g.block_idxs[j]])
    m = len(idx)
    Xu, Yu = X[idx], sd.SY[idx]
    ls = np.array([0.06, 0.06])
    # 80-bit
    Xl = Xu.astype(LD); d = (Xl[:, None, :] - Xl[None, :, :]) / ls.astype(LD)
    Knf = np.exp(-np.sum(d * d, axis=2)); Kl = Knf + LD(0.01) * np.eye(m, dtype=LD)
    Ll = _chol(Kl); Ut = Ll.T
    Wt = _solve_lower(Ll, np.eye(m, dtype=LD))                  # W = L^-1 = U^-T
    At = _solve_upper(Ut.copy(), _solve_lower(Ll, Yu.astype(LD)))
    # fp64 LAPACK
    K64 = np.asarray(Kl, dtype=np.float64)      # (K rounded once from the 80-bit values: the same input for both)
    d64 = (Xu[:, None, :] - Xu[None, :, :]) / ls
    K64 = np.exp(-np.sum(d64 * d64, axis=2)) + 0.01 * np.eye(m)
    U64 = sl.cholesky(K64, lower=False)
    W64 = sl.solve_triangular(U64, np.eye(m), trans='T', lower=False)
    A64 = sl.cho_solve((U64, False), Yu)
    # device
    mp = (m + 15) // 16 * 16
    Ud = np.triu(ctx.debug_fetch(nb + q, 0)[:m, :m])
    Wd = np.tril(ctx.debug_fetch(nb + q, 1)[:m, :m])
    Ad = ctx.debug_fetch(nb + q, 3)[:50, :m].T
    f = lambda a, t: float(np.max(np.abs(a - t)) / np.max(np.abs(t)))
    # the gradient's last stage in numpy fp64 from each side's W and A, and the device's own rows, against the truth
    def grad_np(W, A):
        M = A @ A.T - 50.0 * (W.T @ W)
        Kz = np.exp(-np.sum(d64 * d64, axis=2)); np.fill_diagonal(Kz, 0.0)
        out = np.zeros((m, 2))
        for dd in range(2):
            D = -2.0 * (Xu[:, None, dd] - Xu[None, :, dd]) / ls[dd] ** 2 * Kz
            out[:, dd] = np.sum(M * D, axis=1)
        return out
    Pt = Wt.T @ Wt; Mt = At @ At.T - LD(50) * Pt
    Kzt = Knf.copy(); np.fill_diagonal(Kzt, 0)
    gt = np.zeros((m, 2), dtype=LD)
    for dd in range(2):
        gt[:, dd] = np.sum(Mt * (LD(-2) * (Xl[:, None, dd] - Xl[None, :, dd]) / (LD(ls[dd]) ** 2) * Kzt), axis=1)
    gt = np.asarray(gt, dtype=np.float64)
    gdev = ctx.debug_fetch(nb + q, 4)[:m, :2]
    e = lambda a: float(np.max(np.abs(a - gt)))
    # which stage carries the excess: LAPACK's triangular solves applied to the DEVICE's factor, and the device's K
    mp_ = (m + 15) // 16 * 16
    Ud_full = np.triu(ctx.debug_fetch(nb + q, 0)[:m, :m])
    W_dU = sl.solve_triangular(Ud_full, np.eye(m), trans='T', lower=False)
    A_dU = sl.cho_solve((Ud_full, False), Yu)
    print("     gradient rows |x - true| max:  device %.2e   numpy from the device's W, A %.2e   numpy from LAPACK's W, A %.2e   "
          "LAPACK solves on the DEVICE's U %.2e   (max|g| %.2e)"
          % (e(gdev), e(grad_np(Wd, Ad)), e(grad_np(W64, A64)), e(grad_np(W_dU, A_dU)), np.abs(gt).max()))
    keep.append((q, idx, gt, d64))
    stage_rows.append((e(gdev), e(grad_np(Wd, Ad)), e(grad_np(W64, A64)), e(grad_np(W_dU, A_dU))))
    rows.append((m, f(Ud, Ut), f(U64, Ut), f(Wd, Wt), f(W64, Wt), f(Ad, At), f(A64, At)))
    print("unit %4d m=%3d  U: gpu %.2e lapack %.2e | W: gpu %.2e lapack %.2e | A = K^-1 Y: gpu %.2e lapack %.2e   (max-abs / max|true|)"
          % ((q,) + rows[-1]))
r = np.array(rows)
print("means: U gpu/lapack %.2f   W %.2f   A %.2f" % (r[:, 1].mean() / r[:, 2].mean(), r[:, 3].mean() / r[:, 4].mean(), r[:, 5].mean() / r[:, 6].mean()))
# ... and LAPACK on the DEVICE's kernel matrix (fill-only run: the K pool holds the 64 x 64 blocks ti <= tj)
ctx.debug_run(X, 0)
kerr = []
for (q, idx, gt, d64) in keep:
    m = len(idx)
    Kd = ctx.debug_fetch(nb + q, 0)[:m, :m]
    Kd = np.triu(Kd) + np.triu(Kd, 1).T
    Xu, Yu = X[idx], sd.SY[idx]
    U_k = sl.cholesky(Kd, lower=False)
    W_k = sl.solve_triangular(U_k, np.eye(m), trans='T', lower=False)
    A_k = sl.cho_solve((U_k, False), Yu)
    M = A_k @ A_k.T - 50.0 * (W_k.T @ W_k)
    Kz = np.exp(-np.sum(d64 * d64, axis=2)); np.fill_diagonal(Kz, 0.0)
    out = np.zeros((m, 2))
    for dd in range(2):
        D = -2.0 * (Xu[:, None, dd] - Xu[None, :, dd]) / 0.06 ** 2 * Kz
        out[:, dd] = np.sum(M * D, axis=1)
    K64 = np.exp(-np.sum(d64 * d64, axis=2)) + 0.01 * np.eye(m)
    kerr.append((float(np.max(np.abs(out - gt))), float(np.max(np.abs(Kd - K64))), float(np.max(np.abs(Kd - K64) / K64))))
kerr = np.array(kerr)
print("LAPACK everything on the DEVICE's K: gradient-row error mean %.3e   (max |K_dev - K_numpy| %.2e abs, %.2e relative)"
      % (kerr[:, 0].mean(), kerr[:, 1].max(), kerr[:, 2].max()))
sr = np.array(stage_rows)
print("gradient-row error means over %d units: device %.3e | numpy(device W, A) %.3e | numpy(LAPACK W, A) %.3e | LAPACK solves on device U %.3e"
      % ((len(sr),) + tuple(sr.mean(axis=0))))
g.close()
