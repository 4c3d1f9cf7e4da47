"""Why was the device's Cholesky factor 1.1-1.2x as far from an 80-bit factorisation as LAPACK's (round 2), and what
cures it?  A numpy fp64 emulation of the device's blocked right-looking factorisation (16-row panels, root-free diagonal
tile, unit-triangular substitution) with the trailing update's ORDER OF ACCUMULATION as the variable, against the
80-bit evaluation and LAPACK, on pair units of the north-star configuration (REAL=1; ~40 s of input sampling) or on
synthetic units of the same shape:
    seq    every product enters the running entry by itself (fused multiply-add): what the MFMA chain did      1.16-1.19x
    hier   a step's 16 products summed from zero, then ONE addition into the running entry (since round 3)      0.67x
    hier4  the same per 4 products                                                                               0.66x
    seq4   4 products exact, one rounding (what an MFMA with a single internal rounding would give)              0.62x
    zinit  accumulators start at zero and take all products, K added when a row is taken                         1.07x
    hfrac0.25 / hfrac0.5   hier for the first quarter / half of the steps only                                   0.80x / 0.67x
(gradient-row error relative to LAPACK's, means over 10 REAL units; the synthetic units — uniform points in a 0.2 x 0.1 box —
rank the schemes differently: there every blocked variant beats LAPACK and seq is the best).  CPU only:
    [REAL=1] python tests/diag/cpu_accumulation_order.py [units]"""
import os
import sys

import numpy as np
import scipy.linalg as sl

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from ld_truth import unit_llgrad_ld
rng = np.random.RandomState(5)
LD = np.longdouble
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
def fma(a, b, c):   # ~fused
    return np.asarray(LD(a) * LD(b) + LD(c), dtype=np.float64)
def chol_zinit(K):
    """accumulators start at ZERO and take every product sequentially; the kernel-matrix tile is added when the row is taken"""
    m = len(K); mp = (m + 15) // 16 * 16
    A0 = np.eye(mp); A0[:m, :m] = K
    S = np.zeros((mp, mp))
    U = np.zeros((mp, mp))
    for j in range(0, mp, 16):
        D = A0[j:j+16, j:j+16] - S[j:j+16, j:j+16]
        Uj = np.zeros((16, 16))
        for k in range(16):
            p = D[k, k]; r = D[k, k:].copy(); w = r / p
            for i in range(k + 1, 16):
                D[i, i:] = fma(-w[i - k], r[i - k:], D[i, i:])
            Uj[k, k:] = r / np.sqrt(p); Uj[k, k] = np.sqrt(p)
        U[j:j+16, j:j+16] = Uj
        if j + 16 >= mp: break
        B = A0[j:j+16, j+16:] - S[j:j+16, j+16:]
        dg = np.diag(Uj); G = Uj / dg[:, None]
        Z = B.copy()
        for c in range(16):
            for a in range(c + 1, 16):
                Z[a] = fma(-G[c, a], Z[c], Z[a])
        P = Z / dg[:, None]
        U[j:j+16, j+16:] = P
        T = S[j+16:, j+16:]
        for k in range(16):
            T[:] = fma(P[k][:, None], P[k][None, :], T)
    return U[:m, :m]
def chol_blocked(K, mode):
    if mode == 'zinit': return chol_zinit(K)
    """right-looking, 16-row panels, upper. mode: 'seq' = every product accumulated into the trailing entry one at a time (FMA);
    'hier' = the 16 products of a step summed from zero (FMA chain), then ONE addition into the trailing entry;
    'seq4' = groups of 4 products summed from zero then added (MFMA with internal single rounding?)"""
    m = len(K); mp = (m + 15) // 16 * 16
    A = np.eye(mp); A[:m, :m] = K
    U = np.zeros((mp, mp))
    for j in range(0, mp, 16):
        # diagonal tile: unblocked (LDL-ordered like the device; rounding differences inside the tile are second order)
        D = A[j:j+16, j:j+16].copy()
        Uj = np.zeros((16, 16))
        for k in range(16):
            p = D[k, k]
            r = D[k, k:].copy()
            w = r / p
            for i in range(k + 1, 16):
                D[i, i:] = fma(-w[i - k], r[i - k:], D[i, i:])
            Uj[k, k:] = r / np.sqrt(p)
            Uj[k, k] = np.sqrt(p)
        U[j:j+16, j:j+16] = Uj
        if j + 16 >= mp: break
        # row panel: forward substitution with G = D^-1 U, scale
        B = A[j:j+16, j+16:].copy()
        dg = np.diag(Uj)
        G = Uj / dg[:, None]
        Z = B.copy()
        for c in range(16):
            for a in range(c + 1, 16):
                Z[a] = fma(-G[c, a], Z[c], Z[a])
        P = Z / dg[:, None]
        U[j:j+16, j+16:] = P
        # trailing update
        T = A[j+16:, j+16:]
        if mode == 'seq':
            for k in range(16):
                T[:] = fma(-P[k][:, None], P[k][None, :], T)
        elif mode == 'hier':
            S = np.zeros_like(T)
            for k in range(16):
                S = fma(P[k][:, None], P[k][None, :], S)
            T[:] = T - S
        elif mode.startswith('hfrac'):      # hierarchical for the first fraction of the steps, sequential afterwards
            frac = float(mode[5:])
            if j < frac * mp:
                S = np.zeros_like(T)
                for k in range(16):
                    S = fma(P[k][:, None], P[k][None, :], S)
                T[:] = T - S
            else:
                for k in range(16):
                    T[:] = fma(-P[k][:, None], P[k][None, :], T)
        elif mode.startswith('hlast'):      # sequential first, hierarchical for the LAST fraction
            frac = float(mode[5:])
            if j >= (1 - frac) * mp:
                S = np.zeros_like(T)
                for k in range(16):
                    S = fma(P[k][:, None], P[k][None, :], S)
                T[:] = T - S
            else:
                for k in range(16):
                    T[:] = fma(-P[k][:, None], P[k][None, :], T)
        elif mode == 'hier4':
            for k0 in range(0, 16, 4):
                S = np.zeros_like(T)
                for k in range(k0, k0 + 4):
                    S = fma(P[k][:, None], P[k][None, :], S)
                T[:] = T - S
        elif mode == 'seq4':
            for k0 in range(0, 16, 4):
                S = np.zeros_like(T, dtype=LD)
                for k in range(k0, k0 + 4):
                    S = S + LD(P[k][:, None]) * LD(P[k][None, :])
                T[:] = np.asarray(LD(T) - S, dtype=np.float64)
    return U[:m, :m]
def M_from_U(U, Y, dy):
    m = len(U)
    W = sl.solve_triangular(U, np.eye(m), trans='T', lower=False)
    A = sl.cho_solve((U, False), Y)
    return A @ A.T - dy * (W.T @ W)


MODES = ("seq", "hier", "hier4", "seq4", "zinit", "hfrac0.25", "hfrac0.5")


def real_units(count):
    from gprf_amd.synthetic import SampledData
    from gprf_amd import grid_centers
    sd = SampledData(n=10500, ntrain=10000, lscale=0.06, obs_std=0.02, yd=50, seed=0, use_gpu=False,
                     cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gprf_bench_data"))
    sd.set_centers(grid_centers(100))
    r = np.random.RandomState(3)
    ls = np.array([0.06, 0.06])
    for q in r.choice(len(sd.neighbors), 12, replace=False)[:count]:
        i, j = sd.neighbors[q]
        idx = np.concatenate([sd.block_idxs[i], sd.block_idxs[j]])
        Xu, Yu = sd.X_obs[idx], sd.SY[idx]
        d = (Xu[:, None, :] - Xu[None, :, :]) / ls
        Knf = np.exp(-np.sum(d * d, axis=2))
        yield Xu, Yu, Knf + 0.01 * np.eye(len(idx)), Knf, ls, 0.01


if __name__ == "__main__":
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    units = real_units(count) if os.environ.get("REAL") else (unit(170 + 10 * (it % 8)) for it in range(count))
    res = {k: [] for k in ("lapack",) + MODES}
    for it, (X, Y, K, Knf, ls, nv) in enumerate(units):
        _, gt = unit_llgrad_ld(X, Y, nv, 1.0, ls)
        gt = gt.astype(np.float64)
        Us = {"lapack": sl.cholesky(K, lower=False)}
        for md in MODES:
            Us[md] = chol_blocked(K, md)
        for k, U in Us.items():
            res[k].append(np.max(np.abs(grad_from(M_from_U(U, Y, 50), X, Knf, ls) - gt)))
        print(it, len(K), {k: "%.2e" % v[-1] for k, v in res.items()}, flush=True)
    for k, v in res.items():
        print("%-10s gradient-row error mean %.3e   ratio to LAPACK %.2f" % (k, np.mean(v), np.mean(v) / np.mean(res["lapack"])))
