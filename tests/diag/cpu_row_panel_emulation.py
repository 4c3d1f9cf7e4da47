"""What does the Cholesky's row panel as an explicit tile inverse on the matrix pipe (U_jk = V_jj^T C_jk, round 4) cost in
accuracy against the forward substitution of rounds 1-3, and which correction of V_jj gives it back?  A numpy fp64
emulation of the device's blocked factorisation (16-row panels, root-free diagonal tile, hierarchical trailing update) with
the ROW PANEL as the variable, against the 80-bit evaluation and LAPACK, on the north-star configuration's units:
    subst     forward substitution with the unit triangular G = D^-1 U_jj, scaled by 1 / U_kk (rounds 1-3)
    vinv      V_jj by the column operations of tile_inverse(), U_jk = V_jj^T C_jk summed from zero in k order (round 4)
    vinv_ns   the same with one Newton-Schulz step  V <- V + V (I - U_jj V)   (products summed from zero, k order)
    vinv_ns2  two such steps;  vinv_x  the correctly rounded inverse (80-bit back substitution): what ANY V can give
    vinv_nsl  ...                                   V <- V + (I - V U_jj) V
    vinv_ref  one step of refinement on the panel itself: U_jk += V^T (C_jk - U_jj^T U_jk)
CPU only (~40 s of input sampling, cached):
    python tests/diag/cpu_row_panel_emulation.py [n_pairs] [n_unaries]"""
import os
import sys

import numpy as np
import scipy.linalg as sl

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from ld_truth import unit_llgrad_ld

LD = np.longdouble


def fma(a, b, c):
    return np.asarray(LD(a) * LD(b) + LD(c), dtype=np.float64)


def mm_tn(A, B):
    """sum_k A[k][i] B[k][j], from zero, k ascending, fused: the MFMA chain"""
    S = np.zeros((A.shape[1], B.shape[1]))
    for k in range(A.shape[0]):
        S = fma(A[k][:, None], B[k][None, :], S)
    return S


def tile_inverse(Uj):
    """row r of V: x U = e_r by the column operations of the device's tile_inverse()"""
    rd = 1.0 / np.diag(Uj)
    V = np.eye(16)
    for k in range(16):
        V[:, k] = V[:, k] * rd[k]
        for i in range(k + 1, 16):
            V[:, i] = fma(-Uj[k, i], V[:, k], V[:, i])
    return V


def _inv_ld(Uj):
    """the correctly rounded inverse: back substitution in 80-bit arithmetic"""
    U = Uj.astype(LD)
    V = np.zeros((16, 16), dtype=LD)
    for c in range(16):
        e = np.zeros(16, dtype=LD); e[c] = 1
        for i in range(15, -1, -1):
            V[i, c] = (e[i] - U[i, i + 1:] @ V[i + 1:, c]) / U[i, i]
    return V


def chol_blocked(K, panel):
    m = len(K); mp = (m + 15) // 16 * 16
    A = np.eye(mp); A[:m, :m] = K
    U = np.zeros((mp, mp))
    for j in range(0, mp, 16):
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
        if j + 16 >= mp:
            break
        B = A[j:j+16, j+16:].copy()
        if panel == "subst":
            dg = np.diag(Uj)
            G = Uj / dg[:, None]
            Z = B.copy()
            for c in range(16):
                for a in range(c + 1, 16):
                    Z[a] = fma(-G[c, a], Z[c], Z[a])
            P = Z / dg[:, None]
        else:
            V = tile_inverse(Uj)
            if panel == "vinv_ns":
                R = np.eye(16) - mm_tn(Uj.T.copy(), V)            # I - U V
                V = V + mm_tn(V.T.copy(), R)                      # V + V R
            elif panel == "vinv_ns2":
                for _ in range(2):
                    R = np.eye(16) - mm_tn(Uj.T.copy(), V)
                    V = V + mm_tn(V.T.copy(), R)
            elif panel == "vinv_x":
                V = np.asarray(np.linalg.inv(Uj.astype(LD)) if False else _inv_ld(Uj), dtype=np.float64)
            elif panel == "vinv_nsl":
                R = np.eye(16) - mm_tn(V.T.copy(), Uj)            # I - V U
                V = V + mm_tn(R.T.copy(), V)                      # V + R V
            P = mm_tn(V, B)                                       # V^T B
            if panel == "vinv_ref":
                Rs = B - mm_tn(Uj, P)                             # C - U^T P
                P = P + mm_tn(V, Rs)
        U[j:j+16, j+16:] = P
        T = A[j+16:, j+16:]
        S = np.zeros_like(T)
        for k in range(16):
            S = fma(P[k][:, None], P[k][None, :], S)
        T[:] = T - S
    return U[:m, :m]


def M_from_U(U, Y, dy):
    m = len(U)
    W = sl.solve_triangular(U, np.eye(m), trans='T', lower=False)
    A = sl.cho_solve((U, False), Y)
    return A @ A.T - dy * (W.T @ W)


def grad_from(M, X, Knf, ls):
    Kz = Knf.copy(); np.fill_diagonal(Kz, 0)
    g = np.zeros_like(X)
    for dd in range(2):
        D = -2 * (X[:, None, dd] - X[None, :, dd]) / (ls[dd] ** 2) * Kz
        g[:, dd] = np.sum(M * D, axis=1)
    return g


MODES = ("subst", "vinv", "vinv_ns", "vinv_ns2", "vinv_x", "vinv_nsl", "vinv_ref")


def units(n_pairs, n_unaries):
    from gprf_amd.synthetic import SampledData
    from gprf_amd import grid_centers
    sd = SampledData(n=10500, ntrain=10000, lscale=0.06, obs_std=0.02, yd=50, seed=0, use_gpu=False,
                     cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gprf_bench_data"))
    sd.set_centers(grid_centers(100))
    r = np.random.RandomState(3)
    ls = np.array([0.06, 0.06])
    todo = [("pair", np.concatenate([sd.block_idxs[i], sd.block_idxs[j]]))
            for (i, j) in [sd.neighbors[q] for q in r.choice(len(sd.neighbors), n_pairs, replace=False)]]
    todo += [("unary", sd.block_idxs[b]) for b in r.choice(len(sd.block_idxs), n_unaries, replace=False)]
    for kind, idx in todo:
        Xu, Yu = sd.X_obs[idx], sd.SY[idx]
        d = (Xu[:, None, :] - Xu[None, :, :]) / ls
        Knf = np.exp(-np.sum(d * d, axis=2))
        yield kind, Xu, Yu, Knf + 0.01 * np.eye(len(idx)), Knf, ls, 0.01


if __name__ == "__main__":
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    n_unaries = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    res = {kind: {k: [] for k in ("lapack",) + MODES} for kind in ("pair", "unary")}
    for it, (kind, X, Y, K, Knf, ls, nv) in enumerate(units(n_pairs, n_unaries)):
        _, gt = unit_llgrad_ld(X, Y, nv, 1.0, ls)
        gt = gt.astype(np.float64)
        Us = {"lapack": sl.cholesky(K, lower=False)}
        for md in MODES:
            Us[md] = chol_blocked(K, md)
        for k, U in Us.items():
            res[kind][k].append(np.max(np.abs(grad_from(M_from_U(U, Y, 50), X, Knf, ls) - gt)))
        print(it, kind, len(K), {k: "%.2e" % v[-1] for k, v in res[kind].items()}, flush=True)
    for kind in res:
        if not res[kind]["lapack"]:
            continue
        print("--", kind, len(res[kind]["lapack"]), "units")
        lap = np.array(res[kind]["lapack"])
        for k, v in res[kind].items():
            v = np.array(v)
            print("%-9s |. - true| max %.3e mean %.3e   max/lapack-max %.2f  mean ratio %.2f  worst unit %.2f"
                  % (k, v.max(), v.mean(), v.max() / lap.max(), np.mean(v / lap), np.max(v / lap)))
