"""80-bit (np.longdouble, 64-bit mantissa) evaluation of one unit's gradient from X directly — an
extended-precision yardstick for BOTH the fp64 oracle and the GPU (test infrastructure).  Follows
gprf.py:496-573 for the ("euclidean","se") kernel."""
import numpy as np

LD = np.longdouble


def _chol(K):
    n = K.shape[0]
    L = np.zeros_like(K)
    for j in range(n):
        L[j, j] = np.sqrt(K[j, j] - np.dot(L[j, :j], L[j, :j]))
        if j + 1 < n:
            L[j + 1:, j] = (K[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return L


def _solve_lower(L, B):
    Z = np.zeros_like(B)
    for i in range(L.shape[0]):
        Z[i] = (B[i] - L[i, :i] @ Z[:i]) / L[i, i]
    return Z


def _solve_upper(U, B):
    Z = np.zeros_like(B)
    for i in range(U.shape[0] - 1, -1, -1):
        Z[i] = (B[i] - U[i, i + 1:] @ Z[i + 1:]) / U[i, i]
    return Z


def unit_llgrad_ld(X, Y, nv, sv, ls):
    X = X.astype(LD)
    Y = Y.astype(LD)
    ls = np.asarray(ls, dtype=LD)
    m, dy = Y.shape
    diff = (X[:, None, :] - X[None, :, :]) / ls
    Knf = LD(sv) * np.exp(-np.sum(diff * diff, axis=2))
    K = Knf + LD(nv) * np.eye(m, dtype=LD)
    L = _chol(K)
    P = _solve_upper(L.T.copy(), _solve_lower(L, np.eye(m, dtype=LD)))
    A = P @ Y
    M = A @ A.T - LD(dy) * P
    np.fill_diagonal(Knf, 0)
    g = np.zeros(X.shape, dtype=LD)
    for d in range(X.shape[1]):
        D = LD(-2) * (X[:, None, d] - X[None, :, d]) / (ls[d] * ls[d]) * Knf
        g[:, d] = np.sum(M * D, axis=1)
    ll = -LD(0.5) * np.sum(Y * A) - LD(dy) * np.sum(np.log(np.diag(L))) - LD(0.5) * dy * m * np.log(2 * LD(np.pi))
    return ll, g
