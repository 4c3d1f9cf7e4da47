"""GPU bring-up: per-stage comparison of the HIP pipeline against numpy + the oracle on small problems.
Run on a GPU box:  python tests/diag/gpu_stage_check.py"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gprf_amd import _capi, GPCov, Blocker, grid_centers
from gprf_amd.gprf import GPRF
from oracle.gprf_ref import GPRFRef
from oracle.vector_tree import GPCov as OCov


def stage_check(n, nb, dy, lscale, seed=0, pairs=True):
    rng = np.random.RandomState(seed)
    X = rng.rand(n, 2)
    Y = rng.randn(n, dy)
    cov = GPCov([1.0], [lscale, lscale * 1.1], "euclidean", "se")
    b = Blocker(grid_centers(nb))
    blocks = b.block_clusters(X)
    nbrs = b.neighbors() if pairs else []
    g = GPRF(X, Y, None, cov, 0.01, block_idxs=blocks, neighbors=nbrs)
    ref = GPRFRef(X, Y, None, OCov([1.0], [lscale, lscale * 1.1], "euclidean", "se"), 0.01, block_idxs=blocks, neighbors=nbrs)
    ctx = g._ctx
    g._push_neighbors(nbrs)
    ctx.debug_run(X, 6)
    nt, nl = ctx.num_units()
    worst = {}
    for l in range(nl):
        m, mp, gu = ctx.debug_unit_shape(l)
        if gu < len(blocks):
            idx = blocks[gu]
        else:
            i, j = nbrs[gu - len(blocks)]
            idx = np.concatenate([blocks[i], blocks[j]])
        Xu, Yu = X[idx], Y[idx]
        ll, gX, gC, parts = ref.gaussian_llgrad(Xu, Yu, grad_X=True, grad_cov=True, return_parts=True)
        K = parts["K"]
        Uref = np.linalg.cholesky(K).T
        U = ctx.debug_fetch(l, 0)
        W = ctx.debug_fetch(l, 1)
        Z = ctx.debug_fetch(l, 2)
        At = ctx.debug_fetch(l, 3)
        gXu = ctx.debug_fetch(l, 4)
        sc = ctx.debug_fetch(l, 5)
        e = {}
        e["U"] = np.max(np.abs(np.triu(U[:m, :m]) - Uref)) if m else 0
        Wref = np.linalg.inv(Uref).T if m else np.zeros((0, 0))
        e["W"] = np.max(np.abs(np.tril(W[:m, :m]) - Wref)) if m else 0
        e["Z"] = np.max(np.abs(Z[:m, :dy] - Wref @ Yu)) if m else 0
        e["At"] = np.max(np.abs(At[:dy, :m] - parts["Alpha"].T)) if m else 0
        e["gX"] = np.max(np.abs(gXu[:m, :2] - gX)) if m else 0
        e["ll"] = abs(sc[0] - ll)
        e["logdet"] = abs(sc[1] - parts["logdet"])
        for k, v in e.items():
            worst[k] = max(worst.get(k, 0), v)
        if l < 2 or any(not np.isfinite(v) or v > 1e-6 for v in e.values()):
            print("  unit", l, "g", gu, "m", m, {k: "%.2e" % v for k, v in e.items()}, "info", sc[3])
    print(" worst per stage:", {k: "%.2e" % v for k, v in worst.items()})
    r = g.llgrad(grad_X=True, grad_cov=True)
    o = ref.llgrad(grad_X=True, grad_cov=True)
    print(" full: ll %.10g vs %.10g  |dgX| %.3e (max|gX| %.3e)  gC" % (r[0], o[0], np.max(np.abs(r[1] - o[1])), np.max(np.abs(o[1]))),
          r[2], o[2])
    g.close()


def timing(n=10000, nb=100, dy=50, lscale=0.06, pairs=True, reps=20):
    rng = np.random.RandomState(1)
    X = rng.rand(n, 2)
    Y = rng.randn(n, dy)
    cov = GPCov([1.0], [lscale, lscale], "euclidean", "se")
    b = Blocker(grid_centers(nb))
    blocks = b.block_clusters(X)
    nbrs = b.neighbors() if pairs else []
    g = GPRF(X, Y, None, cov, 0.01, block_idxs=blocks, neighbors=nbrs)
    t = time.time(); r = g.llgrad(grad_X=True); t1 = time.time() - t
    g._ctx.set_timing(True, reset=True)
    ts = []
    for _ in range(reps):
        t = time.time(); r = g.llgrad(grad_X=True); ts.append(time.time() - t)
    st = g._ctx.get_timing()
    print(" timing n=%d nb=%d pairs=%d: first %.1f ms, median %.3f ms; stages(ms) %s; work %s" % (
        n, nb, len(nbrs), t1 * 1e3, np.median(ts) * 1e3, {k: round(v, 4) for k, v in st.items()}, g._ctx.work_estimate()))
    g.close()


if __name__ == "__main__":
    np.set_printoptions(precision=6, linewidth=200)
    print("== stage check n=200 nb=4 dy=10"); stage_check(200, 4, 10, 0.4)
    print("== stage check n=300 nb=4 dy=50 no pairs"); stage_check(300, 4, 50, 0.3, pairs=False)
    print("== stage check n=1000 nb=9 dy=50"); stage_check(1000, 9, 50, 0.15)
    print("== timing"); timing(pairs=False); timing(pairs=True)
