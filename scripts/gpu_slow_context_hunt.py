"""Hunt for a slow context: build the north-star GPRF repeatedly in one process, time 100 sequential evaluations on each,
and for every one print the HIP-event stage times of 30 more — which stage carries the excess when a context is slow?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gprf_amd import GPCov, Blocker, grid_centers
from gprf_amd.gprf import GPRF
rng = np.random.RandomState(1)
n = 10000; X = rng.rand(n, 2); Y = rng.randn(n, 50)
b = Blocker(grid_centers(100)); nb = b.neighbors()
Xs = [np.clip(X + 0.002 * rng.randn(n, 2), 0, 1) for _ in range(10)]
for c in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    g = GPRF(X, Y, b.block_clusters, GPCov([1.0], [0.06, 0.06], "euclidean", "se"), 0.01, neighbors=nb)
    for k in range(20): g.update_X(Xs[k % 10]); g.llgrad(grad_X=True)
    t0 = time.perf_counter()
    for k in range(100): g.update_X(Xs[k % 10]); g.llgrad(grad_X=True)
    ms = 1e3 * (time.perf_counter() - t0) / 100
    g._ctx.set_timing(True, reset=True)
    for k in range(30): g.update_X(Xs[k % 10]); g.llgrad(grad_X=True)
    tm = g._ctx.get_timing(); g._ctx.set_timing(False); tm.pop("count", None)
    t0 = time.perf_counter()
    for k in range(100): g.update_X(Xs[k % 10]); g.llgrad(grad_X=True)
    ms2 = 1e3 * (time.perf_counter() - t0) / 100
    print("context %2d: %.3f ms | again %.3f ms | stages(us) %s" % (c, ms, ms2, {k: round(1e3 * v, 1) for k, v in tm.items()}), flush=True)
    g.close()
