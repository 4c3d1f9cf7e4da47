import sys, os, time
import numpy as np
sys.path.insert(0, '/root/repo')
from gprf_amd import GPCov, Blocker, grid_centers
from gprf_amd.gprf import GPRF, _csr_from_block_idxs
rng = np.random.RandomState(1)
n = 10000; X = rng.rand(n, 2); Y = rng.randn(n, 50)
b = Blocker(grid_centers(100)); nbrs = b.neighbors()
g = GPRF(X, Y, b.block_clusters, GPCov([1.0], [0.06, 0.06], "euclidean", "se"), 0.01, neighbors=nbrs)
g.llgrad(grad_X=True)
def t(f, n=20):
    ts=[]
    for _ in range(n):
        t0=time.perf_counter(); f(); ts.append(time.perf_counter()-t0)
    return np.median(ts)*1e3
X2 = X + 1e-3*rng.randn(n,2)
print("block_clusters %.3f ms" % t(lambda: b.block_clusters(X2)))
bl = b.block_clusters(X2)
print("block_assignment_fast %.3f ms" % t(lambda: b.block_assignment_fast(X2)))
print("csr %.3f ms" % t(lambda: _csr_from_block_idxs(bl)))
ptr, pts = _csr_from_block_idxs(bl)
print("set_blocks %.3f ms" % t(lambda: g._ctx.set_blocks(ptr, pts)))
def f():
    g._ctx.set_blocks(ptr, pts); g._ctx.eval(X2, True, False)
print("set_blocks+eval (rebuild) %.3f ms" % t(f))
print("eval only %.3f ms" % t(lambda: g._ctx.eval(X2, True, False)))
def f2():
    g.update_X(X2); g.llgrad(grad_X=True)
print("update_X+llgrad %.3f ms" % t(f2))
# re-blocking that really changes the partition every time (a few points cross block borders)
ts_u, ts_l = [], []
for it in range(30):
    X3 = X + 2e-3 * rng.randn(n, 2)
    t0 = time.perf_counter(); g.update_X(X3); t1 = time.perf_counter(); g.llgrad(grad_X=True); t2 = time.perf_counter()
    ts_u.append(t1 - t0); ts_l.append(t2 - t1)
print("changing partition: update_X %.3f ms  llgrad %.3f ms" % (np.median(ts_u) * 1e3, np.median(ts_l) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for it in range(20):
    X3 = X + 2e-3 * rng.randn(n, 2); g.update_X(X3); g.llgrad(grad_X=True)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
