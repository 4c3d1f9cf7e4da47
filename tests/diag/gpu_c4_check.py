"""C4-shaped run on one GPU (BASELINE configs[3]: n=80000, 841 blocks + 3192 pairs, yd=50, lscale=0.02, task xcov):
random-normal Y (the N=80500 prior draw is out of reach of the reference's own dense sampler), checks that the
path runs at that size, times it, and spot-checks two units against the oracle."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gprf_amd import GPCov, Blocker, grid_centers
from gprf_amd.gprf import GPRF
rng = np.random.RandomState(2)
n = 80000
X = rng.rand(n, 2); Y = rng.randn(n, 50)
b = Blocker(grid_centers(800)); nbrs = b.neighbors()
t = time.time(); blocks = b.block_clusters(X); print("blocks %d pairs %d (host blocking %.2fs)" % (len(blocks), len(nbrs), time.time() - t))
g = GPRF(X, Y, b.block_clusters, GPCov([1.0], [0.02, 0.02], "euclidean", "se"), 0.01, block_idxs=blocks, neighbors=nbrs)
t = time.time(); ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True); print("first eval %.1f ms  ll=%.6e" % ((time.time() - t) * 1e3, ll))
g._ctx.set_timing(True, reset=True)
ts = []
for _ in range(10):
    t = time.time(); g.llgrad(grad_X=True, grad_cov=True); ts.append(time.time() - t)
st = g._ctx.get_timing()
print("median %.2f ms  stages(us) %s  work %s" % (np.median(ts) * 1e3, {k: round(v * 1e3) for k, v in st.items() if k != 'count'}, g._ctx.work_estimate()))
t = time.time(); g.update_X(X + 1e-4 * rng.randn(n, 2)); g.llgrad(grad_X=True); print("first update_X + llgrad (centres uploaded, everything re-blocked) %.1f ms" % ((time.time() - t) * 1e3))
for step in (1e-4, 1e-4, 1e-7, 1e-7):      # a few points change block / nobody does
    X2 = g.X + step * rng.randn(n, 2)
    t = time.time(); g.update_X(X2); t1 = time.time(); g.llgrad(grad_X=True, grad_cov=True); t2 = time.time()
    print("step %.0e: update_X %.2f ms  llgrad %.2f ms" % (step, (t1 - t) * 1e3, (t2 - t1) * 1e3))
# spot check against the oracle on the Bethe-weighted sum restricted to two units
from oracle.gprf_ref import GPRFRef
from oracle.vector_tree import GPCov as OC
ref = GPRFRef(g.X, Y, None, OC([1.0], [0.02, 0.02], "euclidean", "se"), 0.01, block_idxs=g.block_idxs, neighbors=nbrs)
ctx = g._ctx
ctx.debug_run(g.X, 6)
for l in (3, len(blocks) + 1000):
    m, mp, gu = ctx.debug_unit_shape(l)
    idx = g.block_idxs[gu] if gu < len(blocks) else np.concatenate([g.block_idxs[nbrs[gu - len(blocks)][0]], g.block_idxs[nbrs[gu - len(blocks)][1]]])
    o = ref.gaussian_llgrad(g.X[idx], Y[idx], grad_X=True)
    d = ctx.debug_fetch(l, 4)[:m, :2]; s5 = ctx.debug_fetch(l, 5)
    print("unit %d m=%d  ll rel err %.2e  gX max err %.2e (max|gX| %.2e)" % (gu, m, abs(s5[0] - o[0]) / abs(o[0]), np.max(np.abs(d - o[1])), np.max(np.abs(o[1]))))
g.close()
