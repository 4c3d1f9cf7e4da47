"""C4-shaped problem (n = 80000, 841 blocks + 3192 pairs), both gradients, synchronous evaluations without timers:
   GPRF_DIAG=<form> python scripts/gpu_c4_ab.py [reps]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gprf_amd import GPCov, Blocker, grid_centers
from gprf_amd.gprf import GPRF
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.RandomState(1)
n, nb, dy, ls = 80000, 800, 50, 0.02
X = rng.rand(n, 2); Y = rng.randn(n, dy)
b = Blocker(grid_centers(nb))
g = GPRF(X, Y, b.block_clusters, GPCov([1.0], [ls, ls], "euclidean", "se"), 0.01, neighbors=b.neighbors())
Xs = [np.ascontiguousarray(X + 1e-4 * k * rng.randn(n, 2)) for k in range(3)]
for k in range(3):
    g.update_X(Xs[k]); g.llgrad(grad_X=True, grad_cov=True)
ts = []
for k in range(reps):
    t = time.perf_counter(); g.update_X(Xs[k % 3]); g.llgrad(grad_X=True, grad_cov=True); ts.append(time.perf_counter() - t)
print("C4 xcov, update_X + llgrad: median %.3f ms  min %.3f ms   [%s]" % (np.median(ts) * 1e3, np.min(ts) * 1e3, os.environ.get("GPRF_DIAG", "")))
g.close()
