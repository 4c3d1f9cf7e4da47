"""Stage times with units of 21-30 tiles per edge (16 blocks of ~206 points + their pairs, SE kernel; FILL=1: K through the pool):
    python scripts/gpu_wide_units_time.py [n] [blocks]
(A/B: GPRF_DIAG=potrf_gw=0 sends the units of 21..28 tiles to the generic kernel instead of k_potrf_reg8w)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gprf_amd import Blocker, grid_centers, GPCov
from gprf_amd.gprf import GPRF
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3300
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.RandomState(31)
X = rng.rand(n, 2); Y = rng.randn(n, 50)
b = Blocker(grid_centers(nb))
g = GPRF(X, Y, b.block_clusters, GPCov([1.0], [0.09, 0.11], "euclidean", "se"), 0.02, neighbors=b.neighbors())
sz = [len(u) for u in g.block_idxs]
print("n=%d blocks=%d pairs=%d tiles per edge of the pairs:" % (n, len(sz), len(g.neighbors)), sorted(set((sz[i] + sz[j] + 15) // 16 for i, j in g.neighbors)))
for _ in range(3): g.llgrad(grad_X=True, grad_cov=True)
ts = []
for _ in range(20):
    t = time.time(); g.llgrad(grad_X=True, grad_cov=True); ts.append(time.time() - t)
print("sync eval: median %.3f ms" % (np.median(ts) * 1e3))
g._ctx.set_timing(True, reset=True)
for _ in range(20): g.llgrad(grad_X=True, grad_cov=True)
st = g._ctx.get_timing()
print("stages(us)", {k: round(v * 1e3, 1) for k, v in st.items() if k != "count"})
g.close()
