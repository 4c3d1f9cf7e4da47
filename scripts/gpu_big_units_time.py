"""Stage times of units beyond one workgroup (the blocked multi-launch path): n points of the north-star recipe's shape (random
outputs), `blocks` grid blocks with / without their pairs:
    python scripts/gpu_big_units_time.py [n] [blocks] [pairs 0|1] [reps]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gprf_amd import Blocker, grid_centers, GPCov
from gprf_amd.gprf import GPRF
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 0
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rng = np.random.RandomState(31)
X = rng.rand(n, 2); Y = rng.randn(n, 50)
b = Blocker(grid_centers(nb))
g = GPRF(X, Y, b.block_clusters, GPCov([1.0], [0.06, 0.06], "euclidean", "se"), 0.01, neighbors=b.neighbors() if pairs else [])
sz = np.array([len(u) for u in g.block_idxs])
m = np.concatenate([sz, [sz[i] + sz[j] for i, j in g.neighbors]]).astype(float)
flops = float(np.sum(m ** 3 + 4 * m ** 2 * 50))
g.llgrad(grad_X=True)
ts = []
for _ in range(reps):
    t = time.time(); g.llgrad(grad_X=True); ts.append(time.time() - t)
print("n=%d blocks=%d pairs=%d largest unit %d: sync eval median %.3f ms = %.1f TFLOP/s algorithmic"
      % (n, len(sz), len(g.neighbors), int(m.max()), np.median(ts) * 1e3, flops / np.median(ts) / 1e12))
g._ctx.set_timing(True, reset=True)
for _ in range(reps): g.llgrad(grad_X=True)
st = g._ctx.get_timing()
print("stages(ms)", {k: round(v, 3) for k, v in st.items() if k != "count"})
g.close()
