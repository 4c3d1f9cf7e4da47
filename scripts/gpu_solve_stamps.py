"""Per-step cycle breakdown of k_solve_panel (wave 0 of each unit's Y workgroup), diagnostic build only:
   GPRF_BUILD_DEFS=-DGPRF_PROFILE python gprf_amd/build.py && python scripts/gpu_solve_stamps.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gprf_amd import GPCov, Blocker, grid_centers
from gprf_amd.gprf import GPRF
rng = np.random.RandomState(1)
n = 10000; X = rng.rand(n, 2); Y = rng.randn(n, 50)
b = Blocker(grid_centers(100)); blocks = b.block_clusters(X); nbrs = b.neighbors()
kw = {"shard": (0, int(os.environ["WORLD"]))} if os.environ.get("WORLD") else {}
g = GPRF(X, Y, None, GPCov([1.0], [0.06, 0.06], "euclidean", "se"), 0.01, block_idxs=blocks, neighbors=nbrs, **kw)
g._push_neighbors(nbrs)
ctx = g._ctx
for _ in range(3): ctx.debug_run(X, 3)        # ... up to the solve
nt, nl = ctx.num_units()
rows = np.array([ctx.debug_fetch(l, 6) for l in range(nl)])
for T in (7, 13, 15, 16):
    sel = rows[rows[:, 5] == T]
    if len(sel):
        m = sel[:, :5].mean(axis=0) / T
        print("T=%d units=%d cycles/step:" % (T, len(sel)), " ".join("%s %.0f" % (a, v) for a, v in zip(["stage", "barrier", "fetch", "solve", "update"], m)), " total/step %.0f  total %.0f" % (m.sum(), m.sum() * T),
              " | before the loop %.0f  after it %.0f" % (sel[:, 6].mean(), sel[:, 7].mean()))
