import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STAMPS = os.environ.get("STAMPS", "1")      # which wave's stamps: 1 | 2 | 3
os.environ["GPRF_DIAG"] = "potrf_stamps=" + STAMPS
mode2 = STAMPS == "2"
mode3 = STAMPS == "3"
from gprf_amd import GPCov, Blocker, grid_centers
from gprf_amd.gprf import GPRF
rng = np.random.RandomState(1)
n = 10000; X = rng.rand(n, 2); Y = rng.randn(n, 50)
b = Blocker(grid_centers(100)); blocks = b.block_clusters(X); nbrs = b.neighbors()
g = GPRF(X, Y, None, GPCov([1.0], [0.06, 0.06], "euclidean", "se"), 0.01, block_idxs=blocks, neighbors=nbrs)
g._push_neighbors(nbrs)
ctx = g._ctx
for _ in range(3): ctx.debug_run(X, 1)
nt, nl = ctx.num_units()
rows = np.array([ctx.debug_fetch(l, 6) for l in range(nl)])
v3 = False
tc = 7 if mode2 else (5 if v3 else 4)
for T in (7, 10, 12, 13, 14, 15, 16):
    sel = rows[rows[:, tc] == T]
    if len(sel):
        m = sel[:, :tc].mean(axis=0)
        names = ["B1wait", "panel", "B2+B3wait", "factor", "B4wait"] if v3 else ["panel", "barrier1", "factor", "barrier2"]
        if mode2: names = ["dump", "loads", "subst", "stores", "barrierB", "trailing", "barrierA"]
        if mode3: names = ["copy", "diag", "chain", "rest"]
        print("T=%d units=%d cycles/step:" % (T, len(sel)), " ".join("%s %.0f" % (a, b) for a, b in zip(names, m / (T - 1))),
              " total/step %.0f" % (m.sum() / (T - 1)),
              "" if (mode2 or mode3) else " | prologue %.0f epilogue %.0f loop %.0f" % (sel[:, 5].mean(), sel[:, 6].mean(), m.sum()))
