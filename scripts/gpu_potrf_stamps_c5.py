"""Cholesky step-loop stamps on the seismic shape (GPRF_LIB = a -DGPRF_PROFILE build; STAMPS = 1 | 2 | 3):
    GPRF_LIB=build_variants/libgprf_profile.so STAMPS=3 python scripts/gpu_potrf_stamps_c5.py [n]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STAMPS = os.environ.get("STAMPS", "1")      # which wave's stamps: 1 | 2 | 3
os.environ["GPRF_DIAG"] = "potrf_stamps=" + STAMPS
mode = STAMPS
from gprf_amd import GPCov, seismic
from gprf_amd.gprf import GPRF
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
X = seismic.synthetic_events(n, seed=0)
Y = np.random.RandomState(1).randn(n, 50)
blocks, reblock = seismic.pdtree_cluster(X, 210)
g = GPRF(X, Y, reblock, GPCov([1.0], [40.0, 40.0], "lld", "matern32"), 0.1, neighbor_threshold=0.6)
g._push_neighbors(g.neighbors)
ctx = g._ctx
for _ in range(3): ctx.debug_run(X, 1)
nt, nl = ctx.num_units()
rows = np.array([ctx.debug_fetch(l, 6) for l in range(nl)])
tc = 7 if mode == "2" else 4
names = {"1": ["panel", "barrier1", "factor", "barrier2"], "2": ["s0", "s1", "solve", "s3", "barrierB", "trailing", "barrierA"],
         "3": ["trail-diag", "trail-wait", "chain", "rest"]}[mode]
for T in sorted(set(rows[:, tc].astype(int))):
    sel = rows[rows[:, tc] == T]
    if T > 1 and len(sel) > 3:
        m = sel[:, :tc].mean(axis=0)
        print("T=%d units=%d cycles/step:" % (T, len(sel)), " ".join("%s %.0f" % (a, b) for a, b in zip(names, m / (T - 1))),
              " total/step %.0f" % (m.sum() / (T - 1)),
              (" | prologue %.0f epilogue %.0f loop %.0f" % (sel[:, 5].mean(), sel[:, 6].mean(), m.sum())) if mode == "1" else "")
g.close()
