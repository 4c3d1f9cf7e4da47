"""k_mgrad phase cycles on the seismic configuration's shape (GPRF_LIB = a -DGPRF_PROFILE build):
    GPRF_LIB=build_variants/libgprf_profile.so python scripts/gpu_mgrad_stamps_c5.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gprf_amd import GPCov, seismic
from gprf_amd.gprf import GPRF
n = 20000
X = seismic.synthetic_events(n, seed=0)
Y = np.random.RandomState(1).randn(n, 50)
blocks, reblock = seismic.pdtree_cluster(X, 210)
g = GPRF(X, Y, reblock, GPCov([1.0], [40.0, 40.0], "lld", "matern32"), 0.1, neighbor_threshold=0.6)
g._push_neighbors(g.neighbors)
ctx = g._ctx
for _ in range(3): ctx.debug_run(X, 6)
nt, nl = ctx.num_units()
rows = np.array([ctx.debug_fetch(l, 6) for l in range(nl)])
for name, o in (("first (0,0)", 0), ("last (TB-1,0)", 4)):
    for nch in sorted(set(rows[:, o + 3].astype(int))):
        sel = rows[rows[:, o + 3] == nch]
        if nch > 0 and len(sel) > 3:
            m = sel[:, o:o + 3].mean(axis=0)
            print("%s nch=%d units=%d: prologue %.0f  loop %.0f (%.0f/chunk)  reductions %.0f  total %.0f cycles"
                  % (name, nch, len(sel), m[0], m[1], m[1] / nch, m[2], m.sum()))
g.close()
