"""k_mgrad phase cycles (GPRF_LIB = a -DGPRF_PROFILE build): python scripts/gpu_mgrad_stamps.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gprf_amd import GPCov, Blocker, grid_centers
from gprf_amd.gprf import GPRF
rng = np.random.RandomState(1)
n = 10000; X = rng.rand(n, 2); Y = rng.randn(n, 50)
b = Blocker(grid_centers(100)); blocks = b.block_clusters(X); nbrs = b.neighbors()
g = GPRF(X, Y, None, GPCov([1.0], [0.06, 0.06], "euclidean", "se"), 0.01, block_idxs=blocks, neighbors=nbrs,
         shard=(0, int(os.environ["WORLD"])) if os.environ.get("WORLD") else None)
g._push_neighbors(nbrs)
ctx = g._ctx
for _ in range(3): ctx.debug_run(X, 6)
nt, nl = ctx.num_units()
rows = np.array([ctx.debug_fetch(l, 6) for l in range(nl)])
if os.environ.get("FINE"):      # a -DGPRF_PROFILE -DGPRF_MGRAD_FINE build
    for nch in sorted(set(rows[:, 6].astype(int))):
        sel = rows[rows[:, 6] == nch]
        if nch > 0 and len(sel) > 3:
            m = sel.mean(axis=0)
            if os.environ.get("LOOP"):      # ... -DGPRF_MGRAD_LOOP
                print("bottom-left pair nch=%d units=%d: loop %.0f = per chunk: write+barrier %.0f  fetch issue %.0f  MFMAs %.0f"
                      % (nch, len(sel), m[5], m[0] / nch, m[1] / nch, m[2] / nch))
                continue
            print("bottom-left pair nch=%d units=%d: loop %.0f | barrier %.0f  diag tile %.0f  lower tiles %.0f  row+theta sums %.0f  "
                  "last barrier+stores %.0f" % (nch, len(sel), m[5], m[0], m[1], m[2], m[3], m[4]))
    g.close()
    sys.exit(0)
for name, o in (("first (0,0)", 0), ("last (TB-1,0)", 4)):
    for nch in sorted(set(rows[:, o + 3].astype(int))):
        sel = rows[rows[:, o + 3] == nch]
        if nch > 0 and len(sel) > 3:
            m = sel[:, o:o + 3].mean(axis=0)
            print("%s nch=%d units=%d: prologue %.0f  loop %.0f (%.0f/chunk)  reductions %.0f  total %.0f cycles"
                  % (name, nch, len(sel), m[0], m[1], m[1] / nch, m[2], m.sum()))
g.close()
