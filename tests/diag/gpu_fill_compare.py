"""K through the pool, k_fill_se against the entry-by-entry k_fill<0,0> (GPRF_FILL_VARIANT=0): where do they differ?
Each variant in its own process (the switch is read once); the parent compares the dumps."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from gprf_amd import Blocker, grid_centers, GPCov
from gprf_amd.gprf import GPRF
rng = np.random.RandomState(31)
n = 1800
X = rng.rand(n, 2); Y = rng.randn(n, 7)
b = Blocker(grid_centers(16))
g = GPRF(X, Y, b.block_clusters, GPCov([1.0], [0.09, 0.11], "euclidean", "se"), 0.02, neighbors=b.neighbors())
g._push_neighbors(g.neighbors)
ctx = g._ctx
ctx.debug_run(X, 0)
nt, nl = ctx.num_units()
out = {}
for l in range(nl):
    m, mp, gu = ctx.debug_unit_shape(l)
    out["K%%d" %% l] = ctx.debug_fetch(l, 0)
    out["m%%d" %% l] = m
np.savez(sys.argv[1], **out)
''' % ROOT
paths = []
for v in ("0", "4"):
    p = "/tmp/fillcmp_%s.npz" % v
    env = dict(os.environ, GPRF_FUSED_FILL="0", GPRF_FILL_VARIANT=v)
    subprocess.run([sys.executable, "-c", CHILD, p], env=env, check=True)
    paths.append(p)
a, b = np.load(paths[0]), np.load(paths[1])
nunits = len([k for k in a.files if k.startswith("K")])
tot = 0
shown = 0
for l in range(nunits):
    Ka, Kb, m = a["K%d" % l], b["K%d" % l], int(a["m%d" % l])
    mp = Ka.shape[0]
    mask = np.zeros_like(Ka, dtype=bool)
    nt = (mp + 63) // 64
    for ti in range(nt):
        for tj in range(ti, nt):
            mask[64 * ti:64 * ti + 64, 64 * tj:64 * tj + 64] = True
    d = (Ka != Kb) & mask
    if d.any():
        idx = np.argwhere(d)
        tot += len(idx)
        if shown < 6:
            shown += 1
            ulps = np.abs(Ka[d].view(np.int64) - Kb[d].view(np.int64))
            i, j = idx[0]
            print("unit %d m=%d mp=%d: %d of %d entries differ; ulp distance max %d mean %.2f; first (%d,%d): %r vs %r; block rows %s block cols %s; on diagonal %d; rows>=m %d cols>=m %d" % (
                l, m, mp, len(idx), int(mask.sum()), ulps.max(), ulps.mean(), i, j, Ka[i, j], Kb[i, j], sorted(set((idx[:, 0] // 64).tolist())),
                sorted(set((idx[:, 1] // 64).tolist())), int((idx[:, 0] == idx[:, 1]).sum()), int((idx[:, 0] >= m).sum()), int((idx[:, 1] >= m).sum())))
print("units %d, differing entries %d" % (nunits, tot))
