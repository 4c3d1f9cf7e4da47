"""Stress walk: N sequential evaluations of the north-star shape, every one re-partitioning on the device; prints a digest of
everything returned.  Run it twice, and once with the diagnostic switches off (GPRF_DIAG=fused_build=0,gx_fold=0,part_major=0,one_queue=1): the three digests must be equal (bit-identical variants, no race)."""
import hashlib, sys, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gprf_amd import Blocker, grid_centers, GPCov
from gprf_amd.gprf import GPRF
rng = np.random.RandomState(5)
n = 10000
X = rng.rand(n, 2); Y = rng.randn(n, 50)
b = Blocker(grid_centers(100))
g = GPRF(X, Y, b.block_clusters, GPCov([1.0], [0.06, 0.06], "euclidean", "se"), 0.01, neighbors=b.neighbors())
h = hashlib.sha256()
for it in range(int(sys.argv[1])):
    X = np.clip(X + 0.004 * rng.randn(n, 2), 0.0, 1.0)
    g.update_X(X)
    ll, gX, gC = g.llgrad(grad_X=True, grad_cov=(it % 3 == 0))
    h.update(np.float64(ll).tobytes()); h.update(np.ascontiguousarray(gX).tobytes()); h.update(np.ascontiguousarray(gC).tobytes())
    if not np.isfinite(ll): print("non-finite at", it); break
print("DIGEST", h.hexdigest(), ll)
