"""Stability soak of the two-queue Cholesky (fork released by the large-unit kernel's first workgroup, join by stream
memory operation): (1) 30000 sequential evaluations of one context, (2) 12 contexts evaluated round-robin with their
evaluations enqueued back to back on one torch stream (HIP streams then share hardware queues), (3) two contexts
alternating through the host-in / host-out path.  Prints rates; a hang shows up as the caller's timeout.
    timeout 300 python tests/diag/gpu_soak.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gprf_amd import GPCov, Blocker, grid_centers
from gprf_amd.gprf import GPRF
from gprf_amd import dist as gdist

rng = np.random.RandomState(4)
n = 10000
X = rng.rand(n, 2); Y = rng.randn(n, 50)
b = Blocker(grid_centers(100)); nbrs = b.neighbors()
cov = GPCov([1.0], [0.06, 0.06], "euclidean", "se")
Xs = [np.ascontiguousarray(X + 2e-4 * k * rng.randn(n, 2)) for k in range(10)]

g = GPRF(X, Y, b.block_clusters, cov, 0.01, neighbors=nbrs)
ref = g.llgrad(grad_X=True)
t0 = time.time()
N1 = int(os.environ.get("SOAK_N", "30000"))
for k in range(N1):
    g.update_X(Xs[k % 10]); r = g.llgrad(grad_X=True)
print("(1) %d sequential evaluations: %.0f evals/s, last ll %.6e" % (N1, N1 / (time.time() - t0), r[0]))
g.update_X(X); again = g.llgrad(grad_X=True)
assert again[0] == ref[0] and np.array_equal(again[1], ref[1])

ctxs = [GPRF(Xs[k % 10], Y, None, cov, 0.01, block_idxs=b.block_clusters(Xs[k % 10]), neighbors=nbrs) for k in range(12)]
evs = []
st = torch.cuda.Stream()
for c in ctxs:
    c._push_blocks(); c._push_neighbors(nbrs)
    e = gdist.DeviceEvaluator(c); e.set_X(c.X); evs.append(e)
t0 = time.time()
N2 = 300
for r_ in range(N2):
    for e in evs:
        e.enqueue(True, False, stream=st)
torch.cuda.synchronize()
print("(2) 12 contexts x %d rounds enqueued back to back: %.0f evals/s" % (N2, 12 * N2 / (time.time() - t0)))
first = evs[0].result(True, False)
solo = ctxs[0].llgrad(grad_X=True)
assert np.isclose(first[0], solo[0], rtol=1e-13)
for c in ctxs[2:]:
    c.close()
a, c2 = ctxs[0], ctxs[1]
t0 = time.time()
for k in range(3000):
    a.update_X(Xs[k % 10]); a.llgrad(grad_X=True)
    c2.update_X(Xs[(k + 3) % 10]); c2.llgrad(grad_X=True)
print("(3) two contexts alternating, 6000 evaluations: %.0f evals/s" % (6000 / (time.time() - t0)))
a.close(); c2.close(); g.close()
print("soak ok")
