"""Host-side cost of the sharded (multi-GPU) evaluation path on ONE GPU: a one-rank process group with the all-reduce forced
(GPRF_FORCE_ALLREDUCE=1), DeviceEvaluator.evaluate against the plain host-in / host-out call.
    python scripts/gpu_sharded_overhead.py"""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ["GPRF_FORCE_ALLREDUCE"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from gprf_amd import GPCov, Blocker, grid_centers
from gprf_amd.gprf import GPRF
rng = np.random.RandomState(4)
n = 10000
X = rng.rand(n, 2); Y = rng.randn(n, 50)
b = Blocker(grid_centers(100)); nbrs = b.neighbors()
cov = GPCov([1.0], [0.06, 0.06], "euclidean", "se")
Xs = [np.ascontiguousarray(X + 2e-4 * k * rng.randn(n, 2)) for k in range(10)]
plain = GPRF(X, Y, b.block_clusters, cov, 0.01, neighbors=nbrs)
def rate(fn, N=500):
    for k in range(30): fn(Xs[k % 10])
    t0 = time.perf_counter()
    for k in range(N): fn(Xs[k % 10])
    return (time.perf_counter() - t0) / N * 1e6
print("plain, before the sharded context exists: %.1f us" % rate(lambda Xk: (plain.update_X(Xk), plain.llgrad(grad_X=True))))
shard = GPRF(X, Y, b.block_clusters, cov, 0.01, neighbors=nbrs, shard=(0, 1))
from gprf_amd import dist as gdist
shard._dist_eval = gdist.DeviceEvaluator(shard)
shard._push_blocks(); shard._push_neighbors(shard.neighbors)
for g, name in ((plain, "plain host-in/host-out"), (shard, "sharded path (1 rank, RCCL all-reduce forced)")):
    if g is shard:
        ev = lambda Xk: (g.update_X(Xk), g._llgrad_sharded(np.ascontiguousarray(g.X), True, False))
    else:
        ev = lambda Xk: (g.update_X(Xk), g.llgrad(grad_X=True))
    for k in range(30): ev(Xs[k % 10])
    t0 = time.perf_counter()
    N = 500
    for k in range(N): r = ev(Xs[k % 10])
    print("%-48s %.1f us per evaluation" % (name, (time.perf_counter() - t0) / N * 1e6))
a = plain.llgrad(grad_X=True); plain.update_X(Xs[0]); shard.update_X(Xs[0])
a = plain.llgrad(grad_X=True); c = shard._llgrad_sharded(np.ascontiguousarray(shard.X), True, False)
print("same result:", a[0] == c[0], np.array_equal(a[1], c[1]))
dist.destroy_process_group()
