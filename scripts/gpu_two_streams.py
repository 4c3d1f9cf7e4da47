"""How much does the chip have left when one evaluation's kernels run?  Two independent contexts (same workload) enqueued
on two streams at once vs one after the other: aggregate device-resident evaluations per second."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gprf_amd.synthetic import SampledData
from gprf_amd import grid_centers
from gprf_amd import dist as gdist

sd = SampledData(n=10500, ntrain=10000, lscale=0.06, obs_std=0.02, yd=50, seed=0, use_gpu=True)
sd.set_centers(grid_centers(100))
evs = []
for k in range(2):
    g = sd.build_gprf(local_dist=0.5)
    g._push_neighbors(g.neighbors)
    ev = gdist.DeviceEvaluator(g)
    ev.set_X(sd.X_obs)
    evs.append(ev)
dev = torch.device("cuda", 0)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for ev in evs:
    ev.enqueue(True, False, stream=s1)
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for k in range(N):
    evs[k % 2].enqueue(True, False, stream=s1)
torch.cuda.synchronize()
one = N / (time.perf_counter() - t0)
t0 = time.perf_counter()
for k in range(N // 2):
    evs[0].enqueue(True, False, stream=s1)
    evs[1].enqueue(True, False, stream=s2)
torch.cuda.synchronize()
two = N / (time.perf_counter() - t0)
print("one stream: %.0f evals/s   two streams, two contexts: %.0f evals/s  (x%.2f)" % (one, two, two / one))
for ev in evs:
    ev.g.close()
