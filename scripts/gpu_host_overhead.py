"""Where the sequential evaluation's wall time goes on the host side: python scripts/gpu_host_overhead.py
(north-star configuration; compare GPRF_SYNC=block)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gprf_amd.synthetic import SampledData
from gprf_amd import grid_centers

sd = SampledData(n=10500, ntrain=10000, lscale=0.06, obs_std=0.02, yd=50, seed=0, use_gpu=True)
sd.set_centers(grid_centers(100))
g = sd.build_gprf(local_dist=0.5)
rng = np.random.RandomState(0)
Xs = [np.ascontiguousarray(sd.X_obs + 2e-4 * k * rng.randn(*sd.X_obs.shape)) for k in range(10)]   # a few points change block
for k in range(20):
    g.update_X(Xs[k % 10]); g.llgrad(grad_X=True)
N = 300
t0 = time.perf_counter()
for k in range(N):
    g.update_X(Xs[k % 10]); g.llgrad(grad_X=True)
t_py = (time.perf_counter() - t0) / N
ctx = g._ctx
t0 = time.perf_counter()
for k in range(N):
    ctx.update_eval(Xs[k % 10], True, False)
t_capi = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for k in range(N):
    ctx.eval(Xs[k % 10], True, False)
t_eval = (time.perf_counter() - t0) / N
ctx.set_timing(True, reset=True)
for k in range(50):
    ctx.update_eval(Xs[k % 10], True, False)
tm = ctx.get_timing(); tm.pop("count")
print("sync=%s  GPRF.update_X+llgrad %.1f us | ctx.update_eval %.1f us | ctx.eval (no re-partition) %.1f us | kernels (events) %.1f us  %s"
      % (os.environ.get("GPRF_SYNC", "spin"), t_py * 1e6, t_capi * 1e6, t_eval * 1e6, sum(tm.values()) * 1e3,
         {k: round(v * 1e3, 1) for k, v in tm.items()}))
g.close()
