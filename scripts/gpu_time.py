"""Stage timing on a C3-shaped problem (random Y): python scripts/gpu_time.py [reps]
WORLD=N times rank 0's shard of an N-rank job (partial sums, no all-reduce)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gprf_amd import GPCov, Blocker, grid_centers
from gprf_amd.gprf import GPRF

def timing(n=10000, nb=100, dy=50, lscale=0.06, pairs=True, reps=20, tag=""):
    rng = np.random.RandomState(1)
    X = rng.rand(n, 2); Y = rng.randn(n, dy)
    b = Blocker(grid_centers(nb)); blocks = b.block_clusters(X); nbrs = b.neighbors() if pairs else []
    g = GPRF(X, Y, None, GPCov([1.0], [lscale, lscale], "euclidean", "se"), 0.01, block_idxs=blocks, neighbors=nbrs,
             shard=(0, int(os.environ["WORLD"])) if os.environ.get("WORLD") else None, reduce=False)
    if os.environ.get("RAW"):
        # ablation runs produce garbage (possibly NOT_PD): time the raw C-ABI call and ignore the status
        g._push_neighbors(nbrs)
        ctx = g._ctx
        gc = bool(os.environ.get("GC"))      # GC=1: hyper-parameter gradient too (task xcov)
        ctx.eval(X, True, gc)
        ctx.set_timing(True, reset=True)
        for _ in range(reps):
            ctx.eval(X, True, gc)
        st = ctx.get_timing()
        print("%s pairs=%d RAW: stages(us) %s" % (tag, len(nbrs), {k: round(v * 1e3, 1) for k, v in st.items() if k != "count"}))
        g.close(); return
    g.llgrad(grad_X=True)
    ts0 = []
    for _ in range(reps):
        t = time.time(); g.llgrad(grad_X=True); ts0.append(time.time() - t)
    print("%s untimed sync eval: median %.3f ms  min %.3f ms" % (tag, np.median(ts0) * 1e3, np.min(ts0) * 1e3))
    if os.environ.get("NOSTAGES"):
        g.close(); return
    g._ctx.set_timing(True, reset=True)
    ts = []
    for _ in range(reps):
        t = time.time(); g.llgrad(grad_X=True); ts.append(time.time() - t)
    st = g._ctx.get_timing()
    print("%s pairs=%d: median %.3f ms; stages(us) %s" % (tag, len(nbrs), np.median(ts) * 1e3, {k: round(v * 1e3, 1) for k, v in st.items() if k != "count"}))
    g.close()

if __name__ == "__main__":
    if os.environ.get("C4"):      # BASELINE configs[3]'s shape
        timing(n=80000, nb=800, lscale=0.02, pairs=True, tag=os.environ.get("TAG", "C4"))
        sys.exit(0)
    timing(pairs=True, tag=os.environ.get("TAG", ""))
    if os.environ.get("LOCAL"): timing(pairs=False, tag="local")
