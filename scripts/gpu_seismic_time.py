"""Stage timing of the seismic configuration (BASELINE config 5's shape: great-circle/depth distance, Matern-3/2,
tree blocks of < 210 events, edge threshold 0.6, task xcov) on the stand-in catalogue:
    python scripts/gpu_seismic_time.py [n] [reps]        (comparison with the oracle: tests/diag/gpu_seismic_diag.py)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gprf_amd import GPCov, seismic
from gprf_amd.gprf import GPRF


def main(n=20000, reps=20, yd=50, blocksize=210, threshold=0.6, lscale=40.0):
    X = seismic.synthetic_events(n, seed=0)
    Y = np.random.RandomState(1).randn(n, yd)
    cov = GPCov([1.0], [lscale, lscale], "lld", "matern32")
    t = time.time()
    blocks, reblock = seismic.pdtree_cluster(X, blocksize)
    t_tree = time.time() - t
    t = time.time()
    g = GPRF(X, Y, reblock, cov, 0.1, neighbor_threshold=threshold)
    t_setup = time.time() - t
    sz = [len(b) for b in g.block_idxs]
    pm = max([sz[i] + sz[j] for i, j in g.neighbors] + [0])
    print("n=%d blocks=%d (%d..%d) pairs=%d largest unit=%d  tree %.2fs  GPRF() incl. neighbours %.2fs"
          % (n, len(sz), min(sz), max(sz), len(g.neighbors), pm, t_tree, t_setup))
    g.llgrad(grad_X=True, grad_cov=True)
    ts = []
    for _ in range(reps):
        t = time.time(); g.llgrad(grad_X=True, grad_cov=True); ts.append(time.time() - t)
    print("sync eval (x + cov gradients): median %.3f ms  min %.3f ms" % (np.median(ts) * 1e3, np.min(ts) * 1e3))
    g._ctx.set_timing(True, reset=True)
    for _ in range(reps):
        g.llgrad(grad_X=True, grad_cov=True)
    st = g._ctx.get_timing()
    print("stages(us)", {k: round(v * 1e3, 1) for k, v in st.items() if k != "count"})
    g._ctx.set_timing(False, reset=True)
    # the optimiser's view: callback with re-routing through the tree
    obj = seismic.SeismicObjective(g, X, np.array([[0.1, 1.0, lscale, lscale]]), x_prior=seismic.make_x_prior(X, 2.0))
    x = obj.full0.copy()
    rng = np.random.RandomState(2)
    ts = []
    for _ in range(reps):
        x[:obj.nx] += rng.randn(obj.nx) * 1e-4
        t = time.time(); obj(x); ts.append(time.time() - t)
    print("callback (update_X re-routes all events + update_covs + llgrad + priors): median %.3f ms" % (np.median(ts) * 1e3))
    g.close()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 20000, int(sys.argv[2]) if len(sys.argv) > 2 else 20)
