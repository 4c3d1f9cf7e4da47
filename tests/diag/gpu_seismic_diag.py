"""Where does the seismic configuration leave the oracle?  python tests/diag/gpu_seismic_diag.py"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gprf_amd import GPCov, seismic
from gprf_amd.gprf import GPRF
from oracle.gprf_ref import GPRFRef
from oracle.vector_tree import GPCov as OC


def case(n, blocksize, dfn, wfn, ls, thr, yd=5, seed=0, local=False, scale=1.0):
    X = seismic.synthetic_events(n, seed=seed)
    if dfn == "euclidean":
        X = X.copy()
    Y = np.random.RandomState(1).randn(n, yd)
    blocks, reblock = seismic.pdtree_cluster(X, blocksize)
    cov = GPCov([1.0], ls, dfn, wfn)
    g = GPRF(X, Y, None, cov, 0.1, block_idxs=blocks, neighbor_threshold=thr, neighbors=[] if local else None)
    r = GPRFRef(X, Y, None, OC([1.0], ls, dfn, wfn), 0.1, block_idxs=g.block_idxs, neighbors=g.neighbors)
    a = r.llgrad(grad_X=True, grad_cov=True)
    b = g.llgrad(grad_X=True, grad_cov=True)
    sz = [len(x) for x in blocks]
    pm = max([sz[i] + sz[j] for i, j in g.neighbors] + [max(sz)])
    bad = np.argsort(-np.max(np.abs(a[1] - b[1]), axis=1))[:3]
    print("n=%d bs=%d %s/%s ls=%s pairs=%d largest=%d: ll rel %.1e  gX %.1e  gC %.1e   worst rows %s"
          % (n, blocksize, dfn, wfn, ls, len(g.neighbors), pm, abs(a[0] - b[0]) / abs(a[0]),
             np.max(np.abs(a[1] - b[1])) / np.max(np.abs(a[1])), np.max(np.abs(a[2] - b[2]) / np.abs(a[2])),
             [(int(i), X[i].round(3).tolist(), a[1][i].round(4).tolist(), b[1][i].round(4).tolist()) for i in bad[:2]]))
    g.close()


case(2000, 210, "lld", "matern32", [40.0, 40.0], 0.6)
case(2000, 210, "lld", "matern32", [40.0, 40.0], 0.6, local=True)
case(4000, 210, "lld", "matern32", [40.0, 40.0], 0.6)
case(4000, 400, "lld", "matern32", [40.0, 40.0], 0.6)
case(4000, 400, "lld", "matern32", [40.0, 40.0], 0.6, local=True)
case(8000, 210, "lld", "matern32", [40.0, 40.0], 0.6)
if os.environ.get("BIG"):      # BASELINE config 5's shape, the oracle takes ~40 s
    case(20000, 210, "lld", "matern32", [40.0, 40.0], 0.6, yd=50)
