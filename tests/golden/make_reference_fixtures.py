"""Golden partitions produced BY THE REFERENCE'S OWN CODE (build container only).

Two of the reference's files run under Python 3 as they are: ``/root/reference/pdtree_clustering.py`` (the seismic
principal-direction tree, lines 4-94) and ``/root/reference/block_clustering.py`` (``Blocker``, lines 7-45).  This script
imports them BY PATH — nothing of them is copied into the repository and nothing of them travels to the GPU box — runs them
on seeded inputs and writes their outputs to ``tests/golden/ref_partitions.npz``:

* ``pdtree_cluster`` at block sizes 120 and 210 on the stand-in catalogue (n = 3000, with a cluster straddling the date
  line): the leaf index lists, and ``reblock`` for (a) the build points, (b) slightly and (c) strongly moved points, (d) a
  leaf emptied by moving all its events away, (e) events pushed across the -22 degree longitude cut;
* ``Blocker.block_clusters`` at 4 / 100 / 841 grid centres on uniform points incl. points exactly on centres and points
  equidistant from two and from four centres;
* ``Blocker.neighbors()`` exactly as the reference computes it under THIS numpy (the fragile ``cc[cc > 0]`` filter,
  SURVEY section 8a-11: 180 edges for 100 centres, 1624 for 841), next to the 8-neighbourhood the published objectives
  require (342 / 3192), which the product builds.

The inputs are stored with the outputs (the fixture is self-contained data).  Run:  python tests/golden/make_reference_fixtures.py
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference"


def _by_path(name):
    spec = importlib.util.spec_from_file_location("_ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _pack(lists):
    """list of index arrays -> (ptr, concatenated)"""
    ptr = np.zeros(len(lists) + 1, dtype=np.int64)
    ptr[1:] = np.cumsum([len(a) for a in lists])
    cat = np.concatenate([np.asarray(a, dtype=np.int64) for a in lists]) if len(lists) else np.zeros(0, dtype=np.int64)
    return ptr, cat


def catalogue():
    from gprf_amd.seismic import synthetic_events
    n = 3000
    X = synthetic_events(n, seed=5)
    X[:40, 0] = np.where(np.arange(40) % 2 == 0, 179.9, -179.9) + np.linspace(-0.05, 0.05, 40)   # date-line cluster
    return X


def main():
    pd = _by_path("pdtree_clustering")
    bc = _by_path("block_clustering")
    out = {}

    # ---- pdtree_clustering.py:79-94 ----
    X = catalogue()
    out["pd_X"] = X
    rng = np.random.RandomState(1)
    for bs in (120, 210):
        leaves, reblock = pd.pdtree_cluster(X.copy(), blocksize=bs)
        out["pd%d_leaf_ptr" % bs], out["pd%d_leaf_idx" % bs] = _pack(leaves)
        moves = {"same": X.copy(),
                 "near": X + rng.randn(*X.shape) * [0.3, 0.3, 3.0],
                 "far": X + rng.randn(*X.shape) * [5.0, 5.0, 20.0]}
        emptied = X.copy()
        emptied[np.asarray(leaves[3])] += [40.0, 10.0, 0.0]               # leaf 3 loses all its events
        moves["emptied"] = emptied
        cut = X.copy()
        west = np.argsort(np.abs(((X[:, 0] + 22) % 360 - 22) - (-22.0)))[:30]      # the events nearest the cut from the east
        cut[west, 0] -= 3.0                                                # ... pushed across it (they wrap to +335)
        moves["cut"] = cut
        for name, XX in moves.items():
            keep = XX.copy()
            res = reblock(XX)                                              # (wraps the column in place and restores it)
            assert np.array_equal(XX, keep)
            out["pd%d_%s_X" % (bs, name)] = keep
            out["pd%d_%s_ptr" % (bs, name)], out["pd%d_%s_idx" % (bs, name)] = _pack(res)
        assert any(len(a) == 0 for a in reblock(emptied.copy()))

    # ---- block_clustering.py:7-45 ----
    from gprf_amd import grid_centers
    rng = np.random.RandomState(21)
    for nb in (4, 100, 841):
        C = np.asarray(grid_centers(nb), dtype=np.float64)
        g = int(round(np.sqrt(len(C))))
        n = 4000
        P = rng.rand(n, 2) * 1.1 - 0.05                                    # a few points outside the unit square
        P[:len(C)][:200] = C[:200]                                         # on the centres
        P[200] = 0.5 * (C[0] + C[1])                                       # equidistant from two centres
        P[201] = 0.25 * (C[0] + C[1] + C[g] + C[g + 1])                    # ... from four
        P[202] = 0.5 * (C[0] + C[g])
        b = bc.Blocker(C)
        out["bc%d_X" % nb] = P
        out["bc%d_ptr" % nb], out["bc%d_idx" % nb] = _pack(b.block_clusters(P))
        # the reference's own neighbour rule under this numpy, and the one without diagonal connections
        out["bc%d_ref_neighbors" % nb] = np.asarray(b.neighbors(), dtype=np.int64).reshape(-1, 2)
        out["bc%d_ref_neighbors_axis" % nb] = np.asarray(b.neighbors(diag_connections=False), dtype=np.int64).reshape(-1, 2)
        out["bc%d_centers" % nb] = C
    out["numpy_version"] = np.array(np.__version__)
    np.savez_compressed(os.path.join(HERE, "ref_partitions.npz"), **out)
    for k in sorted(out):
        v = out[k]
        print(k, v.shape, v.dtype)
    for nb in (4, 100, 841):
        print(nb, "centres: reference neighbors()", len(out["bc%d_ref_neighbors" % nb]), "edges; axis-only",
              len(out["bc%d_ref_neighbors_axis" % nb]))


if __name__ == "__main__":
    main()
