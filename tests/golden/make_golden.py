"""Generate the committed golden vectors from the oracle (after tests/test_oracle_kat.py shows the oracle
reproduces the reference's published traces).  Inputs + expected outputs only.
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.gprf_ref import GPRFRef  # noqa: E402
from oracle.harness_ref import BlockerRef, SampledDataRef, grid_centers  # noqa: E402
from oracle.vector_tree import GPCov  # noqa: E402


def pack_blocks(blocks):
    ptr = np.zeros(len(blocks) + 1, dtype=np.int64)
    ptr[1:] = np.cumsum([len(b) for b in blocks])
    pts = np.concatenate([np.asarray(b, dtype=np.int32) for b in blocks]) if ptr[-1] else np.zeros(0, np.int32)
    return ptr, pts.astype(np.int32)


def c1():
    """BASELINE config 1: ntrain=500, ntest=500, 4 blocks, yd=10, lscale=0.4, obs_std=0.04."""
    sd = SampledDataRef(n=1000, ntrain=500, lscale=0.4, obs_std=0.04, yd=10, seed=0)
    sd.set_centers(grid_centers(4))
    out = dict(X_obs=sd.X_obs, SX=sd.SX, SY=sd.SY, theta=np.array([0.01, 1.0, 0.4, 0.4]),
               neighbors=np.array(sd.neighbors, dtype=np.int32), obs_std=0.04)
    out["block_ptr"], out["block_pts"] = pack_blocks(sd.block_idxs)
    for tag, ld in (("local", 1.0), ("gprf", 0.1)):
        g = sd.build_gprf(local_dist=ld, mode="rows")
        ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
        out["ll_" + tag], out["gX_" + tag], out["gC_" + tag] = ll, gX, gC
    g = sd.build_gprf(local_dist=0.1)
    out["ll_allpairs"], out["gX_allpairs"], _ = g.llgrad(local=False, grad_X=True)
    np.savez_compressed(os.path.join(HERE, "c1_small.npz"), **out)


def tiny_parts():
    """60 points, 2 blocks + 1 pair, per-stage matrices of the pair unit (K, L, K^-1, Alpha)."""
    rng = np.random.RandomState(7)
    X = rng.rand(60, 2)
    Y = rng.randn(60, 7)
    cov = GPCov([1.3], [0.35, 0.5], "euclidean", "se")
    blocks = [np.arange(0, 27), np.arange(27, 60)]
    g = GPRFRef(X, Y, None, cov, 0.02, block_idxs=blocks, neighbors=[(1, 0)], mode="rows")
    ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
    idx = np.concatenate([blocks[1], blocks[0]])
    ull, ugX, ugC, parts = g.gaussian_llgrad(X[idx], Y[idx], grad_X=True, grad_cov=True, return_parts=True)
    ptr, pts = pack_blocks(blocks)
    np.savez_compressed(os.path.join(HERE, "tiny_parts.npz"), X=X, Y=Y, theta=np.array([0.02, 1.3, 0.35, 0.5]),
                        block_ptr=ptr, block_pts=pts, neighbors=np.array([[1, 0]], dtype=np.int32), ll=ll, gX=gX, gC=gC,
                        pair_K=parts["K"], pair_L=parts["L"], pair_prec=parts["prec"], pair_Alpha=parts["Alpha"],
                        pair_logdet=parts["logdet"], pair_ll=ull, pair_gX=ugX, pair_gC=ugC)


def lld_toy():
    """64 clustered (lon, lat, depth) points, ("lld","matern32") — PARITY UNPINNED kernel (oracle header)."""
    rng = np.random.RandomState(11)
    lon = 130.0 + rng.randn(64) * 0.3
    lat = -2.0 + rng.randn(64) * 0.3
    dep = np.abs(rng.randn(64)) * 30.0
    X = np.stack([lon, lat, dep], axis=1)
    Y = rng.randn(64, 5)
    cov = GPCov([1.0], [40.0, 20.0], "lld", "matern32")
    blocks = [np.arange(0, 20), np.arange(20, 41), np.arange(41, 64)]
    nbrs = [(1, 0), (2, 1)]
    g = GPRFRef(X, Y, None, cov, 0.1, block_idxs=blocks, neighbors=nbrs, mode="rows")
    ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
    ptr, pts = pack_blocks(blocks)
    np.savez_compressed(os.path.join(HERE, "lld_toy.npz"), X=X, Y=Y, theta=np.array([0.1, 1.0, 40.0, 20.0]),
                        block_ptr=ptr, block_pts=pts, neighbors=np.array(nbrs, dtype=np.int32), ll=ll, gX=gX, gC=gC)


def degenerate():
    """Edge cases the reference handles: an empty block (gprf.py:507-513), a one-point block, a block of
    exactly 16 and 32 points (tile multiples), duplicated points that make K singular to working precision
    with zero noise -> jitchol's jitter path (gpy_linalg.py:81-97)."""
    rng = np.random.RandomState(3)
    X = rng.rand(80, 2)
    Y = rng.randn(80, 4)
    cov = GPCov([1.0], [0.5, 0.5], "euclidean", "se")
    blocks = [np.arange(0, 0), np.arange(0, 1), np.arange(1, 17), np.arange(17, 49), np.arange(49, 80)]
    nbrs = [(1, 0), (2, 1), (3, 2), (4, 3), (4, 0)]
    g = GPRFRef(X, Y, None, cov, 0.01, block_idxs=blocks, neighbors=nbrs, mode="rows")
    ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
    ptr, pts = pack_blocks(blocks)
    out = dict(X=X, Y=Y, theta=np.array([0.01, 1.0, 0.5, 0.5]), block_ptr=ptr, block_pts=pts,
               neighbors=np.array(nbrs, dtype=np.int32), ll=ll, gX=gX, gC=gC)
    # jitter case: duplicate points, no noise
    Xd = rng.rand(24, 2)
    Xd[12:] = Xd[:12]
    Yd = rng.randn(24, 3)
    gd = GPRFRef(Xd, Yd, None, cov, 0.0, block_idxs=[np.arange(24)], neighbors=[], mode="rows")
    lld, gXd, gCd = gd.llgrad(grad_X=True, grad_cov=True)
    out.update(dup_X=Xd, dup_Y=Yd, dup_theta=np.array([0.0, 1.0, 0.5, 0.5]), dup_ll=lld, dup_gX=gXd, dup_gC=gCd)
    np.savez_compressed(os.path.join(HERE, "degenerate.npz"), **out)


if __name__ == "__main__":
    c1(); tiny_parts(); lld_toy(); degenerate()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
