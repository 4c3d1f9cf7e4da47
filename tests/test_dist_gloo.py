"""The N > 1 path on CPU: two processes over gloo.  Each rank takes the units gprf_partition_units gives
it, forms its partial [ll | gradX | gradC] (here with the oracle standing in for the device evaluator — the
GPU evaluation itself is covered by the -m gpu tests, including a two-shard run on one GPU), the partials
are all-reduced with the product's own collective wrapper, and every rank must hold the full result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden, blocks_from_csr
from gprf_amd import dist as gdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _partial(rank, world, z):
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov
    X, Y, th = z["X_obs"], z["SY"], z["theta"]
    blocks = blocks_from_csr(z["block_ptr"], z["block_pts"])
    nbrs = [tuple(int(v) for v in r) for r in z["neighbors"]]
    g = GPRFRef(X, Y, None, GPCov([th[1]], th[2:], "euclidean", "se"), th[0], block_idxs=blocks, neighbors=nbrs)
    mine = gdist.local_units(blocks, nbrs, Y.shape[1], rank, world)
    n, dx = X.shape
    ll, gX, gC = 0.0, np.zeros((n, dx)), np.zeros(4)
    nb = len(blocks)
    for u in mine:
        if u < nb:
            w = 1 - g.neighbor_count[u]
            idx = blocks[u]
        else:
            w = 1
            i, j = nbrs[u - nb]
            idx = np.concatenate([blocks[i], blocks[j]])
        l, gx, gc = g.gaussian_llgrad(X[idx], Y[idx], grad_X=True, grad_cov=True)
        ll += w * l
        np.add.at(gX, idx, w * gx)
        gC += w * gc
    return gdist.pack_out(ll, gX, gC, n, dx, 4), len(mine)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z = load_golden("c1_small.npz")
        buf, n_mine = _partial(rank, world, z)
        t = torch.from_numpy(buf)
        gdist.allreduce_sum_(t)
        bad = gdist.agree_first_bad(7 if rank == 1 else -1)
        none_bad = gdist.agree_first_bad(-1)
        n, dx = z["X_obs"].shape
        ll, gX, gC = gdist.unpack_out(t.numpy(), n, dx, 4, True, True)
        q.put((rank, n_mine, ll, gX, gC, bad, none_bad))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_sum_equals_full():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    z = load_golden("c1_small.npz")
    assert sum(r[1] for r in res) == 4 + 6           # every unit evaluated exactly once
    assert all(r[1] > 0 for r in res)
    for (_, _, ll, gX, gC, bad, none_bad) in res:
        assert np.isclose(ll, z["ll_gprf"], rtol=1e-13)
        assert np.allclose(gX, z["gX_gprf"], rtol=0, atol=1e-9 * np.abs(z["gX_gprf"]).max())
        assert np.allclose(gC, z["gC_gprf"], rtol=1e-11)
        assert bad == 7 and none_bad == -1


def test_pack_unpack_roundtrip():
    gX = np.arange(12.0).reshape(6, 2)
    gC = np.array([1.0, 2.0, 3.0, 4.0])
    buf = gdist.pack_out(-3.5, gX, gC, 6, 2, 4)
    assert buf.shape == (1 + 12 + 4 + 2,)
    ll, a, b = gdist.unpack_out(buf, 6, 2, 4, True, True)
    assert ll == -3.5 and np.array_equal(a, gX) and np.array_equal(b, gC.reshape(1, -1))
    ll, a, b = gdist.unpack_out(buf, 6, 2, 4, False, False)
    assert a.shape == (0, 0) and b.shape == (0, 0)      # gprf.py:275,291
