"""The optimiser-visible multi-GPU path on a one-GPU box: two processes (one per "GPU", both on device 0), gloo
collectives.  ``GPRF(shard=(rank, 2)).llgrad`` must return the ALL-REDUCED result on every rank, so that the
reference's driver (``do_optimization``: scipy L-BFGS-B around the objective callback, gprfopt.py:320-432) runs
unchanged and in lockstep on all ranks — the counterpart of the reference's process-pool fan-out inside llgrad
(gprf.py:218-233, 253-288).

Checked: (1) 5 L-BFGS-B iterations with device re-blocking on every evaluation: traces bit-identical across ranks,
equal to the single-process trace to 1e-12 relative, final X equal; (2) a unit that is not positive definite on ONE
rank: every rank walks jitchol's schedule (gpy_linalg.py:77-97) for the same unit and returns the same numbers — no
rank is left with NaN, none hangs in the next collective; a hopeless unit raises LinAlgError on all ranks."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sdata():
    from gprf_amd.synthetic import SampledData
    from gprf_amd import grid_centers
    sd = SampledData(n=2500, ntrain=2000, lscale=6 / np.sqrt(2000), obs_std=2 / np.sqrt(2000), yd=20, seed=0)
    sd.set_centers(grid_centers(9))
    return sd


def _singular_case():
    """6 blocks in a row; block 2 holds 40 copies of ONE location and the noise variance is 0: its kernel matrix (and
    its two pairs') is exactly singular; jitchol's first jitter, 1e-6 * mean(diag K), rescues each."""
    rng = np.random.RandomState(3)
    n, nb = 360, 6
    X = rng.rand(n, 2) * [1.0, 0.2]
    X[:, 0] = (np.repeat(np.arange(nb), n // nb) + X[:, 0]) / nb
    blocks = [np.arange(b * (n // nb), (b + 1) * (n // nb)) for b in range(nb)]
    X[blocks[2][:40]] = X[blocks[2][0]]
    Y = rng.randn(n, 4)
    nbrs = [(b, b - 1) for b in range(1, nb)]
    return X, Y, blocks, nbrs


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gprf_amd import GPCov, _capi
        from gprf_amd.gprf import GPRF
        from gprf_amd.objective import do_optimization
        out = {}
        sd = _sdata()
        g = sd.build_gprf(local_dist=0.5, shard=(rank, world))
        rx, obj = do_optimization(g, sd.X_obs, None, sd, maxiter=5)
        out["trace"] = [t[2] for t in obj.trace]
        out["x"] = rx
        out["n_local"] = g._ctx.num_units()[1]
        out["blocks"] = [np.asarray(b) for b in g.block_idxs]
        g.close()
        # --- a unit that fails on one rank only
        X, Y, blocks, nbrs = _singular_case()
        Xnan = X.copy()
        Xnan[blocks[2][5], 0] = np.nan                  # no jitter repairs a NaN entry
        for Xc, key in ((X, "jitter"), (Xnan, "hopeless")):
            gs = GPRF(Xc, Y, None, GPCov([1.0], [0.05, 0.05], "euclidean", "se"), 0.0, block_idxs=blocks,
                      neighbors=nbrs, shard=(rank, world))
            try:
                r = gs.llgrad(grad_X=True, grad_cov=True)
                out[key] = (r[0], r[1], r[2], None if gs._jitter is None else gs._jitter.copy())
            except np.linalg.LinAlgError as e:
                out[key] = ("raised", str(e))
            # the collective still works afterwards: nobody is out of step
            t = torch.ones(1, dtype=torch.float64, device="cuda")
            dist.all_reduce(t)
            out[key + "_after"] = float(t.item())
            gs.close()
        # --- a re-partition that grows ONE unit past GPRF_MAX_UNIT points: fatal on the rank that owns it only
        # (the limit lowered to 1024 for this: a real unit of 16385 points costs 4e12 flop to get to)
        os.environ["GPRF_DIAG"] = "max_unit=1024"
        from gprf_amd import Blocker, grid_centers
        rng = np.random.RandomState(8)
        Xa = rng.rand(1200, 2)
        bl = Blocker(grid_centers(4))
        gb = GPRF(Xa, rng.randn(1200, 3), bl.block_clusters, GPCov([1.0], [0.2, 0.2], "euclidean", "se"), 0.01,
                  neighbors=bl.neighbors(), shard=(rank, world))
        gb.llgrad(grad_X=True)
        Xb = Xa.copy()
        Xb[:1100] = np.array(grid_centers(4))[rng.randint(0, 2, 1100)] + 0.01 * rng.randn(1100, 2)   # 1100 points in blocks 0, 1
        gb.update_X(Xb)
        try:
            gb.llgrad(grad_X=True)
            out["too_big"] = "returned"
        except _capi.GprfHipError as e:
            out["too_big"] = "raised: %s" % e
        t = torch.ones(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        out["too_big_after"] = float(t.item())
        out["too_big_owner"] = int(_capi.partition_units(np.array([300] * 4 + [600] * 6, dtype=np.int32), 3, world)[4])
        gb.close()
        del os.environ["GPRF_DIAG"]
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_two_rank_optimisation_and_jitter_in_lockstep():
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a, b = res[0], res[1]
    # every unit evaluated exactly once: 9 blocks + 20 pairs shared out
    assert a["n_local"] + b["n_local"] == 29 and a["n_local"] > 0 and b["n_local"] > 0
    assert len(a["trace"]) >= 5 and a["trace"] == b["trace"]                 # bit-identical on both ranks
    assert np.array_equal(a["x"], b["x"])
    assert all(np.array_equal(u, v) for u, v in zip(a["blocks"], b["blocks"]))
    # ... and the single-process run
    from gprf_amd.objective import do_optimization
    sd = _sdata()
    g = sd.build_gprf(local_dist=0.5)
    rx, obj = do_optimization(g, sd.X_obs, None, sd, maxiter=5)
    one = [t[2] for t in obj.trace]
    g.close()
    assert len(one) == len(a["trace"])
    # (the objective passes through zero along the run: rounding-level differences between the two summation orders are
    # measured against the largest value of the trace, not against the value next to the sign change)
    assert np.allclose(one, a["trace"], rtol=1e-12, atol=1e-12 * np.max(np.abs(one)))
    assert np.allclose(rx, a["x"], rtol=0, atol=1e-9)
    # --- the not-PD unit: same answer on both ranks, equal to the single-process answer
    from gprf_amd import GPCov
    from gprf_amd.gprf import GPRF
    X, Y, blocks, nbrs = _singular_case()
    g1 = GPRF(X, Y, None, GPCov([1.0], [0.05, 0.05], "euclidean", "se"), 0.0, block_idxs=blocks, neighbors=nbrs)
    r1 = g1.llgrad(grad_X=True, grad_cov=True)
    j1 = g1._jitter
    g1.close()
    assert j1 is not None and np.count_nonzero(j1) >= 1                      # the schedule really ran
    for r in (a, b):
        ll, gX, gC, jit = r["jitter"]
        assert np.isfinite(ll) and np.all(np.isfinite(gX)) and np.all(np.isfinite(gC))
        assert np.array_equal(jit, j1)                                       # same units, same jitter on every rank
        assert np.isclose(ll, r1[0], rtol=1e-12) and np.allclose(gX, r1[1], rtol=1e-9, atol=1e-9 * np.abs(r1[1]).max())
        assert r["jitter_after"] == 2.0
    assert a["jitter"][0] == b["jitter"][0] and np.array_equal(a["jitter"][1], b["jitter"][1])
    for r in (a, b):
        assert r["hopeless"][0] == "raised" and "not positive definite" in r["hopeless"][1].lower()
        assert r["hopeless_after"] == 2.0
    # the unit past GPRF_MAX_UNIT: BOTH ranks raise (the owner with the library's message), nobody hangs
    for r in (a, b):
        assert r["too_big"].startswith("raised"), r["too_big"]
        assert r["too_big_after"] == 2.0
    assert any("GPRF_MAX_UNIT" in r["too_big"] for r in (a, b)) and any("another rank" in r["too_big"] for r in (a, b))
