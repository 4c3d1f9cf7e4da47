"""Single-process multi-device evaluation (gprf_create_multi, ``GPRF(devices=...)``): ONE process and one host thread drive
N member contexts — here N logical devices all mapped to GPU 0, which exercises everything but the xGMI hop: sharding inside
the library, the members' assembly kernels storing their partial vectors into slots on the first device, the summing kernel,
one completion flag.  The reference's drivers are one Python process around scipy (gprfopt.py:377-422) with the fan-out
inside llgrad (gprf.py:218-233): this is that shape."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(n=3000, nb=25, dy=6, seed=3):
    from gprf_amd import Blocker, grid_centers, GPCov
    rng = np.random.RandomState(seed)
    X = rng.rand(n, 2)
    Y = rng.randn(n, dy)
    b = Blocker(grid_centers(nb))
    return X, Y, b, GPCov([1.0], [0.1, 0.1], "euclidean", "se")


@pytest.mark.parametrize("ndev", [2, 3])
def test_group_equals_the_sum_of_shards_bit_for_bit_and_the_single_device_result(ndev):
    from gprf_amd.gprf import GPRF
    X, Y, b, cov = _case()
    nbrs = b.neighbors()
    one = GPRF(X, Y, b.block_clusters, cov, 0.01, neighbors=nbrs)
    ref = one.llgrad(grad_X=True, grad_cov=True)
    grp = GPRF(X, Y, b.block_clusters, cov, 0.01, neighbors=nbrs, devices=[0] * ndev)
    got = grp.llgrad(grad_X=True, grad_cov=True)
    assert grp._ctx.num_units() == (one._ctx.num_units()[0], one._ctx.num_units()[0])      # every unit on exactly one member
    # the members' partials, one process per "device" style, added in member order: the same bits
    acc = None
    for r in range(ndev):
        g = GPRF(X, Y, b.block_clusters, cov, 0.01, neighbors=nbrs, shard=(r, ndev), reduce=False)
        p = g.llgrad(grad_X=True, grad_cov=True)
        acc = list(p) if acc is None else [acc[0] + p[0], acc[1] + p[1], acc[2] + p[2]]
        g.close()
    assert got[0] == acc[0] and np.array_equal(got[1], acc[1]) and np.array_equal(got[2], acc[2])
    assert np.isclose(got[0], ref[0], rtol=1e-12)
    assert np.allclose(got[1], ref[1], rtol=0, atol=1e-11 * np.abs(ref[1]).max()) and np.allclose(got[2], ref[2], rtol=1e-10)
    # update_X re-partitions on every member; the walk stays equal to the single-device walk
    rng = np.random.RandomState(9)
    Xk = X
    for _ in range(3):
        Xk = Xk + 0.01 * rng.randn(*X.shape)
        one.update_X(Xk); grp.update_X(Xk)
        a, c = one.llgrad(grad_X=True), grp.llgrad(grad_X=True)
        assert np.isclose(a[0], c[0], rtol=1e-12) and np.allclose(a[1], c[1], rtol=0, atol=1e-11 * np.abs(a[1]).max())
    assert all(np.array_equal(u, v) for u, v in zip(one.block_idxs, grp.block_idxs))
    one.close(); grp.close()


def test_eight_members_on_one_gpu_equal_the_single_device_result():
    """GPRF(devices=[0] * 8): the shape of the one-process 8-GPU run with all eight members on GPU 0 — eight shards, eight
    slots, one summing kernel; every unit on exactly one member, the result equal to the single-device one to 1e-12 over a
    re-partitioning walk."""
    from gprf_amd.gprf import GPRF
    X, Y, b, cov = _case(n=4000, nb=36)
    nbrs = b.neighbors()
    one = GPRF(X, Y, b.block_clusters, cov, 0.01, neighbors=nbrs)
    grp = GPRF(X, Y, b.block_clusters, cov, 0.01, neighbors=nbrs, devices=[0] * 8)
    n_members, on_host, devs = grp._ctx.group_info()[:3]
    assert n_members == 8 and list(devs) == [0] * 8
    rng = np.random.RandomState(2)
    Xk = X
    for _ in range(3):
        one.update_X(Xk); grp.update_X(Xk)
        a, c = one.llgrad(grad_X=True, grad_cov=True), grp.llgrad(grad_X=True, grad_cov=True)
        assert grp._ctx.num_units() == (one._ctx.num_units()[0], one._ctx.num_units()[0])      # every unit on exactly one member
        assert sum(grp._ctx.group_info()[3]) == one._ctx.num_units()[0] and min(grp._ctx.group_info()[3]) > 0
        assert np.isclose(a[0], c[0], rtol=1e-12)
        assert np.allclose(a[1], c[1], rtol=0, atol=1e-12 * np.abs(a[1]).max()) and np.allclose(a[2], c[2], rtol=1e-11)
        Xk = Xk + 0.01 * rng.randn(*X.shape)
    one.close(); grp.close()


def test_group_drives_the_optimiser_callback_like_one_device():
    """gprf_objective over a group: the location prior is added by member 0 alone, the trace of a short L-BFGS-B run equals
    the single-device trace."""
    from gprf_amd.synthetic import SampledData
    from gprf_amd import grid_centers
    from gprf_amd.objective import do_optimization
    sd = SampledData(n=2500, ntrain=2000, lscale=6 / np.sqrt(2000), obs_std=2 / np.sqrt(2000), yd=12, seed=0)
    sd.set_centers(grid_centers(9))
    g1 = sd.build_gprf(local_dist=0.5)
    x1, o1 = do_optimization(g1, sd.X_obs, np.array([[0.13]]), sd, maxiter=4)
    g3 = sd.build_gprf(local_dist=0.5, devices=[0, 0, 0])
    x3, o3 = do_optimization(g3, sd.X_obs, np.array([[0.13]]), sd, maxiter=4)
    t1, t3 = [t[2] for t in o1.trace], [t[2] for t in o3.trace]
    assert len(t1) == len(t3) >= 4 and np.allclose(t1, t3, rtol=1e-11, atol=1e-12 * np.max(np.abs(t1)))
    assert np.allclose(x1, x3, rtol=0, atol=1e-8)
    assert np.isclose(sum(o3.parts), t3[-1], rtol=1e-12)
    g1.close(); g3.close()


def test_group_not_pd_and_too_big_units_are_reported_once(monkeypatch):
    from gprf_amd import GPCov, _capi, Blocker, grid_centers
    from gprf_amd.gprf import GPRF
    rng = np.random.RandomState(3)
    n, nb = 360, 6
    X = rng.rand(n, 2) * [1.0, 0.2]
    X[:, 0] = (np.repeat(np.arange(nb), n // nb) + X[:, 0]) / nb
    blocks = [np.arange(b * (n // nb), (b + 1) * (n // nb)) for b in range(nb)]
    X[blocks[2][:40]] = X[blocks[2][0]]                 # 40 copies of one location, zero noise: singular
    Y = rng.randn(n, 4)
    nbrs = [(b, b - 1) for b in range(1, nb)]
    cov = GPCov([1.0], [0.05, 0.05], "euclidean", "se")
    g1 = GPRF(X, Y, None, cov, 0.0, block_idxs=blocks, neighbors=nbrs)
    r1 = g1.llgrad(grad_X=True, grad_cov=True)
    g2 = GPRF(X, Y, None, cov, 0.0, block_idxs=blocks, neighbors=nbrs, devices=[0, 0])
    r2 = g2.llgrad(grad_X=True, grad_cov=True)          # jitchol's schedule ran over the group
    assert np.array_equal(g1._jitter, g2._jitter) and np.count_nonzero(g2._jitter) >= 1
    assert np.isclose(r1[0], r2[0], rtol=1e-12) and np.allclose(r1[1], r2[1], rtol=1e-9, atol=1e-9 * np.abs(r1[1]).max())
    g1.close(); g2.close()
    # a re-partition that grows a unit past GPRF_MAX_UNIT on one member: the library's message through the front context
    # (the limit lowered to 1024 for this: a real unit of 16385 points costs 4e12 flop to get to)
    monkeypatch.setenv("GPRF_DIAG", "max_unit=1024")
    Xa = rng.rand(1200, 2)
    bl = Blocker(grid_centers(4))
    gb = GPRF(Xa, rng.randn(1200, 3), bl.block_clusters, GPCov([1.0], [0.2, 0.2], "euclidean", "se"), 0.01,
              neighbors=bl.neighbors(), devices=[0, 0])
    gb.llgrad(grad_X=True)
    Xb = Xa.copy()
    Xb[:1100] = np.array(grid_centers(4))[rng.randint(0, 2, 1100)] + 0.01 * rng.randn(1100, 2)
    gb.update_X(Xb)
    with pytest.raises(_capi.GprfHipError, match="GPRF_MAX_UNIT"):
        gb.llgrad(grad_X=True)
    gb.close()
    with pytest.raises(_capi.GprfHipError, match="multi-device group"):
        g = GPRF(Xa, rng.randn(1200, 3), bl.block_clusters, GPCov([1.0], [0.2, 0.2], "euclidean", "se"), 0.01,
                 neighbors=bl.neighbors(), devices=[0, 0])
        try:
            g._ctx.set_shard(0, 2)
        finally:
            g.close()


def test_host_staged_slots_give_the_same_bits(monkeypatch):
    """No peer access between two devices -> the members' partial vectors meet in pinned host memory instead of the first
    device's memory (forced here: one GPU).  Same sums in the same order: the same bits as the peer-store form."""
    from gprf_amd.gprf import GPRF
    X, Y, b, cov = _case(n=2000, nb=16)
    nbrs = b.neighbors()
    peer = GPRF(X, Y, b.block_clusters, cov, 0.01, neighbors=nbrs, devices=[0, 0])
    a = peer.llgrad(grad_X=True, grad_cov=True)
    nm, on_host, devs, units = peer._ctx.group_info()
    assert nm == 2 and not on_host and devs == [0, 0] and sum(units) == peer._ctx.num_units()[0]
    monkeypatch.setenv("GPRF_GROUP_HOST_SLOTS", "1")
    host = GPRF(X, Y, b.block_clusters, cov, 0.01, neighbors=nbrs, devices=[0, 0])
    monkeypatch.delenv("GPRF_GROUP_HOST_SLOTS")
    c = host.llgrad(grad_X=True, grad_cov=True)
    assert host._ctx.group_info()[1] is True
    assert a[0] == c[0] and np.array_equal(a[1], c[1]) and np.array_equal(a[2], c[2])
    rng = np.random.RandomState(4)
    Xk = X + 0.01 * rng.randn(*X.shape)
    peer.update_X(Xk); host.update_X(Xk)
    a, c = peer.llgrad(grad_X=True), host.llgrad(grad_X=True)
    assert a[0] == c[0] and np.array_equal(a[1], c[1])
    peer.close(); host.close()


def test_group_over_distinct_devices_when_the_box_has_them():
    """The xGMI hop itself: members on DIFFERENT physical GPUs (skipped on one-GPU boxes — the hop has never run there)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from gprf_amd.gprf import GPRF
    X, Y, b, cov = _case()
    nbrs = b.neighbors()
    one = GPRF(X, Y, b.block_clusters, cov, 0.01, neighbors=nbrs)
    grp = GPRF(X, Y, b.block_clusters, cov, 0.01, neighbors=nbrs, devices=[0, 1])
    rng = np.random.RandomState(2)
    Xk = X
    for _ in range(20):      # repeated evaluations: a stale slot from the evaluation before would show
        Xk = Xk + 0.005 * rng.randn(*X.shape)
        one.update_X(Xk); grp.update_X(Xk)
        a, c = one.llgrad(grad_X=True, grad_cov=True), grp.llgrad(grad_X=True, grad_cov=True)
        assert np.isclose(a[0], c[0], rtol=1e-12) and np.allclose(a[1], c[1], rtol=0, atol=1e-11 * np.abs(a[1]).max())
        assert np.allclose(a[2], c[2], rtol=1e-10)
    assert grp._ctx.group_info()[2] == [0, 1]
    one.close(); grp.close()


def test_create_failures_say_why():
    from gprf_amd import _capi
    with pytest.raises(_capi.GprfHipError, match="sees .* HIP device"):
        _capi.Context(100, 2, 3, 0, 0, devices=[0, 4096])
    with pytest.raises(_capi.GprfHipError, match="dx must be 3"):
        _capi.Context(100, 2, 3, 1, 1, device=0)
