#!/usr/bin/env python
"""bench.py — GPRF objective+gradient evaluations per second on MI355X.

Metric (BASELINE.json): "GPRF objective+gradient evals/sec, n=10000 nblocks=100 yd=50".
Workload: n=10000, 100 grid blocks, yd=50, lscale=0.06, obs_std=0.02, noise 0.01, local_dist=0.5 (the full
GPRF objective: 100 unary units + 342 neighbouring block-pair units), task x (gradient w.r.t. X), synthetic
inputs regenerated from seed 0 by the reference's recipe (gprf_amd/synthetic.py).

One "step" = one objective+gradient evaluation of every unit: gather -> K fill -> Cholesky -> triangular
solves -> gradient reduce -> Bethe-weighted assembly (+ one all-reduce when N > 1), with X, Y, hypers and the
unit tables already resident in HBM and the result left in HBM (gprf_eval_device).  Steps cycle over 10
distinct X (the first L-BFGS-B iterates, each with its own re-blocking), enqueued back to back on one
stream.  N > 1 ("strong" scaling): the SAME evaluation's units are sharded over the ranks (LPT on
m^3 + 4 m^2 dy) and each step ends with a single RCCL all-reduce of 1 + n*dx + ncov doubles.

Rank 0 prints ONE JSON line.  Extra keys: "roofline" (dominant kernel, HIP-event timed inside the timed
region), "cpu_baseline" (the oracle's reference-shaped CPU port on this box's host cores), "stages_ms",
"sync_evals_per_s" (one host sync + D2H per evaluation, as an optimiser would call it),
"host_inclusive_evals_per_s" (Python update_X incl. host re-blocking + llgrad with H2D/D2H), and
"local_gp_evals_per_s" (BASELINE configs[1], no pairs).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix = vector peak (vendor data sheet; 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
FP64_MFMA_MEASURED_TFLOPS = 47.8   # scripts/mfma_f64_peak.hip on the box (profiles/r01_mfma_f64_peak.txt): clock under load
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ntrain", type=int, default=10000)
    ap.add_argument("--nblocks", type=int, default=100)
    ap.add_argument("--yd", type=int, default=50)
    ap.add_argument("--lscale", type=float, default=0.06)
    ap.add_argument("--obs-std", type=float, default=0.02)
    ap.add_argument("--local-dist", type=float, default=0.5)
    ap.add_argument("--task", default="x", choices=["x", "xcov"])
    ap.add_argument("--distinct-x", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only-north-star", action="store_true",
                    help="profiling runs: skip the secondary legs (fill-rate, synchronous, host-inclusive, local-GP, CPU) "
                         "so that every kernel launch of the process has the north-star shapes")
    ap.add_argument("--no-stage-timing", action="store_true",
                    help="diagnostic: no HIP events between the kernels in the timed region (no roofline then)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def algorithmic_flops(sizes, dy):
    """F(m) = m^3 + 4 m^2 dy, split by kernel."""
    m = np.asarray(sizes, dtype=np.float64)
    return {
        "potrf": float(np.sum(m ** 3 / 3.0)),
        "solve": float(np.sum(m ** 3 / 3.0 + m ** 2 * dy)),
        "at": float(np.sum(m ** 2 * dy)),
        "grad": float(np.sum(m ** 3 / 3.0 + 2.0 * m ** 2 * dy)),
        "fill_bytes": float(np.sum(8.0 * m ** 2)),
        "total": float(np.sum(m ** 3 + 4.0 * m ** 2 * dy)),
    }


def cpu_baseline(sd, local_dist, seconds, grad_cov):
    """The oracle (CPU port of the reference path) timed on this box's host cores.  (A) reference-shaped:
    serial Python loop over units, one Python->C call per (point, coordinate) for the kernel derivative rows
    (gprf.py:556-561), LAPACK dpotrf+dtrtri+dpotri+dpotrs as pdinv/dpotrs do; BLAS threads = all cores.
    (B) the same arithmetic with the per-row calls hoisted into C."""
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    out = {}
    for mode in ("rows", "matrix"):
        g = GPRFRef(sd.X_obs, sd.SY, sd.reblock, GPCov([1.0], [sd.lscale, sd.lscale], "euclidean", "se"),
                    sd.noise_var, block_idxs=sd.block_idxs, neighbors=sd.neighbors if local_dist < 1.0 else [],
                    mode=mode)
        n_done, t0 = 0, time.time()
        while True:
            g.update_X(sd.X_obs)
            g.llgrad(grad_X=True, grad_cov=grad_cov)
            n_done += 1
            if time.time() - t0 > seconds / 2:
                break
        out[mode] = (n_done / (time.time() - t0), n_done)
    return {"value": out["rows"][0], "unit": "evals/s", "cores": threads, "kind": "port",
            "sample": "%d full evaluation(s) of the same workload (all %d units), oracle mode=rows "
                      "(reference-shaped per-row derivative calls), BLAS threads=%d" % (out["rows"][1], g.n_blocks + len(g.neighbors), threads),
            "vectorised_value": out["matrix"][0],
            "vectorised_sample": "%d full evaluation(s), oracle mode=matrix" % out["matrix"][1]}


TIMING_PERIOD = 7


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # test hook for 1-GPU boxes: GPRF_BENCH_ONE_GPU=1 lets N ranks time-share GPU 0 with gloo collectives, so that the
    # multi-rank code path of this file can be exercised end to end; the numbers of such a run mean nothing
    one_gpu = os.environ.get("GPRF_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("GPRF_FORCE_ALLREDUCE") == "1":   # the latter: exercise the RCCL path on one GPU
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from gprf_amd import grid_centers, _capi
    from gprf_amd import dist as gdist
    from gprf_amd.objective import Objective
    from gprf_amd.synthetic import SampledData

    # ---------------- inputs (reference recipe, seed 0); rank 0 samples, everyone gets the same bytes
    ntrain, ntest = args.ntrain, 500
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), "gprf_bench_data")
    if rank == 0:
        sd = SampledData(n=ntrain + ntest, ntrain=ntrain, lscale=args.lscale, obs_std=args.obs_std, yd=args.yd,
                         seed=0, use_gpu=True, cache_dir=cache)
    if world > 1:
        dist.barrier()
        if rank != 0:
            sd = SampledData(n=ntrain + ntest, ntrain=ntrain, lscale=args.lscale, obs_std=args.obs_std, yd=args.yd,
                             seed=0, use_gpu=True, cache_dir=cache)
        ySY = torch.as_tensor(sd.SY, device=dev)
        dist.broadcast(ySY, 0)
        sd.SY = np.ascontiguousarray(ySY.cpu().numpy())
    sd.set_centers(grid_centers(args.nblocks))
    grad_cov = args.task == "xcov"
    n, dx = sd.X_obs.shape

    # ---------------- 10 distinct X: the first L-BFGS-B iterates (rank 0, unsharded), then broadcast
    nX = args.distinct_x
    Xs = np.zeros((nX, n, dx))
    if rank == 0:
        g0 = sd.build_gprf(local_dist=args.local_dist, device=local_rank)
        obj = Objective(g0, sd.X_obs, None, sd)
        seen = []

        class _Enough(Exception):
            pass

        def f(x):
            seen.append(x[:n * dx].reshape(n, dx).copy())
            if len(seen) >= nX:
                raise _Enough
            return obj(x)
        import scipy.optimize
        try:
            scipy.optimize.minimize(f, obj.full0, jac=True, method="l-bfgs-b", options={"ftol": 1e-6, "maxiter": 200})
        except _Enough:
            pass
        while len(seen) < nX:
            seen.append(seen[-1].copy())
        Xs[:] = np.stack(seen[:nX])
        g0.close()
    if world > 1:
        t = torch.as_tensor(Xs, device=dev)
        dist.broadcast(t, 0)
        Xs = t.cpu().numpy()

    # ---------------- one sharded context per distinct X (its own blocks, tables and workspace in HBM)
    evs, sizes_all = [], []
    for k in range(nX):
        blocks = sd.reblock(Xs[k])
        nbrs = sd.neighbors if args.local_dist < 1.0 else []
        from gprf_amd.gprf import GPRF
        g = GPRF(Xs[k], sd.SY, None, sd.cov, sd.noise_var, block_idxs=blocks, neighbors=nbrs, device=local_rank,
                 shard=(rank, world))
        g._push_neighbors(nbrs)
        ev = gdist.DeviceEvaluator(g)
        ev.set_X(Xs[k])
        evs.append(ev)
        sizes_all.append(gdist.unit_sizes(blocks, nbrs))
    torch.cuda.synchronize()
    # every evaluation of the timed region goes on ONE stream, strictly one after the other, as an
    # optimiser would issue them (no overlap between consecutive evaluations)
    run_stream = torch.cuda.Stream(device=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- warmup, then EXACTLY --steps timed steps
    for k in range(args.warmup):
        evs[k % nX].enqueue(True, grad_cov, stream=run_stream)
    barrier()
    # HIP events between the kernels cost ~3 us of stream time each (25 us per evaluation with all 8 of them):
    # the stage durations are sampled on every TIMING_PERIOD-th evaluation of the timed region (a period coprime
    # with the number of distinct X, so every context is sampled)
    for ev in evs:
        ev.g._ctx.set_timing(True, reset=True)
        ev.g._ctx.set_timing(False)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        e = evs[k % nX]
        sampled = (not args.no_stage_timing) and k % TIMING_PERIOD == 0
        if sampled:
            e.g._ctx.set_timing(True)
        e.enqueue(True, grad_cov, stream=run_stream)
        if sampled:
            e.g._ctx.set_timing(False)
    barrier()
    elapsed = time.perf_counter() - t0
    for ev in evs:
        rc, bad = ev.g._ctx.eval_status()
        assert rc == _capi.GPRF_OK, "unit %d not positive definite" % bad
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if args.no_stage_timing:
        if rank == 0:
            print(json.dumps({"diagnostic": "no HIP events between kernels in the timed region", "value": args.steps / elapsed,
                              "unit": "evals/s", "ms_per_step": 1e3 * elapsed / args.steps, "n_gpus": world}))
        return
    # per-stage averages over the timed region (HIP events on the launch stream), this rank's shard
    used = [ev for i, ev in enumerate(evs) if i < args.steps]
    stage = {}
    cnt = 0
    for ev in used:
        try:
            tm = ev.g._ctx.get_timing()
        except _capi.GprfHipError:      # a context the sampling never reached (very short runs)
            continue
        c = tm.pop("count")
        cnt += c
        for kname, v in tm.items():
            stage[kname] = stage.get(kname, 0.0) + v * c
    stage = {kname: v / max(cnt, 1) for kname, v in stage.items()}
    for ev in evs:
        ev.g._ctx.set_timing(False)

    value = args.steps / elapsed
    result = None
    if rank == 0:
        # algorithmic work of this rank's shard, averaged over the distinct X
        fl = {}
        for k in range(nX):
            owner = _capi.partition_units(sizes_all[k], args.yd, world)
            f = algorithmic_flops(sizes_all[k][owner == 0], args.yd)
            for a, b in f.items():
                fl[a] = fl.get(a, 0.0) + b / nX
        total_all = float(np.mean([algorithmic_flops(s, args.yd)["total"] for s in sizes_all]))
        compute_stages = ["potrf", "solve", "at", "grad"]
        dom = max(compute_stages + ["fill"], key=lambda s: stage[s])
        if dom == "fill":
            ach = fl["fill_bytes"] / (stage["fill"] * 1e-3) / 1e9
            roof = {"kernel": "k_fill", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None}
        else:
            ach = fl[dom] / (stage[dom] * 1e-3) / 1e12
            roof = {"kernel": "k_" + dom, "bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / FP64_PEAK_TFLOPS, "traffic": None}
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (separate --pmc runs of this
        # same command; (2*FETCH_SIZE + WRITE_SIZE) KiB, read side doubled as the guide prescribes for gfx950)
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01f_traffic.json")))
            key = {"potrf": "k_potrf_reg_gen", "solve": "k_solve_panel", "at": "k_at", "grad": "k_mgrad", "fill": "k_fill"}[dom]
            if world == 1 and args.ntrain == 10000 and args.nblocks == 100 and args.local_dist < 1.0:
                roof["traffic"] = tr[key]["bytes_per_launch"]
                roof["traffic_source"] = "profiles/r01f_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
        except Exception:
            pass
        roof["avg_launch_ms"] = stage[dom]
        roof["algorithmic_per_launch"] = fl["fill_bytes"] if dom == "fill" else fl[dom]
        if world == 1 and not args.only_north_star:
            # K is generated inside k_potrf_reg on this configuration and k_fill does not run; the fill kernel's
            # HBM write rate is measured on the side by forcing the K pool back for a few evaluations
            os.environ["GPRF_FUSED_FILL"] = "0"
            e = evs[0]
            e.g._ctx.set_timing(True, reset=True)
            for _ in range(8):
                e.enqueue(True, grad_cov, stream=run_stream)
            torch.cuda.synchronize()
            tmf = e.g._ctx.get_timing()
            e.g._ctx.set_timing(False, reset=True)
            del os.environ["GPRF_FUSED_FILL"]
            roof["fill_kernel"] = {"GBps": fl["fill_bytes"] / (tmf["fill"] * 1e-3) / 1e9, "ms": tmf["fill"],
                                   "potrf_ms_reading_K": tmf["potrf"],
                                   "note": "k_fill timed with GPRF_FUSED_FILL=0; by default k_potrf_reg generates K "
                                           "and it never exists in HBM (stages_ms.fill is then two event records)"}
        roof["mfma_f64_measured_peak"] = FP64_MFMA_MEASURED_TFLOPS
        roof["whole_eval_TFLOPs"] = total_all * value / 1e12
        roof["whole_eval_frac_of_fp64_peak"] = total_all * value / 1e12 / (FP64_PEAK_TFLOPS * world)
        result = {
            "metric": "GPRF objective+gradient evals/sec, n=%d nblocks=%d yd=%d" % (ntrain, args.nblocks, args.yd),
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "n=%d nblocks=%d yd=%d lscale=%g obs_std=%g local_dist=%g (%d unary + %d pair units) task=%s"
                                   % (ntrain, args.nblocks, args.yd, args.lscale, args.obs_std, args.local_dist,
                                      args.nblocks, len(sd.neighbors) if args.local_dist < 1.0 else 0, args.task),
                       "distinct_X": nX, "parallelism": "units sharded over %d rank(s), 1 all-reduce/eval" % world},
            "roofline": roof,
            **({"note": "GPRF_BENCH_ONE_GPU=1 test run: all ranks time-share one GPU over gloo; not a measurement"} if one_gpu else {}),
            "stages_ms": {k2: round(v, 5) for k2, v in stage.items()},
            "stage_timing": "HIP events between the kernels on every %d-th evaluation of the timed region (%d sampled)"
                            % (TIMING_PERIOD, cnt),
        }

    # ---------------- secondary rates (N = 1 only): synchronous, host-inclusive, local-GP config
    if world == 1 and not args.only_north_star:
        ev = evs[0]
        ts = []
        for k in range(min(args.steps, 50)):
            t1 = time.perf_counter()
            evs[k % nX].enqueue(True, grad_cov, stream=run_stream)
            evs[k % nX].result(True, grad_cov)
            ts.append(time.perf_counter() - t1)
        result["sync_evals_per_s"] = 1.0 / float(np.median(ts))
        gh = sd.build_gprf(local_dist=args.local_dist, device=local_rank)
        gh.llgrad(grad_X=True)
        ts = []
        for k in range(min(args.steps, 30)):
            t1 = time.perf_counter()
            gh.update_X(Xs[k % nX])
            gh.llgrad(grad_X=True, grad_cov=grad_cov)
            ts.append(time.perf_counter() - t1)
        result["host_inclusive_evals_per_s"] = 1.0 / float(np.median(ts))
        gh.close()
        gl = sd.build_gprf(local_dist=1.0, device=local_rank)
        el = gdist.DeviceEvaluator(gl)
        gl._push_neighbors(gl.neighbors)
        el.set_X(sd.X_obs)
        for _ in range(10):
            el.enqueue(True, grad_cov, stream=run_stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(100):
            el.enqueue(True, grad_cov, stream=run_stream)
        torch.cuda.synchronize()
        result["local_gp_evals_per_s"] = 100.0 / (time.perf_counter() - t1)
        gl.close()
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(sd, args.local_dist, args.cpu_seconds, grad_cov)
            result["speedup_vs_cpu_port"] = value / result["cpu_baseline"]["value"]

    for ev in evs:
        ev.g.close()
    if rank == 0:
        print(json.dumps(result))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
