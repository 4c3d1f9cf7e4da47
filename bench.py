#!/usr/bin/env python
"""bench.py — GPRF objective+gradient evaluations per second on MI355X.

Metric (BASELINE.json): "GPRF objective+gradient evals/sec, n=10000 nblocks=100 yd=50".
Workload: n=10000, 100 grid blocks, yd=50, lscale=0.06, obs_std=0.02, noise 0.01, local_dist=0.5 (the full
GPRF objective: 100 unary units + 342 neighbouring block-pair units), task x (gradient w.r.t. X), synthetic
inputs regenerated from seed 0 by the reference's recipe (gprf_amd/synthetic.py).

One "step" = ONE objective+gradient evaluation exactly as the reference's L-BFGS-B callback issues it
(gprfopt.py:377-417): host X in -> GPRF.update_X(X) (re-runs the block function: re-partition, here on
the device) -> GPRF.llgrad(grad_X=True) -> host (ll, gradX) out, the evaluation completely finished — result on the
host, all-reduce included when N > 1 — before the next one starts.  Steps cycle over 10 distinct X (the first
L-BFGS-B iterates: every step re-partitions).  `value` = steps / wall time of that sequential loop.
N > 1 ("strong" scaling): the SAME evaluation's units are sharded over the ranks (LPT on m^3 + 4 m^2 dy); every rank
holds X, Y, theta; each evaluation ends with a single RCCL all-reduce of 1 + n*dx + ncov (+2 status) doubles.
`python bench.py --gpus N` (no launcher) starts that N-rank job ITSELF, as a child process (torch.distributed.run, one rank
per GPU) before this process touches the GPU, relays rank 0's line and adds the single-process figure
(gprf_create_multi: one process, N devices) from a second child; under torch.distributed.run it is one of the ranks.
`value` = steps / the MEDIAN wall time of REPS repetitions of the timed loop (every repetition: exactly --steps
evaluations between barrier + synchronize; all samples in "ms_per_step_samples").

Rank 0 prints ONE JSON line.  Extra keys: "roofline" (dominant kernel, HIP-event timed inside the timed region),
"cpu_baseline" (the oracle's CPU port on this box's host cores), "stages_ms", "device_resident_evals_per_s" (the
same evaluations enqueued back to back with X / result left in HBM and no host synchronisation: the kernels' own
rate), "llgrad_only_evals_per_s" (blocks fixed, no update_X), "local_gp_evals_per_s" (BASELINE configs[1], no pairs)
and "c4_evals_per_s" (BASELINE configs[3]: n=80000, 841 blocks + 3192 pairs, task xcov — the configuration whose
work is large enough to shard), measured the same sequential way at every N.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix = vector peak (vendor data sheet; 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
FP64_MFMA_MEASURED_TFLOPS = 47.8   # scripts/mfma_f64_peak.hip on the box (profiles/r01_mfma_f64_peak.txt): the all-MFMA micro-benchmark, clock lowered under that load (a real kernel exceeds it: k_big_gemm's long-K pass 56)
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md
TRAFFIC_FILES = [os.path.join("profiles", "r06_traffic.json"), os.path.join("profiles", "r05_traffic.json")]      # newest first
STAGE_PASS_EVALS = 60     # evaluations of the separate pass that times every kernel of every evaluation


def _fill_counter_gbps():
    """(WRITE_SIZE + 2 FETCH_SIZE) per launch / the kernel's average duration, from the committed rocprofv3 passes of k_fill_se
    (scripts/profile_fill.sh -> profiles/r06_fill_counters.json); None when they were never taken"""
    try:
        for name in ("r06_fill_counters.json", "r05_fill_counters.json"):      # newest first
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                with open(path) as f:
                    return float(json.load(f)["counter_GBps"])
        return None
    except (OSError, ValueError, KeyError):
        return None


FILL_COUNTER_GBPS = _fill_counter_gbps()
REPS = 7                  # repetitions of the timed loop; `value` is their median (a slow leg shows in the samples, not in the headline)


def source_hash():
    """sha256 over the native sources (gprf_amd/build.py): a committed rocprofv3 traffic figure counts for THIS code only if it
    was profiled on the same sources"""
    from gprf_amd import build as hip_build
    return hip_build.source_hash()


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
    ap.add_argument("--no-c4", action="store_true", help="skip the BASELINE configs[3] leg (n=80000)")
    ap.add_argument("--single-process", action="store_true",
                    help="drive the --gpus N devices from THIS one process (gprf_create_multi: no torchrun, no inter-process "
                         "collective) — the shape of the reference's own single-process drivers")
    ap.add_argument("--no-c5", action="store_true", help="skip the BASELINE configs[4]-shaped leg (seismic stand-in, n=20000)")
    ap.add_argument("--only-north-star", action="store_true",
                    help="profiling runs: skip the secondary legs so that every kernel launch of the process has the "
                         "north-star shapes")
    ap.add_argument("--no-stage-timing", action="store_true",
                    help="diagnostic: skip the separate per-kernel timing pass (no roofline then)")
    ap.add_argument("--source-hash", action="store_true", help="print the native sources' hash (profile_run.sh) and exit")
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
    ap.add_argument("--no-parity", action="store_true", help="skip the parity sample against the oracle")
    ap.add_argument("--reps", type=int, default=REPS, help="repetitions of the timed loop (value = median)")
    ap.add_argument("--no-single-process-leg", action="store_true",
                    help="self-launched N > 1 run: skip the second child (one process driving the N devices)")
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


# ------------------------------------------------------------------------------------------------ CPU baseline
_POOL_STATE = {}


def _pool_init(X, Y, lscale, nv, blocks, nbrs):
    from threadpoolctl import threadpool_limits
    _POOL_STATE["limit"] = threadpool_limits(limits=1)      # (kept alive; the environment variables set by the parent
                                                            # already start the BLAS of a spawned worker single-threaded)
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov
    _POOL_STATE["g"] = GPRFRef(X, Y, None, GPCov([1.0], [lscale, lscale], "euclidean", "se"), nv, block_idxs=blocks,
                               neighbors=nbrs, mode="matrix")


def _pool_unit(u):
    g = _POOL_STATE["g"]
    nb = g.n_blocks
    if u < nb:
        r = g.llgrad_unary(u, grad_X=True)
    else:
        i, j = g.neighbors[u - nb]
        r = g.llgrad_joint(i, j, grad_X=True)
    return float(r[0])


def cpu_baseline(sd, local_dist, seconds, grad_cov):
    """The oracle (CPU port of the reference path, kind "port") timed on this box's host cores.

    (A) reference-shaped, THE WHOLE EVALUATION: update_X's host re-blocking, then the serial loop over ALL units, one
        Python->C call per (point, coordinate) for the kernel derivative rows (gprf.py:556-561), LAPACK
        dpotrf+dtrtri+dpotri+dpotrs as pdinv/dpotrs do (oracle mode="rows"), the Bethe-weighted assembly (gprf.py:253-273);
        BLAS threads = 1 (`value`: >= 3 whole evaluations, median) and = all cores (as many whole evaluations as the
        budget allows, at least one).  Nothing is extrapolated.  `sample_estimate_evals_per_s` is the every-6th-unit
        estimate earlier rounds reported, kept as a cross-check of those records only.
    (B) best-effort CPU: the same arithmetic with the per-row calls hoisted into C (mode="matrix"), ALL units fanned
        over a process pool (one BLAS thread per worker), 3 full evaluations (median)."""
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov
    from threadpoolctl import threadpool_limits
    cores = os.cpu_count() or 1
    nbrs = sd.neighbors if local_dist < 1.0 else []
    g = GPRFRef(sd.X_obs, sd.SY, sd.reblock, GPCov([1.0], [sd.lscale, sd.lscale], "euclidean", "se"), sd.noise_var,
                block_idxs=sd.block_idxs, neighbors=nbrs, mode="rows")
    nb, nu = g.n_blocks, g.n_blocks + len(nbrs)
    sizes = [len(b) for b in sd.block_idxs] + [len(sd.block_idxs[i]) + len(sd.block_idxs[j]) for (i, j) in nbrs]
    order = np.argsort(sizes, kind="stable")

    def whole_eval():
        t0 = time.perf_counter()
        g.update_X(sd.X_obs)                    # re-runs the block function (gprf.py:171-172): part of one evaluation
        g.llgrad(grad_X=True, grad_cov=grad_cov)
        return time.perf_counter() - t0

    rows = {}
    t_budget_end = time.perf_counter() + seconds
    for threads, min_reps in ((1, 3), (cores, 1)):
        with threadpool_limits(limits=threads):
            ts = [whole_eval()]
            while len(ts) < 3 and (len(ts) < min_reps or time.perf_counter() + ts[-1] < t_budget_end):
                ts.append(whole_eval())
        rows[threads] = (1.0 / float(np.median(ts)), len(ts))
    # the earlier rounds' estimator (every 6th unit of the size-sorted list, extrapolated): a cross-check only
    stride = 6
    sample = order[stride // 2::stride]
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        for u in sample:
            if u < nb:
                g.llgrad_unary(int(u), grad_X=True, grad_cov=grad_cov)
            else:
                i, j = nbrs[int(u) - nb]
                g.llgrad_joint(i, j, grad_X=True, grad_cov=grad_cov)
        est = (time.perf_counter() - t0) * (nu / float(len(sample)))
    t_blocking = []
    for _ in range(3):
        t0 = time.perf_counter()
        g.update_X(sd.X_obs)
        t_blocking.append(time.perf_counter() - t0)
    out = {"value": rows[1][0], "unit": "evals/s", "cores": 1, "kind": "port",
           "sample": "the WHOLE evaluation, nothing extrapolated: oracle mode=rows (reference-shaped: update_X re-blocking + serial loop "
                     "over all %d units, per-(point,coordinate) derivative calls, dpotrf+dtrtri+dpotri+dpotrs, Bethe assembly), "
                     "1 BLAS thread, %d whole evaluations (median)" % (nu, rows[1][1]),
           "rows_blas1_evals_per_s": rows[1][0], "rows_blas_all_evals_per_s": rows[cores][0],
           "rows_blas_all_reps": rows[cores][1], "host_cores": cores,
           "sample_estimate_evals_per_s": 1.0 / (est + float(np.median(t_blocking)))}
    # (B) process pool over ALL units, vectorised per-unit arithmetic
    saved = {}
    try:
        import multiprocessing as mp
        nproc = min(cores, 64)
        ctx = mp.get_context("spawn")      # (never fork a process that has initialised the GPU)
        saved = dict((k, os.environ.get(k)) for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"))
        for k in saved:
            os.environ[k] = "1"
        with ctx.Pool(nproc, initializer=_pool_init,
                      initargs=(sd.X_obs, sd.SY, sd.lscale, sd.noise_var, sd.block_idxs, nbrs)) as pool:
            # (bounded: a pool that crawls — oversubscribed BLAS threads on a big host — must not eat the bench's minutes)
            pool.map_async(_pool_unit, range(nu), chunksize=max(1, nu // (4 * nproc))).get(timeout=90)      # warm-up
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                pool.map_async(_pool_unit, [int(u) for u in order[::-1]], chunksize=1).get(timeout=30)      # largest first
                ts.append(time.perf_counter() - t0 + float(np.median(t_blocking)))
        out["pool_value"] = 1.0 / float(np.median(ts))
        out["pool_sample"] = "oracle mode=matrix, all %d units over a %d-process pool (1 BLAS thread each), 3 full evaluations (median)" % (nu, nproc)
    except Exception as e:      # the pool is a side figure: never lose the bench line over it
        out["pool_value"] = None
        out["pool_sample"] = "process pool failed: %r" % (e,)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return out


def parity_sample(g, sd, X, nbrs, grad_cov, n_sample=48):
    """`n_sample` units of the timed configuration (every k-th of the size-sorted unit list: unaries and pairs, smallest to
    largest), evaluated by the device (per-unit results through gprf_debug_run / gprf_debug_fetch) and by the oracle
    (GPRFRef.gaussian_llgrad: gprf.py:496-591 on LAPACK) on identical inputs.  BASELINE.json's tolerance is "gradient max-abs
    error < 1e-8" at max|gradX| ~ 2e5; two fp64 evaluations of these units differ by 1-3e-8, so the
    relative figure is carried next to the absolute one."""
    from oracle.gprf_ref import GPRFRef
    from oracle.vector_tree import GPCov
    blocks = sd.reblock(X)
    ref = GPRFRef(X, sd.SY, None, GPCov([1.0], [sd.lscale, sd.lscale], "euclidean", "se"), sd.noise_var, block_idxs=blocks,
                  neighbors=nbrs, mode="matrix")
    g.update_X(X)
    g.llgrad(grad_X=True, grad_cov=grad_cov)
    ctx = g._ctx
    ctx.debug_run(np.ascontiguousarray(X), 6)
    nb = len(blocks)
    units = [(b, None) for b in range(nb)] + [(i, j) for (i, j) in nbrs]
    sizes = np.array([len(blocks[i]) + (len(blocks[j]) if j is not None else 0) for (i, j) in units])
    order = np.argsort(sizes, kind="stable")
    pick = order[np.unique(np.linspace(0, len(order) - 1, min(n_sample, len(order))).round().astype(int))]
    e_abs, e_rel, e_ll, gmax_all = 0.0, 0.0, 0.0, 0.0
    for u in pick:
        i, j = units[int(u)]
        idx = blocks[i] if j is None else np.concatenate([blocks[i], blocks[j]])
        m = len(idx)
        if m == 0:
            continue
        dx = X.shape[1]
        d_gx = ctx.debug_fetch(int(u), 4)[:m, :dx]
        d_ll = float(ctx.debug_fetch(int(u), 5)[0])
        o_ll, o_gx, _ = ref.gaussian_llgrad(X[idx], sd.SY[idx], grad_X=True)
        gm = float(np.max(np.abs(o_gx)))
        e = float(np.max(np.abs(d_gx - o_gx)))
        e_abs, e_rel, gmax_all = max(e_abs, e), max(e_rel, e / gm), max(gmax_all, gm)
        e_ll = max(e_ll, abs(d_ll - o_ll) / abs(o_ll))
    return {"units": int(len(pick)), "unit_points": [int(sizes[pick].min()), int(sizes[pick].max())],
            "max_abs_gX_vs_oracle": e_abs, "rel": e_rel, "ll_rel": e_ll, "max_abs_gX": gmax_all,
            "tolerance": "north_star: gradient max-abs error < 1e-8 — below the reference path's own rounding at max|gradX| ~ 2e5 "
                         "(fp64 LAPACK is 1-2e-8 per unit from an 80-bit evaluation, tests/test_gpu_northstar.py); asserted in the tests: "
                         "per unit <= 3.0e-8 absolute and <= 1.15x the oracle's own distance from the 80-bit truth, ll relative 1e-12",
            "checker": "oracle.gprf_ref.GPRFRef.gaussian_llgrad (gprf.py:496-591 restated on LAPACK), per unit, outside the timed region"}



def git_head():
    try:
        return subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


def _last_json_line(text):
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and ('"metric"' in line or '"value"' in line):      # (the --no-stage-timing line has no "metric")
            try:
                return json.loads(line)
            except ValueError:
                pass
    return None


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: THIS process never touches the GPU (no torch import, no HIP call) — it
    starts the N-rank job as a child (`python -m torch.distributed.run --nproc-per-node N bench.py ...`: one rank per GPU,
    one RCCL all-reduce per evaluation, the north-star design), relays rank 0's JSON line and exits with the child's code;
    a second child times the single-process form (gprf_create_multi) and its figure rides along as an extra key."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    me = os.path.abspath(__file__)
    argv = [a for a in sys.argv[1:] if a != "--no-single-process-leg"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), me] + argv
    t0 = time.perf_counter()
    # (bounded: a stuck rendezvous or collective initialisation must not hang whoever called this for good)
    limit = float(os.environ.get("GPRF_BENCH_CHILD_TIMEOUT_S", "3000"))
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=limit)
    except subprocess.TimeoutExpired as e:
        def _txt(b):
            return b.decode(errors="replace") if isinstance(b, bytes) else (b or "")
        sys.stderr.write("bench.py: the %d-rank child did not finish within %.0f s\n--- its stdout (tail)\n%s\n--- its stderr (tail)\n%s\n"
                         % (args.gpus, limit, _txt(e.stdout)[-2000:], _txt(e.stderr)[-4000:]))
        sys.exit(124)
    line = _last_json_line(r.stdout or "")
    if r.returncode != 0 or line is None:
        sys.stdout.write(r.stdout or "")
        sys.stderr.write("bench.py: the %d-rank child exited with code %d%s\n--- its stderr (tail)\n%s\n"
                         % (args.gpus, r.returncode, "" if line is not None else " and printed no result line", (r.stderr or "")[-4000:]))
        sys.exit(r.returncode if r.returncode != 0 else 1)
    line["launched_by"] = "bench.py itself: child `python -m torch.distributed.run --nproc-per-node %d` (%.0f s)" % (
        args.gpus, time.perf_counter() - t0)
    if not args.no_single_process_leg:
        sp = {"value": None}
        try:
            cmd2 = [sys.executable, me, "--single-process", "--only-north-star", "--no-cpu-baseline"] + argv
            r2 = subprocess.run(cmd2, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
            l2 = _last_json_line(r2.stdout or "")
            if r2.returncode == 0 and l2 is not None:
                sp = {"value": l2["value"], "unit": l2["unit"], "ms_per_step": l2["ms_per_step"],
                      "ms_per_step_samples": l2.get("ms_per_step_samples"), "parallelism": l2["config"]["parallelism"],
                      "group": l2["config"].get("group")}
            else:
                sp["error"] = "rc %d: %s" % (r2.returncode, (r2.stderr or "")[-400:])
        except Exception as e:      # a side figure: never lose the headline over it
            sp["error"] = repr(e)
        line["single_process"] = sp
        line["single_process_evals_per_s"] = sp["value"]
    print(json.dumps(line))
    sys.exit(0)


def main():
    args = parse()
    if args.source_hash:
        print(source_hash())
        return
    if (args.gpus > 1 and not args.single_process and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ
            and "LOCAL_RANK" not in os.environ):
        launch_ranks(args)      # (does not return)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    devices = None
    if args.single_process:
        if world != 1:
            sys.exit("bench.py --single-process runs as ONE process (no torch.distributed.run)")
        # (GPRF_BENCH_ONE_GPU=1: N logical members on GPU 0 — exercises the path on a one-GPU box, the numbers mean nothing)
        devices = [0] * args.gpus if os.environ.get("GPRF_BENCH_ONE_GPU") == "1" else list(range(args.gpus))
    elif world != args.gpus:
        args.gpus = world       # under a launcher the launcher's world size is the truth

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
    # next to the GPU: the zero-copy evaluation is 25-55 % slower from the other socket of a two-socket box, and which socket a
    # process starts on is the scheduler's choice (gprf_amd/numa.py; GPRF_NUMA_PIN=0 leaves the affinity alone)
    from gprf_amd import numa
    affinity_before = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    numa_node, numa_cpus = numa.pin_to_gpu_node(local_rank)
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
    from gprf_amd.gprf import GPRF
    from gprf_amd.objective import Objective
    from gprf_amd.synthetic import SampledData

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def sample(ntrain, lscale, obs_std, nblocks):
        """reference recipe, seed 0; rank 0 samples (N x N prior Cholesky on its GPU), everyone gets the same bytes"""
        cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), "gprf_bench_data")
        kw = dict(n=ntrain + 500, ntrain=ntrain, lscale=lscale, obs_std=obs_std, yd=args.yd, seed=0, use_gpu=True, cache_dir=cache)
        sd = SampledData(**kw) if rank == 0 else None
        if world > 1:
            dist.barrier()
            if rank != 0:
                sd = SampledData(**kw)      # (cached by rank 0 when TMPDIR is shared; the broadcast below decides)
            ySY = torch.as_tensor(sd.SY, device=dev)
            dist.broadcast(ySY, 0)
            sd.SY = np.ascontiguousarray(ySY.cpu().numpy())
        sd.set_centers(grid_centers(nblocks))
        return sd

    def sequential_rate(g, Xlist, steps, warmup, grad_cov, reps=1):
        """THE metric's loop: update_X + llgrad, one evaluation finished before the next starts, nothing else in the
        timed region (no events).  `reps` repetitions, each EXACTLY `steps` evaluations between barrier + synchronize
        (max over ranks); -> (evals/s at the median repetition, its ms per step, ms per step of every repetition)"""
        nXl = len(Xlist)
        for k in range(warmup):
            g.update_X(Xlist[k % nXl])
            g.llgrad(grad_X=True, grad_cov=grad_cov)
        samples = []
        for _ in range(max(1, reps)):
            barrier()
            t0 = time.perf_counter()
            for k in range(steps):
                g.update_X(Xlist[k % nXl])
                g.llgrad(grad_X=True, grad_cov=grad_cov)
            barrier()
            el = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            samples.append(el)
        el = float(np.median(samples))
        return steps / el, 1e3 * el / steps, [1e3 * e / steps for e in samples]

    # ---------------- inputs
    ntrain = args.ntrain
    sd = sample(ntrain, args.lscale, args.obs_std, args.nblocks)
    grad_cov = args.task == "xcov"
    n, dx = sd.X_obs.shape
    nbrs = sd.neighbors if args.local_dist < 1.0 else []

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
    Xlist = [np.ascontiguousarray(Xs[k]) for k in range(nX)]
    sizes_all = [gdist.unit_sizes(sd.reblock(Xs[k]), nbrs) for k in range(nX)]

    # ---------------- the timed region: the reference's GPRF object, sharded over the ranks
    g = (sd.build_gprf(local_dist=args.local_dist, devices=devices) if devices is not None else
         sd.build_gprf(local_dist=args.local_dist, device=local_rank, shard=(rank, world)))
    n_members = len(devices) if devices is not None else world      # devices working on ONE evaluation
    value, ms_per_step, ms_samples = sequential_rate(g, Xlist, args.steps, args.warmup, grad_cov, args.reps)
    # who really took part: ranks of the collective's backend, devices this process sees, units per shard
    if devices is not None:
        nm, on_host, devs_used, shard_units = g._ctx.group_info()
        group_info = {"members": nm, "devices": devs_used, "partials_meet": "pinned host memory (no peer access)" if on_host
                      else "fine-grained memory of device %d (peer stores)" % devs_used[0]}
        backend, rccl_ranks = None, None
    else:
        group_info = None
        nloc = g._ctx.num_units()[1]
        if world > 1:
            tl = torch.zeros(world, dtype=torch.int64, device=dev)
            tl[rank] = nloc
            dist.all_reduce(tl)
            shard_units = [int(v) for v in tl.tolist()]
            backend = dist.get_backend()
            rccl_ranks = dist.get_world_size() if backend == "nccl" else 0
        else:
            shard_units, backend, rccl_ranks = [nloc], None, None
    if args.no_stage_timing:
        if rank == 0:
            print(json.dumps({"diagnostic": "headline only (no per-kernel timing pass)", "value": value,
                              "unit": "evals/s", "ms_per_step": ms_per_step, "n_gpus": n_members,
                              "ms_per_step_samples": [round(v, 5) for v in ms_samples], "library": _capi.runtime_config()}))
        g.close()
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    # per-kernel durations: a SEPARATE pass of the same sequential loop with HIP events between the kernels of EVERY
    # evaluation (recorded on the stream the kernels are launched on); the headline above carries none of it
    g._ctx.set_timing(True, reset=True)
    t_ev0 = time.perf_counter()
    for k in range(STAGE_PASS_EVALS):
        g.update_X(Xlist[k % nX])
        g.llgrad(grad_X=True, grad_cov=grad_cov)
    barrier()
    ms_with_events = 1e3 * (time.perf_counter() - t_ev0) / STAGE_PASS_EVALS
    tm = g._ctx.get_timing()
    g._ctx.set_timing(False)
    cnt = tm.pop("count")
    stage = dict(tm)

    result = None
    if rank == 0:
        # algorithmic work of this rank's shard, averaged over the distinct X
        fl = {}
        for k in range(nX):
            owner = _capi.partition_units(sizes_all[k], args.yd, len(devices) if devices is not None else world)
            f_ = algorithmic_flops(sizes_all[k][owner == 0], args.yd)
            for a, b in f_.items():
                fl[a] = fl.get(a, 0.0) + b / nX
        total_all = float(np.mean([algorithmic_flops(s, args.yd)["total"] for s in sizes_all]))
        compute_stages = ["potrf", "solve", "at", "grad"]
        # fixed rule: the dominant kernel is the LONGEST stage of the per-kernel timing pass
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
        # (traffic: only when the committed profile was taken on exactly these native sources — otherwise null and why)
        try:
            tfile = next(f_ for f_ in TRAFFIC_FILES if os.path.exists(os.path.join(ROOT, f_)))
            tr = json.load(open(os.path.join(ROOT, tfile)))
            # (the Cholesky stage is two kernels side by side: their traffic adds up)
            # (whichever of a stage's kernels ran: the wide or the narrow At kernel; k_gx_finalize only where it is launched)
            keys = {"potrf": ("k_potrf_reg_gen", "k_potrf_reg8_gen", "k_potrf_reg2_gen"), "solve": ("k_solve_panel",), "at": ("k_at", "k_at_wide"),
                    "grad": ("k_mgrad", "k_gx_finalize"), "fill": ("k_fill",)}[dom]
            keys = tuple(k_ for k_ in keys if k_ in tr)
            if n_members == 1 and args.ntrain == 10000 and args.nblocks == 100 and args.local_dist < 1.0:
                # the committed counters always ride along; `traffic_sources_match` says whether they were taken on exactly
                # these native sources (scripts/gpu_round6_final.sh refuses to finish a round when they were not)
                roof["traffic"] = sum(tr[k_]["bytes_per_launch"] for k_ in keys)
                roof["traffic_sources_match"] = tr.get("source_hash") == source_hash()
                roof["traffic_source"] = "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; profiled sources %s, commit %s; these sources %s)" % (
                    tfile, tr.get("source_hash"), tr.get("commit", "?"), source_hash())
            else:
                roof["traffic_source"] = "null: the committed counters are of the single-device north-star run"
        except Exception as e:
            roof["traffic_source"] = "null: %r" % (e,)
        roof["avg_launch_ms"] = stage[dom]
        roof["algorithmic_per_launch"] = fl["fill_bytes"] if dom == "fill" else fl[dom]
        # the Cholesky and the gradient stage are within a few per cent of each other: the same figures for every compute
        # stage, so that the line reads the same whichever of them is the longer one in a given run
        roof["all_stages"] = {s_: {"ms": round(stage[s_], 5), "TFLOPs": round(fl[s_] / (stage[s_] * 1e-3) / 1e12, 3),
                                   "frac": round(fl[s_] / (stage[s_] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4)}
                              for s_ in compute_stages if stage[s_] > 0}
        wk = min(roof["all_stages"], key=lambda s_: roof["all_stages"][s_]["frac"])
        roof["worst"] = {"kernel": "k_" + wk, "frac": roof["all_stages"][wk]["frac"]}
        roof["mfma_f64_measured_peak"] = FP64_MFMA_MEASURED_TFLOPS
        kernels_ms = sum(stage[s] for s in stage)
        # (stage times are rank 0's / member 0's kernels — 1 / n_members of the units: every figure that divides the WHOLE
        # evaluation's work or wall time by them is defined only when one device does all of it)
        roof["whole_eval_TFLOPs_kernels"] = total_all / (kernels_ms * 1e-3) / 1e12 if n_members == 1 else None
        roof["whole_eval_frac_of_fp64_peak_kernels"] = (total_all / (kernels_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if n_members == 1 else None
        roof["stages_of"] = "the only device" if n_members == 1 else "shard 0 of %d (its share of the units, its kernels)" % n_members
        result = {
            "metric": "GPRF objective+gradient evals/sec, n=%d nblocks=%d yd=%d" % (ntrain, args.nblocks, args.yd),
            "value": value, "unit": "evals/s", "n_gpus": n_members, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "repetitions": len(ms_samples), "ms_per_step_samples": [round(v, 5) for v in ms_samples],
            "value_is": "steps / median wall time over the repetitions (each: exactly `steps` evaluations between barrier + synchronize)",
            "rccl_ranks": rccl_ranks, "collective_backend": backend, "devices_seen": torch.cuda.device_count(),
            "shard_units": shard_units,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "n=%d nblocks=%d yd=%d lscale=%g obs_std=%g local_dist=%g (%d unary + %d pair units) task=%s"
                                   % (ntrain, args.nblocks, args.yd, args.lscale, args.obs_std, args.local_dist,
                                      args.nblocks, len(nbrs), args.task),
                       "distinct_X": nX,
                       "step": "host X in -> update_X (device re-partition + table rebuild) -> llgrad -> host result out, "
                               "sequential, one synchronisation per evaluation",
                       "parallelism": ("units sharded over %d device(s) driven by ONE process; partial sums meet on device 0 (peer "
                                       "stores + one summing kernel)" % len(devices)) if devices is not None else
                                      "units sharded over %d rank(s), 1 all-reduce/eval" % world,
                       "host": ("process pinned to the GPU's NUMA node %d (%d cpus)" % (numa_node, numa_cpus)) if numa_node >= 0
                               else "CPU affinity left as found",
                       "group": group_info, "library": _capi.runtime_config(),
                       "strong_scaling_note": "the 442 units of this configuration are ONE round of workgroups on one GPU and its floor at any N is one "
                                              "16-tile unit's chain through four stages: c4_evals_per_s (n=80000, 4033 units) is the "
                                              "configuration whose work shards (DESIGN.md section 7)"},
            "roofline": roof,
            **({"note": "GPRF_BENCH_ONE_GPU=1 test run: all ranks / members time-share one GPU; not a measurement"} if one_gpu else {}),
            "stages_ms": {k2: round(v, 5) for k2, v in stage.items()},
            "kernels_ms_per_eval": round(kernels_ms, 5),
            "stage_timing": "separate pass behind the timed region: %d sequential evaluations with HIP events between the kernels "
                            "of every one (%.4f ms per evaluation with the events; the headline has none); 'gather' includes "
                            "the re-partition and table-build kernels.  With its timers on the library runs the stages LAUNCH-WIDE, one "
                            "kernel after the other: these are per-kernel durations (the roofline's).  The timed region itself "
                            "pipelines the two size classes' Cholesky -> substitution -> At -> gradient on two queues "
                            "(config.library class_depth; DESIGN.md section 4.7), so ms_per_step is SHORTER than the stages' sum; "
                            "'launch_wide' is the same loop with that pipelining off" % (cnt, ms_with_events),
            "host_gap_ms": None,      # (meaningful only where the stages run one after the other: launch_wide.host_gap_ms)
        }
        # the same timed loop with the by-class pipelining off (GPRF_DIAG solve_class=0): what the stage times above add up to
        if n_members == 1:
            prev = os.environ.get("GPRF_DIAG")
            os.environ["GPRF_DIAG"] = (prev + "," if prev else "") + "solve_class=0"
            try:
                lw_value, lw_ms, lw_samples = sequential_rate(g, Xlist, args.steps, args.warmup, grad_cov, 3)
            finally:
                if prev is None:
                    os.environ.pop("GPRF_DIAG", None)
                else:
                    os.environ["GPRF_DIAG"] = prev
            result["launch_wide"] = {"value": lw_value, "ms_per_step": lw_ms, "ms_per_step_samples": [round(v, 5) for v in lw_samples],
                                     "host_gap_ms": round(lw_ms - kernels_ms, 5),
                                     "note": "GPRF_DIAG=solve_class=0: every stage one launch over all units, the queues joined behind the "
                                             "Cholesky (rounds 1-5); bit-identical results (tests/test_gpu_variants.py)"}
            result["pipelining_gain"] = lw_ms / ms_per_step

    # ---------------- secondary figures
    if not args.only_north_star:
        # blocks fixed (no update_X): the llgrad-only rate of SURVEY 8d
        g.update_X(Xlist[0])
        g.llgrad(grad_X=True, grad_cov=grad_cov)
        barrier()
        t1 = time.perf_counter()
        for _ in range(50):
            g.llgrad(grad_X=True, grad_cov=grad_cov)
        barrier()
        if rank == 0:
            result["llgrad_only_evals_per_s"] = 50.0 / (time.perf_counter() - t1)
    if n_members == 1 and not args.only_north_star:
        # the kernels' own rate: the same evaluations enqueued back to back on one stream, X / result resident in HBM,
        # no host synchronisation inside the loop (one context per distinct X, its tables built beforehand)
        run_stream = torch.cuda.Stream(device=dev)
        evs = []
        for k in range(nX):
            gk = GPRF(Xlist[k], sd.SY, None, sd.cov, sd.noise_var, block_idxs=sd.reblock(Xlist[k]), neighbors=nbrs,
                      device=local_rank)
            gk._push_neighbors(nbrs)
            ev = gdist.DeviceEvaluator(gk)
            ev.set_X(Xlist[k])
            evs.append(ev)
        for k in range(20):
            evs[k % nX].enqueue(True, grad_cov, stream=run_stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(args.steps):
            evs[k % nX].enqueue(True, grad_cov, stream=run_stream)
        torch.cuda.synchronize()
        result["device_resident_evals_per_s"] = args.steps / (time.perf_counter() - t1)
        # K is generated inside k_potrf_reg on this configuration and k_fill does not run; the fill kernel's HBM
        # write rate is measured on the side by forcing the K pool back for a few evaluations
        # (gprf_debug_run with stop_after = 0: gather + the fill of EVERY unit, nothing else)
        e = evs[0]
        for _ in range(2):      # (the K pool's allocation and first touch are not the kernel's time)
            e.g._ctx.debug_run(Xlist[0], 0)
        torch.cuda.synchronize()
        e.g._ctx.set_timing(True, reset=True)
        for _ in range(8):
            e.g._ctx.debug_run(Xlist[0], 0)
        torch.cuda.synchronize()
        tmf = e.g._ctx.get_timing()
        e.g._ctx.set_timing(False, reset=True)
        f0 = algorithmic_flops(sizes_all[0], args.yd)
        # bytes k_fill_se really writes: the 64 x 64 blocks ti <= tj of every unit (diagonal blocks whole)
        written = 0.0
        for mp_ in (sizes_all[0] + 15) // 16 * 16:
            r0 = np.arange(0, mp_, 64)
            written += 8.0 * float(np.sum(np.minimum(64, mp_ - r0) * (mp_ - r0)))
        result["roofline"]["fill_kernel"] = {
            "GBps": f0["fill_bytes"] / (tmf["fill"] * 1e-3) / 1e9, "written_GBps": written / (tmf["fill"] * 1e-3) / 1e9,
            "ms": tmf["fill"], "counter_GBps": FILL_COUNTER_GBPS,
            "note": "k_fill_se timed through the library's fill-only debug run; by default k_potrf_reg generates K and it never "
                    "exists in HBM (stages_ms.fill is then two event records).  GBps = algorithmic 8 m^2 per unit; written_GBps = "
                    "the 64-row block rows from their diagonal block to the unit's edge, which is what it stores; counter_GBps = (WRITE_SIZE + FETCH_SIZE) / duration from the committed "
                    "rocprofv3 pass of this kernel (profiles/r05_fill_rocprof_summary.txt), null until taken"}
        for ev in evs:
            ev.g.close()
        # BASELINE configs[1]: no pairs
        gl = sd.build_gprf(local_dist=1.0, device=local_rank)
        result["local_gp_evals_per_s"] = sequential_rate(gl, Xlist, 100, 10, grad_cov, 3)[0]
        gl.close()
    g.close()

    # ---------------- the reference's coarse partitions of the SAME data (gprfopt_analyze.py:237-238): 9 blocks + 20 pairs
    # (units of ~1100 / ~2200 points) and ONE block, the full GP on all 10000 points — units beyond one workgroup, through
    # the blocked multi-launch path (k_big_*); gprf_results.tgz has the reference's own seconds per evaluation beside them
    if n_members == 1 and not args.only_north_star and args.ntrain == 10000:
        big = {}
        # (round 5: + the 25- and 49-block partitions — pairs of ~800 / ~450 points, whose Cholesky and substitution moved to the
        # blocked path / the 32-tile register kernel)
        for nbk, ld, tag, ref_s in ((49, 0.1, "49 blocks + 156 pairs", 13.3), (25, 0.1, "25 blocks + 72 pairs", 27.6),
                                    (9, 0.1, "9 blocks + 20 pairs", 85.33), (1, 1.0, "1 block (full GP)", 233.55)):
            sd.set_centers(grid_centers(nbk))
            gb = sd.build_gprf(local_dist=ld, device=local_rank)
            rate, ms, smp = sequential_rate(gb, Xlist[:3], 6 if nbk > 9 else (3 if nbk == 9 else 2), 1, grad_cov, 1)
            szb = gdist.unit_sizes(sd.block_idxs, sd.neighbors if ld < 1.0 else [])
            big[tag] = {"evals_per_s": rate, "ms_per_eval": ms, "largest_unit_points": int(szb.max()),
                        "algorithmic_TFLOPs": algorithmic_flops(szb, args.yd)["total"] * rate / 1e12,
                        "reference_published_s_per_eval": ref_s}
            gb.close()
        sd.set_centers(grid_centers(args.nblocks))
        result["big_units"] = big

    # ---------------- time to solution: the reference's whole optimisation run on the north-star data (gprfopt.py:377-432:
    # scipy L-BFGS-B, ftol 1e-6, maxiter 200, task x) — outside the timed region; how much of an evaluation's wall time is
    # the device and how much scipy's host loop around 20000 variables
    if n_members == 1 and rank == 0 and not args.only_north_star and args.ntrain == 10000 and args.task == "x":
        from gprf_amd.objective import do_optimization
        go = sd.build_gprf(local_dist=args.local_dist, device=local_rank)
        do_optimization(go, sd.X_obs, None, sd, maxsec=None, maxiter=3)      # (warm: workspace, instantiations)
        t0 = time.perf_counter()
        z_fin, ob = do_optimization(go, sd.X_obs, None, sd, maxsec=None, maxiter=200)
        wall = time.perf_counter() - t0
        n_ev = len(ob.trace)
        err = float(np.mean(np.sqrt(np.sum((z_fin.reshape(-1, 2) - sd.SX) ** 2, axis=1))))
        result["optimize_c3"] = {
            "evals": n_ev, "wall_s": wall, "ms_per_eval_wall": 1e3 * wall / n_ev, "gpu_ms_per_eval": result["ms_per_step"],
            "host_ms_per_eval": 1e3 * wall / n_ev - result["ms_per_step"],
            "first_objective": ob.trace[0][2], "final_objective": ob.trace[-1][2], "final_mean_location_error": err,
            "reference_published": {"evals": 89, "wall_s": 650.03, "first_objective": -6563678.10,
                                    "final_objective": 409688.60, "final_mean_location_error": 0.00363347,
                                    "run": "10000_10500_100_0.060000_0.020000_0.1000_50_l-bfgs-b_x_-1_0.0100_s0_gprf0"},
            "note": "scipy.optimize L-BFGS-B (ftol 1e-6, maxiter 200) driving gprf_objective; host = scipy's own loop + Python"}
        go.close()

    # ---------------- BASELINE configs[3] (n=80000, 841 blocks + 3192 pairs, task xcov), same sequential loop, every N
    if not args.only_north_star and not args.no_c4 and args.ntrain == 10000:
        sd4 = sample(80000, 0.02, 0.002, 800)
        g4 = (sd4.build_gprf(local_dist=0.5, devices=devices) if devices is not None else
              sd4.build_gprf(local_dist=0.5, device=local_rank, shard=(rank, world)))
        rng = np.random.RandomState(1)
        X4 = [np.ascontiguousarray(sd4.X_obs + 0.25 * sd4.obs_std * k * rng.randn(*sd4.X_obs.shape)) for k in range(3)]
        c4, c4ms, c4samples = sequential_rate(g4, X4, 30, 5, True, 3)
        if rank == 0:
            s4 = gdist.unit_sizes(sd4.block_idxs, sd4.neighbors)
            result["c4_evals_per_s"] = c4
            result["c4"] = {"workload": "n=80000 nblocks=800(841) yd=50 lscale=0.02 obs_std=0.002 local_dist=0.5 (841 unary + %d pair "
                                        "units) task=xcov, prior draw by dense fp64 Cholesky at N=80500 on the GPU" % len(sd4.neighbors),
                            "ms_per_eval": c4ms, "steps": 30, "distinct_X": 3, "ms_per_eval_samples": [round(v, 4) for v in c4samples],
                            "algorithmic_TFLOPs": algorithmic_flops(s4, 50)["total"] * c4 / 1e12}
        g4.close()
        del sd4

    # ---------------- BASELINE configs[4]'s shape on the stand-in catalogue (the ISC file is not distributed): great-circle /
    # depth distance, Matern-3/2, split-tree blocks of < 210 events, edge threshold 0.6, task xcov; same sequential loop
    # (update_X re-routes every event through the tree on the device)
    if not args.only_north_star and not args.no_c5 and args.ntrain == 10000:
        from gprf_amd import GPCov, seismic
        n5 = 20000
        X5 = seismic.synthetic_events(n5, seed=0)
        Y5 = np.random.RandomState(1).randn(n5, 50)
        blocks5, reblock5 = seismic.pdtree_cluster(X5, 210)
        kw5 = dict(devices=devices) if devices is not None else dict(device=local_rank, shard=(rank, world))
        g5 = GPRF(X5, Y5, reblock5, GPCov([1.0], [40.0, 40.0], "lld", "matern32"), 0.1, neighbor_threshold=0.6, **kw5)
        rng5 = np.random.RandomState(2)
        X5s = [np.ascontiguousarray(X5 + 1e-4 * k * rng5.randn(*X5.shape)) for k in range(3)]
        c5, c5ms, c5samples = sequential_rate(g5, X5s, 40, 5, True, 3)
        g5._ctx.set_timing(True, reset=True)
        for k in range(12):
            g5.update_X(X5s[k % 3])
            g5.llgrad(grad_X=True, grad_cov=True)
        barrier()
        st5 = g5._ctx.get_timing()
        g5._ctx.set_timing(False)
        st5.pop("count")
        if rank == 0:
            sz5 = gdist.unit_sizes(g5.block_idxs, g5.neighbors)
            fl5 = algorithmic_flops(sz5, 50)
            result["c5_evals_per_s"] = c5
            result["c5"] = {"workload": "STAND-IN catalogue (synthetic_events, n=%d), lld / matern32, split-tree blocks < 210 (%d blocks), "
                                        "threshold 0.6 (%d pairs, largest unit %d points), yd=50, task xcov" % (
                                            n5, len(g5.block_idxs), len(g5.neighbors), int(sz5.max())),
                            "ms_per_eval": c5ms, "steps": 40, "distinct_X": 3, "ms_per_eval_samples": [round(v, 4) for v in c5samples],
                            "stages_ms": {k2: round(v, 5) for k2, v in st5.items()},
                            "algorithmic_TFLOPs": fl5["total"] * c5 / 1e12,
                            "fill_GBps": (fl5["fill_bytes"] / (st5["fill"] * 1e-3) / 1e9) if (n_members == 1 and st5["fill"] > 0) else None}
        g5.close()
        del g5
        # the same at the catalogue's paper scale (about 1e5 events: leaves of 195 points, pairs of 390 = 25 tiles per edge —
        # the eight-wave Cholesky with its overflow tiles waiting in the U pool, the single-buffer forward substitution)
        n6 = 100000
        X6 = seismic.synthetic_events(n6, seed=0)
        Y6 = np.random.RandomState(1).randn(n6, 50)
        blocks6, reblock6 = seismic.pdtree_cluster(X6, 210)
        g6 = GPRF(X6, Y6, reblock6, GPCov([1.0], [40.0, 40.0], "lld", "matern32"), 0.1, neighbor_threshold=0.6, **kw5)
        rng6 = np.random.RandomState(3)
        X6s = [np.ascontiguousarray(X6 + 1e-4 * k * rng6.randn(*X6.shape)) for k in range(2)]
        c6, c6ms, c6samples = sequential_rate(g6, X6s, 10, 2, True, 3)
        g6._ctx.set_timing(True, reset=True)
        for k in range(6):
            g6.update_X(X6s[k % 2])
            g6.llgrad(grad_X=True, grad_cov=True)
        barrier()
        st6 = g6._ctx.get_timing()
        g6._ctx.set_timing(False)
        st6.pop("count")
        if rank == 0:
            sz6 = gdist.unit_sizes(g6.block_idxs, g6.neighbors)
            fl6 = algorithmic_flops(sz6, 50)
            result["c5_paper_scale"] = {"workload": "STAND-IN catalogue at paper scale (synthetic_events, n=%d), lld / matern32, split-tree blocks "
                                                    "< 210 (%d blocks), threshold 0.6 (%d pairs, largest unit %d points), yd=50, task xcov" % (
                                                        n6, len(g6.block_idxs), len(g6.neighbors), int(sz6.max())),
                                        "evals_per_s": c6, "ms_per_eval": c6ms, "steps": 10, "distinct_X": 2,
                                        "ms_per_eval_samples": [round(v, 4) for v in c6samples],
                                        "stages_ms": {k2: round(v, 5) for k2, v in st6.items()},
                                        "algorithmic_TFLOPs": fl6["total"] * c6 / 1e12}
        g6.close()
        del g6, X6, Y6, X6s

    # ---------------- parity figures of THIS run: a sample of the timed configuration's units, the device's per-unit
    # log-likelihood and gradient rows against the oracle's on the same inputs (the oracle is the checker here, nothing else).
    # Behind every timed leg, on a context of its own, and with ONE BLAS thread: a LAPACK thread pool left spinning on the
    # cores this process is pinned to would disturb whatever is timed after it
    if rank == 0 and n_members == 1 and not args.no_parity:
        from threadpoolctl import threadpool_limits
        sd.set_centers(grid_centers(args.nblocks))
        gp = sd.build_gprf(local_dist=args.local_dist, device=local_rank)
        with threadpool_limits(limits=1):
            result["parity"] = parity_sample(gp, sd, Xlist[0], nbrs, grad_cov)
        gp.close()

    if n_members == 1 and not args.only_north_star and not args.no_cpu_baseline:
        if affinity_before is not None:
            os.sched_setaffinity(0, affinity_before)      # the CPU baseline may use every core of the box
        result["cpu_baseline"] = cpu_baseline(sd, args.local_dist, args.cpu_seconds, grad_cov)
        result["speedup_vs_cpu_port"] = value / result["cpu_baseline"]["value"]
        if result["cpu_baseline"].get("pool_value"):
            result["speedup_vs_cpu_pool"] = value / result["cpu_baseline"]["pool_value"]

    if rank == 0:
        print(json.dumps(result))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
