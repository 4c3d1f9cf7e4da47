"""bench.py's multi-rank code path end to end on a one-GPU box: two ranks time-share GPU 0 (GPRF_BENCH_ONE_GPU=1, gloo
collectives instead of RCCL).  Checks the contract of the JSON line, not its numbers."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_prints_one_contract_line():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, GPRF_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "14", "--warmup", "2",
           "--ntrain", "2000", "--nblocks", "16", "--yd", "8"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d
    assert d["n_gpus"] == 2 and d["steps"] == 14 and d["warmup"] == 2 and d["value"] > 0
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-9
    assert "not a measurement" in d["note"] and d["roofline"]["frac"] > 0
    # every compute stage's figures ride along, whichever stage is the longest in this run
    assert set(d["roofline"]["all_stages"]) <= {"potrf", "solve", "at", "grad"} and d["roofline"]["all_stages"]
    assert all(v["ms"] > 0 and v["frac"] > 0 for v in d["roofline"]["all_stages"].values())


def test_single_process_multi_device_bench_line():
    """bench.py --single-process --gpus 2: ONE process drives two members (both on GPU 0 here) through gprf_create_multi."""
    env = dict(os.environ, GPRF_BENCH_ONE_GPU="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--single-process", "--gpus", "2", "--steps", "10", "--warmup", "2",
           "--ntrain", "2000", "--nblocks", "16", "--yd", "8", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "ONE process" in d["config"]["parallelism"]
    assert d["roofline"]["worst"]["kernel"].startswith("k_") and 0 < d["roofline"]["worst"]["frac"] < 1


def test_plain_python_bench_gpus_2_launches_the_ranks_itself():
    """`python bench.py --gpus 2` with NO launcher — how a scaling driver that knows only the N = 1 command line would
    call it: bench.py starts the two-rank job itself as a child process (before touching the GPU), relays rank 0's line,
    adds the single-process figure from a second child, exits 0."""
    env = dict(os.environ, GPRF_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "2", "--reps", "3",
           "--ntrain", "2000", "--nblocks", "16", "--yd", "8", "--no-c4", "--no-c5", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["warmup"] == 2 and d["value"] > 0
    assert "torch.distributed.run" in d["launched_by"]
    # the record says who took part: gloo here (two ranks time-sharing one GPU), so no RCCL ranks are claimed
    assert d["collective_backend"] == "gloo" and d["rccl_ranks"] == 0 and d["devices_seen"] >= 1
    assert len(d["shard_units"]) == 2 and sum(d["shard_units"]) == 16 + len_pairs(16)
    assert d["repetitions"] == 3 and len(d["ms_per_step_samples"]) == 3
    assert abs(d["ms_per_step"] - sorted(d["ms_per_step_samples"])[1]) < 1e-4          # the median repetition
    sp = d["single_process"]
    assert sp["value"] and sp["value"] == d["single_process_evals_per_s"] and "ONE process" in sp["parallelism"]
    assert sp["group"]["members"] == 2


def test_eight_ranks_dry_run_on_one_gpu():
    """`python bench.py --gpus 8` as the scaling driver calls it, on a one-GPU box: eight ranks time-share GPU 0 over gloo
    (GPRF_BENCH_ONE_GPU=1).  Nothing N = 2 never touched may stop the first real 8-GPU run: eight shards, every unit on
    exactly one rank, one contract line, rc 0."""
    env = dict(os.environ, GPRF_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2", "--reps", "3",
           "--ntrain", "3000", "--nblocks", "36", "--yd", "8", "--no-c4", "--no-c5", "--no-cpu-baseline",
           "--no-single-process-leg"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 5 and d["value"] > 0
    assert d["collective_backend"] == "gloo" and d["rccl_ranks"] == 0
    assert len(d["shard_units"]) == 8 and sum(d["shard_units"]) == 36 + len_pairs(36) and min(d["shard_units"]) > 0


def len_pairs(nblocks):
    """pairs of the 8-neighbourhood on the g x g grid of block centres"""
    g = int(round(nblocks ** 0.5))
    return 2 * g * (g - 1) + 2 * (g - 1) * (g - 1)
