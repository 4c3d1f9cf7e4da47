"""A/B of how a host-in / host-out evaluation moves its data and learns that it is done (GPRF_IO_MODE), on an idle host and
under a deliberately busy one:
    0  zero-copy: kernels read X from / write the result to pinned host memory; completion = a polled word (k_done)
    1  copies:    X by hipMemcpyAsync H2D, result in HBM + hipMemcpyAsync D2H, hipStreamSynchronize
    2  mixed:     X zero-copy in, result in HBM + hipMemcpyAsync D2H, completion = a polled word behind the copy
Every cell = one fresh process running bench.py's headline loop (REPS repetitions of STEPS evaluations, no instrumentation);
printed: the median ms per evaluation, min .. max over the repetitions.  The load = LOAD processes streaming memory
(np.copyto over 128 MB) pinned to the GPU's NUMA node — the socket the bench process itself is pinned to.
Usage (GPU box):  python3 scripts/gpu_io_mode_ab.py [launches per cell]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LAUNCHES = int(sys.argv[1]) if len(sys.argv) > 1 else 3
STEPS, REPS = 200, 7
BUSY = "import numpy as np\na = np.ones(1 << 24); b = a.copy()\nwhile True:\n    np.copyto(b, a)\n"


def node_cpus():
    """cpus of the first GPU's NUMA node, from sysfs (this process never touches the GPU)"""
    import glob
    from gprf_amd import numa
    for path in sorted(glob.glob("/sys/class/drm/card*/device/numa_node")):
        try:
            node = int(open(path).read().strip())
            if node >= 0:
                return sorted(numa._node_cpus(node) & os.sched_getaffinity(0))
        except (OSError, ValueError):
            pass
    return sorted(os.sched_getaffinity(0))


def run_cell(mode):
    env = dict(os.environ, GPRF_IO_MODE=str(mode))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--only-north-star", "--no-stage-timing", "--steps", str(STEPS),
           "--warmup", "20", "--reps", str(REPS)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    for line in reversed(r.stdout.splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise RuntimeError(r.stderr[-1500:])


def main():
    cpus = node_cpus()
    nload = min(len(cpus), int(os.environ.get("LOAD", "48")))
    print("GPU 0's NUMA node: %d cpus; load = %d memory-streaming processes on them" % (len(cpus), nload))
    for busy in (False, True):
        procs = []
        if busy:
            for k in range(nload):
                p = subprocess.Popen([sys.executable, "-c", BUSY])
                try:
                    os.sched_setaffinity(p.pid, {cpus[k % len(cpus)]})
                except OSError:
                    pass
                procs.append(p)
            time.sleep(3.0)
        try:
            for mode in (0, 1, 2):
                meds, lo, hi = [], 1e9, 0.0
                for _ in range(LAUNCHES):
                    d = run_cell(mode)
                    meds.append(d["ms_per_step"])
                    lo, hi = min(lo, min(d["ms_per_step_samples"])), max(hi, max(d["ms_per_step_samples"]))
                print("host %-5s io_mode %d (%s): medians of %d launches %s  | repetitions min %.4f max %.4f  | max/min of medians %.3f"
                      % ("BUSY" if busy else "idle", mode, d["library"].get("io_mode"), LAUNCHES, " ".join("%.4f" % m for m in meds), lo, hi,
                         max(meds) / min(meds)), flush=True)
        finally:
            for p in procs:
                p.kill()
            for p in procs:
                p.wait()


if __name__ == "__main__":
    main()
