"""Build libgprf_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# GPRF_LIB: diagnostic builds (ablation / profile variants) live next to the product library under another name
LIB = os.environ.get("GPRF_LIB") or os.path.join(HERE, "libgprf_hip.so")
# one translation unit per stage (round 6: the 5000-line kernel file compiled 2 m 46 s as one unit; split, the files compile
# side by side and a kernel variant recompiles one of them)
# (slowest first: they start first)
SOURCES = ["gprf_solve_wide32.hip", "gprf_solve_wide.hip", "gprf_solve.hip", "gprf_solve_class.hip", "gprf_potrf.hip", "gprf_mgrad.hip", "gprf_big.hip",
           "gprf_fill.hip", "gprf_tables.hip", "gprf_capi.hip"]
HEADERS = ["gprf_kernels.h", "gprf_dev.h", "gprf_solve_panel.h", os.path.join("..", "..", "include", "gprf_hip.h")]


def source_hash():
    """sha256 over the native sources and headers (sorted by name): a committed rocprofv3 counter figure counts for THIS code
    only if it was taken on the same sources — the GPU box has no git to ask (bench.py, scripts/prof_summary.py)"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        h.update(os.path.basename(f).encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _flag_tag():
    """what LIB was built WITH, besides its sources: the define / flag set (GPRF_BUILD_DEFS).  Stored next to the library
    (LIB + ".tag"), so that `GPRF_BUILD_DEFS=-DGPRF_PROFILE python gprf_amd/build.py` rebuilds a library that is up to date by
    its mtimes but was compiled with other defines — and the next plain build restores the product library."""
    import hashlib
    return hashlib.sha256(" ".join(_flags()).encode()).hexdigest()[:16]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    try:
        with open(LIB + ".tag") as f:
            return f.read().strip() != _flag_tag()
    except OSError:
        return False        # (a library without a tag file: built before round 6, or shipped alone — trusted by its mtimes)


def have_compiler():
    return os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))


def build(force=False, verbose=False):
    """Compile to a temporary file next to LIB and rename it into place, under an exclusive lock: with one process per
    GPU every rank may find the library stale at import — one of them builds, the others wait and find it fresh; nobody
    ever maps a half-written file."""
    if not force and not _stale():
        return LIB
    import fcntl
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():      # somebody else built it while this process waited
                return LIB
            return _build_locked(verbose, force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _flags():
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-pthread",
             # k_potrf_reg keeps tiles in explicitly numbered AGPRs behind inline asm: the compiler must never park a
             # spilled VGPR in an AGPR of its own choosing (tests/test_isa_invariants.py checks the ISA)
             "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"]
    # diagnostic builds: GPRF_BUILD_DEFS="-DGPRF_PROFILE" compiles the in-kernel cycle stamps in
    return os.environ.get("GPRF_BUILD_DEFS", "").split() + flags


def _compiler_id(hipcc):
    """what the object cache is keyed on besides the flags: another hipcc / ROCm must not reuse old objects"""
    try:
        return subprocess.run([hipcc, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60).stdout.decode()
    except (OSError, subprocess.SubprocessError):
        return "unknown"


def _build_locked(verbose, force=False):
    """One object per source (kept under csrc/_obj, named by a hash of the flags, the library name and the compiler's
    version: a change to the host layer alone does not recompile the kernels' two minutes), then one link.  ``force``
    recompiles everything; objects of other flag sets (diagnostic builds) older than a week are removed."""
    import hashlib
    import time
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = _flags()
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    tag = hashlib.sha256((" ".join(flags) + os.path.basename(LIB) + _compiler_id(hipcc)).encode()).hexdigest()[:12]
    for old in os.listdir(objdir):
        path = os.path.join(objdir, old)
        if old.endswith(".o") and ("." + tag + ".") not in old and time.time() - os.path.getmtime(path) > 7 * 86400:
            os.remove(path)
    objs, jobs = [], []
    for f in SOURCES:
        src = os.path.join(CSRC, f)
        obj = os.path.join(objdir, "%s.%s.o" % (os.path.splitext(f)[0], tag))
        # (strictly newer: a header touched within the same clock tick as the object counts as changed)
        if force or not (os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t)):
            jobs.append((src, obj))
        objs.append(obj)

    def compile_one(job):
        src, obj = job
        tmp_o = obj + ".tmp.%d" % os.getpid()
        cmd = [hipcc] + flags + ["-c", "-o", tmp_o, src]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd)
            os.replace(tmp_o, obj)
        finally:
            if os.path.exists(tmp_o):
                os.remove(tmp_o)

    if jobs:
        # the files side by side (hipcc is one process per file; GPRF_BUILD_JOBS bounds them, default: the cores, at most 8)
        from concurrent.futures import ThreadPoolExecutor
        nj = max(1, min(len(jobs), int(os.environ.get("GPRF_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 1)))
        with ThreadPoolExecutor(nj) as ex:
            list(ex.map(compile_one, jobs))
    tmp = "%s.tmp.%d" % (LIB, os.getpid())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-pthread",
           "-Wl,--no-undefined",      # a symbol one source file declares and the other forgot to define fails HERE
           "-o", tmp] + objs
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
        with open(LIB + ".tag", "w") as f:
            f.write(_flag_tag() + "\n")
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == "__main__":
    import sys
    # `python -m gprf_amd.build`: build if stale; `--force`: recompile every source whatever the object cache says
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
