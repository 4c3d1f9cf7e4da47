"""Build libgprf_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# GPRF_LIB: diagnostic builds (ablation / profile variants) live next to the product library under another name
LIB = os.environ.get("GPRF_LIB") or os.path.join(HERE, "libgprf_hip.so")
SOURCES = ["gprf_kernels.hip", "gprf_capi.hip"]
HEADERS = ["gprf_kernels.h", os.path.join("..", "..", "include", "gprf_hip.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


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
    objs = []
    for f in SOURCES:
        src = os.path.join(CSRC, f)
        obj = os.path.join(objdir, "%s.%s.o" % (os.path.splitext(f)[0], tag))
        # (>=: a header touched within the same clock tick as the object counts as newer)
        if force or not (os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t)):
            cmd = [hipcc] + flags + ["-c", "-o", obj + ".tmp.%d" % os.getpid(), src]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
                print(" ".join(cmd))
            try:
                subprocess.check_call(cmd)
                os.replace(obj + ".tmp.%d" % os.getpid(), obj)
            finally:
                if os.path.exists(obj + ".tmp.%d" % os.getpid()):
                    os.remove(obj + ".tmp.%d" % os.getpid())
        objs.append(obj)
    tmp = "%s.tmp.%d" % (LIB, os.getpid())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-pthread",
           "-Wl,--no-undefined",      # a symbol one source file declares and the other forgot to define fails HERE
           "-o", tmp] + objs
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == "__main__":
    import sys
    # `python -m gprf_amd.build`: build if stale; `--force`: recompile every source whatever the object cache says
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
