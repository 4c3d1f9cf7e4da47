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
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = "%s.tmp.%d" % (LIB, os.getpid())
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-pthread",
           "-Wl,--no-undefined",      # a symbol one source file declares and the other forgot to define fails HERE
           # k_potrf_reg keeps tiles in explicitly numbered AGPRs behind inline asm: the compiler must never park a
           # spilled VGPR in an AGPR of its own choosing (tests/test_isa_invariants.py checks the ISA)
           "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0",
           "-o", tmp] + [os.path.join(CSRC, f) for f in SOURCES]
    # diagnostic builds: GPRF_BUILD_DEFS="-DGPRF_PROFILE" compiles the in-kernel cycle stamps in
    cmd[1:1] = os.environ.get("GPRF_BUILD_DEFS", "").split()
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose="-v" in sys.argv))
