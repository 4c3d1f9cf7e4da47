"""Compile-time invariants of the hand-scheduled Cholesky kernels, checked on the gfx950 ISA that hipcc emits
(no GPU needed).  k_potrf_reg keeps its accumulator tiles in explicitly numbered AGPRs behind volatile asm:
that is only sound while the compiler itself never touches an AGPR in that kernel and never spills."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gprf_amd", "csrc", "gprf_potrf.hip")      # the Cholesky kernels' own translation unit
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _compile(tmp_path_factory, src):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("isa") / (os.path.basename(src) + ".s"))
    # (the same source file and flags as the shipped object: gprf_amd/build.py)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                           "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0",       # as gprf_amd/build.py
                           "-o", out, src], stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    return _compile(tmp_path_factory, SRC)


@pytest.fixture(scope="module")
def isa_big(tmp_path_factory):
    return _compile(tmp_path_factory, os.path.join(ROOT, "gprf_amd", "csrc", "gprf_big.hip"))      # the blocked path's kernels


def _function(lines, mangled_part):
    start = [i for i, l in enumerate(lines) if mangled_part in l and l.startswith("_Z") and "; @" in l]
    assert len(start) == 1, "kernel %s not found exactly once" % mangled_part
    end = [i for i, l in enumerate(lines) if i > start[0] and ".Lfunc_end" in l][0]
    return lines[start[0]:end]


def _setting(lines, mangled_part, key):
    for l in lines:
        m = re.match(r"\s*\.set\s+(\S+)\.%s,\s*(\d+)" % key, l)
        if m and mangled_part in m.group(1):
            return int(m.group(2))
    raise AssertionError("no .set %s for %s" % (key, mangled_part))


# every instantiation of the register-resident Cholesky: K generated, two workgroups per CU (four waves of 256 registers, 20
# slots); K generated / K read, eight waves of 256 registers (the largest units); K read, waiting tiles in the U pool
@pytest.mark.parametrize("inst,slots", [("12k_potrf_reg2ILi4ELi20ELb1EE", 20), ("12k_potrf_reg8ILi20ELb1EE", 20),
                                        ("12k_potrf_reg8ILi20ELb0EE", 20), ("13k_potrf_reg8wILi20EE", 20)])
def test_potrf_reg_accumulators_are_untouched_by_the_compiler(isa, inst, slots):
    body = _function(isa, inst)
    inasm, outside, stubs = False, [], 0
    for l in body:
        if "#ASMSTART" in l:
            inasm = True
            continue
        if "#ASMEND" in l:
            inasm = False
            continue
        code = l.split(";")[0]
        if inasm:
            stubs += bool(re.search(r"\ba\[", code))
        elif re.search(r"\ba\[?\d", code):
            outside.append(l.strip())
    assert stubs > 9 * slots                # the tile stubs are there (every slot x set / get / MFMA)
    assert not outside, outside[:5]         # ... and nothing else names an AGPR
    assert not [l for l in body if "scratch_" in l]          # no spills in the step loop or anywhere else
    assert _setting(isa, inst, "num_agpr") == 8 * slots
    # the unified register file holds 512 per SIMD lane: two waves with 20 slots + 96 VGPRs each
    assert _setting(isa, inst, "num_vgpr") <= 96


def test_generic_potrf_does_not_spill(isa):
    body = _function(isa, "7k_potrfENS")
    assert not [l for l in body if "scratch_" in l]


def _longest_exposed_run(body):
    """scripts/isa_serial_loads.py's figure: the longest run of [vector loads, s_waitcnt vmcnt(0)] groups with no MFMA or
    barrier in between — memory round trips that nothing overlaps."""
    chain = best = 0
    pending = False
    for l in body:
        s = l.strip()
        if s.startswith(("global_load", "buffer_load", "flat_load")) and "lds" not in s.split()[0]:
            pending = True
        elif s.startswith("s_waitcnt") and "vmcnt(0)" in s:
            if pending:
                chain += 1
                best = max(best, chain)
                pending = False
        elif s.startswith(("v_mfma", "s_barrier")):
            chain = 0
    return best


def test_blocked_path_gemm_keeps_its_loads_in_flight(isa_big):
    """Round 5: k_big_gemm's operand fetch (a select around every load) and its epilogue (element-wise C -= acc) had compiled to
    load / s_waitcnt vmcnt(0) / load chains — 65 exposed round trips in a row, 39.9 ms instead of 33.6 for the 10000-point
    unit.  The branch-free fetch and the batched epilogue must stay that way: a handful of waits, and
    the chunk loop's prefetch outstanding across its barrier (a counted vmcnt, not 0)."""
    body = _function(isa_big, "10k_big_gemmENS")
    assert _longest_exposed_run(body) <= 8
    assert any(re.search(r"s_waitcnt vmcnt\((8|9|1\d)\)", l) for l in body)      # eight loads stay in flight behind the wait
    assert not [l for l in body if "scratch_" in l]
    assert _longest_exposed_run(_function(isa_big, "12k_big_updateENS")) <= 2
