"""Compile the oracle's C restatement (gcc, host only).  ORACLE — test infrastructure."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "treegp_kernels.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libtreegp_oracle.so")


def build(force=False):
    """gcc -O2 (no -ffast-math: the oracle must keep IEEE semantics) -> oracle/_build/libtreegp_oracle.so"""
    os.makedirs(OUT_DIR, exist_ok=True)
    if (not force) and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=gnu99", "-ffp-contract=off", "-o", LIB, SRC, "-lm"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
