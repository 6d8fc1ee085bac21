"""Build the C part of the oracle (TEST INFRASTRUCTURE): gcc -> oracle/libocpg_oracle.so.

The reference's own native code (models/ops/src/**) is CUDA-only (needs <cuda.h>, THC atomics, nvcc;
setup.py:47 raises without CUDA) -> unbuildable here, so there is no ``oracle/_ref`` build for this
project; the reference is exercised through its Python path instead (tests/golden/ref_import.py).
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "msda_oracle.c")
LIB = os.path.join(HERE, "libocpg_oracle.so")


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    cmd = ["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
