"""Build libocpg_hip.so (hipcc, gfx950 only) in-tree: ocpg_amd/lib/libocpg_hip.so.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the GPU box
with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(os.path.dirname(HERE), "lib")
LIB = os.path.join(LIBDIR, "libocpg_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-std=c++17", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]
# gemm.hip plans GEMMs through hipBLASLt (soname libhipblaslt.so.1: inside a torch process the copy torch already loaded)
LINK = ["-L/opt/rocm/lib", "-lhipblaslt"]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sources()
    deps = srcs + glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(HERE, "..", "..", "include", "*.h"))
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in deps):
        return LIB
    cmd = [HIPCC] + FLAGS + ["-o", LIB] + srcs + LINK
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
