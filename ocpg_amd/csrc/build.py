"""Build libocpg_hip.so (hipcc, gfx950 only) in-tree: ocpg_amd/lib/libocpg_hip.so.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the GPU box
with the repo snapshot (it is git-ignored, not gpurun-ignored).  Each .hip is compiled to its own object
(ocpg_amd/lib/obj/, in parallel, only when it or a header changed) and the objects are linked into the library.
"""
import glob
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(os.path.dirname(HERE), "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libocpg_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["-O3", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# gemm.hip plans GEMMs through hipBLASLt (soname libhipblaslt.so.1: inside a torch process the copy torch already loaded)
LINK = ["-L/opt/rocm/lib", "-lhipblaslt"]
EXTRA = os.environ.get("OCPG_HIPCC_FLAGS", "").split()     # experiment switches (-DEXP_...)


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def _headers():
    return sorted(glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(HERE, "..", "..", "include", "*.h")))


def _stamp(src, hdr_digest):
    h = hashlib.sha1()
    h.update(open(src, "rb").read())
    h.update(hdr_digest)
    h.update(" ".join(CFLAGS + EXTRA).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sources()
    hd = hashlib.sha1()
    for h in _headers():
        hd.update(open(h, "rb").read())
    hd = hd.digest()
    todo, objs = [], []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        st = _stamp(s, hd)
        try:
            fresh = os.path.exists(o) and open(o + ".stamp").read() == st
        except OSError:
            fresh = False
        if force or not fresh:
            todo.append((s, o, st))

    def one(job):
        s, o, st = job
        cmd = [HIPCC] + CFLAGS + EXTRA + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(o + ".stamp", "w") as f:
            f.write(st)

    if todo:
        with ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            list(ex.map(one, todo))
    # the link has its own stamp: the sorted object list with each object's stamp + the link line.  A renamed / deleted source, a
    # link that failed after the objects were written, or a library older than its objects all relink.
    for stale in glob.glob(os.path.join(OBJDIR, "*.o")):
        if stale not in objs:                   # object of a source that no longer exists
            os.remove(stale)
            if os.path.exists(stale + ".stamp"):
                os.remove(stale + ".stamp")
    link_cmd = [HIPCC, "--offload-arch=gfx950", "-fPIC", "-shared"] + EXTRA + ["-o", LIB] + objs + LINK
    lh = hashlib.sha1(" ".join(link_cmd).encode())
    for o in objs:
        lh.update(open(o + ".stamp").read().encode())
    lstamp, lfile = lh.hexdigest(), LIB + ".stamp"
    try:
        linked = (os.path.exists(LIB) and open(lfile).read() == lstamp
                  and os.path.getmtime(LIB) >= max(os.path.getmtime(o) for o in objs))
    except OSError:
        linked = False
    if force or todo or not linked:
        if verbose:
            print(" ".join(link_cmd), flush=True)
        if os.path.exists(lfile):
            os.remove(lfile)
        subprocess.check_call(link_cmd)
        with open(lfile, "w") as f:
            f.write(lstamp)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
