"""Diagnostic: per-phase cycle shares of k_scatter_col3 (library built with -DEXP_STAMPS; OCPG_HIP_LIB points to it): wave 0 of every
workgroup, summed in registers, one flush per workgroup (slots 8..15 of the stamp table)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocpg_amd import _lib
from ocpg_amd._lib import lib, stream_ptr
sys.argv = [sys.argv[0]]
os.environ["GV_PATHS"] = "0"
os.environ.setdefault("GV_MODES", "ring")
os.environ["ITERS"] = "10"
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
import runpy
L.ocpg_debug_stamps(buf, 1)
runpy.run_path(os.path.join(os.path.dirname(__file__), "bench_msda_gv.py"))
torch.cuda.synchronize()
L.ocpg_debug_stamps(buf, 0)
names = ["setup+qg", "loads+stage+bin atomics+barrier", "direct path", "scan / task lists + barrier", "item writes + barrier", "wide tasks",
         "narrow tasks + flush issue"]
tot = sum(buf[8 + i] for i in range(7))
for i, nm in enumerate(names):
    print(f"{nm:36s} {100.0 * buf[8 + i] / max(tot, 1):5.1f} %   {buf[8 + i] / 11 / 2400 / 8 / 10:9.0f} ticks per workgroup")
