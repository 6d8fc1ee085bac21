"""Diagnostic: per-phase cycle shares of the column scatter kernel (library built with -DEXP_STAMPS; OCPG_HIP_LIB points to it)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MSDA_MODES", "ring")
import torch
import runpy
from ocpg_amd import _lib
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
ns = runpy.run_path(os.path.join(os.path.dirname(__file__), "bench_msda.py"))     # warms up and times
torch.cuda.synchronize()
L.ocpg_debug_stamps(buf, 1)
value, sh, lsi, loc, attn = ns["inputs"](ns["S"], os.environ["MSDA_MODES"].split(",")[0])
sh._ocpg_host = ns["shapes"]
go = torch.randn(ns["N"], ns["S"], 256, device="cuda")
from ocpg_amd.models.ops.functions import ms_deform_attn_backward
n = 10
for _ in range(n):
    ms_deform_attn_backward(value, sh, lsi, loc, attn, go)
torch.cuda.synchronize()
L.ocpg_debug_stamps(buf, 0)
names = ["tile_setup", "prologue loads+stage", "bin", "scan", "drop", "accumulate+flush issue", "end barrier"]
tot = sum(buf[i] for i in range(7))
for i, nm in enumerate(names):
    print(f"{nm:26s} {buf[i] / n / 1e6:10.2f} Mcycles/launch  {100.0 * buf[i] / tot:5.1f} %")
