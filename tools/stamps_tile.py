"""Diagnostic: per-phase cycle shares of the output-tiled grad_value kernels (library built with OCPG_HIPCC_FLAGS=-DEXP_STAMPS)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import runpy
import torch
from ocpg_amd import _lib
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
os.environ["ITERS"] = "5"
ns = runpy.run_path(os.path.join(os.path.dirname(__file__), "bench_msda_gv.py"))
torch.cuda.synchronize()
L.ocpg_debug_stamps_tile(buf, 1)
loc = ns["ring_loc"](0.0, 0.0).to("cuda")
gv = torch.zeros(ns["N"], ns["S"], 8, 32, device="cuda")
os.environ["OCPG_MSDA_TILE"] = "1"
n = 5
for _ in range(n):
    ns["run"](loc, ns["attn"], ns["go"], gv, 1)
torch.cuda.synchronize()
L.ocpg_debug_stamps_tile(buf, 0)
names = ["A setup", "A prefilter", "A exam+count+stage", "A scan", "A drop", "A accumulate", "A end barrier", "A store",
         "B stage go", "B far pass", "B exam+count", "B scan", "B drop", "B accumulate", "B end barrier", "B flush"]
for lo, hi in ((0, 8), (8, 16)):
    tot = sum(buf[i] for i in range(lo, hi)) or 1
    for i in range(lo, hi):
        if names[i] != "-":
            print(f"{names[i]:18s} {buf[i] / (2 * n) / 1e6:10.2f} Mcycles/launch  {100.0 * buf[i] / tot:5.1f} %")
