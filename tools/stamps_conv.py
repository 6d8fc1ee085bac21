"""In-kernel phase stamps of the halo-staged 3x3 convolution (build with OCPG_HIPCC_FLAGS=-DEXP_STAMPS): where a K step's cycles go."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OCPG_CONV3X3_HALO"] = "1"
import numpy as np
import torch
from ocpg_amd._lib import LIB_PATH, check, lib
dev = torch.device("cuda:0")
n, c, h, w = 10, 256, int(os.environ.get("CONV_H", "24")), int(os.environ.get("CONV_W", "40"))
x = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
wt = (torch.randn(c, 3, 3, c, device=dev) * 0.02).to(torch.bfloat16)
y = torch.empty(n, h, w, c, device=dev, dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
L = lib()
for _ in range(5):
    check(L.ocpg_conv3x3_mfma_fwd(x.data_ptr(), wt.data_ptr(), None, None, 0, n, h, w, c, c, 1, y.data_ptr(), st), "fwd")
torch.cuda.synchronize()
raw = ctypes.CDLL(LIB_PATH)
nblk = min(1024, n * ((h + 5) // 6) * ((w + 19) // 20))
buf = (ctypes.c_ulonglong * (1024 * 4 * 8))()
raw.ocpg_debug_conv_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert raw.ocpg_debug_conv_stamps(buf, 1024 * 4 * 8) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 4, 8)[:nblk].astype(np.float64)
steps = 36.0
names = ["fetch issue", "fragment reads + MFMA issue", "wait loads + park", "barrier", "whole kernel"]
print(f"{nblk} workgroups (grid y = 0), s_memtime ticks (100 MHz -> x 24 = shader cycles at 2.4 GHz) per K step, mean over waves:")
for q in range(4):
    print(f"  {names[q]:30s} {a[:, :, q].mean() / steps:8.1f} ticks/step")
print(f"  {names[4]:30s} {a[:, :, 4].mean():8.1f} ticks total = {a[:, :, 4].mean() / steps:.1f} per step")
