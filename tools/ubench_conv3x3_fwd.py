"""conv3x3_mfma forward alone at the ResNet-101 layer3 shape (10 x 24 x 40, 256 -> 256, bf16): the target of PMC passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocpg_amd._lib import check, lib
dev = torch.device("cuda:0")
n, c, h, w = 10, int(os.environ.get("CONV_C", "256")), int(os.environ.get("CONV_H", "24")), int(os.environ.get("CONV_W", "40"))
x = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
wt = (torch.randn(c, 3, 3, c, device=dev) * 0.02).to(torch.bfloat16)
scale, shift = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
y = torch.empty(n, h, w, c, device=dev, dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
L = lib()
for _ in range(int(os.environ.get("CONV_ITERS", "20"))):
    check(L.ocpg_conv3x3_mfma_fwd(x.data_ptr(), wt.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1, n, h, w, c, c, 1, y.data_ptr(), st), "fwd")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    L.ocpg_conv3x3_mfma_fwd(x.data_ptr(), wt.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1, n, h, w, c, c, 1, y.data_ptr(), st)
e1.record()
torch.cuda.synchronize()
print("conv3x3_mfma fwd %d ch %dx%d: %.1f us" % (c, h, w, e0.elapsed_time(e1) / 50 * 1e3))
