"""Debug: which MSO test makes test_fused_linear_bias_relu_dropout[8200-0.0-bf16] fail afterwards, and where the outputs differ."""
import sys
import pytest
import torch

sel = sys.argv[1]
if sel != "none":
    pytest.main(["tests/test_mso_gpu.py", "-q", "-m", "gpu", "-k", sel, "-p", "no:cacheprovider"])
from ocpg_amd.models.ops.functions import fused_ln_func as f
from ocpg_amd import _lib
dev = torch.device("cuda", 0)
torch.manual_seed(0)
rows, k, c, dtype = 8200, 64, 1032, torch.bfloat16
x = torch.randn(rows, k, device=dev).to(dtype)
w = (torch.randn(c, k, device=dev) * 0.2).to(dtype)
b = torch.randn(c, device=dev).to(dtype)
h0 = f.LinearBiasReluDropout.apply(x, w, b, 0.0, (99, 7), 5).float()
h1 = torch.relu(torch.nn.functional.linear(x, w, b)).float()
h2 = torch.relu(x.float() @ w.float().t() + b.float())
d = (h0 - h1).abs()
print(sel, "fused vs torch-bf16", (h0 - h1).norm().item() / h1.norm().item(), "| fused vs fp32", (h0 - h2).norm().item() / h2.norm().item(),
      "| torch-bf16 vs fp32", (h1 - h2).norm().item() / h2.norm().item())
bad = d > 0.05 * h1.abs().max()
print("  bad elements", int(bad.sum()), "rows with bad", int(bad.any(1).sum()), "cols with bad", int(bad.any(0).sum()))
if bad.any():
    r = bad.any(1).nonzero().flatten()
    cc = bad.any(0).nonzero().flatten()
    print("  rows", r[:10].tolist(), "...", r[-5:].tolist(), " cols", cc[:10].tolist(), "...", cc[-5:].tolist())
L = _lib.lib()
print("  plans", L.ocpg_gemm_plans(), "tuned", L.ocpg_gemm_tuned(None), "rejected", L.ocpg_gemm_tune_rejected())
