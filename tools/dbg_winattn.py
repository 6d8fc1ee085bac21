import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ocpg_amd.models.ops.functions.win_attn_func import window_attention
dev = torch.device("cuda:0")
torch.manual_seed(0)
def ref(qkv, bias, region, scale):
    bw, n, _, h, d = qkv.shape
    q, k, v = qkv.float().permute(2, 0, 3, 1, 4)
    s = (q * scale) @ k.transpose(-1, -2) + bias[None]
    if region is not None:
        nw = region.shape[0]
        m = torch.zeros(nw, n, n, device=qkv.device).masked_fill(region[:, None, :] != region[:, :, None], -100.0)
        s = (s.view(bw // nw, nw, h, n, n) + m[None, :, None]).view(bw, h, n, n)
    return (s.softmax(-1) @ v).transpose(1, 2).reshape(bw, n, h * d)
cases = [(8, 4, 245, 3), (644, 322, 245, 3), (168, 84, 245, 6), (48, 24, 245, 12), (12, 6, 245, 24), (40, 40, 392, 32)]
for dt in (torch.float32, torch.bfloat16, torch.float16):
    for (bw, nw, n, h) in cases:
        for use_region in (False, True):
            print("case", dt, bw, nw, n, h, use_region, flush=True)
            qkv = torch.randn(bw, n, 3, h, 32, device=dev).to(dt).requires_grad_(True)
            bias = (torch.randn(h, n, n, device=dev) * 0.1).requires_grad_(True)
            region = torch.randint(0, 4, (nw, n), device=dev, dtype=torch.int32) if use_region else None
            out = window_attention(qkv, bias, region, 32 ** -0.5, nw if use_region else 1)
            go = torch.randn_like(out)
            gq, gb = torch.autograd.grad((out.float() * go.float()).sum(), (qkv, bias))
            torch.cuda.synchronize()
            if bw <= 48:
                q2 = qkv.detach().clone().requires_grad_(True); b2 = bias.detach().clone().requires_grad_(True)
                o2 = ref(q2, b2, region, 32 ** -0.5)
                gq2, gb2 = torch.autograd.grad((o2 * go.float()).sum(), (q2, b2))
                print("   err out %.2e dqkv %.2e dbias %.2e" % ((out.float() - o2).abs().max(), (gq.float() - gq2.float()).abs().max(), (gb - gb2).abs().max()), flush=True)
print("done")
