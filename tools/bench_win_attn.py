"""Micro-benchmark: window attention kernels at the Swin-T stage shapes of config #4 (2 clips x 5 frames of 384x640, bf16) and the
Swin-B stage-1 shape of config #5 (fp16, N = 392); OCPG_WIN_ATTN_MFMA=0/1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ocpg_amd.models.ops.functions.win_attn_func import window_attention
dev = torch.device("cuda:0")
def run(name, bw, n, h, dtype, iters=10):
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(bw, n, 3, h, 32, device=dev, generator=g).to(dtype).requires_grad_(True)
    bias = torch.randn(h, n, n, device=dev, generator=g).requires_grad_(True)
    region = torch.randint(0, 3, (max(1, bw // 2), n), device=dev, generator=g).int()
    nw = region.shape[0]
    go = torch.randn(bw, n, h * 32, device=dev, generator=g).to(dtype)
    out = {}
    for mode in os.environ.get("WIN_ATTN_MODES", "0,1").split(","):
        os.environ["OCPG_WIN_ATTN_MFMA"] = mode
        for _ in range(3):
            o = window_attention(qkv, bias, region, 32 ** -0.5, nw); o.backward(go)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for _ in range(iters):
            e[0].record(); o = window_attention(qkv, bias, region, 32 ** -0.5, nw); e[1].record(); o.backward(go); e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        out[mode] = (tf / iters * 1e3, tb / iters * 1e3)
    flops = 4.0 * bw * h * n * n * 32
    out.setdefault("0", (float("nan"), float("nan")))
    out.setdefault("1", (float("nan"), float("nan")))
    print(f"{name:28s} bw={bw:5d} N={n} H={h:2d}: fwd valu {out['0'][0]:7.1f} us  mfma {out['1'][0]:7.1f} us ({flops / out['1'][0] / 1e6:6.1f} TFLOP/s) | "
          f"bwd valu {out['0'][1]:7.1f} us  mfma {out['1'][1]:7.1f} us", flush=True)
SHAPES = [("swin-t stage1 (bf16)", 644, 245, 3, torch.bfloat16), ("swin-t stage2 (bf16)", 168, 245, 6, torch.bfloat16),
          ("swin-t stage3 (bf16)", 48, 245, 12, torch.bfloat16), ("swin-t stage4 (bf16)", 12, 245, 24, torch.bfloat16),
          ("swin-b stage1 (fp16, N=392)", 270, 392, 4, torch.float16)]
only = os.environ.get("WIN_ATTN_ONLY")          # substring of the shape's name (PMC passes: one shape, one mode)
for sh in SHAPES:
    if not only or only in sh[0]:
        run(*sh)
