"""Diagnostic: HIP-event timing of the fused transformer-glue kernels at the encoder shapes of config #2 (N=10 frames)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocpg_amd.models.ops.functions import fused_ln_func as f
dev = torch.device("cuda:0")
R = 51000
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
# bias+relu+dropout on [R, 1024] bf16 (kernels only: call the C entry points)
from ocpg_amd._lib import lib
L = lib(); st = torch.cuda.current_stream().cuda_stream
a = torch.randn(R, 1024, device=dev).bfloat16(); bias = torch.randn(1024, device=dev).bfloat16(); h = torch.empty_like(a)
gh = torch.randn(R, 1024, device=dev).bfloat16(); ga = torch.empty_like(a); db = torch.empty(L.ocpg_bias_relu_dropout_bwd_slots(R, 1024, 1), 1024, device=dev)
t = timeit(lambda: L.ocpg_bias_relu_dropout_fwd(a.data_ptr(), bias.data_ptr(), R, 1024, 0.1, 1, 2, 1, h.data_ptr(), st))
print("brd_fwd  bf16 [51000,1024] p=0.1: %.1f us  %.2f TB/s" % (t, 2 * a.numel() * 2 / t / 1e6))
t = timeit(lambda: L.ocpg_bias_relu_dropout_bwd(gh.data_ptr(), h.data_ptr(), R, 1024, 0.1, 1, ga.data_ptr(), db.data_ptr(), st))
print("brd_bwd  bf16 [51000,1024]: %.1f us  %.2f TB/s" % (t, 3 * a.numel() * 2 / t / 1e6))
for xdt in (torch.float32, torch.bfloat16):
    x = torch.randn(R, 256, device=dev).to(xdt); res = torch.randn(R, 256, device=dev); g = torch.ones(256, device=dev); be = torch.zeros(256, device=dev)
    y = torch.empty(R, 256, device=dev); stats = torch.empty(2, R, device=dev); gy = torch.randn(R, 256, device=dev)
    gx = torch.empty_like(x); gres = torch.empty_like(res); dgb = torch.empty(L.ocpg_dropout_add_ln_bwd_slots(R), 2, 256, device=dev)
    dt = 0 if xdt == torch.float32 else 1; xb = x.element_size()
    t = timeit(lambda: L.ocpg_dropout_add_ln_fwd(x.data_ptr(), res.data_ptr(), g.data_ptr(), be.data_ptr(), R, 256, 1e-5, 0.1, 1, 2, dt, y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), st))
    print("dal_fwd x=%s [51000,256]: %.1f us  %.2f TB/s" % (str(xdt)[6:], t, R * 256 * (xb + 8) / t / 1e6))
    t = timeit(lambda: L.ocpg_dropout_add_ln_bwd(gy.data_ptr(), x.data_ptr(), res.data_ptr(), g.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), R, 256, 0.1, 1, 2, dt, gx.data_ptr(), gres.data_ptr(), dgb.data_ptr(), st))
    print("dal_bwd x=%s [51000,256]: %.1f us  %.2f TB/s" % (str(xdt)[6:], t, R * 256 * (2 * xb + 12) / t / 1e6))
