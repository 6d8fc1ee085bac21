"""Diagnostic (r4): Video-Swin relative-position bias (both layouts) -- csrc/layernorm.hip relpos_bias vs the gather + copies chain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ocpg_amd.models.video_swin_transformer as vs
dev = torch.device("cuda:0")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for dim, heads in ((128, 4), (256, 8), (512, 16), (1024, 32)):
    wa = vs.WindowAttention3D(dim, (8, 7, 7), heads, qkv_bias=True).to(dev)
    n = 392
    g = torch.randn(heads, n, n, device=dev).transpose(1, 2)          # what the attention backward hands back
    res = []
    for on in (True, False):
        vs._RELPOS_KERNEL = on

        def run():
            wa.relative_position_bias_table.grad = None
            b = wa.relative_position_bias(n)
            bt = wa.__dict__.pop("_bias_t", None)
            bc = b.float().contiguous()
            if bt is None:
                bt = bc.transpose(1, 2).contiguous()
            bc.backward(g)
        res.append(timeit(run))
    print("heads %2d: kernel %6.1f us, gather + copies %6.1f us (forward in both layouts + backward)" % (heads, res[0], res[1]))
