"""Diagnostic: options for the two backward contractions of a 1x1 conv on channels-last bf16 maps (GPU-busy us)."""
import torch, torch.nn.functional as F
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
N = 10
shapes = [(64, 256, 96, 160), (256, 64, 96, 160), (128, 512, 48, 80), (512, 128, 48, 80), (256, 1024, 24, 40), (1024, 256, 24, 40),
          (512, 2048, 12, 20), (2048, 512, 12, 20), (2048, 256, 12, 20), (1024, 256, 24, 40), (512, 256, 48, 80)]
def busy(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n): fn()
        torch.cuda.synchronize()
    ks = [e for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA]
    return sum(e.device_time_total for e in ks) / n, sum(e.count for e in ks) // n
for cin, cout, h, w in shapes:
    M = N * h * w
    x = torch.randn(N, cin, h, w, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 1, 1, device=dev, dtype=torch.bfloat16) * 0.05)
    gy = torch.randn(N, cout, h, w, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x2, gy2, w2 = x.permute(0, 2, 3, 1).reshape(M, cin), gy.permute(0, 2, 3, 1).reshape(M, cout), wt.view(cout, cin)
    opts = {
        "dW mm(gy^T,x)": lambda: torch.mm(gy2.t(), x2),
        "dW mm(x^T,gy)^T": lambda: torch.mm(x2.t(), gy2).t(),
        "dW conv_bwd": lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1],
        "dX mm": lambda: torch.mm(gy2, w2),
        "dX conv_bwd": lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0],
    }
    for S in (8, 32):
        if M % S == 0:
            opts["dW bmm split %d" % S] = (lambda S=S: torch.bmm(gy2.view(S, M // S, cout).transpose(1, 2), x2.view(S, M // S, cin)).sum(0))
            opts["dW bmm split %d f32" % S] = (lambda S=S: torch.bmm(gy2.view(S, M // S, cout).transpose(1, 2), x2.view(S, M // S, cin), out_dtype=torch.float32).sum(0))
    ref = torch.mm(gy2.float().t(), x2.float())
    line = []
    for k, f in opts.items():
        try:
            r = f()
            if k.startswith("dW"):
                err = (r.float().reshape(cout, cin) - ref).abs().max().item() / ref.abs().max().item()
                assert err < 3e-2, (k, err)
            t, nk = busy(f)
            line.append("%s %.0f(%d)" % (k, t, nk))
        except Exception as e:
            line.append("%s ERR %s" % (k, str(e)[:40]))
    print("%4d->%4d @%3dx%3d | " % (cin, cout, h, w) + " | ".join(line), flush=True)
