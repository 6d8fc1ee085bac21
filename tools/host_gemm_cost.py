"""Diagnostic: host cost per call of small GPU ops (GPU not the bottleneck): mm vs elementwise vs ctypes kernel."""
import time, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
def host(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6, (time.perf_counter() - t0) / n * 1e6
for dt in (torch.float32, torch.bfloat16):
    a = torch.randn(256, 256, device=dev, dtype=dt); b = torch.randn(256, 256, device=dev, dtype=dt); c = torch.empty(256, 256, device=dev, dtype=dt)
    bias = torch.randn(256, device=dev, dtype=dt)
    a3 = torch.randn(8, 256, 256, device=dev, dtype=dt)
    for name, fn in (("add", lambda: torch.add(a, b)), ("mm", lambda: torch.mm(a, b)), ("mm out=", lambda: torch.mm(a, b, out=c)),
                     ("mm a.t()", lambda: torch.mm(a.t(), b)), ("addmm", lambda: torch.addmm(bias, a, b)), ("bmm", lambda: torch.bmm(a3, a3)),
                     ("linear", lambda: torch.nn.functional.linear(a, b, bias))):
        h, w = host(fn)
        print("%-10s %-10s host %.1f us/call (wall incl. drain %.1f)" % (str(dt)[6:], name, h, w))
from ocpg_amd.models.ops.functions.gemm_func import gemm, gemm_tn_split
for dt in (torch.float32, torch.bfloat16):
    a = torch.randn(256, 256, device=dev, dtype=dt); b = torch.randn(256, 256, device=dev, dtype=dt); bias = torch.randn(256, device=dev, dtype=dt)
    for ta in (False, True):
        for tb in (False, True):
            ref = (a.t() if ta else a).float() @ (b.t() if tb else b).float() + bias.float()
            err = (gemm(a, b, ta, tb, bias).float() - ref).abs().max().item() / ref.abs().max().item()
            assert err < (1e-5 if dt == torch.float32 else 2e-2), (dt, ta, tb, err)
    big_a = torch.randn(9600, 128, device=dev, dtype=dt); big_b = torch.randn(9600, 64, device=dev, dtype=dt)
    ref = big_a.float().t() @ big_b.float()
    err = (gemm_tn_split(big_a, big_b, 6).float() - ref).abs().max().item() / ref.abs().max().item()
    assert err < (1e-5 if dt == torch.float32 else 2e-2), err
    for name, fn in (("ocpg gemm", lambda: gemm(a, b)), ("ocpg gemm bias tb", lambda: gemm(a, b, False, True, bias))):
        h, w = host(fn)
        print("%-10s %-18s host %.1f us/call (wall incl. drain %.1f)" % (str(dt)[6:], name, h, w))
