"""Diagnostic: 1x1 conv + frozen BN (+skip) + ReLU: GEMM-epilogue path vs GEMM + bn_act kernel (GPU events), ResNet-101 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocpg_amd._lib import lib
L = lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (c, co, hw, use_skip) in ((256, 64, 96 * 160, False), (64, 256, 96 * 160, True), (512, 128, 48 * 80, False), (128, 512, 48 * 80, True),
                              (1024, 256, 24 * 40, False), (256, 1024, 24 * 40, True), (2048, 512, 12 * 20, False), (512, 2048, 12 * 20, True)):
    m = 10 * hw
    for dt, code in ((torch.bfloat16, 1), (torch.float32, 0)):
        x = torch.randn(m, c, device=dev).to(dt); w = (torch.randn(co, c, device=dev) * 0.05).to(dt)
        scale = torch.rand(co, device=dev) + 0.5; shift = torch.randn(co, device=dev) * 0.1
        skip = torch.randn(m, co, device=dev).to(dt) if use_skip else None
        y1 = torch.empty(m, co, device=dev, dtype=dt); y2 = torch.empty_like(y1)
        sp = None if skip is None else skip.data_ptr()
        rc = L.ocpg_gemm_bn_act(x.data_ptr(), w.data_ptr(), y1.data_ptr(), scale.data_ptr(), shift.data_ptr(), sp, 1, code, m, co, c, st)
        def two():
            L.ocpg_gemm(x.data_ptr(), w.data_ptr(), y2.data_ptr(), None, code, code, 0, 1, m, co, c, c, c, co, 1, 0, 0, 0, 1.0, 0.0, st)
            L.ocpg_bn_act_fwd(y2.data_ptr(), scale.data_ptr(), shift.data_ptr(), sp, y2.data_ptr(), m, co, 1, 1, code, st)
        two(); torch.cuda.synchronize()
        t2 = timeit(two)
        if rc == 0:
            err = (y1.float() - y2.float()).abs().max().item() / y2.float().abs().max().item()
            t1 = timeit(lambda: L.ocpg_gemm_bn_act(x.data_ptr(), w.data_ptr(), y1.data_ptr(), scale.data_ptr(), shift.data_ptr(), sp, 1, code, m, co, c, st))
            print("%4d->%4d M=%6d skip=%d %-8s epilogue %.1f us | gemm+bn_act %.1f us | rel diff %.2e" % (c, co, m, use_skip, str(dt)[6:], t1, t2, err))
        else:
            print("%4d->%4d M=%6d skip=%d %-8s epilogue rc=%d | gemm+bn_act %.1f us" % (c, co, m, use_skip, str(dt)[6:], rc, t2))
