"""Diagnostic: time + check ocpg_dynmask_fwd_f32 at the config-#2 shape (all 4 decoder layers in one launch).
usage: bench_dynmask.py [lib.so ...]   (default: the in-tree library)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda:0")
BT, Q, C, H, W, S = 10, 20, 256, 48, 80, 8
torch.manual_seed(0)
feats = torch.randn(BT, C, H, W, device=dev)
NP = (C + 2) * 16 + 256 + 32
params = torch.randn(BT * Q, NP, device=dev) * 0.1
refpix = torch.rand(BT * Q, 2, device=dev) * torch.tensor([W * S, H * S], device=dev)

def reference():
    w0 = params[:, :(C + 2) * 16].reshape(BT, Q * 16, C + 2)
    x = torch.bmm(w0[..., :C].double(), feats.view(BT, C, H * W).double()).view(BT * Q, 16, H, W)
    xs = (torch.arange(W, device=dev) * S + S // 2).double(); ys = (torch.arange(H, device=dev) * S + S // 2).double()
    wx = w0[..., C].reshape(BT * Q, 16).double(); wy = w0[..., C + 1].reshape(BT * Q, 16).double()
    relx = refpix[:, 0:1].double() - xs; rely = refpix[:, 1:2].double() - ys
    x = x + wx[:, :, None, None] * relx[:, None, None, :] + wy[:, :, None, None] * rely[:, None, :, None]
    b0 = params[:, -32:-16].double(); b1 = params[:, -16:].double()
    pre = x + b0[:, :, None, None]
    w1 = params[:, (C + 2) * 16:(C + 2) * 16 + 256].reshape(BT * Q, 16, 16).double()
    return torch.bmm(w1, pre.clamp(min=0).flatten(2)).view(BT * Q, 16, H, W) + b1[:, :, None, None], pre

ref_out, ref_pre = reference()
libs = sys.argv[1:] or [os.path.join(ROOT, "ocpg_amd", "lib", "libocpg_hip.so")]
for path in libs:
    L = ctypes.CDLL(path)
    f = L.ocpg_dynmask_fwd_f32
    f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 6 + [ctypes.c_void_p] * 3
    out = torch.empty(BT * Q, 16, H, W, device=dev); pre = torch.empty_like(out)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: f(feats.data_ptr(), params.data_ptr(), refpix.data_ptr(), BT, Q, C, H, W, S, out.data_ptr(), pre.data_ptr(), st)
    assert call() == 0
    torch.cuda.synchronize()
    e_out = (out.double() - ref_out).abs().max().item(); e_pre = (pre.double() - ref_pre).abs().max().item()
    for _ in range(5): call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): call()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    flops = BT * Q * H * W * (2 * 16 * (C + 2) + 2 * 256)
    print("%-28s %8.1f us  %6.1f TFLOP/s  max|err| out %.2e pre %.2e (|out| max %.1f)" % (os.path.basename(path), us, flops / us / 1e6, e_out, e_pre, ref_out.abs().max().item()))
