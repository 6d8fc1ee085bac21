"""Debug: per-gradient relative error of the native MSO path under autocast vs the fp32 library path; unit convs with fp32 storage."""
import torch
import torch.nn.functional as F
from ocpg_amd.models import decoder
from ocpg_amd.models.decoder import MSO
from ocpg_amd.models.ops.functions.mso_func import conv3x3_n16
from ocpg_amd.util.misc import NestedTensor

dev = torch.device("cuda", 0)
torch.manual_seed(5)
bt, n, h, w = 2, 3, 12, 20
mso = MSO(mask_dim=16, img_dim=(32, 64)).to(dev)
f4 = torch.randn(bt, 32, 2 * h, 2 * w, device=dev).contiguous(memory_format=torch.channels_last)
f8 = torch.randn(bt, 64, h, w, device=dev).contiguous(memory_format=torch.channels_last)
pms = [torch.randn(bt, 16, h, w, device=dev) for _ in range(n)]
go = torch.randn(n * bt, 1, 2 * h, 2 * w, device=dev)


def run(native, amp_dt):
    decoder.NATIVE = native
    mso.zero_grad(set_to_none=True)
    a4, a8 = f4.clone().requires_grad_(True), f8.clone().requires_grad_(True)
    ps = [p.clone().requires_grad_(True) for p in pms]
    feats = [NestedTensor(a4, None), NestedTensor(a8, None)]
    with torch.autocast("cuda", dtype=amp_dt, enabled=amp_dt is not None):
        out = mso.forward_multi(ps, feats, stacked=True)
    (out.float() * go).sum().backward()
    grads = {k: v.grad.clone() for k, v in mso.named_parameters()}
    grads.update(f4=a4.grad.clone(), f8=a8.grad.clone(), **{f"pm{i}": p.grad.clone() for i, p in enumerate(ps)})
    return out.float().detach(), grads


ref_out, ref_g = run(False, None)
for amp in (None, torch.bfloat16, torch.float16):
    for native in (True, False):
        out, g = run(native, amp)
        print("amp", amp, "native", native, "out", ((out - ref_out).abs().max() / ref_out.abs().max()).item())
        for k, want in ref_g.items():
            print("    %-22s %.2e" % (k, ((g[k].float() - want).abs().max() / want.abs().max()).item()))

# unit: fp32 storage, 16-bit compute, C = 32 / 64
for c in (32, 64):
    for dt, code in ((torch.bfloat16, 1), (torch.float16, 2)):
        x = torch.randn(2, 12, 20, c, device=dev).requires_grad_(True)
        wt = (torch.randn(16, 9, c, device=dev) / 10).requires_grad_(True)
        gg = torch.randn(2, 12, 20, 16, device=dev).to(dt).float()
        out = conv3x3_n16(x, wt, None, None, None, True, code)
        out.backward(gg)
        xr = F.relu(x.detach()).to(dt).double()
        wr = wt.detach().to(dt).double().requires_grad_(True)
        ref = F.conv2d(xr.permute(0, 3, 1, 2), wr.view(16, 3, 3, c).permute(0, 3, 1, 2), None, padding=1).permute(0, 2, 3, 1)
        ref.backward(gg.double())
        print("unit", c, dt, ((out.double() - ref).abs().max() / ref.abs().max()).item(), ((wt.grad.double() - wr.grad).abs().max() / wr.grad.abs().max()).item())
