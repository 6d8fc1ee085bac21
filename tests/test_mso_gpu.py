"""csrc/mso.hip (MSO's convolutions, weight gradients and x2 resize on channels-last maps) against plain PyTorch.

The checker is torch's own conv2d / interpolate evaluated in float64 on the CPU (exact to ~1e-15), with the operands rounded to the
compute type first where the kernel rounds them (bf16 / fp16 operands, fp32 accumulation): the comparison then only sees fp32
summation-order noise, so the bound is tight for every compute type.  Reference semantics: models/decoder.py:22-46.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CODES = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


def _round(t, dt):
    return t.to(dt).double()


def _ref_conv(x, w, bias, addend, residual, relu_in, dt):
    """x [NB,H,W,C], w [co,9,C] (float64 CPU tensors carrying requires_grad) -> [NB,H,W,co] float64"""
    co, _, c = w.shape
    xi = F.relu(x) if relu_in else x
    # straight-through rounding: values rounded to the compute type, gradients as the identity (the kernels' backward rounds its own operands)
    xi = xi + (_round(xi.detach(), dt) - xi.detach())
    wi = w + (_round(w.detach(), dt) - w.detach())
    out = F.conv2d(xi.permute(0, 3, 1, 2), wi.view(co, 3, 3, c).permute(0, 3, 1, 2), None, padding=1).permute(0, 2, 3, 1)
    if bias is not None:
        out = out + bias
    if addend is not None:
        out = out + addend.repeat(x.shape[0] // addend.shape[0], 1, 1, 1)
    if residual is not None:
        out = out + residual
    return out


CASES = [
    # NB, H, W, C, co, relu_in, bias, NA, residual, x dtype
    (2, 8, 16, 16, 16, True, True, None, True, torch.float32),
    (6, 13, 21, 16, 16, True, False, 2, False, torch.float32),        # ragged tiles, addend shared by 3 sets of 2 images
    (3, 9, 35, 24, 16, False, True, None, False, torch.float32),      # 24 channels: a half-filled second group
    (2, 11, 19, 20, 5, True, True, None, False, torch.float32),       # C % 8 != 0: scalar staging; 5 output channels
    (2, 17, 33, 16, 1, False, True, None, False, torch.float32),      # out_conv
    (2, 10, 18, 136, 16, True, True, None, False, None),              # several LDS stages + a ragged last one; x in the compute dtype
    (1, 24, 40, 512, 16, True, True, None, False, None),              # the stride-8 feature half at its real width
]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES)
def test_conv3x3_n16_forward_backward(dev, case, dt):
    from ocpg_amd.models.ops.functions.mso_func import conv3x3_n16
    nb, h, wd, c, co, relu_in, has_bias, na, has_res, xdt = case
    xdt = dt if xdt is None else xdt
    g = torch.Generator().manual_seed(nb * 1000 + h * 10 + c)
    x = torch.randn(nb, h, wd, c, generator=g).to(xdt)
    w = (torch.randn(co, 9, c, generator=g) / (3 * c ** 0.5)).to(dt)
    bias = torch.randn(co, generator=g) if has_bias else None
    addend = torch.randn(na, h, wd, co, generator=g) if na else None
    res = torch.randn(nb, h, wd, co, generator=g) if has_res else None
    # the incoming gradient is representable in the compute type (the kernels round it for their products, as the autocast
    # convolution's grad_output is in that type; bias / addend / residual gradients take it as it comes)
    go = torch.randn(nb, h, wd, co, generator=g).to(dt).float()

    leaves = [t.double().requires_grad_(True) if t is not None else None for t in (x, w, bias, addend, res)]
    ref = _ref_conv(*leaves, relu_in, dt)
    ref.backward(go.double())

    dl = [t.to(dev).requires_grad_(True) if t is not None else None for t in (x, w, bias, addend, res)]
    out = conv3x3_n16(dl[0], dl[1], dl[2], dl[3], dl[4], relu_in, CODES[dt])
    assert out.dtype == torch.float32 and tuple(out.shape) == (nb, h, wd, co)
    out.backward(go.to(dev))
    scale = ref.abs().max().item()
    tol = 2e-5 if dt == torch.float32 else 1e-4                      # fp32 accumulation of exactly representable products
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() <= tol * scale
    names = ("x", "w", "bias", "addend", "residual")
    for name, a, b in zip(names, dl, leaves):
        if a is None:
            continue
        got, want = a.grad.detach().cpu().double(), b.grad
        assert got.shape == want.shape
        s = want.abs().max().item() + 1e-12
        # gradients returned in a 16-bit type (x, w in that type) carry that type's rounding
        gtol = tol if a.grad.dtype == torch.float32 else (8e-3 if a.grad.dtype == torch.bfloat16 else 1e-3)
        if name == "x" and dt != torch.float32:
            gtol = max(gtol, 1e-4)
        err = (got - want).abs().max().item()
        assert err <= gtol * s, (name, err, s)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("atomic", [True, False])
def test_res_block_equals_two_convolutions(dev, dt, atomic, monkeypatch):
    """ResBlockN16 (one node, skip gradient added in the input-gradient kernel's epilogue) == the two Conv3x3N16 nodes it fuses, in
    both weight-gradient modes (bands added with atomics / partial products summed)."""
    from ocpg_amd.models.ops.functions import mso_func as m
    monkeypatch.setattr(m, "WGRAD_ATOMIC", atomic)
    g = torch.Generator().manual_seed(3)
    nb, na, h, w, c = 6, 2, 19, 27, 16
    p = torch.randn(nb, h, w, c, generator=g).to(dev)
    sh = torch.randn(na, h, w, c, generator=g).to(dev)
    wm = (torch.randn(c, 9, c, generator=g) / 12).to(dt).to(dev)
    w2 = (torch.randn(c, 9, c, generator=g) / 12).to(dt).to(dev)
    b2 = torch.randn(c, generator=g).to(dev)
    go = torch.randn(nb, h, w, c, generator=g).to(dev)
    res = []
    for fused in (True, False):
        leaves = [t.clone().requires_grad_(True) for t in (p, sh, wm, w2, b2)]
        if fused:
            out = m.res_block_n16(*leaves, CODES[dt])
        else:
            y = m.conv3x3_n16(leaves[0], leaves[2], None, leaves[1], None, True, CODES[dt])
            out = m.conv3x3_n16(y, leaves[3], leaves[4], None, leaves[0], True, CODES[dt])
        out.backward(go)
        res.append([out.detach()] + [t.grad for t in leaves])
    for a, b, name in zip(res[0], res[1], ("out", "p", "shared", "wm", "w2", "b2")):
        tol = 2e-5 if a.dtype == torch.float32 else 8e-3
        assert (a.float() - b.float()).abs().max().item() <= tol * b.float().abs().max().item(), name


@pytest.mark.parametrize("shape", [(3, 12, 20, 16, 24, 40), (2, 13, 7, 16, 25, 13), (1, 5, 9, 8, 15, 31), (2, 48, 80, 16, 96, 160)])
def test_bilinear_nhwc_equals_interpolate(dev, shape):
    from ocpg_amd.models.ops.functions.mso_func import bilinear_nhwc
    nb, h, w, c, ho, wo = shape
    g = torch.Generator().manual_seed(h * w)
    x = torch.randn(nb, h, w, c, generator=g)
    go = torch.randn(nb, ho, wo, c, generator=g)
    xr = x.double().requires_grad_(True)
    ref = F.interpolate(xr.permute(0, 3, 1, 2), size=(ho, wo), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    ref.backward(go.double())
    xd = x.to(dev).requires_grad_(True)
    out = bilinear_nhwc(xd, (ho, wo))
    out.backward(go.to(dev))
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() <= 1e-5
    assert (xd.grad.cpu().double() - xr.grad).abs().max().item() <= 1e-5 * max(1.0, xr.grad.abs().max().item())


@pytest.mark.parametrize("amp", [None, torch.bfloat16, torch.float16])
def test_mso_native_equals_library_path(dev, amp, monkeypatch):
    """The whole block (forward_multi: 3 mask sets over shared features, and forward: one set) through csrc/mso.hip against the same
    module through F.conv2d / F.interpolate in fp32, values and every gradient."""
    from ocpg_amd.models import decoder
    from ocpg_amd.models.decoder import MSO
    from ocpg_amd.util.misc import NestedTensor
    torch.manual_seed(5)
    bt, n, h, w = 2, 3, 12, 20
    mso = MSO(mask_dim=16, img_dim=(32, 64)).to(dev)
    f4 = torch.randn(bt, 32, 2 * h, 2 * w, device=dev).contiguous(memory_format=torch.channels_last)
    f8 = torch.randn(bt, 64, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    pms = [torch.randn(bt, 16, h, w, device=dev) for _ in range(n)]
    go = torch.randn(n * bt, 1, 2 * h, 2 * w, device=dev)

    def run(native, amp_dt):
        monkeypatch.setattr(decoder, "NATIVE", native)
        mso.zero_grad(set_to_none=True)
        a4, a8 = f4.clone().requires_grad_(True), f8.clone().requires_grad_(True)
        ps = [p.clone().requires_grad_(True) for p in pms]
        feats = [NestedTensor(a4, None), NestedTensor(a8, None)]
        with torch.autocast("cuda", dtype=amp_dt, enabled=amp_dt is not None):
            out = mso.forward_multi(ps, feats, stacked=True)
            one = mso(ps[1], feats)
        (out.float() * go).sum().backward()
        grads = {k: v.grad.clone() for k, v in mso.named_parameters()}
        grads.update(f4=a4.grad.clone(), f8=a8.grad.clone(), **{f"pm{i}": p.grad.clone() for i, p in enumerate(ps)})
        return out.float().detach(), one.float().detach(), grads

    ref_out, ref_one, ref_g = run(False, None)
    out, one, g = run(True, amp)
    s = ref_out.abs().max().item()
    assert (ref_one - ref_out[2:4]).abs().max().item() <= 1e-4 * s
    rel = lambda a, b: (a.float() - b.float()).norm().item() / (b.float().norm().item() + 1e-12)
    if amp is None:
        bound = lambda k: 2e-5
    else:
        # 16-bit operands flip a few ReLU masks, which moves gradients by percents in ANY implementation: the yardstick is the
        # library path (F.conv2d under the same autocast) against the same fp32 run -- the kernels must not be further away than
        # that (norm-relative: single flipped units make the max norm jumpy)
        lib_out, _, lib_g = run(False, amp)
        lib_err = {k: rel(lib_g[k], want) for k, want in ref_g.items()}
        lib_err["out"] = rel(lib_out, ref_out)
        bound = lambda k: 1.5 * lib_err[k] + 2e-3
    assert rel(out, ref_out) <= bound("out")
    assert rel(one, out[2:4]) <= 1e-5
    for k, want in ref_g.items():
        assert rel(g[k], want) <= bound(k), (k, rel(g[k], want), bound(k))
