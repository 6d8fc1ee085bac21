"""Window attention on the matrix cores (csrc/win_attn_mfma.hip) against the tensor-op formulation of
WindowAttention3D.forward (models/video_swin_transformer.py:138-169) in fp32, for bf16 / fp16 storage: forward, log-sum-exp and
all gradients; shapes = the clamped (5,7,7) window (N = 245, padding keys in the last tile, shift regions), the full (8,7,7)
window (N = 392) and a tiny ragged one.  The vector-ALU kernels (OCPG_WIN_ATTN_MFMA=0) are run on the same inputs as the yardstick:
the matrix-core path may not be further from fp32 than they are by more than the storage rounding."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(qkv, bias, region, scale, nw):
    bw, n, _, h, hd = qkv.shape
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3).float() for i in range(3))          # [bw, h, n, hd]
    s = (q * scale) @ k.transpose(-1, -2) + bias[None].float()
    if region is not None:
        r = region[torch.arange(bw, device=qkv.device) % nw]                         # [bw, n]
        s = s + torch.where(r[:, None, :, None] != r[:, None, None, :], -100.0, 0.0)
    p = torch.softmax(s, -1)
    return (p @ v).permute(0, 2, 1, 3).reshape(bw, n, h * hd)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("bw,nw,n,h,shift", [(6, 3, 245, 3, True), (4, 2, 392, 4, True), (4, 4, 245, 6, False), (3, 1, 37, 2, True)])
def test_mfma_window_attention(dev, dtype, bw, nw, n, h, shift):
    from ocpg_amd.models.ops.functions.win_attn_func import window_attention
    g = torch.Generator(device=dev).manual_seed(n * 7 + h)
    qkv32 = torch.randn(bw, n, 3, h, 32, device=dev, generator=g)
    bias = torch.randn(h, n, n, device=dev, generator=g) * 0.5
    region = (torch.randint(0, 3, (nw, n), device=dev, generator=g).int() if shift else None)
    go = torch.randn(bw, n, h * 32, device=dev, generator=g)
    scale = 32 ** -0.5
    qkv_lp = qkv32.to(dtype)
    # fp32 reference on the SAME (rounded) inputs
    a = qkv_lp.float().requires_grad_(True)
    b = bias.clone().requires_grad_(True)
    want = _reference(a, b, region, scale, nw)
    want_g = torch.autograd.grad((want * go).sum(), (a, b))
    res = {}
    for mode in ("1", "0"):
        os.environ["OCPG_WIN_ATTN_MFMA"] = mode
        try:
            x = qkv_lp.clone().requires_grad_(True)
            bb = bias.clone().requires_grad_(True)
            out = window_attention(x, bb, region, scale, nw)
            gx, gb = torch.autograd.grad((out.float() * go).sum(), (x, bb))
            res[mode] = (out.float(), gx.float(), gb)
        finally:
            os.environ.pop("OCPG_WIN_ATTN_MFMA", None)
    eps = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
    for name, got, ref_v, valu in (("out", res["1"][0], want, res["0"][0]), ("dqkv", res["1"][1], want_g[0], res["0"][1]),
                                   ("dbias", res["1"][2], want_g[1], res["0"][2])):
        scale_v = ref_v.abs().max().item()
        e_mfma, e_valu = (got - ref_v).abs().max().item(), (valu - ref_v).abs().max().item()
        assert e_mfma <= 1.5 * e_valu + 6 * eps * scale_v, (name, e_mfma, e_valu, scale_v)


@pytest.mark.parametrize("xdt,odt", [(torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16), (torch.float16, torch.float16),
                                     (torch.float32, torch.float32)])
@pytest.mark.parametrize("rows,c", [(1000, 96), (333, 192), (64, 384), (17, 768), (5, 1024), (2, 100), (5000, 128)])
def test_fused_layernorm_low_precision(dev, xdt, odt, rows, c):
    """csrc/layernorm.hip against F.layer_norm in fp32 on the same (rounded) input: y to the output dtype's rounding; dx, dgamma, dbeta
    against autograd of the fp32 formulation."""
    from ocpg_amd.models.ops.functions.layernorm_func import LayerNormLP
    g = torch.Generator(device=dev).manual_seed(rows + c)
    x = (torch.randn(rows, c, device=dev, generator=g) * 2 + 0.5).to(xdt)
    w = torch.randn(c, device=dev, generator=g)
    b = torch.randn(c, device=dev, generator=g)
    go = torch.randn(rows, c, device=dev, generator=g).to(odt)
    xa, wa, ba = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = LayerNormLP.apply(xa, wa, ba, 1e-5, odt)
    assert y.dtype == odt
    gx, gw, gb = torch.autograd.grad((y.float() * go.float()).sum(), (xa, wa, ba))
    assert gx.dtype == xdt
    xr, wr, br = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (c,), wr, br, 1e-5)
    rx, rw, rb = torch.autograd.grad((yr * go.float()).sum(), (xr, wr, br))
    eps_o = {torch.float32: 2e-6, torch.bfloat16: 2 ** -8, torch.float16: 2 ** -11}[odt]
    eps_x = {torch.float32: 2e-6, torch.bfloat16: 2 ** -8, torch.float16: 2 ** -11}[xdt]
    assert (y.float() - yr).abs().max().item() <= eps_o * yr.abs().max().item() + 1e-6
    assert (gx.float() - rx).abs().max().item() <= eps_x * rx.abs().max().item() + 1e-5
    assert (gw - rw).abs().max().item() <= 2e-5 * rw.abs().max().item() + 1e-4
    assert (gb - rb).abs().max().item() <= 2e-5 * rb.abs().max().item() + 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("n,heads", [(392, 4), (245, 3), (37, 16)])
def test_relative_position_bias_kernel_equals_table_lookup(n, heads):
    """csrc/layernorm.hip relpos_bias (round 4): bias and its transpose from the table in one launch == the reference's
    table[index[:N, :N].reshape(-1)].reshape(N, N, -1).permute(2, 0, 1) (video_swin_transformer.py:151-153), full and clamped windows; the
    table gradient from a contiguous gradient and from one whose TRANSPOSE is contiguous (what the attention backward hands back)."""
    import ocpg_amd.models.video_swin_transformer as vs
    dev = torch.device("cuda:0")
    torch.manual_seed(n)
    wa = vs.WindowAttention3D(32 * heads, (8, 7, 7), heads, qkv_bias=True).to(dev)
    with torch.no_grad():
        wa.relative_position_bias_table.normal_()
    idx = wa.relative_position_index[:n, :n].reshape(-1)
    want = wa.relative_position_bias_table[idx].view(n, n, -1).permute(2, 0, 1)
    bias = wa.relative_position_bias(n)
    bias_t = wa.__dict__.pop("_bias_t")
    assert bias.is_contiguous() and bias_t.is_contiguous()
    assert torch.equal(bias, want) and torch.equal(bias_t, want.transpose(1, 2))
    for g in (torch.randn(heads, n, n, device=dev), torch.randn(heads, n, n, device=dev).transpose(1, 2)):
        got, = torch.autograd.grad(bias, wa.relative_position_bias_table, g, retain_graph=True)
        ref, = torch.autograd.grad(want, wa.relative_position_bias_table, g, retain_graph=True)
        assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-6
