"""Parity of the HIP MSDeformAttn op (through the C ABI) against the oracle and the reference's golden vectors.

Protocol of the reference's own test (models/ops/test.py): fp64 forward allclose (:32-44), fp32 forward
rtol 1e-2 / atol 1e-3 (:47-60), fp64 gradcheck for D in {30,32,64,71,1025,2048,3096} (:63-86).
"""
import pytest
import torch

from cases import MSDA_CASES, level_start, msda_case_inputs
import synth

pytestmark = pytest.mark.gpu


def _fn():
    from ocpg_amd.models.ops.functions import MSDeformAttnFunction
    return MSDeformAttnFunction


def test_native_library_is_loaded(dev):
    from ocpg_amd import _lib
    assert _lib.lib().ocpg_hip_version().startswith(b"ocpg_hip")


def test_testpy_forward_double_float(golden, dev):
    g = golden("msda_testpy")
    shapes, ls = g["shapes"].to(dev), g["level_start"].to(dev)
    out = _fn().apply(g["d_value"].double().to(dev), shapes, ls, g["d_loc"].double().to(dev), g["d_attn"].double().to(dev), 2)
    assert torch.allclose(out.cpu(), g["d_out"])                       # test.py:40
    out = _fn().apply(g["f_value"].to(dev), shapes, ls, g["f_loc"].to(dev), g["f_attn"].to(dev), 2)
    assert torch.allclose(out.cpu(), g["f_out"], rtol=1e-2, atol=1e-3)  # test.py:56
    assert torch.allclose(out.cpu(), g["f_out"], rtol=1e-5, atol=1e-8)  # and much tighter than the reference asks


@pytest.mark.parametrize("ch", [30, 32, 64, 71, 1025])
def test_testpy_gradient_vectors(golden, dev, ch):
    g = golden("msda_testpy")
    shapes, ls = g["shapes"].to(dev), g["level_start"].to(dev)
    v, l, a = (g[f"g{ch}_{k}"].double().to(dev).requires_grad_(True) for k in ("value", "loc", "attn"))
    go = synth.rand(f"testpy_go_{ch}", g[f"g{ch}_out"].shape).double().to(dev)
    out = _fn().apply(v, shapes, ls, l, a, 2)
    assert torch.allclose(out.cpu(), g[f"g{ch}_out"])
    gv, gl, ga = torch.autograd.grad((out * go).sum(), (v, l, a))
    assert torch.allclose(gv.cpu(), g[f"g{ch}_gv"], rtol=1e-9, atol=1e-12)
    assert torch.allclose(gl.cpu(), g[f"g{ch}_gl"], rtol=1e-9, atol=1e-12)
    assert torch.allclose(ga.cpu(), g[f"g{ch}_ga"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("ch", [30, 32, 64, 71, 1025, 2048, 3096])
def test_testpy_gradcheck(dev, ch):
    """test.py:63-78 check_gradient_numerical, same shapes."""
    N, M, Lq, L, P = 1, 2, 2, 2, 2
    shapes, ls = level_start([(6, 4), (3, 2)])
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3 + ch)
    value = (torch.rand(N, S, M, ch) * 0.01).double().to(dev).requires_grad_(True)
    loc = torch.rand(N, Lq, M, L, P, 2).double().to(dev).requires_grad_(True)
    attn = torch.rand(N, Lq, M, L, P) + 1e-5
    attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).double().to(dev).requires_grad_(True)
    # fp64 atomics make the op run-to-run non-bitwise (summation order): nondet_tol covers the last bits
    assert torch.autograd.gradcheck(_fn().apply, (value, shapes.to(dev), ls.to(dev), loc, attn, 2), nondet_tol=1e-10)


@pytest.mark.parametrize("case", MSDA_CASES, ids=[c["name"] for c in MSDA_CASES])
def test_cases_vs_golden_and_oracle(golden, dev, case):
    from oracle import msda as om
    g = golden("msda_cases")
    n = case["name"]
    value, shapes, ls, loc, attn, go = msda_case_inputs(case)
    v, l, a = (x.to(dev).requires_grad_(True) for x in (value, loc, attn))
    out = _fn().apply(v, shapes.to(dev), ls.to(dev), l, a, 64)
    gv, gl, ga = torch.autograd.grad((out * go.to(dev)).sum(), (v, l, a))
    # fp32 HIP vs fp64 reference truth
    assert torch.allclose(out.cpu().double(), g[f"{n}_out64"], rtol=1e-4, atol=1e-5)
    assert torch.allclose(gv.cpu().double(), g[f"{n}_gv64"], rtol=1e-4, atol=1e-5)
    assert torch.allclose(ga.cpu().double(), g[f"{n}_ga64"], rtol=1e-4, atol=2e-5)
    wh = torch.stack([shapes[:, 1], shapes[:, 0]], -1).double()[None, None, None, :, None, :]
    on_edge = ((loc.double() * wh - 0.5) == -1.0).any(-1, keepdim=True).expand_as(gl)
    assert torch.allclose(gl.cpu().double()[~on_edge], g[f"{n}_gl64"][~on_edge], rtol=1e-3, atol=2e-4)
    # fp32 HIP vs fp32 C oracle (same arithmetic, different summation order)
    oc = om.msda_c_forward(value, shapes, ls, loc, attn)
    assert torch.allclose(out.cpu(), oc, rtol=1e-5, atol=1e-6)
    ogv, ogl, oga = om.msda_c_backward(value, shapes, ls, loc, attn, go)
    assert torch.allclose(gv.cpu(), ogv, rtol=1e-4, atol=1e-5)
    assert torch.allclose(gl.cpu(), ogl, rtol=1e-3, atol=1e-4)
    assert torch.allclose(ga.cpu(), oga, rtol=1e-4, atol=1e-5)
    # fp64 HIP (generic kernels) vs fp64 truth
    v, l, a = (x.double().to(dev).requires_grad_(True) for x in (value, loc, attn))
    out = _fn().apply(v, shapes.to(dev), ls.to(dev), l, a, 64)
    gv, gl, ga = torch.autograd.grad((out * go.double().to(dev)).sum(), (v, l, a))
    assert torch.allclose(out.cpu(), g[f"{n}_out64"], rtol=1e-10, atol=1e-12)
    assert torch.allclose(gv.cpu(), g[f"{n}_gv64"], rtol=1e-9, atol=1e-11)
    assert torch.allclose(gl.cpu()[~on_edge], g[f"{n}_gl64"][~on_edge], rtol=1e-8, atol=1e-9)
    assert torch.allclose(ga.cpu(), g[f"{n}_ga64"], rtol=1e-9, atol=1e-11)


def _full_size_inputs(dev, Lq=None, N=5):
    """BASELINE config #2 encoder shapes: N=5 frames, levels 48x80..6x10, M=8, D=32, P=4."""
    shapes, ls = level_start([(48, 80), (24, 40), (12, 20), (6, 10)])
    S = int(shapes.prod(1).sum())
    Lq = Lq or S
    g = torch.Generator(device="cpu").manual_seed(11)
    value = torch.randn(N, S, 8, 32, generator=g)
    loc = torch.rand(N, Lq, 8, 4, 4, 2, generator=g) * 1.1 - 0.05
    attn = torch.softmax(torch.randn(N, Lq, 8, 16, generator=g), -1).view(N, Lq, 8, 4, 4)
    return value.to(dev), shapes.to(dev), ls.to(dev), loc.to(dev), attn.to(dev), S


def test_full_size_properties(dev):
    """Size-independent properties at BASELINE config #2 size: linearity in value / attention, and
    <out, go> == <value, grad_value> (adjointness of the scatter to the gather)."""
    value, shapes, ls, loc, attn, S = _full_size_inputs(dev)
    f = _fn().apply
    out = f(value, shapes, ls, loc, attn, 64)
    assert out.shape == (5, S, 256) and torch.isfinite(out).all()
    out2 = f(2.5 * value, shapes, ls, loc, attn, 64)
    assert torch.allclose(out2, 2.5 * out, rtol=1e-5, atol=1e-5)
    out3 = f(value, shapes, ls, loc, 0.5 * attn, 64)
    assert torch.allclose(out3, 0.5 * out, rtol=1e-5, atol=1e-5)
    v = value.clone().requires_grad_(True)
    a = attn.clone().requires_grad_(True)
    go = torch.randn_like(out)
    o = f(v, shapes, ls, loc, a, 64)
    gv, ga = torch.autograd.grad((o * go).sum(), (v, a))
    lhs = (o.double() * go.double()).sum()
    assert torch.allclose(lhs, (gv.double() * value.double()).sum(), rtol=1e-4)
    assert torch.allclose(lhs, (ga.double() * attn.double()).sum(), rtol=1e-4)


def test_full_size_matches_c_oracle_on_a_slice(dev):
    """Same full-size call, one frame checked element-wise against the C oracle (keeps the CPU part to seconds)."""
    from oracle import msda as om
    value, shapes, ls, loc, attn, S = _full_size_inputs(dev, N=1)
    out = _fn().apply(value, shapes, ls, loc, attn, 64)
    oc = om.msda_c_forward(value.cpu(), shapes.cpu(), ls.cpu(), loc.cpu(), attn.cpu())
    assert torch.allclose(out.cpu(), oc, rtol=1e-4, atol=1e-5)
    go = torch.randn_like(out)
    from ocpg_amd.models.ops.functions import ms_deform_attn_backward
    gv, gl, ga = ms_deform_attn_backward(value, shapes, ls, loc, attn, go)
    ogv, ogl, oga = om.msda_c_backward(value.cpu(), shapes.cpu(), ls.cpu(), loc.cpu(), attn.cpu(), go.cpu())
    assert torch.allclose(gv.cpu(), ogv, rtol=1e-3, atol=1e-4)
    assert torch.allclose(gl.cpu(), ogl, rtol=1e-3, atol=1e-3)
    assert torch.allclose(ga.cpu(), oga, rtol=1e-3, atol=1e-4)


def test_edge_inputs(dev):
    """Empty query set, N not a multiple of the reference's im2col_step, non-contiguous input error."""
    f = _fn().apply
    shapes, ls = level_start([(3, 4)])
    value = torch.randn(3, 12, 2, 8, device=dev)
    loc = torch.rand(3, 0, 2, 1, 2, 2, device=dev)
    attn = torch.rand(3, 0, 2, 1, 2, device=dev)
    out = f(value, shapes.to(dev), ls.to(dev), loc, attn, 64)
    assert out.shape == (3, 0, 16)
    big = torch.randn(3, 12, 2, 16, device=dev)[..., ::2]
    with pytest.raises(RuntimeError, match="contiguous"):
        f(big, shapes.to(dev), ls.to(dev), torch.rand(3, 1, 2, 1, 2, 2, device=dev), torch.rand(3, 1, 2, 1, 2, device=dev), 64)
    # N = 65 > 64 and not a multiple of 64: the reference raises (ms_deform_attn_cuda.cu:50-52), we do not chunk
    value = torch.randn(65, 12, 2, 8, device=dev)
    loc = torch.rand(65, 5, 2, 1, 2, 2, device=dev)
    attn = torch.rand(65, 5, 2, 1, 2, device=dev)
    out = f(value, shapes.to(dev), ls.to(dev), loc, attn, 64)
    from oracle import msda as om
    assert torch.allclose(out.cpu(), om.msda_c_forward(value.cpu(), shapes, ls, loc.cpu(), attn.cpu()), rtol=1e-5, atol=1e-6)
