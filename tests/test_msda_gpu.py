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


# ---- round 2: the self-attention kernels (column-tile scatter + gather-only row kernel; LDS-window forward) -------------
def _local_inputs(dev, N, shapes_l, M=8, D=32, P=4, noise=1.5, outliers=0.02, seed=5):
    """Encoder-like sampling pattern (reference initialisation ms_deform_attn.py:64-78: the head's direction, 1..P pixels
    away from the query's own pixel) plus gaussian noise and a few far outliers, so that samples fall inside the kernels'
    LDS windows, on their edges and outside them (direct global path), and outside the map."""
    import math
    shapes, ls = level_start(shapes_l)
    S = int(shapes.prod(1).sum())
    L = len(shapes_l)
    g = torch.Generator().manual_seed(seed)
    refs = []
    for (h, w) in shapes_l:
        ys, xs = torch.meshgrid(torch.linspace(0.5, h - 0.5, h) / h, torch.linspace(0.5, w - 0.5, w) / w, indexing="ij")
        refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(refs, 0)[None, :, None, None, None, :]
    th = torch.arange(M) * (2 * math.pi / M)
    grid = torch.stack([th.cos(), th.sin()], -1)
    grid = grid / grid.abs().max(-1, keepdim=True)[0]
    off = grid.view(1, 1, M, 1, 1, 2) * torch.arange(1, P + 1).view(1, 1, 1, 1, P, 1)
    off = off.expand(N, S, M, L, P, 2) + noise * torch.randn(N, S, M, L, P, 2, generator=g)
    norm = torch.tensor([[w, h] for h, w in shapes_l], dtype=torch.float32).view(1, 1, 1, L, 1, 2)
    loc = ref + off / norm
    far = torch.rand(N, S, M, L, P, 1, generator=g) < outliers
    loc = torch.where(far, torch.rand(N, S, M, L, P, 2, generator=g) * 1.3 - 0.15, loc).contiguous()
    value = torch.randn(N, S, M, D, generator=g)
    attn = torch.softmax(torch.randn(N, S, M, L * P, generator=g), -1).view(N, S, M, L, P)
    go = torch.randn(N, S, M * D, generator=g)
    return value, shapes, ls, loc, attn, go


class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        import os
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *exc):
        import os
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("shapes_l", [[(48, 80), (24, 40), (12, 20), (6, 10)],      # config #2
                                       [(60, 108), (30, 54), (15, 27), (8, 14)],     # config #5 (ragged level sizes)
                                       [(32, 32), (16, 16), (8, 8)],                 # config #1: 3 levels
                                       [(13, 7), (5, 9)]],                           # non-pyramid: levels that do not nest
                         ids=["cfg2", "cfg5", "cfg1_3lvl", "ragged"])
def test_self_attention_kernels_vs_c_oracle(dev, shapes_l):
    """Forward + backward on locality-structured inputs, element-wise against the C oracle (one frame keeps the CPU side
    to seconds), for every path that serves Lq == S: default (row forward, column scatter + row gather backward), the
    output-tiled grad_value kernels (round 3, csrc/msda_tile.hip, OCPG_MSDA_TILE=1), the LDS-window forward (OCPG_MSDA_FWD=col)
    and the round-1 kernels (OCPG_MSDA_COL=0).  The inputs hold samples within the tile margin, beyond it (far corners: atomics of the
    coarse kernel) and outside the map."""
    from oracle import msda as om
    from ocpg_amd.models.ops.functions import ms_deform_attn_backward, ms_deform_attn_forward
    value, shapes, ls, loc, attn, go = _local_inputs(dev, 1, shapes_l)
    oc = om.msda_c_forward(value, shapes, ls, loc, attn)
    ogv, ogl, oga = om.msda_c_backward(value, shapes, ls, loc, attn, go)
    dv, dl, da, dg = (t.to(dev) for t in (value, loc, attn, go))
    ds, dls = shapes.to(dev), ls.to(dev)
    ds._ocpg_host = shapes
    for env in ({}, {"OCPG_MSDA_COL_LP": "2"}, {"OCPG_MSDA_COL_LP": "1"}, {"OCPG_MSDA_TILE": "1"}, {"OCPG_MSDA_FWD": "col"}, {"OCPG_MSDA_COL": "0"}):
        with _env(**env):
            out = ms_deform_attn_forward(dv, ds, dls, dl, da)
            gv, gl, ga = ms_deform_attn_backward(dv, ds, dls, dl, da, dg)
        assert torch.allclose(out.cpu(), oc, rtol=1e-4, atol=1e-5), env
        assert torch.allclose(gv.cpu(), ogv, rtol=1e-3, atol=1e-4), env
        # grad_loc is a difference of corner values scaled by the level size (|g| up to ~1e3 here): fp32 summation order
        # shows at ~2e-6 of the largest element (the round-1 kernels and the C oracle differ by the same amount)
        assert (gl.cpu() - ogl).abs().max() <= 2e-5 * ogl.abs().max(), env
        assert torch.allclose(ga.cpu(), oga, rtol=1e-3, atol=1e-4), env


def test_self_attention_backward_paths_agree_at_bench_size(dev):
    """N = 10 frames (what bench.py runs: 2 clips x 5 frames): column scatter + row gather against the round-1 tiled
    kernel (itself pinned to the oracle above), plus the adjointness <out, go> == <value, grad_value>."""
    from ocpg_amd.models.ops.functions import ms_deform_attn_backward, ms_deform_attn_forward
    value, shapes, ls, loc, attn, go = (t.to(dev) if i != 1 else t for i, t in
                                        enumerate(_local_inputs(dev, 10, [(48, 80), (24, 40), (12, 20), (6, 10)], seed=9)))
    ds, dls = shapes.to(dev), ls.to(dev)
    ds._ocpg_host = shapes
    out = ms_deform_attn_forward(value, ds, dls, loc, attn)
    gv, gl, ga = ms_deform_attn_backward(value, ds, dls, loc, attn, go)
    with _env(OCPG_MSDA_COL="0"):
        gv0, gl0, ga0 = ms_deform_attn_backward(value, ds, dls, loc, attn, go)
    scale = gv0.abs().max()
    assert (gv - gv0).abs().max() <= 2e-5 * scale
    with _env(OCPG_MSDA_TILE="1"):                                  # the output-tiled kernels (tile stores + coarse / far atomics)
        gv1, _, _ = ms_deform_attn_backward(value, ds, dls, loc, attn, go)
    assert (gv1 - gv0).abs().max() <= 2e-5 * scale
    assert (gl - gl0).abs().max() <= 2e-5 * gl0.abs().max() and torch.allclose(ga, ga0, rtol=1e-3, atol=1e-4)
    lhs = (out.double() * go.double()).sum()
    assert torch.allclose(lhs, (gv.double() * value.double()).sum(), rtol=1e-4)


def test_self_attention_without_host_shapes(dev):
    """A foreign caller that has no host copy of the shapes still gets the same results (row kernels)."""
    from ocpg_amd.models.ops.functions import ms_deform_attn_backward, ms_deform_attn_forward
    value, shapes, ls, loc, attn, go = _local_inputs(dev, 1, [(16, 24), (8, 12)])
    dv, dl, da, dg = (t.to(dev) for t in (value, loc, attn, go))
    a, b_ = shapes.to(dev), shapes.to(dev)
    a._ocpg_host = shapes
    out1, out2 = ms_deform_attn_forward(dv, a, ls.to(dev), dl, da), ms_deform_attn_forward(dv, b_, ls.to(dev), dl, da)
    assert torch.allclose(out1, out2, rtol=1e-5, atol=1e-6)
    g1 = ms_deform_attn_backward(dv, a, ls.to(dev), dl, da, dg)
    g2 = ms_deform_attn_backward(dv, b_, ls.to(dev), dl, da, dg)
    for x, y in zip(g1, g2):
        assert torch.allclose(x, y, rtol=1e-4, atol=1e-5)


def test_grad_value_path_selection_follows_the_offsets(dev):
    """ocpg_msda_bwd_value_sel_f32 (round 4): the call site's state picks the column scatter or the output-tiled kernels ON THE DEVICE from
    the previous call's share of far samples.  A site that sees spread-out ("trained-like") offsets moves to the tiled kernels on its
    second call, moves back once the offsets are local again, and every call -- whichever family ran, including the calls on which the
    state changes -- returns the same grad_value as the fixed path (both are pinned to the C oracle above)."""
    from ocpg_amd.models.ops.functions import ms_deform_attn_backward
    shapes_l = [(48, 80), (24, 40), (12, 20), (6, 10)]
    near = [t.to(dev) if i != 1 else t for i, t in enumerate(_local_inputs(dev, 2, shapes_l, noise=0.3, outliers=0.0, seed=3))]
    wide = [t.to(dev) if i != 1 else t for i, t in enumerate(_local_inputs(dev, 2, shapes_l, noise=4.0, outliers=0.08, seed=4))]
    shapes = near[1]
    ds, dls = shapes.to(dev), near[2]
    ds._ocpg_host = shapes
    state = torch.zeros(8, dtype=torch.int32, device=dev)
    want = {}
    for name, (value, _, _, loc, attn, go) in (("near", near), ("wide", wide)):
        want[name] = ms_deform_attn_backward(value, ds, dls, loc, attn, go)[0]
    seen = []
    for name in ("near", "wide", "wide", "wide", "near", "near", "near"):
        value, _, _, loc, attn, go = near if name == "near" else wide
        ran = int(state[3])                                  # the path THIS call takes
        gv = ms_deform_attn_backward(value, ds, dls, loc, attn, go, sel_state=state)[0]
        st = state.tolist()
        seen.append((name, ran, st[3], st[6], st[7]))
        assert (gv - want[name]).abs().max() <= 2e-5 * want[name].abs().max(), seen
        assert st[0] == 0 and st[1] == 0 and st[2] == 0 and st[5] == 0, st     # counters and tickets are reset by the last workgroups
        assert st[7] > 0 and 0 <= st[6] <= st[7], st
    paths = [r for _, r, _, _, _ in seen]
    # call 1 (near) column, proposes column; call 2 (wide) still column, proposes tiled; calls 3-4 tiled; call 5 (near) still tiled, proposes
    # column; calls 6-7 column
    assert paths == [0, 0, 1, 1, 1, 0, 0], seen
    assert seen[1][3] * 100 > 6 * seen[1][4] and seen[0][3] * 100 < 6 * seen[0][4], seen      # far shares: wide above, near below the 6 % threshold (the column kernel's count)


@pytest.mark.parametrize("pad", [False, True])
def test_fused_front_end_equals_the_unfused_module(dev, pad):
    """MSDeformAttn at the config-#2 encoder shape (8 heads x 32, 4 levels x 4 points, 2 frames) with trained-looking projections: the fused
    front end (ocpg_msda_fused_*: softmax, `reference + offset`, softmax backward and gradient layout inside the kernels) against the same
    module with OCPG_MSDA_FUSED_FRONT off (ATen softmax / add / split-cat around the plain entry points): output, locations, weights and
    the gradients of query, source and all parameters."""
    from ocpg_amd.models.ops.modules import MSDeformAttn
    from ocpg_amd.models.ops.modules import ms_deform_attn as mod_file
    shapes_l = [(48, 80), (24, 40), (12, 20), (6, 10)]
    shapes, ls = level_start(shapes_l)
    S = int(shapes.prod(1).sum())
    g = torch.Generator().manual_seed(11)
    m = MSDeformAttn(256, 4, 8, 4)
    with torch.no_grad():
        m.sampling_offsets.weight.copy_(torch.randn(m.sampling_offsets.weight.shape, generator=g) * 0.05)
        m.attention_weights.weight.copy_(torch.randn(m.attention_weights.weight.shape, generator=g) * 0.2)
        m.attention_weights.bias.copy_(torch.randn(m.attention_weights.bias.shape, generator=g) * 0.5)
    m.to(dev)
    N = 2
    refs = []
    for (h, w) in shapes_l:
        ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")
        refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(refs, 0)[None, :, None, :].expand(N, S, 4, 2).contiguous().to(dev)
    q0 = torch.randn(N, S, 256, generator=g).to(dev)
    src0 = torch.randn(N, S, 256, generator=g).to(dev)
    go = torch.randn(N, S, 256, generator=g).to(dev)
    mask = None
    if pad:
        mask = torch.zeros(N, S, dtype=torch.bool)
        mask[1, 3000:3600] = True
        mask = mask.to(dev)
    ds, dls = shapes.to(dev), ls.to(dev)
    ds._ocpg_host = shapes
    res = {}
    for fused in (True, False):
        old = mod_file.FUSED_FRONT
        mod_file.FUSED_FRONT = fused
        try:
            q, src = q0.clone().requires_grad_(True), src0.clone().requires_grad_(True)
            m._sel_state.zero_()
            out, loc, attn = m(q, ref, src, ds, dls, mask)
            grads = torch.autograd.grad((out * go).sum(), [q, src] + list(m.parameters()))
            res[fused] = (out, loc, attn) + tuple(grads)
        finally:
            mod_file.FUSED_FRONT = old
    names = ["out", "loc", "attn", "gq", "gsrc"] + ["g_" + n for n, _ in m.named_parameters()]
    for nm, a, b in zip(names, res[True], res[False]):
        tol = 2e-5 * b.abs().max().item() + 1e-7
        assert (a - b).abs().max().item() <= tol, (nm, (a - b).abs().max().item(), b.abs().max().item())
