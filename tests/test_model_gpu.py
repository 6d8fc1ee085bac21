"""The product model on the MI355X (real HIP MSDeformAttn op through the C ABI) against the reference's golden vectors."""
import pytest
import torch

import model_checks
import module_checks as mc

pytestmark = pytest.mark.gpu


def test_lfm(golden, dev):
    mc.check_lfm(golden("lfm"), dev, rtol=2e-4, atol=2e-5)


def test_fusion(golden, dev):
    mc.check_fusion(golden("fusion"), dev, rtol=2e-4, atol=2e-5)


def test_msda_module(golden, dev):
    mc.check_msda_module(golden("msda_module"), dev, rtol=2e-4, atol=2e-5)


def test_transformer(golden, dev):
    mc.check_transformer(golden("transformer"), dev, rtol=5e-4, atol=5e-5)


def test_dynmask_mso(golden, dev):
    mc.check_dynmask_mso(golden("dynmask_mso"), dev, rtol=2e-4, atol=2e-4)


@pytest.fixture()
def strict_hip(monkeypatch):
    """The head_dim-32 module fixtures must be served by the production HIP kernels: any library fallback raises, and the
    census of successful C-ABI calls says which entry points ran (the 8-head / head_dim-8 fixtures above
    only reach the SDPA fallback and msda_fwd_fast<2>)."""
    from ocpg_amd import _lib
    from ocpg_amd.models import fallbacks
    monkeypatch.setenv("OCPG_STRICT_HIP", "1")
    fallbacks.reset()
    calls = _lib.census(True)
    yield calls
    _lib.census(False)
    assert fallbacks.snapshot() == {}


def test_fusion_head_dim_32(golden, dev, strict_hip):
    """segmentation.py:95-113 at 2 heads x 32: reference vectors through csrc/attn_smallk.hip (fwd + bwd, key padding)."""
    mc.check_fusion(golden("fusion_d32"), dev, rtol=2e-4, atol=2e-5)
    assert strict_hip.get("ocpg_attn_smallk_fwd", 0) >= 1 and strict_hip.get("ocpg_attn_smallk_bwd", 0) >= 1, strict_hip


def test_msda_module_head_dim_32(golden, dev, strict_hip):
    """ms_deform_attn.py:80-118 at 2 heads x 32 (2-d and 4-d reference points, padding mask): the <8>-lane kernels."""
    mc.check_msda_module(golden("msda_module_d32"), dev, rtol=2e-4, atol=2e-5)
    assert strict_hip.get("ocpg_msda_fwd_f32", 0) >= 2, strict_hip
    assert any(k.startswith("ocpg_msda_bwd") for k in strict_hip), strict_hip


def test_msda_module_fused_front_end_vs_reference_vectors(golden, dev, strict_hip):
    """The same reference vectors (ms_deform_attn.py:80-118 run by the reference at 2 heads x 32) through the round-4 FUSED front end: with the
    host copy of the level shapes attached (as the product's transformer does) the self-attention case computes softmax + `reference +
    offset` inside ocpg_msda_fused_fwd_f32, the softmax backward + the [d offsets | d logits] layout inside ocpg_msda_fused_bwd_qproj_f32, and
    grad_value through the path-selecting entry point; output, sampling locations, attention weights and every gradient must match."""
    mc.check_msda_module(golden("msda_module_d32"), dev, rtol=2e-4, atol=2e-5, host_shapes=True)
    for sym in ("ocpg_msda_fused_fwd_f32", "ocpg_msda_fused_bwd_qproj_f32", "ocpg_msda_bwd_value_sel_f32"):
        assert strict_hip.get(sym, 0) >= 1, (sym, strict_hip)


def test_transformer_head_dim_32(golden, dev, strict_hip):
    """deformable_transformer.py:134-217 (2 enc + 2 dec layers) at 2 heads x 32: MSDeformAttn <8> kernels, the decoder's
    self-attention through attn_smallk, residual / LayerNorm / FFN glue through csrc/fused_ln.hip."""
    mc.check_transformer(golden("transformer_d32"), dev, rtol=5e-4, atol=5e-5)
    for sym in ("ocpg_msda_fwd_f32", "ocpg_attn_smallk_fwd", "ocpg_attn_smallk_bwd", "ocpg_dropout_add_ln_fwd", "ocpg_bias_relu_dropout_fwd"):
        assert strict_hip.get(sym, 0) >= 1, (sym, strict_hip)
    # the encoder's self-attention: fused front end + path-selecting grad_value (round 4); the decoder's cross-attention: the generic entry points
    for sym in ("ocpg_msda_fused_fwd_f32", "ocpg_msda_fused_bwd_qproj_f32", "ocpg_msda_bwd_value_sel_f32", "ocpg_msda_bwd_f32"):
        assert strict_hip.get(sym, 0) >= 1, (sym, strict_hip)


@pytest.mark.parametrize("bt,q,c,h,w", [(2, 20, 256, 48, 80), (1, 5, 32, 40, 52), (3, 7, 10, 5, 9)])
def test_dynmask_kernel_vs_oracle(dev, bt, q, c, h, w):
    """ocpg_dynmask_fwd_f32 (+ its GEMM-shaped backward) against the literal restatement of ocpg.py:475-549 on the CPU:
    the config-#2 map with all 4 decoder layers' queries in one launch (multi-pixel-per-lane kernel), a map with a
    ragged last strip, and a tiny odd-channel case (one-pixel-per-lane kernel, C % 4 != 0)."""
    from oracle import ocpg_ref
    from ocpg_amd.models.ops.functions.dynmask_func import dynamic_mask
    gen = torch.Generator().manual_seed(bt * 1000 + q)
    feats = torch.randn(1, bt, c, h, w, generator=gen)
    params = torch.randn(1, bt * q, (c + 2) * 16 + 256 + 32, generator=gen) * 0.1
    refs = torch.rand(1, bt * q, 2, generator=gen)
    go = torch.randn(1, bt * q, 16, h, w, generator=gen)
    size = torch.tensor([h * 8 - 3, w * 8 - 5])
    cfg = {"dynamic_mask_channels": 16, "controller_layers": 2}
    f0, p0, r0 = (x.clone().requires_grad_(True) for x in (feats, params, refs))
    want = ocpg_ref.dynamic_mask_with_coords(cfg, f0, p0, r0, [size])
    (want * go).sum().backward()
    f1, p1, r1 = (x.to(dev).requires_grad_(True) for x in (feats, params, refs))
    refpix = r1 * torch.stack([size[1], size[0]]).float().to(dev)
    got = dynamic_mask(f1[0], p1[0], refpix[0], 8)
    (got * go[0].to(dev)).sum().backward()
    scale = want.abs().max().item()
    assert (got.cpu() - want[0]).abs().max().item() <= 2e-5 * scale + 1e-5
    for a, b_, name in ((f1.grad, f0.grad, "feats"), (p1.grad, p0.grad, "params"), (r1.grad, r0.grad, "ref")):
        err = (a.cpu() - b_).abs().max().item()
        assert err <= 1e-4 * b_.abs().max().item() + 1e-5, (name, err, b_.abs().max().item())


def test_matcher_criterion(golden, dev):
    mc.check_matcher_crit(golden("matcher_crit"), dev, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("tag", ["nopad", "pad"])
def test_train_step_matches_reference(golden, dev, tag):
    """fp32 parity mode: mask logits <= 1e-3 abs (north star), matcher indices bit-exact, 18 losses, all grad norms."""
    res = model_checks.run_train_step(golden("e2e_tiny"), tag, dev, rtol=1e-3, atol=1e-4)
    print(res)


@pytest.mark.parametrize("fixture,tag", [("e2e_d32", "pad"), ("e2e_cfg1", "nopad")])
def test_train_step_matches_reference_bench_kernels(golden, dev, fixture, tag, monkeypatch):
    """e2e_d32: the same step at head_dim 32 (hidden 64 / 2 heads) -> the encoder's MSDeformAttn runs the kernels of the
    BASELINE configurations (msda_fwd_fast<8>, column-tile scatter + gather-row backward), asserted below.
    e2e_cfg1: BASELINE config #1's shapes (ONE frame, 256x256, THREE feature levels, ONE query, hidden 256 / 8 heads)."""
    from ocpg_amd.models.ops.functions import ms_deform_attn_func as f
    seen = []
    real = f.lib

    class Spy:
        def __getattr__(self, name):
            fn = getattr(real(), name)

            def call(*a):
                rc = fn(*a)
                seen.append((name, rc))
                return rc
            return call
    monkeypatch.setattr(f, "lib", lambda: Spy())
    res = model_checks.run_train_step(golden(fixture), tag, dev, rtol=1e-3, atol=1e-4)
    print(res)
    names = {n for n, rc in seen if rc == 0}
    # the dedicated self-attention kernels served the encoder (rc 0 = launched, not "-2000 unsupported"): with 4 levels x 4 points the fused
    # front end (round 4) + the path-selecting grad_value entry, otherwise (config #1: 3 levels) the round-2 pair of backward entries
    fused = {"ocpg_msda_fused_fwd_f32", "ocpg_msda_fused_bwd_qproj_f32", "ocpg_msda_bwd_value_sel_f32"}
    assert fused <= names or {"ocpg_msda_bwd_value_f32", "ocpg_msda_bwd_locattn_f32"} <= names \
        or {"ocpg_msda_bwd_value_sel_f32", "ocpg_msda_bwd_locattn_f32"} <= names, names


@pytest.mark.parametrize("tag", ["nopad", "pad"])
def test_train_step_with_tagged_masks(golden, dev, tag):
    """Same reference vectors with the padding mask tagged by its host-known valid extents: memoised level masks / position
    encodings / valid ratios / reference grid; for 'nopad' also the transformer's no-padding shortcut (masked_fill skipped)."""
    model_checks.run_train_step(golden("e2e_tiny"), tag, dev, rtol=1e-3, atol=1e-4, tag_masks=True)
    model_checks.run_train_step(golden("e2e_tiny"), tag, dev, rtol=1e-3, atol=1e-4, tag_masks=True)      # second run: cache hits


def test_train_step_channels_last_layout(golden, dev):
    """The bench layout (channels-last convs: NHWC frozen-BN kernels, 1x1 convs as hipBLASLt GEMMs) against the same
    reference vectors, fp32, same bounds."""
    model_checks.run_train_step(golden("e2e_tiny"), "nopad", dev, rtol=1e-3, atol=1e-4, channels_last=True)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("bias", [False, True])
def test_pointwise_conv_as_gemm(dev, dtype, bias):
    """amp_cache.Conv2d's GEMM path for 1x1/stride-1 convs of channels-last maps == the convolution (MIOpen) path."""
    from ocpg_amd.models import amp_cache
    torch.manual_seed(0)
    conv = amp_cache.Conv2d(96, 160, 1, bias=bias).to(dev, dtype)
    x = torch.randn(3, 96, 13, 17, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
    go = torch.randn(3, 160, 13, 17, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
    res = []
    for on in (True, False):
        amp_cache.GEMM_1X1 = on
        try:
            xi = x.clone().requires_grad_(True)
            conv.zero_grad()
            y = conv(xi)
            assert (type(y.grad_fn).__name__ == "Conv1x1AsGemmBackward") == on
            y.backward(go)
            res.append([y.detach().float(), xi.grad.float(), conv.weight.grad.float()] + ([conv.bias.grad.float()] if bias else []))
        finally:
            amp_cache.GEMM_1X1 = True
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for a, b_ in zip(*res):
        assert a.shape == b_.shape and (a - b_).abs().max().item() <= tol * b_.abs().max().item() + 1e-6


def _probe_keep_mask(rows, cols, p, rng, dev):
    """The keep/scale pattern csrc/fused_ln.hip generates for (seed, offset) on a [rows, cols] tensor (1/(1-p) or 0)."""
    from ocpg_amd.models.ops.functions import fused_ln_func as f
    ones = torch.ones(rows, 1, device=dev)
    w = torch.ones(cols, 1, device=dev)
    return f.LinearBiasReluDropout.apply(ones, w, torch.zeros(cols, device=dev), p, rng, 1)


@pytest.mark.parametrize("xdtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,c,p", [(1023, 256, 0.0), (37, 1024, 0.0), (5, 20, 0.0), (777, 256, 0.1), (64, 2048, 0.3)])
def test_fused_dropout_add_layernorm(dev, xdtype, rows, c, p):
    """LayerNorm(res + dropout(x)) in one pass each way == the three-op formulation with the SAME mask (recovered from
    the generator through a probe call), outputs and all four gradients; p = 0 is the parity configuration."""
    from ocpg_amd.models.ops.functions import fused_ln_func as f
    torch.manual_seed(0)
    norm = torch.nn.LayerNorm(c).to(dev)
    norm.weight.data.uniform_(0.5, 1.5), norm.bias.data.normal_(0, 0.2)
    x = torch.randn(3, rows, c, device=dev).to(xdtype)
    res = torch.randn(3, rows, c, device=dev)
    go = torch.randn(3, rows, c, device=dev)
    rng = (1234567, 42)
    keep = _probe_keep_mask(3 * rows, c, p, rng, dev).view(3, rows, c) if p > 0 else torch.ones_like(res)
    if p > 0:
        frac = (keep > 0).float().mean().item()
        assert abs(frac - (1 - p)) < 0.02 and torch.all((keep == 0) | ((keep - 1 / (1 - p)).abs() < 1e-6))
    out = []
    for fused in (True, False):
        xi, ri = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
        norm.zero_grad()
        y = f.dropout_add_layer_norm(xi, ri, norm, p, rng) if fused else norm(ri + xi.float() * keep)
        y.backward(go)
        out.append([y.detach(), xi.grad.float(), ri.grad, norm.weight.grad.clone(), norm.bias.grad.clone()])
    tol = 2e-5 if xdtype == torch.float32 else 1e-2
    for a, b_, name in zip(out[0], out[1], ("y", "gx", "gres", "dgamma", "dbeta")):
        t = tol if name == "gx" else 2e-5
        assert (a - b_).abs().max().item() <= t * b_.abs().max().item() + 1e-6, (name, (a - b_).abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,p", [(8200, 0.0), (515, 0.1)])
def test_fused_linear_bias_relu_dropout(dev, dtype, rows, p):
    """dropout(relu(x W^T + b)): GEMM + one fused pass == the three-op formulation with the same mask; gradients incl. bias."""
    from ocpg_amd.models.ops.functions import fused_ln_func as f
    torch.manual_seed(0)
    k, c = 64, 1024 + 8
    x = torch.randn(rows, k, device=dev).to(dtype)
    w = (torch.randn(c, k, device=dev) * 0.2).to(dtype)
    b = torch.randn(c, device=dev).to(dtype)
    go = torch.randn(rows, c, device=dev).to(dtype)
    rng = (99, 7)
    keep = _probe_keep_mask(rows, c, p, rng, dev).to(dtype) if p > 0 else torch.ones(rows, c, device=dev, dtype=dtype)
    out = []
    for fused in (True, False):
        xi, wi, bi = (t.clone().requires_grad_(True) for t in (x, w, b))
        h = f.LinearBiasReluDropout.apply(xi, wi, bi, p, rng, 1 if rows < 4096 else 5) if fused else torch.relu(torch.nn.functional.linear(xi, wi, bi)) * keep
        h.backward(go)
        out.append([h.detach().float(), xi.grad.float(), wi.grad.float(), bi.grad.float()])
    for a, b_, name in zip(out[0], out[1], ("h", "gx", "gw", "gb")):
        if dtype == torch.float32:
            assert (a - b_).abs().max().item() <= 2e-5 * b_.abs().max().item() + 1e-6, (name, (a - b_).abs().max().item(), b_.abs().max().item())
        else:   # bf16 / fp16: the bias is added after (fused) vs before (addmm) the rounding of the GEMM result; a pre-activation within
            #     an ulp of zero flips its ReLU (~0.25 % of the units here: ~3-5 % of the gradient norm) -> norm-relative bound
            assert (a - b_).norm().item() <= 6e-2 * b_.norm().item(), (name, (a - b_).norm().item(), b_.norm().item())


@pytest.mark.parametrize("num_classes", [1, 7])
def test_hip_matcher_cost_equals_torch_formulas(dev, num_classes):
    """csrc/matcher.hip (one launch for all layers / clips / queries, strided mask view) against the matcher's tensor-op
    cost matrix: values to fp32 rounding, argmin identical; invalid frames, multi-class labels, a malformed box counted."""
    import synth
    from ocpg_amd.models import matcher as mm
    torch.manual_seed(0)
    m = mm.HungarianMatcher(cost_class=2, cost_bbox=5, cost_giou=2, cost_mask=2, cost_dice=5, num_classes=num_classes).to(dev)
    lr, b, t, q, H, W = 3, 2, 3, 5, 64, 96
    targets = synth.synthetic_targets(b, t, H, W, dev)
    targets[1]["valid"] = torch.tensor([1, 0, 1], device=dev)
    targets[1]["labels"] = torch.tensor([3, 0, 5], device=dev)
    gen = torch.Generator(device=dev).manual_seed(11)
    logits = torch.randn(lr, b, t, q, num_classes, device=dev, generator=gen)
    boxes = torch.rand(lr, b, t, q, 4, device=dev, generator=gen) * 0.4 + 0.2
    base = torch.randn(b, t, lr, q, H // 2, W // 2, device=dev, generator=gen) * 2        # the model's [b,t,l,q,...] layout
    masks = base.permute(2, 0, 1, 3, 4, 5)                                               # [l,b,t,q,h,w] strided view
    res = []
    for on in (True, False):
        mm.HIP_MATCHER = on
        try:
            res.append(m.cost_matrix_stacked(logits, boxes, masks, targets))
        finally:
            mm.HIP_MATCHER = True
    assert torch.allclose(res[0], res[1], rtol=2e-5, atol=2e-6), (res[0] - res[1]).abs().max()
    assert torch.equal(res[0].argmin(2), res[1].argmin(2))
    flag = mm._BOX_ERRORS[torch.device(dev) if not isinstance(dev, torch.device) else dev]
    before = int(flag.item())
    bad = boxes.clone()
    bad[0, 0, 0, 0, 2] = -0.1                     # negative width -> x1 < x0
    m.cost_matrix_stacked(logits, bad, masks, targets)
    assert int(flag.item()) > before
    flag.zero_()


def test_lfm_fused_spectral_gate(dev):
    """LFMResizeAdaptive with the fused spectral-gate kernel == the tensor-op formulation (complex product, real/imag split,
    cat): output, input gradient and every parameter gradient; first call (own Gaussian) and chained call (resized map)."""
    from ocpg_amd.models import modules
    torch.manual_seed(0)
    lfm = modules.LFMResizeAdaptive(32, 7).to(dev)
    x1 = torch.randn(3, 32, 23, 37, device=dev)
    x2 = torch.randn(3, 32, 12, 19, device=dev)
    res = []
    for on in (True, False):
        modules.FUSED_GATE = on
        try:
            a, b_ = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
            lfm.zero_grad()
            y1, g = lfm(a)
            y2, _ = lfm(b_, g)
            (y1.square().mean() + y2.square().mean()).backward()
            res.append([y1.detach(), y2.detach(), a.grad, b_.grad] + [p.grad.clone() for p in lfm.parameters()])
        finally:
            modules.FUSED_GATE = True
    for u, v in zip(*res):
        assert (u - v).abs().max().item() <= 2e-5 * v.abs().max().item() + 1e-7, ((u - v).abs().max().item(), v.abs().max().item())


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("xdt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("r,cin,cout,bias", [(50, 256, 256, True), (50, 256, 4, True), (200, 256, 4416, True), (10, 256, 1, False),
                                             (18, 256, 512, True), (50, 256, 384, True), (130, 128, 70, True), (1, 64, 64, True)])
def test_small_linear_kernel(dev, xdt, r, cin, cout, bias, relu):
    """csrc/small_linear.hip (few-row Linear, one launch each way) == autocast's cast + addmm and its autograd backward: same bf16
    operands, fp32 accumulation -> y, gx (in x's dtype), gw, gb to one bf16 ulp of the fp32 result."""
    from ocpg_amd.models import amp_cache
    g = torch.Generator(device=dev).manual_seed(r * 131 + cout)
    x = torch.randn(2, r // 2 if r % 2 == 0 else r, cin, device=dev, generator=g).to(xdt)
    x = x if r % 2 == 0 else x[:1]
    w = (torch.randn(cout, cin, device=dev, generator=g) * cin ** -0.5).bfloat16()
    b = torch.randn(cout, device=dev, generator=g).bfloat16() if bias else None
    go = torch.randn(*x.shape[:-1], cout, device=dev, generator=g).bfloat16()
    res = []
    for mine in (True, False):
        xi, wi = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        bi = b.clone().requires_grad_(True) if bias else None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if mine:
                assert amp_cache._small_linear_ok(xi, wi, bi)
                y = amp_cache.SmallLinearFunction.apply(xi, wi, bi, relu)
            else:
                y = torch.nn.functional.linear(xi, wi, bi)
                y = torch.relu(y) if relu else y
        assert y.dtype == torch.bfloat16
        grads = torch.autograd.grad((y.float() * go.float()).sum(), [xi, wi] + ([bi] if bias else []))
        assert grads[0].dtype == xdt
        res.append([y.float()] + [t.float() for t in grads])
    for name, a, b_ in zip(("y", "gx", "gw", "gb"), res[0], res[1]):
        tol = 2 ** -7 * b_.abs().max().item() + 1e-6
        assert (a - b_).abs().max().item() <= tol, (name, (a - b_).abs().max().item(), b_.abs().max().item())


@pytest.mark.parametrize("xdt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,c,h,w", [(10, 256, 45, 80), (2, 256, 50, 33), (1, 256, 25, 61), (3, 256, 23, 40), (2, 256, 6, 10), (1, 64, 1, 3), (2, 512, 17, 15)])
@pytest.mark.parametrize("cl_out", [False, True])
def test_groupnorm_channels_last_kernel(dev, xdt, n, c, h, w, cl_out, monkeypatch):
    """csrc/groupnorm.hip (channels-last map in its own dtype -> fp32 planes, or (cl_out, round 4) -> an fp32 channels-last map, with a
    non-contiguous gradient coming back; one launch each way) == F.group_norm in fp32 on the same
    stored values (what autocast computes for the reference's GroupNorm(32, 256), models/ocpg.py:108-119) and its autograd backward:
    y, dgamma, dbeta to fp32 rounding; dx to fp32 rounding for fp32 maps and to one ulp of the map's dtype otherwise.  Maps of >= 1 500
    pixels take the pixel-tiled kernels (two launches each way, ragged last tile included), smaller ones the one-launch kernels."""
    from ocpg_amd.models.ops.functions import groupnorm_func
    from ocpg_amd.models.ops.functions.groupnorm_func import GroupNorm, eligible
    monkeypatch.setattr(groupnorm_func, "CL_OUT", cl_out)
    gen = torch.Generator(device=dev).manual_seed(n * 1000 + c + h)
    gn = GroupNorm(c // 8, c).to(dev)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(c, device=dev, generator=gen))
        gn.bias.copy_(torch.randn(c, device=dev, generator=gen))
    x = (torch.randn(n, c, h, w, device=dev, generator=gen) * 1.5 + 0.7).to(xdt).contiguous(memory_format=torch.channels_last)
    go = torch.randn(n, c, h, w, device=dev, generator=gen)
    res = []
    for mine in (True, False):
        xi = x.clone(memory_format=torch.preserve_format).requires_grad_(True)
        gn.zero_grad()
        if mine:
            assert eligible(xi, gn)
            y = gn(xi)
            # channels-last only where the LFM behind it runs its own transforms on this map size (else planes, for rocFFT)
            want_cl = cl_out and bool(__import__("ocpg_amd._lib", fromlist=["lib"]).lib().ocpg_lfm_dft_supported(h, w))
            assert (y.permute(0, 2, 3, 1) if want_cl else y).is_contiguous() and y.dtype == torch.float32 and y.shape == (n, c, h, w)
        else:
            y = torch.nn.functional.group_norm(xi.float(), gn.num_groups, gn.weight, gn.bias, gn.eps)
        (y * go).sum().backward()
        if mine:
            assert xi.grad.dtype == xdt and xi.grad.is_contiguous(memory_format=torch.channels_last)
        res.append([y.detach(), xi.grad.float(), gn.weight.grad.clone(), gn.bias.grad.clone()])
    ulp = {torch.float32: 2e-5, torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10}[xdt]
    for name, a, b_, tol in zip(("y", "dx", "dgamma", "dbeta"), res[0], res[1], (2e-5, ulp, 1e-4, 1e-4)):
        bound = tol * b_.abs().max().item() + 1e-6
        assert (a - b_).abs().max().item() <= bound, (name, (a - b_).abs().max().item(), b_.abs().max().item())
    # a map that is not channels-last, or whose groups are not 8 channels wide, takes ATen's path (same module, same parameters)
    assert not eligible(x.contiguous(), gn)
    assert not eligible(x, torch.nn.GroupNorm(c // 4, c).to(dev))


@pytest.mark.parametrize("amp", [None, torch.bfloat16])
def test_lfm_channels_last_gate(dev, amp):
    """LFM with the gate output / inverse-FFT input in channels-last memory (csrc/spectral.hip c2p / p2c, 1x1 convs as GEMMs)
    == the NCHW formulation: output, input gradient, every parameter gradient; ragged sizes (channels and pixels not multiples of
    the 64 x 64 tile); plus the raw kernels against tensor ops."""
    from ocpg_amd.models import modules
    from ocpg_amd.models.ops.functions import spectral_func as sf
    torch.manual_seed(0)
    # raw kernels
    n, c, h, w = 2, 36, 5, 13
    spec = torch.complex(torch.randn(n, c, h, w, device=dev), torch.randn(n, c, h, w, device=dev)).requires_grad_(True)
    coef = torch.rand(n, device=dev).requires_grad_(True)
    high = torch.rand(h, w, device=dev)
    z = sf.spectral_gate_cl(spec, coef, high, torch.float32)
    assert z.shape == (n, 2 * c, h, w) and z.is_contiguous(memory_format=torch.channels_last)
    want = spec * (1 - coef.view(n, 1, 1, 1) * high)
    want = torch.cat([want.real, want.imag], 1)
    assert (z - want).abs().max().item() <= 1e-6
    go = torch.randn_like(want)
    for a, b_ in zip(torch.autograd.grad((z * go).sum(), (spec, coef)), torch.autograd.grad((want * go).sum(), (spec, coef))):
        assert (a - b_).abs().max().item() <= 2e-5 * b_.abs().max().item() + 1e-6
    y = torch.randn(n, 2 * c, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    cplx = sf.pair_to_complex(y)
    want_c = torch.complex(*torch.chunk(y, 2, dim=1))
    assert torch.equal(torch.view_as_real(cplx), torch.view_as_real(want_c))
    gc = torch.complex(torch.randn(n, c, h, w, device=dev), torch.randn(n, c, h, w, device=dev))
    ga, = torch.autograd.grad(torch.view_as_real(cplx * gc.conj()).select(-1, 0).sum(), y)
    gb, = torch.autograd.grad(torch.view_as_real(want_c * gc.conj()).select(-1, 0).sum(), y)
    assert (ga - gb).abs().max().item() <= 1e-6
    # the block
    lfm = modules.LFMResizeAdaptive(36, 7).to(dev)
    x1 = torch.randn(3, 36, 23, 37, device=dev)
    x2 = torch.randn(3, 36, 12, 19, device=dev)

    def run(on, dtype):
        modules.GATE_NHWC = on
        try:
            a, b_ = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
            lfm.zero_grad()
            with torch.autocast("cuda", dtype=dtype, enabled=dtype is not None):
                y1, g = lfm(a)
                y2, _ = lfm(b_, g)
            (y1.float().square().mean() + y2.float().square().mean()).backward()
            return [y1.detach().float(), y2.detach().float(), a.grad, b_.grad] + [p.grad.clone() for p in lfm.parameters()]
        finally:
            modules.GATE_NHWC = True
    if amp is None:
        for u, v in zip(run(True, None), run(False, None)):
            assert (u - v).abs().max().item() <= 3e-5 * v.abs().max().item() + 1e-7, ((u - v).abs().max().item(), v.abs().max().item())
    else:
        ref, nchw, cl = run(False, None), run(False, amp), run(True, amp)
        for r, a, b_ in zip(ref, nchw, cl):
            scale = r.abs().max().item()
            assert (b_ - r).abs().max().item() <= 1.5 * (a - r).abs().max().item() + 1e-2 * scale + 1e-7, \
                ((b_ - r).abs().max().item(), (a - r).abs().max().item(), scale)


@pytest.mark.parametrize("n,c,h,w", [(2, 36, 5, 13), (3, 70, 9, 14), (2, 64, 12, 20), (1, 130, 7, 16), (2, 256, 48, 80), (2, 8, 1, 6), (1, 5, 16, 1),
                                     (1, 40, 128, 3), (2, 33, 6, 121)])
def test_lfm_dft_kernels_equal_torch_fft(dev, n, c, h, w):
    """csrc/lfm_dft.hip (both LFM transforms on the channels-last map, round 4) against torch.fft on the same tensors, forward and every
    gradient: z = cat(Re, Im)(fft2(x) * (1 - coef * high)) and x + ifft2(complex(y[:, :C], y[:, C:])).real (models/modules.py:44-56).
    Odd / prime / composite lengths (5 = 1 x 5, 13 = 1 x 13, 14 = 2 x 7, 80 = 8 x 10, 121 = 11 x 11, 128 = 8 x 16), lengths 1, channel
    counts that are not multiples of the 64-channel slab."""
    from ocpg_amd.models.ops.functions import spectral_func as sf
    torch.manual_seed(h * 131 + w)
    assert sf.dft_supported(h, w)
    x = torch.randn(n, c, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    coef = torch.rand(n, device=dev).requires_grad_(True)
    high = torch.rand(h, w, device=dev)
    z = sf.lfm_spectrum(x, coef, high, torch.float32)
    assert z.shape == (n, 2 * c, h, w) and z.permute(0, 2, 3, 1).is_contiguous()
    spec = torch.fft.fft2(x) * (1 - coef.view(n, 1, 1, 1) * high)
    want = torch.cat([spec.real, spec.imag], 1)
    scale = want.abs().max().item()
    assert (z - want).abs().max().item() <= 2e-6 * scale * max(1.0, (h * w) ** 0.5 / 8), ((z - want).abs().max().item(), scale)
    go = torch.randn_like(want)
    for name, a, b_ in zip(("gx", "gcoef"), torch.autograd.grad((z * go).sum(), (x, coef)), torch.autograd.grad((want * go).sum(), (x, coef))):
        assert (a - b_).abs().max().item() <= 3e-5 * b_.abs().max().item() + 1e-6, (name, (a - b_).abs().max().item(), b_.abs().max().item())
    y = torch.randn(n, 2 * c, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = sf.lfm_inverse(y, x)
    want_r = x + torch.fft.ifft2(torch.complex(*torch.chunk(y, 2, dim=1)), s=(h, w)).real
    assert r.shape == want_r.shape and (r - want_r).abs().max().item() <= 3e-6 * want_r.abs().max().item()
    gr = torch.randn_like(want_r)
    for name, a, b_ in zip(("gy", "gx"), torch.autograd.grad((r * gr).sum(), (y, x)), torch.autograd.grad((want_r * gr).sum(), (y, x))):
        assert (a - b_).abs().max().item() <= 3e-5 * b_.abs().max().item() + 1e-7, (name, (a - b_).abs().max().item(), b_.abs().max().item())
    assert not sf.dft_supported(23, 37) and not sf.dft_supported(8, 34) and not sf.dft_supported(130, 8)


@pytest.mark.parametrize("amp", [None, torch.bfloat16])
def test_text_gate_batch_first_equals_token_major(dev, amp):
    """VisionLanguageFusionModule.forward_batch_first (round 4: the visual tokens are the channels-last map's own memory [b, (t h w), c],
    one short-key attention launch per clip) == forward() on the reference's token-major [t, h, w, b, c] layout
    (models/segmentation.py:95-113 of the reference): output and the gradients of both inputs and all parameters, with a padded text."""
    from ocpg_amd import _lib
    from ocpg_amd.models.segmentation import VisionLanguageFusionModule
    torch.manual_seed(3)
    b, t, h, w, c, lk = 3, 2, 5, 7, 256, 9
    fuse = VisionLanguageFusionModule(c, 8).to(dev)
    vis = torch.randn(b, t * h * w, c, device=dev)
    text = torch.randn(lk, b, c, device=dev)
    pos = torch.randn(lk, b, c, device=dev)
    pad = torch.zeros(b, lk, dtype=torch.bool, device=dev)
    pad[0, 6:], pad[2, 4:] = True, True
    go = torch.randn(b, t * h * w, c, device=dev)
    res = []
    for bf in (True, False):
        v, tx = vis.clone().requires_grad_(True), text.clone().requires_grad_(True)
        fuse.zero_grad()
        calls = _lib.census(True)
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            if bf:
                out = fuse.forward_batch_first(v, tx, pad, pos)
            else:
                tok = v.view(b, t, h, w, c).permute(1, 2, 3, 0, 4)
                out = fuse(visual=tok, text=tx, text_key_padding_mask=pad, text_pos=pos).view(t * h * w, b, c).transpose(0, 1)
        (out.float() * go).sum().backward()
        _lib.census(False)
        assert calls.get("ocpg_attn_smallk_fwd", 0) == (b if bf else 1) and calls.get("ocpg_attn_smallk_bwd", 0) == (b if bf else 1), calls
        res.append([out.detach().float(), v.grad, tx.grad] + [p.grad.clone() for p in fuse.parameters()])
    tol = 2e-5 if amp is None else 2e-2
    for i, (a, r) in enumerate(zip(*res)):
        assert (a - r).abs().max().item() <= tol * r.abs().max().item() + 1e-6, (i, (a - r).abs().max().item(), r.abs().max().item())


@pytest.mark.parametrize("amp", [None, torch.bfloat16])
def test_lfm_block_own_transforms_equal_library_fft(dev, amp):
    """The LFM block with csrc/lfm_dft.hip (default) == the same block on rocFFT between the transposing kernels (OCPG_LFM_DFT=0), two
    chained levels (the second resizes the first's Gaussian), output / input gradients / every parameter gradient; under autocast both
    are held against the fp32 run, the new path no further from it than the library path + the bf16 storage noise."""
    from ocpg_amd import _lib
    from ocpg_amd.models import modules
    torch.manual_seed(1)
    lfm = modules.LFMResizeAdaptive(72, 7).to(dev)
    x1 = torch.randn(3, 72, 24, 40, device=dev).contiguous(memory_format=torch.channels_last)
    x2 = torch.randn(3, 72, 12, 20, device=dev)

    def run(on, dtype):
        modules.DFT_CL = on
        try:
            a, b_ = x1.clone(memory_format=torch.preserve_format).requires_grad_(True), x2.clone().requires_grad_(True)
            lfm.zero_grad()
            calls = _lib.census(True)
            with torch.autocast("cuda", dtype=dtype, enabled=dtype is not None):
                y1, g = lfm(a)
                y2, _ = lfm(b_, g)
            (y1.float().square().mean() + y2.float().square().mean()).backward()
            _lib.census(False)
            assert (calls.get("ocpg_lfm_spectrum_fwd", 0), calls.get("ocpg_lfm_spectrum_inv", 0)) == ((4, 4) if on else (0, 0)), calls
            return [y1.detach().float(), y2.detach().float(), a.grad, b_.grad] + [p.grad.clone() for p in lfm.parameters()]
        finally:
            modules.DFT_CL = True
    if amp is None:
        for u, v in zip(run(True, None), run(False, None)):
            assert (u - v).abs().max().item() <= 3e-5 * v.abs().max().item() + 1e-7, ((u - v).abs().max().item(), v.abs().max().item())
    else:
        ref, lib_, own = run(False, None), run(False, amp), run(True, amp)
        for r, a, b_ in zip(ref, lib_, own):
            scale = r.abs().max().item()
            assert (b_ - r).abs().max().item() <= 1.5 * (a - r).abs().max().item() + 1e-2 * scale + 1e-7, \
                ((b_ - r).abs().max().item(), (a - r).abs().max().item(), scale)


@pytest.mark.parametrize("amp", [None, torch.bfloat16])
def test_lfm_laplace_mean_without_convolution(dev, amp):
    """LFM coefficient branch: nine window means + one small matrix product (csrc/lfm.hip) == conv3x3(valid) followed by the
    spatial mean (reference models/modules.py:36-39): block output, input gradient and every parameter gradient; the raw
    identity is also checked on odd sizes including the 3x3 map (a single output position)."""
    from ocpg_amd.models import modules
    from ocpg_amd.models.ops.functions.spectral_func import conv3x3_valid_spatial_mean
    torch.manual_seed(0)
    for (n, c, h, w) in ((2, 5, 3, 3), (3, 7, 4, 9), (2, 16, 23, 37)):
        x = torch.randn(n, c, h, w, device=dev, requires_grad=True)
        wt = torch.randn(6, c, 3, 3, device=dev, requires_grad=True)
        bs = torch.randn(6, device=dev, requires_grad=True)
        want = torch.nn.functional.conv2d(x, wt, bs).mean(dim=(2, 3))
        got = conv3x3_valid_spatial_mean(x, wt, bs, False)
        assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item() + 1e-6
        go = torch.randn_like(want)
        for a, b_ in zip(torch.autograd.grad((got * go).sum(), (x, wt, bs)), torch.autograd.grad((want * go).sum(), (x, wt, bs))):
            assert (a - b_).abs().max().item() <= 2e-5 * b_.abs().max().item() + 1e-6
    lfm = modules.LFMResizeAdaptive(32, 7).to(dev)
    x1 = torch.randn(3, 32, 23, 37, device=dev)
    x2 = torch.randn(3, 32, 12, 19, device=dev)
    def run(on, dtype):
        modules.LAPLACE_MEAN = on
        try:
            a, b_ = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
            lfm.zero_grad()
            with torch.autocast("cuda", dtype=dtype, enabled=dtype is not None):
                y1, g = lfm(a)
                y2, _ = lfm(b_, g)
            (y1.float().square().mean() + y2.float().square().mean()).backward()
            return [y1.detach().float(), y2.detach().float(), a.grad, b_.grad] + [p.grad.clone() for p in lfm.parameters()]
        finally:
            modules.LAPLACE_MEAN = True
    if amp is None:
        for u, v in zip(run(True, None), run(False, None)):
            assert (u - v).abs().max().item() <= 2e-5 * v.abs().max().item() + 1e-7, ((u - v).abs().max().item(), v.abs().max().item())
    else:
        # bf16: the convolution rounds every one of its outputs (and output gradients) to bf16 before the mean, the window form
        # does not: judged against the fp32 block, the window form must be at least as close as the convolution form
        ref, conv, mine = run(False, None), run(False, amp), run(True, amp)
        for r, cv, mn in zip(ref, conv, mine):
            scale = r.abs().max().item()
            assert (mn - r).abs().max().item() <= 1.5 * (cv - r).abs().max().item() + 1e-2 * scale + 1e-7, \
                ((mn - r).abs().max().item(), (cv - r).abs().max().item(), scale)


@pytest.mark.parametrize("num_classes", [1, 7])
def test_hip_det_losses_equal_torch_formulas(dev, num_classes):
    """csrc/det_loss.hip (focal classification + L1 + GIoU, all layers per launch, GIoU gradient by forward-mode duals)
    against the criterion's tensor-op formulation: the three losses per layer and the gradients w.r.t. logits and boxes;
    invalid frames, multi-class labels, boxes that do not overlap the target (hull term active) and that contain it."""
    import synth
    import cases
    from ocpg_amd.models import build_model, criterion as crit_mod
    torch.manual_seed(0)
    args = cases.default_args(device=str(dev), **cases.TINY)
    _, crit, _ = build_model(args)
    crit.to(dev)
    crit.num_classes = num_classes
    lr, b, t, q = 3, 2, 3, 5
    targets = synth.synthetic_targets(b, t, 32, 48, dev)
    targets[1]["valid"] = torch.tensor([1, 0, 1], device=dev)
    targets[1]["labels"] = torch.tensor([3, 0, 5], device=dev)
    gen = torch.Generator(device=dev).manual_seed(5)
    logits = torch.randn(lr, b, t, q, num_classes, device=dev, generator=gen) * 2
    boxes = torch.rand(lr, b, t, q, 4, device=dev, generator=gen) * 0.3 + 0.1
    boxes[0, 0, :, :, :2] = 0.85                       # far from the target: no intersection
    boxes[1, 1, :, :, :] = torch.tensor([0.375, 0.375, 0.6, 0.6], device=dev)      # contains the target
    src = torch.randint(0, q, (lr, b), device=dev, generator=gen)
    nb = torch.tensor(5.0, device=dev)
    w = torch.tensor([[1.0, 2.0, 3.0]], device=dev).t() * torch.arange(1, lr + 1, device=dev)
    res = []
    for on in (True, False):
        crit_mod.HIP_DET_LOSSES = on
        try:
            lg, bx = logits.clone().requires_grad_(True), boxes.clone().requires_grad_(True)
            if on:
                from ocpg_amd.models.ops.functions import mask_loss_func
                valid = torch.stack([tg["valid"] for tg in targets])
                labels = None if num_classes == 1 else torch.stack([tg["labels"] for tg in targets])
                det = mask_loss_func.det_losses(lg, bx, src, valid, labels, torch.stack([tg["boxes"] for tg in targets]), nb, crit.focal_alpha)
            else:
                ce = crit._labels_stacked(lg, src, targets, nb)
                l1, gi = crit._boxes_stacked(bx, src, targets, nb)
                det = torch.stack([ce, l1, gi])
            g = torch.autograd.grad((det * w).sum(), (lg, bx))
            res.append((det.detach(), g))
        finally:
            crit_mod.HIP_DET_LOSSES = True
    (d1, g1), (d0, g0) = res
    assert torch.allclose(d1, d0, rtol=2e-5, atol=1e-6), (d1, d0)
    for a, b_, name in zip(g1, g0, ("logits", "boxes")):
        assert (a - b_).abs().max().item() <= 5e-5 * b_.abs().max().item() + 1e-7, (name, (a - b_).abs().max().item(), b_.abs().max().item())


@pytest.mark.parametrize("case", ["regular", "degenerate"])
def test_hip_mask_losses_equal_torch_formulas(dev, case):
    """csrc/levelset.hip + csrc/proj.hip (all layers per launch) against the criterion's own tensor-op restatement of
    segmentation.py:203-315 on the same inputs: the six mask losses of 3 stacked layers and the gradients w.r.t. both mask
    resolutions and the level-set features.  'degenerate': saturated logits (ties in the row/column maxima, zero
    foreground mass -> the 1e-5 clamps are active), a frame whose box region is empty, odd sizes."""
    import synth
    from ocpg_amd.models import build_model, criterion as crit_mod
    import cases
    torch.manual_seed(0)
    args = cases.default_args(device=str(dev), **cases.TINY)
    _, crit, _ = build_model(args)
    crit.to(dev)
    lr, b, t = 3, 2, 3
    H, W = (64, 96) if case == "regular" else (32, 64)
    targets = synth.synthetic_targets(b, t, H, W, dev)
    targets[0]["weights"] = torch.rand(t, H, W, device=dev) * targets[0]["masks"]
    if case == "degenerate":
        targets[1]["boxes"] = torch.tensor([[0.375, 0.375, 0.25, 0.25], [0.5, 0.5, 0.0, 0.0], [0.2, 0.7, 0.3, 0.5]], device=dev)
    gen = torch.Generator(device=dev).manual_seed(7)
    pm = torch.randn(lr, b, t, H, W, device=dev, generator=gen) * 3
    pml = torch.randn(lr, b, t, H // 2, W // 2, device=dev, generator=gen) * 3
    if case == "degenerate":
        pm[0, 0] = 40.0          # saturated: every pixel ties for the maximum, p (1 - p) == 0
        pm[1, 1, 0] = -40.0      # zero foreground mass in a frame
        pml[0, 1] = -40.0
        pml[2, 0, 1] = 40.0
    ls = torch.randn(b, t, 12, H // 2, W // 2, device=dev, generator=gen)
    warm = torch.tensor([0.3, 0.6, 1.0], device=dev)
    nb = torch.tensor(float(b * t), device=dev)
    res = []
    for on in (True, False):
        crit_mod.HIP_MASK_LOSSES = on
        try:
            leaves = [x.clone().requires_grad_(True) for x in (pm, pml, ls)]
            d, _ = crit._masks_stacked(leaves[0], leaves[1], leaves[2], targets, nb, warm)
            wts = {k: 1.0 + 0.1 * i for i, k in enumerate(sorted(d))}
            total = sum((d[k] * torch.arange(1, lr + 1, device=dev)).sum() * wts[k] for k in d)
            grads = torch.autograd.grad(total, leaves)
            res.append(({k: v.detach() for k, v in d.items()}, grads))
        finally:
            crit_mod.HIP_MASK_LOSSES = True
    (d1, g1), (d0, g0) = res
    for k in d0:
        assert torch.allclose(d1[k], d0[k], rtol=2e-4, atol=1e-6), (k, d1[k], d0[k])
    for a, b_, name in zip(g1, g0, ("pm", "pml", "ls_features")):
        err = (a - b_).abs().max().item()
        assert err <= 2e-4 * b_.abs().max().item() + 1e-9, (name, err, b_.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("stride,dil", [(1, 1), (2, 1), (1, 2)])
def test_conv3x3_bn_relu_fused_epilogue(dev, dtype, stride, dil):
    """3x3 conv + frozen BN + ReLU as im2col (HIP) + one GEMM with the BN/ReLU epilogue (the path taken for >= 256 channels on
    small maps) against MIOpen conv + the bn_act kernel: output, input gradient, weight gradient."""
    from ocpg_amd.models import amp_cache, backbone
    torch.manual_seed(1)
    conv = amp_cache.Conv2d(256, 64, 3, stride=stride, padding=dil, dilation=dil, bias=False).to(dev)
    conv.to(memory_format=torch.channels_last)
    bn = backbone.FrozenBatchNorm2d(64).to(dev)
    bn.weight.uniform_(0.5, 1.5), bn.bias.normal_(0, 0.1), bn.running_mean.normal_(0, 0.1), bn.running_var.uniform_(0.5, 1.5)
    conv, bn = conv.to(dtype), bn.to(dtype)
    x = torch.randn(3, 256, 13, 18, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
    res = []
    for fused in (True, False):
        backbone.FUSED_CONV3X3_BN = fused
        try:
            xi = x.clone().requires_grad_(True)
            conv.zero_grad()
            y = backbone.conv_bn_act(conv, bn, xi, None, True)
            assert (type(y.grad_fn).__name__ == "Conv3x3BNActBackward") == fused
            go = torch.randn(y.shape, device=dev, dtype=dtype, generator=torch.Generator(device=dev).manual_seed(2))
            y.backward(go)
            res.append([y.detach().float(), xi.grad.float(), conv.weight.grad.float()])
        finally:
            backbone.FUSED_CONV3X3_BN = False       # the module default (opt-in path)
    for a, b_ in zip(*res):
        assert a.shape == b_.shape
        if dtype == torch.float32:
            assert (a - b_).abs().max().item() <= 5e-5 * b_.abs().max().item() + 1e-6
        else:       # bf16: ReLU flips at pre-activations within an ulp of zero (see test_bottleneck_fused_conv_bn_act)
            assert (a - b_).norm().item() <= 5e-2 * b_.norm().item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("project", [False, True])
def test_bottleneck_fused_conv_bn_act(dev, dtype, project):
    """Bottleneck with the fused 1x1-conv + frozen-BN (+ skip) + ReLU nodes (plan-cached hipBLASLt GEMM + bn_act in place)
    against the same block run module by module (MIOpen convs + separate bn_act): outputs and every gradient."""
    from ocpg_amd.models import amp_cache, backbone
    torch.manual_seed(3)
    blk = backbone.Bottleneck(64 if project else 128, 32, 1, 1, project).to(dev)
    for m in blk.modules():
        if isinstance(m, backbone.FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.5), m.bias.normal_(0, 0.1), m.running_mean.normal_(0, 0.1), m.running_var.uniform_(0.5, 1.5)
        if isinstance(m, torch.nn.Conv2d):
            m.to(memory_format=torch.channels_last)
    blk = blk.to(dtype)
    x = torch.randn(3, 64 if project else 128, 19, 23, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
    go = torch.randn(3, 128, 19, 23, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
    res = []
    for fused in (True, False):
        backbone.FUSED_CONV_BN, gemm = fused, amp_cache.GEMM_1X1
        amp_cache.GEMM_1X1 = fused
        try:
            xi = x.clone().requires_grad_(True)
            blk.zero_grad()
            y = blk(xi)
            assert (type(y.grad_fn).__name__ == "Conv1x1BNActBackward") == fused
            y.backward(go)
            res.append([y.detach().float(), xi.grad.float()] + [p.grad.float() for p in blk.parameters()])
        finally:
            backbone.FUSED_CONV_BN, amp_cache.GEMM_1X1 = True, gemm
    assert len(res[0]) == len(res[1]) >= 5
    for a, b_ in zip(*res):
        assert a.shape == b_.shape
        if dtype == torch.float32:
            assert (a - b_).abs().max().item() <= 2e-5 * b_.abs().max().item() + 1e-6
        else:   # bf16: a pre-activation within an ulp of zero flips its ReLU between the two paths and moves single gradient
            #     entries by O(1) (measured: max-abs 2.2 of 4.7 with identical fp32 results) -> norm-relative bound
            assert (a - b_).norm().item() <= 5e-2 * b_.norm().item()


@pytest.mark.parametrize("cin,width,stride,project", [(512, 128, 1, False), (256, 128, 2, True)])
def test_bottleneck_premasked_input_gradient(dev, cin, width, stride, project):
    """Round 4: conv2's input-gradient kernel (conv3x3_mfma<DGRAD>) applies conv1's frozen-BN + ReLU backward in its epilogue and conv1's
    backward skips its bn_act_bwd launch (conv_bn_func.PREMASK).  Same bottleneck, bf16 channels-last, with the fusion on and off: output
    identical, every gradient equal up to the one bf16 rounding the fusion removes; with it ON the census shows one bn_act_bwd less."""
    from ocpg_amd import _lib
    from ocpg_amd.models import backbone
    from ocpg_amd.models.ops.functions import conv_bn_func
    torch.manual_seed(5)
    blk = backbone.Bottleneck(cin, width, stride, 1, project).to(dev)
    for m in blk.modules():
        if isinstance(m, backbone.FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.5), m.bias.normal_(0, 0.1), m.running_mean.normal_(0, 0.1), m.running_var.uniform_(0.5, 1.5)
        if isinstance(m, torch.nn.Conv2d):
            m.to(memory_format=torch.channels_last)
    blk = blk.to(torch.bfloat16)
    x = torch.randn(2, cin, 14, 18, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ho, wo = (14 - 1) // stride + 1, (18 - 1) // stride + 1
    go = torch.randn(2, width * 4, ho, wo, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res, counts = [], []
    for on in (True, False):
        old = conv_bn_func.PREMASK
        conv_bn_func.PREMASK = on
        try:
            conv_bn_func.reset_skip_tokens()
            xi = x.clone().requires_grad_(True)
            blk.zero_grad()
            calls = _lib.census(True)
            y = blk(xi)
            y.backward(go)
            torch.cuda.synchronize()
            counts.append(dict(calls))
            _lib.census(False)
            res.append([y.detach().float(), xi.grad.float()] + [p.grad.float() for p in blk.parameters()])
        finally:
            conv_bn_func.PREMASK = old
    dgrad = lambda c: c.get("ocpg_conv3x3_mfma_dgrad_w", 0) + c.get("ocpg_conv3x3_mfma_dgrad_masked", 0)      # noqa: E731  (own / transposed weight)
    assert dgrad(counts[0]) == 1 and dgrad(counts[1]) == 1, counts
    assert counts[0].get("ocpg_bn_act_bwd", 0) == counts[1].get("ocpg_bn_act_bwd", 0) - 1, counts
    assert torch.equal(res[0][0], res[1][0])
    for a, b_ in zip(res[0][1:], res[1][1:]):
        assert (a - b_).norm().item() <= 1e-2 * b_.norm().item(), ((a - b_).norm().item(), b_.norm().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("stride,dil,bias,hw", [(1, 1, False, (13, 17)), (2, 1, True, (13, 16)), (2, 1, False, (12, 17)), (1, 2, True, (9, 11))])
def test_conv3x3_as_im2col_gemm(dev, dtype, stride, dil, bias, hw):
    """amp_cache.Conv2d's im2col (HIP) + GEMM path for 3x3 convs of channels-last maps == the convolution (MIOpen) path:
    odd/even sizes under stride 2, dilation 2 (the reference's --dilation layer4), bias, three storage types."""
    from ocpg_amd.models import amp_cache
    torch.manual_seed(0)
    conv = amp_cache.Conv2d(32, 48, 3, stride=stride, padding=dil, dilation=dil, bias=bias).to(dev, dtype).to(memory_format=torch.channels_last)
    x = torch.randn(3, 32, *hw, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
    from ocpg_amd.models.ops.functions import conv_gemm_func
    res = []
    for on in (True, False):
        amp_cache.GEMM_3X3 = conv_gemm_func.ALWAYS = on
        try:
            xi = x.clone().requires_grad_(True)
            conv.zero_grad()
            y = conv(xi)
            assert (type(y.grad_fn).__name__ == "Conv3x3AsGemmBackward") == on
            go = torch.randn(y.shape, device=dev, dtype=dtype, generator=torch.Generator(device=dev).manual_seed(5))
            y.backward(go)
            res.append([y.detach().float(), xi.grad.float(), conv.weight.grad.float()] + ([conv.bias.grad.float()] if bias else []))
        finally:
            amp_cache.GEMM_3X3, conv_gemm_func.ALWAYS = True, False
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for a, b_ in zip(*res):
        assert a.shape == b_.shape and (a - b_).abs().max().item() <= tol * b_.abs().max().item() + 1e-6


@pytest.mark.parametrize("dtype,rows", [(torch.float32, 10200), (torch.bfloat16, 9600), (torch.float32, 8191 * 2 + 1)])
def test_token_linear_row_split_weight_gradient(dev, dtype, rows):
    """amp_cache.linear (row-split batched GEMM for the weight gradient) == F.linear, incl. a row count with no divisor."""
    from ocpg_amd.models import amp_cache
    import torch.nn.functional as F
    torch.manual_seed(1)
    x = torch.randn(2, rows // 2, 64, device=dev, dtype=dtype)
    w = (torch.randn(48, 64, device=dev, dtype=dtype) * 0.1).requires_grad_(True)
    b = torch.randn(48, device=dev, dtype=dtype).requires_grad_(True)
    go = torch.randn(2, rows // 2, 48, device=dev, dtype=dtype)
    res = []
    for fn in (amp_cache.linear, F.linear):
        xi = x.clone().requires_grad_(True)
        y = fn(xi, w, b)
        res.append([y.detach().float()] + [g.float() for g in torch.autograd.grad((y.float() * go.float()).sum(), (xi, w, b))])
    assert type(amp_cache.linear(x.clone().requires_grad_(True), w, b).grad_fn).__name__ == "ViewBackward0"
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for a, b_ in zip(*res):
        assert (a - b_).abs().max().item() <= tol * b_.abs().max().item() + 1e-6


def test_inference_loop_and_collate_match_reference(golden, dev):
    """Rows f3 / f4 on the GPU: the video inference loop (clip chopping, best query, un-pad, resize, sigmoid, object merge) against
    the outputs of the reference's own inference_davis.py:203-261 statements; collate against the reference's util/misc.py."""
    model_checks.check_collate(golden("infer_collate"))
    # mask probabilities: |d sigmoid| <= |d logit| / 4 with mask logits held to 1e-3; labels: threshold / argmax ties may flip a pixel
    model_checks.check_inference_loop(golden("infer_collate"), dev, atol_logits=2e-4, atol_masks=5e-4, label_mismatch=1e-4)


@pytest.mark.parametrize("tag", ["nopad", "pad"])
def test_eval_tail_matches_reference(golden, dev, tag):
    model_checks.run_eval(golden("e2e_tiny"), tag, dev, rtol=1e-3, atol=1e-4)


# ---- Video-Swin backbone (BASELINE configs #4 / #5) ----------------------------------------------------------
def test_swin_window_attention_and_masks(golden, dev):
    import swin_checks as sc
    sc.check_window_attention(golden("swin3d"), dev, rtol=5e-4, atol=5e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_swin_window_attention_full_window_n392(golden, dev, dtype):
    """BASELINE config #5's attention shape (N = 392) in fp32 (csrc/win_attn.hip), in fp16 storage (the reference's --amp) and in bf16
    (config #4's dtype) -- the 16-bit ones through the matrix-core kernels (csrc/win_attn_mfma.hip), which the census checks."""
    import swin_checks as sc
    from ocpg_amd import _lib
    c = _lib.census(True)
    try:
        sc.check_window_attention_n392(golden("swin_n392"), dev, dtype)
    finally:
        _lib.census(False)
    assert c.get("ocpg_win_attn_fwd", 0) == 2, c
    assert c.get("ocpg_win_attn_bwd_mfma" if dtype != torch.float32 else "ocpg_win_attn_bwd", 0) == 2 and \
        (dtype == torch.float32 or "ocpg_win_attn_bwd" not in c), c


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_swin_window_attention_mfma_vs_reference_vectors(golden, dev, dtype):
    """The matrix-core window attention on the reference's own swin3d.npz vectors (245-token clamped window, shifted-window regions)."""
    import swin_checks as sc
    from ocpg_amd import _lib
    c = _lib.census(True)
    try:
        sc.check_window_attention_16bit(golden("swin3d"), dev, dtype)
    finally:
        _lib.census(False)
    assert c.get("ocpg_win_attn_bwd_mfma", 0) == 2 and "ocpg_win_attn_bwd" not in c, c


def test_swin_block_and_backbone(golden, dev):
    import swin_checks as sc
    sc.check_block_and_merging(golden("swin3d"), dev, rtol=5e-4, atol=1e-4)
    sc.check_backbone(golden("swin3d"), dev, rtol=1e-3, atol=2e-4)


def test_e2e_with_video_swin(golden, dev):
    import swin_checks as sc
    sc.check_e2e_swin(golden("e2e_swin"), dev, rtol=1e-3, atol=1e-4)


def test_fused_autocast_param_cast_equals_per_op_casts(golden, dev):
    """amp_cache (one fused cast of all parameters per forward) must give what autocast's own per-op casts give.  The
    casts themselves are checked bit-exactly; the end-to-end bf16 step is not reproducible run to run on this stack
    (library GEMM/conv kernels: ~2e-2 norm-relative on the tiny model's mask logits between two identical runs), so the
    on-vs-off difference is bounded by the measured off-vs-off noise floor."""
    import cases
    from ocpg_amd.models import amp_cache
    from ocpg_amd.util.misc import NestedTensor
    g = golden("e2e_tiny")
    meta = g.meta

    def run(enabled):
        amp_cache.ENABLED = enabled
        # this test isolates the CAST: kernels that only exist on the cached-copy side (the split-K neck convolution, round 4) would add their
        # own summation-order differences to the on-vs-off comparison; they have their own tests (test_conv3x3_splitk_kernel)
        splitk, amp_cache.SPLITK_3X3 = amp_cache.SPLITK_3X3, False
        try:
            torch.manual_seed(0)
            args, model, crit = model_checks.build_product(meta, dev)
            B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
            x, mask, targets = cases.e2e_inputs(B, T, H, W, meta["nopad_sizes"], dev)
            model.train(), crit.train()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(NestedTensor(x, mask), model_checks.text_for(B, dev), targets)
                losses, *_ = crit(out, targets)
                total = sum(losses[k] * crit.weight_dict[k] for k in losses if k in crit.weight_dict)
            total.backward()
            gn = {k: p.grad.float().clone() for k, p in model.named_parameters() if p.grad is not None}
            if enabled:     # the fused cast itself: every cached copy is the bf16 rounding of its parameter, bit for bit
                with amp_cache.scope(model), torch.autocast("cuda", dtype=torch.bfloat16):
                    for k, p in model.named_parameters():
                        c = amp_cache.lookup(p)
                        if c is not p:
                            assert c.dtype == torch.bfloat16 and torch.equal(c, p.detach().to(torch.bfloat16)), k
            return out["pred_masks"].detach().float().cpu(), total.item(), gn
        finally:
            amp_cache.ENABLED = True
            amp_cache.SPLITK_3X3 = splitk

    (m0, t0, g0), (m0b, t0b, g0b), (m1, t1, g1) = run(False), run(False), run(True)
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-20))
    floor_m, floor_t = rel(m0b, m0), abs(t0b - t0) / abs(t0)
    assert rel(m1, m0) <= 3 * floor_m + 1e-3, (rel(m1, m0), floor_m)
    assert abs(t1 - t0) <= 3 * floor_t * abs(t0) + 1e-2 * abs(t0), (t1, t0, t0b)
    assert set(g1) == set(g0)
    # gradients: some (LFM gates, deep backbone convs) move by tens of percent between two IDENTICAL bf16 runs, so compare the
    # on-vs-off change of every parameter's gradient with its own off-vs-off change
    d_on = {k: rel(g1[k], g0[k]) for k in g0 if g0[k].norm() > 0}
    d_off = {k: rel(g0b[k], g0[k]) for k in d_on}
    bad = [(k, d_on[k], d_off[k]) for k in d_on if d_on[k] > 4 * d_off[k] + 0.1]
    assert len(bad) <= 0.05 * len(d_on), bad[:5]
    med = lambda d: sorted(d.values())[len(d) // 2]
    assert med(d_on) <= 3 * med(d_off) + 2e-2, (med(d_on), med(d_off))


def test_matcher_and_criterion_kernels_at_config2_size_vs_oracle(dev):
    """csrc/matcher.hip, det_loss.hip, proj.hip, levelset.hip, masked_ce.hip at BASELINE config #2 size (2 clips x 5 frames
    x 384x640, 5 queries, 4 decoder layers) against the ORACLE's restatement of matcher.py / criterion.py /
    segmentation.py (oracle/ocpg_ref.py: matcher, criterion), not against the product's own tensor-op path: matched
    indices bit-exact for every layer, all 36 losses, and the gradient of the weighted total w.r.t. every prediction."""
    import cases
    import synth
    from oracle import ocpg_ref
    from ocpg_amd.models import build_model
    B, T, Q, H, W, LR = 2, 5, 5, 384, 640, 4
    args = cases.default_args(device=str(dev), backbone="resnet50", num_frames=T, num_queries=Q, dec_layers=LR)
    model, crit, _ = build_model(args)
    crit.to(dev).train()
    cfg = ocpg_ref.cfg_from_args(args)
    targets = synth.synthetic_targets(B, T, H, W)
    targets[1]["valid"] = torch.tensor([1, 0, 1, 1, 1])
    targets[1]["boxes"] = torch.tensor([[0.3, 0.4, 0.2, 0.3], [0.6, 0.5, 0.3, 0.2], [0.5, 0.5, 0.4, 0.4], [0.4, 0.6, 0.3, 0.5],
                                        [0.55, 0.45, 0.25, 0.35]])
    targets[0]["weights"] = synth.rand("c2_heat", (T, H, W), uniform=True) * targets[0]["masks"]
    g = {k: synth.rand("c2_" + k, s, uniform=u) for k, s, u in (
        ("logits", (LR, B, T, Q, 1), False), ("boxes", (LR, B, T, Q, 4), True), ("qmasks", (LR, B, T, Q, H // 2, W // 2), False),
        ("pm", (LR, B, T, H, W), False), ("pml", (LR, B, T, H // 2, W // 2), False), ("ls", (B, T, 12, H // 2, W // 2), False))}
    g["boxes"] = g["boxes"] * 0.5 + 0.2
    g["qmasks"] = g["qmasks"] * 2
    # ---- oracle (CPU): per-layer argmin, then the 36 losses + gradients
    idx = [ocpg_ref.matcher(cfg, g["logits"][l], g["boxes"][l], g["qmasks"][l], targets) for l in range(LR)]
    leaves = {k: g[k].clone().requires_grad_(True) for k in ("logits", "boxes", "pm", "pml", "ls")}

    def layer(l):
        return {"pred_logits": leaves["logits"][l], "pred_boxes": leaves["boxes"][l], "pred_masks": leaves["pm"][l],
                "pred_masks_low": leaves["pml"][l], "ls_features": leaves["ls"]}
    ref_out = dict(layer(0), main_idx=torch.stack(idx[0]), aux_outputs=[layer(l) for l in range(1, LR)],
                   aux_idx=[torch.stack(idx[l]) for l in range(1, LR)])
    ref_losses = ocpg_ref.criterion(cfg, ref_out, targets, it0=0)
    wd = ocpg_ref.weight_dict(cfg)
    ref_total = sum(v * wd[k] for k, v in ref_losses.items() if k in wd)
    ref_grads = torch.autograd.grad(ref_total, [leaves[k] for k in ("logits", "boxes", "pm", "pml", "ls")])
    # ---- product (GPU)
    tg = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    d = {k: v.to(dev) for k, v in g.items()}
    cost = model.matcher.to(dev).cost_matrix_stacked(d["logits"], d["boxes"], d["qmasks"], tg)           # [LR, B, Q]
    got_idx = cost.argmin(2).cpu()
    assert torch.equal(got_idx, torch.stack([torch.cat(i) for i in idx])), (got_idx, idx)
    pl = {k: d[k].clone().requires_grad_(True) for k in ("logits", "boxes", "pm", "pml", "ls")}

    def player(l):
        return {"pred_logits": pl["logits"][l], "pred_boxes": pl["boxes"][l], "pred_masks": pl["pm"][l], "pred_masks_low": pl["pml"][l]}
    as_ind = lambda row: [(row[i].reshape(1).to(dev), torch.zeros(1, dtype=torch.long, device=dev)) for i in range(B)]   # noqa: E731
    out = dict(player(0), ls_features=pl["ls"], frames=torch.zeros(B, T, 3, H // 2, W // 2, device=dev),
               main_matcher_index=as_ind(got_idx[0]), aux_outputs=[player(l) for l in range(1, LR)],
               aux_matcher_index=[as_ind(got_idx[l]) for l in range(1, LR)])
    crit.iter = 0
    losses, *_ = crit(out, tg)
    assert set(losses) == set(ref_losses)
    for k, v in ref_losses.items():
        a, b_ = float(losses[k].detach()), float(v.detach())
        assert abs(a - b_) <= 2e-4 * abs(b_) + 2e-6, (k, a, b_)
    total = crit.weighted_sum(losses)
    assert abs(float(total) - float(ref_total)) <= 2e-4 * abs(float(ref_total))
    grads = torch.autograd.grad(total, [pl[k] for k in ("logits", "boxes", "pm", "pml", "ls")])
    for k, a, b_ in zip(("logits", "boxes", "pm", "pml", "ls"), grads, ref_grads):
        err = (a.cpu() - b_).abs().max().item()
        assert err <= 2e-3 * b_.abs().max().item() + 1e-9, (k, err, b_.abs().max().item())


@pytest.mark.parametrize("where", ["gpu"])
def test_reference_train_iteration_and_checkpoint(golden, dev, where):
    model_checks.check_reference_iteration(golden("train_step"), dev)
    model_checks.check_reference_checkpoint(golden("ckpt_ref_manifest"), dev)


def test_bench_mode_against_fp32_reference(golden, dev):
    """The configuration bench.py times (bf16 autocast, channels-last convs, amp_cache, fused kernels) against the REFERENCE's
    fp32 vectors of the head_dim-32 fixture, with the bounds stated here -- not against itself.

    What run-to-run noise is (tools/bf16_noise.py, measured on MI355X): with MIOpen free to pick its solvers the first
    forward op that differs between two identical runs is backbone.0.body.layer2.0.conv2 (3x3, stride 2, bf16), by 7e-6
    relative; every later bf16 rounding re-quantises that perturbation to bf16's 4e-3 and the random-weight network
    (mask logits up to +-108: saturated sigmoids) amplifies it to 2.8e-2 on the mask logits and to tens of percent on the
    cancellation-dominated gradients (LFM gate: laplace / fc).  With torch.backends.cudnn.deterministic = True (MIOpen's
    deterministic solvers) the forward is bit-identical run to run and the gradients agree to <= 2e-2 (float atomics of the
    backward).  The distance to the fp32 reference is of the same size as that noise: it is what bf16 arithmetic costs here."""
    import cases
    from ocpg_amd.util.misc import NestedTensor
    g = golden("e2e_d32")
    meta, tag = g.meta, "pad"
    B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
    old = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        def run():
            args, model, crit = model_checks.build_product(meta, dev)
            model_checks.to_channels_last(model)
            x, mask, targets = cases.e2e_inputs(B, T, H, W, meta[f"{tag}_sizes"], dev)
            model.train(), crit.train()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(NestedTensor(x, mask), model_checks.text_for(B, dev), targets)
                losses, *_ = crit(out, targets)
                total = crit.weighted_sum(losses)
            total.backward()
            return out, {k: float(v.detach()) for k, v in losses.items()}, float(total.detach()), \
                {k: p.grad.float().norm().item() for k, p in model.named_parameters() if p.grad is not None}
        out, losses, total, gn = run()
        out2, _, total2, _ = run()
    finally:
        torch.backends.cudnn.deterministic = old
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-20))      # noqa: E731
    pm = out["pred_masks"].detach().float().cpu()
    assert torch.equal(pm, out2["pred_masks"].detach().float().cpu()) and total == total2      # deterministic solvers: bit-identical forward
    assert torch.equal(torch.cat([i[0] for i in out["main_matcher_index"]]).cpu(), g[f"{tag}_main_idx"].flatten())
    aux = torch.stack([torch.cat([i[0] for i in a]) for a in out["aux_matcher_index"]]).cpu()
    assert torch.equal(aux, g[f"{tag}_aux_idx"].flatten(1))
    assert rel(pm, g[f"{tag}_pred_masks"]) <= 0.1                                               # measured 4.6e-2
    assert rel(out["pred_boxes"].detach().float().cpu(), g[f"{tag}_pred_boxes"]) <= 2e-2
    assert abs(total - g[f"{tag}_total"].item()) <= 1e-2 * abs(g[f"{tag}_total"].item())       # measured 1.6e-3
    for k, v in meta[f"{tag}_losses"].items():
        assert abs(losses[k] - v) <= 3e-2 * abs(v) + 1e-4, (k, losses[k], v)                    # measured <= 1.1e-2
    ref = {k: v for k, v in meta[f"{tag}_grad_norms"].items() if v and k in gn and v > 1e-9}
    d = sorted(abs(gn[k] - v) / v for k, v in ref.items())
    assert d[len(d) // 2] <= 0.25, d[len(d) // 2]                                              # median, measured 0.12
    assert d[int(0.9 * len(d))] <= 1.0, d[int(0.9 * len(d))]                                   # the LFM gate gradients sit in the tail


def _mha_core_reference(q, k, v, pad, scale, H, keep=None):
    """softmax(q k^T * scale + key padding) [* keep] v in plain tensor ops on [L, B, C] inputs (fp32)."""
    Lq, B, C = q.shape
    Lk, hd = k.shape[0], C // H
    qh, kh, vh = (t.float().reshape(t.shape[0], B, H, hd).permute(1, 2, 0, 3) for t in (q, k, v))      # [B,H,L,hd]
    s = torch.matmul(qh, kh.transpose(-1, -2)) * scale
    if pad is not None:
        s = s.masked_fill(pad[:, None, None, :], float("-inf"))
    p = s.softmax(-1)
    if keep is not None:
        p = p * keep
    return torch.matmul(p, vh).permute(2, 0, 1, 3).reshape(Lq, B, C)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Lq,B,H,Lk,padded", [(19200, 2, 8, 9, False), (1237, 3, 8, 20, True), (5, 10, 8, 5, False), (300, 2, 2, 7, True),
                                               (77, 1, 4, 32, True)])
def test_small_key_attention_kernel(dev, dtype, Lq, B, H, Lk, padded):
    """csrc/attn_smallk.hip (fusion gate: 19 200 x B tokens against <= ~20 text tokens; decoder self-attention: 5 x 5) against
    the tensor-op formulation of nn.MultiheadAttention's core: output, dq, dk, dv; key padding (incl. a padded FIRST key);
    strided q / k views as the packed q-k projection produces them."""
    from ocpg_amd.models.ops.functions import attn_smallk_func as f
    C = H * 32
    g = torch.Generator(device="cpu").manual_seed(Lq + Lk)
    qk = torch.randn(Lq, B, 2 * C, generator=g).to(dev).to(dtype)
    q = qk[..., :C]                                                       # strided view (row stride 2C)
    k = (torch.randn(Lk, B, C, generator=g) * 1.5).to(dev).to(dtype)
    v = torch.randn(Lk, B, C, generator=g).to(dev).to(dtype)
    go = torch.randn(Lq, B, C, generator=g).to(dev).to(dtype)
    pad = None
    if padded:
        pad = torch.zeros(B, Lk, dtype=torch.bool, device=dev)
        pad[0, Lk - 2:] = True
        if B > 1:
            pad[1, 0] = True
    scale = 32 ** -0.5
    res = []
    for hip in (True, False):
        qi, ki, vi = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
        out = f.attention(qi, ki, vi, pad, scale, H) if hip else _mha_core_reference(qi, ki, vi, pad, scale, H).to(dtype)
        assert out is not None
        grads = torch.autograd.grad((out.float() * go.float()).sum(), (qi, ki, vi))
        res.append([out.detach().float()] + [x.float() for x in grads])
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for a, b_, name in zip(res[0], res[1], ("out", "dq", "dk", "dv")):
        assert (a - b_).abs().max().item() <= tol * b_.abs().max().item() + 1e-6, (name, (a - b_).abs().max().item(), b_.abs().max().item())


def test_small_key_attention_dropout(dev):
    """Attention-weight dropout (decoder self-attention in training): the mask is a pure function of (seed, offset), keeps
    ~1-p of the weights scaled by 1/(1-p), and the backward uses the SAME mask (recovered here with one-hot values)."""
    from ocpg_amd.models.ops.functions import attn_smallk_func as f
    Lq, B, H, Lk, p = 640, 3, 8, 5, 0.3
    C = H * 32
    g = torch.Generator(device="cpu").manual_seed(3)
    q, k = torch.randn(Lq, B, C, generator=g).to(dev), torch.randn(Lk, B, C, generator=g).to(dev)
    rng = (20240607, 11)
    onehot = torch.zeros(Lk, B, H, 32, device=dev)
    for j in range(Lk):
        onehot[j, :, :, j] = 1.0
    probs = f.attention(q, k, onehot.view(Lk, B, C), None, 32 ** -0.5, H, p, rng).view(Lq, B, H, 32)[..., :Lk]     # dropped weights
    clean = f.attention(q, k, onehot.view(Lk, B, C), None, 32 ** -0.5, H, 0.0).view(Lq, B, H, 32)[..., :Lk]
    keep = torch.where(probs != 0, torch.full_like(probs, 1 / (1 - p)), torch.zeros_like(probs))
    assert torch.allclose(probs, clean * keep, rtol=1e-5, atol=1e-7)
    assert abs((keep > 0).float().mean().item() - (1 - p)) < 0.02
    assert torch.equal(probs, f.attention(q, k, onehot.view(Lk, B, C), None, 32 ** -0.5, H, p, rng).view(Lq, B, H, 32)[..., :Lk])
    v = torch.randn(Lk, B, C, generator=g).to(dev)
    go = torch.randn(Lq, B, C, generator=g).to(dev)
    res = []
    for hip in (True, False):
        qi, ki, vi = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
        out = f.attention(qi, ki, vi, None, 32 ** -0.5, H, p, rng) if hip else \
            _mha_core_reference(qi, ki, vi, None, 32 ** -0.5, H, keep.permute(1, 2, 0, 3))
        res.append([out.detach()] + list(torch.autograd.grad((out * go).sum(), (qi, ki, vi))))
    for a, b_, name in zip(res[0], res[1], ("out", "dq", "dk", "dv")):
        assert (a - b_).abs().max().item() <= 2e-5 * b_.abs().max().item() + 1e-6, name


@pytest.mark.parametrize("n,c,co,h,w,stride", [(10, 256, 256, 24, 40, 1), (2, 128, 128, 48, 80, 1), (3, 256, 256, 48, 80, 2),
                                                (2, 512, 512, 12, 20, 1), (1, 128, 256, 7, 9, 2), (2, 192, 320, 5, 6, 1)])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("own_weight", [True, False])
def test_conv3x3_mfma_kernel(dev, n, c, co, h, w, stride, relu, own_weight, monkeypatch):
    _conv3x3_mfma_case(dev, n, c, co, h, w, stride, relu, own_weight, own_weight, monkeypatch)


@pytest.mark.parametrize("n,c,co,h,w,stride", [(2, 128, 128, 9, 70, 1), (1, 256, 128, 11, 67, 2), (3, 64, 192, 4, 3, 1), (1, 128, 64, 1, 1, 1)])
def test_conv3x3_own_weight_gradient_segments_and_ragged_maps(dev, n, c, co, h, w, stride, monkeypatch):
    """csrc/conv3x3_wgrad.hip on maps wider than one 64- / 32-pixel segment, odd sizes under stride 2, maps smaller than the kernel."""
    _conv3x3_mfma_case(dev, n, c, co, h, w, stride, True, True, True, monkeypatch)


@pytest.mark.parametrize("n,c,co,h,w,stride", [(10, 256, 256, 24, 40, 1), (2, 512, 512, 12, 20, 1), (2, 256, 512, 21, 30, 2)])
def test_conv3x3_body_split_k(dev, n, c, co, h, w, stride, monkeypatch):
    """The opt-in split-K forward / input gradient of the body's 3x3 convolutions (OCPG_CONV3X3_SPLITK; DESIGN section 5: measured slower
    in the step): K chains split over the grid, BN + ReLU / mask epilogue in the summing pass -- same results as the un-split kernels."""
    from ocpg_amd import _lib
    from ocpg_amd.models.ops.functions import conv_bn_func as f
    monkeypatch.setattr(f, "BODY_SPLITK", True)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    assert int(_lib.lib().ocpg_conv3x3_mfma_body_splits(n * ho * wo, co, c)) > 1
    calls = _lib.census(True)
    try:
        _conv3x3_mfma_case(dev, n, c, co, h, w, stride, True, True, True, monkeypatch)
    finally:
        _lib.census(False)
    assert calls.get("ocpg_conv3x3_mfma_fwd_bn_splitk", 0) == 1, calls


def _conv3x3_mfma_case(dev, n, c, co, h, w, stride, relu, own_weight, own_wgrad, monkeypatch):
    """csrc/conv3x3_mfma.hip (implicit-GEMM bf16 MFMA 3x3 conv + frozen-BN affine + ReLU in the epilogue, its input-gradient
    twin, im2col + GEMM weight gradient) against F.conv2d in fp32 on the same bf16-rounded operands followed by the affine:
    ResNet-101 layer2/3/4 shapes incl. the stride-2 blocks, a ragged map (tiles with masked rows) and channel counts that are
    not multiples of the 128-wide output tile."""
    from ocpg_amd.models.ops.functions import conv_bn_func as f
    # own_weight (round 4): the input gradient reads the weight as it lies, through transposing LDS loads (ocpg_conv3x3_mfma_dgrad_w);
    # False: from a transposed copy (ocpg_conv3x3_mfma_dgrad_masked)
    # own_wgrad (round 4): the weight gradient straight from the two maps (csrc/conv3x3_wgrad.hip); False: im2col + GEMM
    monkeypatch.setattr(f, "DGRAD_OWN_WEIGHT", own_weight)
    monkeypatch.setattr(f, "WGRAD_OWN", own_wgrad)
    g = torch.Generator(device="cpu").manual_seed(n * 1000 + c + h)
    x = torch.randn(n, c, h, w, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    scale = (torch.rand(co, generator=g) + 0.5).to(dev)
    shift = (torch.randn(co, generator=g) * 0.1).to(dev)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    go = torch.randn(n, co, ho, wo, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xi, wi = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    y = f.conv3x3_mfma_bn_act(xi, wi, scale, shift, relu, stride, 1)
    assert y.shape == (n, co, ho, wo) and y.is_contiguous(memory_format=torch.channels_last)
    gx, gw = torch.autograd.grad(y, (xi, wi), go)
    xr, wr = x.float().requires_grad_(True), wt.float().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr, None, stride, 1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    yr = yr.relu() if relu else yr
    # the backward of the product starts from the bf16-ROUNDED forward output's ReLU mask; use the same mask for the reference
    gxr, gwr = torch.autograd.grad(yr, (xr, wr), go.float())
    rel = lambda a, b: float((a.float() - b).norm() / (b.norm() + 1e-20))      # noqa: E731
    assert rel(y, yr) <= 6e-3, rel(y, yr)                                     # bf16 output rounding: 2^-9 relative per element
    assert rel(gx, gxr) <= 1.5e-2, rel(gx, gxr)
    assert rel(gw, gwr) <= 1.5e-2, rel(gw, gwr)
    assert (y.float() - yr).abs().max().item() <= 2e-2 * yr.abs().max().item() + 1e-3


@pytest.mark.parametrize("n,c,co,h,w,stride", [(10, 2048, 256, 12, 20, 2), (2, 512, 128, 7, 9, 1), (1, 256, 64, 5, 5, 2)])
def test_conv3x3_splitk_kernel(dev, n, c, co, h, w, stride):
    """csrc/conv3x3_mfma.hip SPLITK (round 4: the neck's stride-2 level input_proj[3], models/ocpg.py:119-123, off MIOpen): K split over
    the grid, fp32 partial tiles, a summing pass with the bias; input gradient by the MFMA kernel, weight gradient from the patch matrix the
    forward wrote.  Against F.conv2d in fp32 on the same bf16-rounded operands: the config-#2 shape, a stride-1 shape with ragged tiles
    and a tiny one."""
    from ocpg_amd.models.ops.functions import conv_bn_func as f
    from ocpg_amd import _lib
    assert int(_lib.lib().ocpg_conv3x3_mfma_splits(n, h, w, c, co, stride)) > 1
    g = torch.Generator(device="cpu").manual_seed(n * 100 + c + h)
    x = torch.randn(n, c, h, w, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = (torch.randn(co, generator=g) * 0.1).to(dev).to(torch.bfloat16)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    go = torch.randn(n, co, ho, wo, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xi, wi, bi = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = f.conv3x3_splitk(xi, wi, bi, stride)
    assert y.shape == (n, co, ho, wo) and y.is_contiguous(memory_format=torch.channels_last)
    gx, gw, gb = torch.autograd.grad(y, (xi, wi, bi), go)
    xr, wr, br = x.float().requires_grad_(True), wt.float().requires_grad_(True), b.float().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr, br, stride, 1)
    gxr, gwr, gbr = torch.autograd.grad(yr, (xr, wr, br), go.float())
    rel = lambda a, b_: float((a.float() - b_).norm() / (b_.norm() + 1e-20))      # noqa: E731
    assert rel(y, yr) <= 6e-3, rel(y, yr)                                     # bf16 output rounding
    assert rel(gx, gxr) <= 1.5e-2 and rel(gw, gwr) <= 1.5e-2 and rel(gb, gbr) <= 1.5e-2, (rel(gx, gxr), rel(gw, gwr), rel(gb, gbr))


@pytest.mark.timeout(900)
def test_full_size_step_vs_oracle(dev):
    """BASELINE config #2 at FULL size (ResNet-101, 4+4 layers, 5 queries, one clip of 5 x 384 x 640, fp32, dropout off): the product's
    forward + criterion on the GPU against oracle/ocpg_ref.py (the CPU restatement of the reference, pinned by the golden vectors) with
    the same weights and inputs: mask logits, boxes, class logits, matcher indices (bit-exact) and every loss term."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p_ in (root, os.path.join(root, "tests", "golden")):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    import bench
    import synth
    from oracle import ocpg_ref
    from ocpg_amd.models import build_model
    from ocpg_amd.models.text_encoder.text_encoder import PrecomputedText
    from ocpg_amd.util.misc import NestedTensor
    T, H, W = 5, 384, 640
    args = bench.model_args(dev, "resnet101", amp=False)
    args.dropout = 0.0
    model, crit, _ = build_model(args)
    sd = synth.synth_state_dict(synth.shapes_of(model), seed=7)
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and not missing.missing_keys
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0
    model.to(dev), crit.to(dev)
    model.train(), crit.train()
    g = torch.Generator().manual_seed(3)
    clip = torch.randn(1, T, 3, H, W, generator=g)
    mask = torch.zeros(1, T, H, W, dtype=torch.bool)
    feats, sent = torch.randn(1, 9, 768, generator=g), torch.randn(1, 768, generator=g)
    pad = torch.zeros(1, 9, dtype=torch.bool)
    targets = synth.synthetic_targets(1, T, H, W)
    # ---- oracle, CPU
    P = {k: v.clone() for k, v in sd.items()}
    cfg = ocpg_ref.cfg_from_args(bench.model_args("cpu", "resnet101", amp=False))
    with torch.no_grad():
        o_out, o_losses, o_total = ocpg_ref.train_step_loss(P, cfg, clip, mask, (feats, sent, pad), targets)
        # ---- the REFEREE: the same restatement evaluated in float64.  "The GPU is as close to the truth as the CPU
        # oracle" becomes a measured statement: |GPU - fp64| against |CPU fp32 - fp64|, per output.
        d64 = torch.float64
        with ocpg_ref.real(d64):
            r_out, r_losses, r_total = ocpg_ref.train_step_loss(ocpg_ref.as_real(P, d64), cfg, clip.double(), mask,
                                                                ocpg_ref.as_real((feats, sent, pad), d64), ocpg_ref.as_real(targets, d64))
    assert torch.equal(r_out["main_idx"], o_out["main_idx"]) and all(torch.equal(a_, b_) for a_, b_ in zip(r_out["aux_idx"], o_out["aux_idx"])), \
        "the fp32 oracle and its fp64 evaluation disagree on the assignment"
    # ---- product, GPU
    tg = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in t.items()} for t in targets]
    crit.iter = 0
    with torch.no_grad():
        out = model(NestedTensor(clip.to(dev), mask.to(dev)), PrecomputedText(feats.to(dev), sent.to(dev), pad.to(dev)), tg)
        losses, *_ = crit(out, tg)
        total = sum(losses[k] * crit.weight_dict[k] for k in losses if k in crit.weight_dict)
    # matcher indices: bit-exact
    assert [int(i[0].flatten()[0]) for i in out["main_matcher_index"]] == [int(v) for v in o_out["main_idx"].tolist()]
    for layer_idx, want_idx in zip(out["aux_matcher_index"], o_out["aux_idx"]):
        assert [int(i[0].flatten()[0]) for i in layer_idx] == [int(v) for v in want_idx.tolist()]
    for name in ("pred_logits", "pred_boxes", "pred_masks", "pred_masks_low"):
        a, b_ = out[name].float().cpu(), o_out[name].float()
        assert a.shape == b_.shape, (name, a.shape, b_.shape)
        err, scale = (a - b_).abs().max().item(), b_.abs().max().item()
        r_ = r_out[name]
        e_gpu, e_cpu = (a.double() - r_).abs().max().item(), (b_.double() - r_).abs().max().item()
        print(f"{name}: max|GPU - fp32 oracle| {err:.3e} at max|ref| {scale:.3e};  vs the fp64 referee: GPU {e_gpu:.3e}, CPU fp32 oracle {e_cpu:.3e}")
        assert err <= 1e-3 + 2e-5 * scale, (name, err, scale)           # north star: mask logits <= 1e-3 (fp32), plus rounding at large magnitudes
        # against the truth the product is held to the literal 1e-3, or -- where fp32 round-off at these magnitudes exceeds it for ANY
        # fp32 evaluation -- to three times the CPU fp32 oracle's own distance from the referee.  (The GPU step is not bit-reproducible
        # -- float atomics in the MSDeformAttn / weight-gradient sums -- so its distance moves from run to run: seven runs of this test in
        # round 4 gave 1.45 .. 2.07 x the CPU oracle's 1.07e-3 on pred_masks_low, at logits of magnitude 819, i.e. 1.9 .. 2.7e-6 relative;
        # a 2 x bound failed one run in seven.)
        assert e_gpu <= max(1e-3, 3.0 * e_cpu), (name, "GPU further from the fp64 referee than fp32 round-off explains", e_gpu, e_cpu)
    bad = []
    for k, v in o_losses.items():
        a = float(losses[k])
        if abs(a - float(v)) > 2e-4 * abs(float(v)) + 1e-6:
            bad.append((k, a, float(v)))
    assert not bad, bad[:6]
    assert abs(float(total) - float(o_total)) <= 1e-4 * abs(float(o_total)), (float(total), float(o_total))
    print(f"total: GPU {float(total):.6f}, CPU fp32 oracle {float(o_total):.6f}, fp64 referee {float(r_total):.6f}")
    assert abs(float(total) - float(r_total)) <= max(2.0 * abs(float(o_total) - float(r_total)), 1e-5 * abs(float(r_total)))

    # ---- the SAME weights and clip in bench mode (bf16 autocast, channels-last convs: what bench.py times) against the fp32 oracle:
    # the matcher's integer result must not move, and every output / loss stays within a stated bf16 bound (8-bit mantissa through
    # 33 bottlenecks + neck + 4 + 4 layers; VERDICT r2: "bf16 at full size is never compared with anything")
    del model, out, losses
    args16 = bench.model_args(dev, "resnet101", amp=True)
    args16.dropout = 0.0
    model16, crit16, _ = build_model(args16)
    model16.load_state_dict(sd, strict=False)
    for m in model16.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0
    model16.to(dev), crit16.to(dev)
    model_checks.to_channels_last(model16)
    model16.train(), crit16.train()
    crit16.iter = 0
    costs = []
    real = model16.matcher.cost_matrix_stacked
    model16.matcher.cost_matrix_stacked = lambda *a, **k: costs.append(real(*a, **k)) or costs[-1]      # spy: the [layers, B, q] matching costs
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out16 = model16(NestedTensor(clip.to(dev), mask.to(dev)), PrecomputedText(feats.to(dev), sent.to(dev), pad.to(dev)), tg)
        losses16, *_ = crit16(out16, tg)
        total16 = crit16.weighted_sum(losses16)
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-20))      # noqa: E731
    # outputs that do not depend on the assignment: every query's class logit and box
    for name, bound in (("pred_logits", 5e-2), ("pred_boxes", 2e-2)):
        a, b_ = out16[name].float().cpu(), o_out[name].float()
        print(f"bf16 {name}: rel L2 {rel(a, b_):.3e}, max|err| {(a - b_).abs().max().item():.3e} at max|ref| {b_.abs().max().item():.3e}")
        assert rel(a, b_) <= bound, (name, rel(a, b_))
    # the integer assignment: a query may only change where bf16 rounding can reach, i.e. where the cost of the fp32 winner is within
    # a bf16-sized margin of the bf16 minimum (random-init queries start from near-identical reference points: SURVEY section 7
    # "argmin bit-exactness ... report the gap")
    cost = torch.cat([c.float().cpu() for c in costs], 0)                   # [layers, B, q]
    want = torch.stack(list(o_out["aux_idx"]) + [o_out["main_idx"]]).long() if len(o_out["aux_idx"]) else o_out["main_idx"][None].long()
    got = cost.argmin(2)
    moved = []
    for l in range(cost.shape[0]):
        for b_i in range(cost.shape[1]):
            w, g_ = int(want[l, b_i]), int(got[l, b_i])
            if w != g_:
                gap = float(cost[l, b_i, w] - cost[l, b_i, g_])
                moved.append((l, b_i, w, g_, gap, gap / (abs(float(cost[l, b_i, g_])) + 1e-9)))
                assert gap / (abs(float(cost[l, b_i, g_])) + 1e-9) <= 2e-2, ("assignment moved by more than bf16 rounding explains", moved[-1])
    print("bf16 assignment vs fp32 oracle: %d of %d moved %s" % (len(moved), cost.shape[0] * cost.shape[1], moved))
    if not moved:
        for name, bound in (("pred_masks", 1e-1), ("pred_masks_low", 1e-1)):
            a, b_ = out16[name].float().cpu(), o_out[name].float()
            print(f"bf16 {name}: rel L2 {rel(a, b_):.3e}")
            assert rel(a, b_) <= bound, (name, rel(a, b_))
        assert abs(float(total16) - float(o_total)) <= 2e-2 * abs(float(o_total)), (float(total16), float(o_total))
    print(f"bf16 total {float(total16):.4f} vs oracle {float(o_total):.4f}")
    assert float(total16) == float(total16) and abs(float(total16) - float(o_total)) <= 0.25 * abs(float(o_total))


@pytest.mark.timeout(900)
def test_full_size_gradients_vs_oracle(dev):
    """Same full-size configuration as test_full_size_step_vs_oracle, now the BACKWARD: the gradient of the weighted total w.r.t. every
    trainable parameter against the oracle's autograd result (CPU, OpenMP C MSDeformAttn backward).  Single bilinear samples that sit
    on a pixel boundary flip their cell on a one-ulp difference (section 2 of DESIGN.md), so the bar is per-tensor relative L2 error
    (<= 3e-2 for >= 97 % of the tensors, median <= 1.5e-3; the flips make both move from run to run with the GEMM kernels the plan
    cache happens to time fastest -- measured over five runs: median 1.3e-4 .. 5.1e-4, 97th percentile 2.6e-3 .. 1.0e-2, worst =
    `ls_feat_viz.bias`, a gradient that cancels to ~0) plus the global gradient norm (<= 1e-3; measured equal to 5 digits every time)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p_ in (root, os.path.join(root, "tests", "golden")):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    import bench
    import synth
    from oracle import ocpg_ref
    from ocpg_amd.models import build_model
    from ocpg_amd.models.text_encoder.text_encoder import PrecomputedText
    from ocpg_amd.util.misc import NestedTensor
    T, H, W = 5, 384, 640
    args = bench.model_args(dev, "resnet101", amp=False)
    args.dropout = 0.0
    model, crit, _ = build_model(args)
    sd = synth.synth_state_dict(synth.shapes_of(model), seed=11)
    model.load_state_dict(sd, strict=False)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0
    model.to(dev), crit.to(dev)
    model.train(), crit.train()
    g = torch.Generator().manual_seed(5)
    clip = torch.randn(1, T, 3, H, W, generator=g)
    mask = torch.zeros(1, T, H, W, dtype=torch.bool)
    feats, sent = torch.randn(1, 9, 768, generator=g), torch.randn(1, 768, generator=g)
    pad = torch.zeros(1, 9, dtype=torch.bool)
    targets = synth.synthetic_targets(1, T, H, W)
    trainable = {k for k, p in model.named_parameters() if p.requires_grad}
    trainable |= {k.replace("transformer.decoder.bbox_embed.", "bbox_embed.") for k in trainable}       # shared module, two state_dict names
    P = {k: v.clone().requires_grad_(k in trainable) for k, v in sd.items()}
    cfg = ocpg_ref.cfg_from_args(bench.model_args("cpu", "resnet101", amp=False))
    _, _, o_total = ocpg_ref.train_step_loss(P, cfg, clip, mask, (feats, sent, pad), targets)
    o_total.backward()
    tg = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in t.items()} for t in targets]
    crit.iter = 0
    out = model(NestedTensor(clip.to(dev), mask.to(dev)), PrecomputedText(feats.to(dev), sent.to(dev), pad.to(dev)), tg)
    losses, *_ = crit(out, tg)
    total = sum(losses[k] * crit.weight_dict[k] for k in losses if k in crit.weight_dict)
    total.backward()
    assert abs(float(total) - float(o_total)) <= 1e-4 * abs(float(o_total))
    rel, sq_a, sq_b = [], 0.0, 0.0
    for k, p in model.named_parameters():
        if not p.requires_grad:
            continue
        ko = k if P[k].grad is not None else k.replace("transformer.decoder.bbox_embed.", "bbox_embed.")      # shared module, two state_dict names
        assert p.grad is not None and P[ko].grad is not None, k
        a, b_ = p.grad.detach().float().cpu(), P[ko].grad.float()
        sq_a += float(a.double().square().sum())
        sq_b += float(b_.double().square().sum())
        rel.append((float((a - b_).norm() / (b_.norm() + 1e-20)), k))
    rel.sort()
    n = len(rel)
    print(f"{n} tensors: median rel L2 {rel[n // 2][0]:.2e}, 97th pct {rel[int(0.97 * n)][0]:.2e}, worst {rel[-1][0]:.2e} ({rel[-1][1]}); "
          f"grad norm {sq_a ** 0.5:.5g} vs {sq_b ** 0.5:.5g}")
    assert abs(sq_a ** 0.5 - sq_b ** 0.5) <= 1e-3 * sq_b ** 0.5
    assert rel[n // 2][0] <= 1.5e-3 and rel[int(0.97 * n)][0] <= 3e-2, rel[-5:]


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("shape,size", [((3, 8, 48, 80), (192, 320)), ((2, 16, 48, 80), (96, 160)), ((2, 3, 37, 53), (41, 97)), ((1, 2, 5, 7), (5, 7)),
                                        ((1, 1, 1, 4), (3, 9))])
def test_bilinear_resize_as_matmul(dev, align, shape, size):
    """models/resample.bilinear_resize == F.interpolate(mode="bilinear") for both align_corners conventions, up- and down-sampling,
    degenerate sizes; value and input gradient."""
    from ocpg_amd.models.resample import bilinear_resize
    g = torch.Generator(device=dev).manual_seed(sum(shape) + size[0])
    x = torch.randn(*shape, device=dev, generator=g)
    go = torch.randn(*shape[:2], *size, device=dev, generator=g)
    a = x.clone().requires_grad_(True)
    b = x.clone().requires_grad_(True)
    ya = bilinear_resize(a, size, align)
    yb = torch.nn.functional.interpolate(b, size=size, mode="bilinear", align_corners=align)
    assert (ya - yb).abs().max().item() <= 5e-6 * yb.abs().max().item() + 2e-6          # fp32 GEMM summation order + weight rounding
    ga, = torch.autograd.grad((ya * go).sum(), a)
    gb, = torch.autograd.grad((yb * go).sum(), b)
    assert (ga - gb).abs().max().item() <= 2e-5 * gb.abs().max().item() + 1e-6


def test_clip_adamw_kernels_equal_torch(dev):
    """ocpg_amd.optim.ClipAdamW (csrc/adamw.hip: total norm, clip coefficient, AdamW update in three launches) against
    torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW (engine.py:100-106, main.py:76-99) over four steps: four groups with their own
    learning rates, odd sizes, a channels-last 4-D parameter, a step where the norm is below the threshold and steps where it clips;
    the state dict round-trips into a plain torch.optim.AdamW."""
    from ocpg_amd.optim import ClipAdamW
    torch.manual_seed(5)
    shapes = [(257, 33), (64, 32, 3, 3), (5,), (1000, 130), (7, 9, 11), (2049,)]

    def make():
        ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
        ps[1].data = ps[1].data.contiguous(memory_format=torch.channels_last)
        return ps
    a = make()
    b = [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in a]
    groups = lambda ps: [{"params": ps[:2], "lr": 1e-2}, {"params": ps[2:3], "lr": 5e-3}, {"params": ps[3:5], "lr": 1e-3}, {"params": ps[5:], "lr": 2e-2}]   # noqa: E731
    mine = ClipAdamW(groups(a), lr=1e-2, weight_decay=5e-4)
    ref = torch.optim.AdamW(groups(b), lr=1e-2, weight_decay=5e-4, foreach=False, fused=False)
    for it, (scale, max_norm) in enumerate(((1e-4, 0.1), (1.0, 0.1), (3.0, 0.1), (1.0, 0.0))):
        gs = [torch.randn(s, device=dev) * scale for s in shapes]
        for p, q, g in zip(a, b, gs):
            p.grad = g.clone().contiguous(memory_format=torch.channels_last) if g.dim() == 4 else g.clone()
            q.grad = p.grad.clone(memory_format=torch.preserve_format)
        norm = mine.step_clip(max_norm)
        # the reference clips with clip_grad_norm_'s rule, coef = min(1, max_norm / (norm + 1e-6)), on the EXACT norm (fp64): torch's own
        # fp32 foreach norm of these 170 k elements is ~1e-5 off, which the second moments would show as 2e-5
        want = torch.sqrt(sum(q.grad.double().square().sum() for q in b))
        if max_norm > 0:
            coef = min(1.0, max_norm / (float(want) + 1e-6))
            for q in b:
                q.grad.mul_(coef)
        ref.step()
        assert abs(float(norm) - float(want)) <= 2e-6 * float(want), (it, float(norm), float(want))
        for i, (p, q) in enumerate(zip(a, b)):
            assert torch.allclose(p, q, rtol=2e-6, atol=2e-7), (it, i, (p - q).abs().max().item())
            for k in ("exp_avg", "exp_avg_sq"):
                assert torch.allclose(mine.state[p][k], ref.state[q][k], rtol=2e-6, atol=1e-9), (it, i, k)
    sd = mine.state_dict()
    assert all(float(v["step"]) == 4.0 for v in sd["state"].values())
    plain = torch.optim.AdamW(groups([torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in a]), lr=1e-2, weight_decay=5e-4)
    plain.load_state_dict(sd)                 # interchangeable with torch's optimizer (checkpoint wire format, row f2)


@pytest.mark.parametrize("rows,cin,cout,bias", [(50, 256, 256, True), (50, 256, 128, True), (1, 64, 5, False), (130, 512, 384, True), (64, 128, 64, True)])
def test_small_linear_f32_kernels_equal_f_linear(dev, rows, cin, cout, bias):
    """csrc/small_linear_f32.hip (the fp32 islands' few-row Linears: MSDeformAttn's projections over the decoder's query rows, reference
    models/deformable_transformer.py:329-332 + models/ops/modules/ms_deform_attn.py:96-115) == F.linear in fp32: output, input gradient,
    weight gradient, bias gradient, to fp32 rounding; served through amp_cache.linear outside autocast only."""
    from ocpg_amd import _lib
    from ocpg_amd.models import amp_cache
    torch.manual_seed(rows + cin)
    x = torch.randn(2, rows, cin, device=dev)
    w = torch.nn.Parameter(torch.randn(cout, cin, device=dev) * 0.1)
    b = torch.nn.Parameter(torch.randn(cout, device=dev)) if bias else None
    go = torch.randn(2, rows, cout, device=dev)
    res = []
    for fn in (amp_cache.linear, torch.nn.functional.linear):
        xi = x.clone().requires_grad_(True)
        calls = _lib.census(True)
        y = fn(xi, w, b)
        gs = torch.autograd.grad((y * go).sum(), [xi, w] + ([b] if bias else []))
        _lib.census(False)
        if fn is amp_cache.linear:
            assert calls.get("ocpg_small_linear_f32_fwd", 0) == 1 and calls.get("ocpg_small_linear_f32_bwd", 0) == 1, calls
        res.append([y.detach()] + list(gs))
    for a, r in zip(*res):
        assert a.dtype == torch.float32 and (a - r).abs().max().item() <= 2e-5 * r.abs().max().item() + 1e-6, ((a - r).abs().max().item(), r.abs().max().item())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        calls = _lib.census(True)
        amp_cache.linear(x, w, b)
        _lib.census(False)
        assert "ocpg_small_linear_f32_fwd" not in calls


def test_clip_adamw_under_grad_scaler_equals_torch(dev):
    """`scaler.step(ClipAdamW, max_norm=...)` (unscale, norm, clip, skip-on-overflow and AdamW on the device, no found_inf.item()) against
    the reference's sequence scaler.unscale_ + clip_grad_norm_ + scaler.step(torch.optim.AdamW) + scaler.update() (engine.py:98-106):
    five steps of which the third carries an inf gradient -- both skip it, both halve the scale, and the bias corrections of the later
    steps count FOUR steps, not five."""
    from ocpg_amd.optim import ClipAdamW
    torch.manual_seed(6)
    shapes = [(257, 33), (64, 32, 3, 3), (5,), (1000, 130), (2049,)]
    a = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    groups = lambda ps: [{"params": ps[:2], "lr": 1e-2}, {"params": ps[2:], "lr": 1e-3}]   # noqa: E731
    mine = ClipAdamW(groups(a), lr=1e-2, weight_decay=5e-4)
    ref = torch.optim.AdamW(groups(b), lr=1e-2, weight_decay=5e-4, foreach=False, fused=False)
    sa, sb = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=2), torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=2)
    scales = []
    for it, (mag, max_norm) in enumerate(((1e-4, 0.1), (1.0, 0.1), (1.0, 0.1), (3.0, 0.1), (1.0, 0.0))):
        gs = [torch.randn(s, device=dev) * mag for s in shapes]
        if it == 2:
            gs[3][17, 5] = float("inf")
        for scaler, ps in ((sa, a), (sb, b)):
            scaler.scale(torch.zeros((), device=dev))                    # (creates the scale tensor, as scaler.scale(loss) does)
            for p, g in zip(ps, gs):
                p.grad = g * scaler.get_scale()
        before = [p.detach().clone() for p in a]
        sa.step(mine, max_norm=max_norm)
        sa.update()
        sb.unscale_(ref)
        want = torch.sqrt(sum(q.grad.double().square().sum() for q in b))
        if max_norm > 0 and it != 2:
            for q in b:
                q.grad.mul_(min(1.0, max_norm / (float(want) + 1e-6)))
        sb.step(ref)
        sb.update()
        scales.append((sa.get_scale(), sb.get_scale()))
        if it == 2:
            assert all(torch.equal(p, q) for p, q in zip(a, before)), "an overflowed step must leave the parameters alone"
        else:
            assert abs(float(mine.grad_norm) - float(want)) <= 4e-6 * float(want), (it, float(mine.grad_norm), float(want))
        for i, (p, q) in enumerate(zip(a, b)):
            assert torch.allclose(p, q, rtol=4e-6, atol=4e-7), (it, i, (p - q).abs().max().item())
            for k in ("exp_avg", "exp_avg_sq"):
                assert torch.allclose(mine.state[p][k], ref.state[q][k], rtol=4e-6, atol=1e-9), (it, i, k)
    assert all(x == y for x, y in scales) and scales[2][0] < scales[1][0], scales
    assert mine.steps_taken() == 4
    assert all(float(v["step"]) == 4.0 for v in mine.state_dict()["state"].values())
    # and back to un-scaled stepping: the count carries over
    for p in a:
        p.grad = torch.randn_like(p)
    mine.step_clip(0.1)
    assert mine.steps_taken() == 5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(9600, 256), (51000, 384), (777, 100), (40000, 2048), (130, 96), (5, 8)])
def test_colsum_partials_equals_sum(dev, dtype, shape):
    """ocpg_colsum_partials (16-byte loads, (vector column, row phase) threads, LDS fold; scalar path when C is not a whole number of
    vectors): the partial rows add up to the fp64 column sums of the matrix as stored."""
    from ocpg_amd._lib import check, lib
    r, c = shape
    g = torch.Generator().manual_seed(r + c)
    x = torch.randn(r, c, generator=g).to(dtype).to(dev)
    code = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dtype]
    nb = int(lib().ocpg_colsum_blocks(r))
    assert 1 <= nb <= 256
    part = torch.full((nb, c), float("nan"), dtype=torch.float32, device=dev)
    check(lib().ocpg_colsum_partials(x.data_ptr(), r, c, code, part.data_ptr(), torch.cuda.current_stream().cuda_stream), "ocpg_colsum_partials")
    want = x.double().sum(0)
    got = part.double().sum(0)
    assert torch.isfinite(part).all()
    assert (got - want).abs().max().item() <= 1e-5 * (x.double().abs().sum(0).max().item() + 1.0)


def test_deferred_partial_sums_finish_inside_the_gradient_cast(dev):
    """amp_cache.defer_sum: the row-split weight-gradient partials [S, Co, K] and the fp32 column-sum partials behind bias gradients of
    single-use layers are finished by the ONE launch that casts all gradients to fp32 (csrc/multi_cast.hip: multi_cast_sum; bias partials in
    256-element chunks since round 4).  A 256 -> 2048 -> 256 token MLP over 16 384 rows (34-way and 2 048-wide cases of the encoder FFN in
    miniature) under bf16 autocast, with the deferral on and off: identical fp32 gradients up to the summation order."""
    from ocpg_amd.models import amp_cache
    torch.manual_seed(2)
    net = torch.nn.Sequential(amp_cache.Linear(256, 2048), torch.nn.ReLU(), amp_cache.Linear(2048, 256)).to(dev)
    amp_cache.mark_single_use(net[0], net[2])
    x = torch.randn(16384, 256, device=dev)
    go = torch.randn(16384, 256, device=dev)
    res = []
    for on in (True, False):
        old = amp_cache.DEFER_SUM
        amp_cache.DEFER_SUM = on
        try:
            net.zero_grad(set_to_none=True)
            calls = __import__("ocpg_amd._lib", fromlist=["census"]).census(True)
            with torch.autocast("cuda", dtype=torch.bfloat16), amp_cache.scope(net):
                y = net(x)
            (y.float() * go).sum().backward()
            torch.cuda.synchronize()
            seen = dict(calls)
            __import__("ocpg_amd._lib", fromlist=["census"]).census(False)
            assert not amp_cache._PARTIALS
            assert (seen.get("ocpg_multi_cast_sum", 0) >= 1) == on, seen
            res.append([p.grad.clone() for p in net.parameters()])
            assert all(g.dtype == torch.float32 for g in res[-1])
        finally:
            amp_cache.DEFER_SUM = old
    for a, b_ in zip(*res):
        assert (a - b_).abs().max().item() <= 2e-2 * b_.abs().max().item() + 1e-6, ((a - b_).abs().max().item(), b_.abs().max().item())
        assert (a - b_).norm().item() <= 4e-3 * b_.norm().item()
