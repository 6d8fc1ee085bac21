"""Video-Swin product-vs-golden checks (shared by the CPU and GPU suites; pure torch ops + the model's HIP ops on GPU)."""
import torch

import cases
import synth
from module_checks import close


def _load(mod, shapes, seed):
    missing = mod.load_state_dict(synth.synth_state_dict(shapes, seed=seed), strict=False)
    assert not missing.unexpected_keys
    assert all("relative_position_index" in k for k in missing.missing_keys), missing.missing_keys


def check_window_attention(g, dev, rtol=1e-4, atol=1e-5):
    import ocpg_amd.models.video_swin_transformer as vs
    wa = vs.WindowAttention3D(24, (2, 3, 3), 3, qkv_bias=True)
    _load(wa, g.meta["wa_shapes"], 6)
    wa.to(dev)
    x = synth.rand("wa_x", (8, 18, 24)).to(dev).requires_grad_(True)
    mask = vs.compute_mask(2, 6, 6, (2, 3, 3), (0, 1, 1), dev)
    close(mask, g["wa_mask"], 0, 0, "compute_mask small")
    for tag, m in (("nomask", None), ("mask", mask)):
        y = wa(x, m)
        close(y, g[f"wa_{tag}_y"], rtol, atol, f"wa {tag} y")
        grads = torch.autograd.grad((y * synth.rand("wa_go", y.shape).to(dev)).sum(), [x] + list(wa.parameters()))
        close(grads[0], g[f"wa_{tag}_gx"], rtol * 10, atol * 10, f"wa {tag} gx")
        for (k, _), gg in zip(wa.named_parameters(), grads[1:]):
            close(gg, g[f"wa_{tag}_gp_{k}"], rtol * 10, atol * 50, f"wa {tag} gp {k}")
    wb = vs.WindowAttention3D(24, (8, 7, 7), 3, qkv_bias=True)
    _load(wb, g.meta["wb_shapes"], 7)
    wb.to(dev)
    close(wb(synth.rand("wb_x", (2, 245, 24)).to(dev), None), g["wb_y"], rtol, atol, "clamped-window attention")
    # head_dim 32: on the GPU this is the fused HIP kernel (csrc/win_attn.hip), on the CPU the generic path
    wc = vs.WindowAttention3D(96, (8, 7, 7), 3, qkv_bias=True)
    _load(wc, g.meta["wc_shapes"], 11)
    wc.to(dev)
    blk = vs.SwinTransformerBlock3D(96, 3, window_size=(8, 7, 7), shift_size=(4, 3, 3))
    regionc = blk._plan(5, 14, 14, dev)[3]                                     # region ids of the 4 shifted windows
    maskc = vs.compute_mask(5, 14, 14, (5, 7, 7), (0, 3, 3), dev) if dev.type != "cuda" else None
    xw = synth.rand("wc_x", (8, 245, 96)).to(dev).requires_grad_(True)
    for tag, use_mask in (("nomask", False), ("mask", True)):
        y = wc(xw, maskc if use_mask else None, regionc if use_mask else None)
        close(y, g[f"wc_{tag}_y"], rtol, atol, f"wc {tag} y")
        grads = torch.autograd.grad((y * synth.rand("wc_go", y.shape).to(dev)).sum(), [xw] + list(wc.parameters()))
        close(grads[0], g[f"wc_{tag}_gx"], rtol * 10, atol * 10, f"wc {tag} gx")
        for (k, _), gg in zip(wc.named_parameters(), grads[1:]):
            ref = g.meta[f"wc_{tag}_grad_norms"][k]
            assert abs(gg.norm().item() - ref) <= 2e-3 * abs(ref) + 1e-5, (tag, k, gg.norm().item(), ref)
            if "bias_table" in k:
                close(gg, g[f"wc_{tag}_gtable"], rtol * 10, atol * 50, f"wc {tag} bias-table grad")
    # integer buffers must equal the reference's by construction
    close(vs.compute_mask(5, 14, 21, (5, 7, 7), (0, 3, 3), dev), g["mask_5_14_21"], 0, 0, "mask 5x14x21")
    close(vs.compute_mask(8, 7, 14, (4, 7, 7), (2, 0, 3), dev), g["mask_8_7_14"], 0, 0, "mask 8x7x14")


def check_block_and_merging(g, dev, rtol=1e-4, atol=2e-5):
    import ocpg_amd.models.video_swin_transformer as vs
    blk = vs.SwinTransformerBlock3D(24, 3, window_size=(8, 7, 7), shift_size=(4, 3, 3), mlp_ratio=2.0)
    _load(blk, g.meta["blk_shapes"], 8)
    blk.to(dev)
    x = synth.rand("blk_x", (2, 5, 10, 13, 24)).to(dev).requires_grad_(True)
    y = blk(x)
    close(y, g["blk_y"], rtol, atol, "shifted block y")
    grads = torch.autograd.grad((y * synth.rand("blk_go", y.shape).to(dev)).sum(), [x] + list(blk.parameters()))
    close(grads[0], g["blk_gx"], rtol * 10, atol * 10, "shifted block gx")
    for (k, _), gg in zip(blk.named_parameters(), grads[1:]):
        ref = g.meta["blk_grad_norms"][k]
        assert abs(gg.norm().item() - ref) <= 2e-3 * abs(ref) + 1e-5, (k, gg.norm().item(), ref)
    pm = vs.PatchMerging(24)
    _load(pm, g.meta["pm_shapes"], 9)
    pm.to(dev)
    close(pm(synth.rand("pm_x", (2, 3, 5, 7, 24)).to(dev)), g["pm_y"], rtol, atol, "patch merging")


def check_backbone(g, dev, rtol=2e-4, atol=5e-5):
    import ocpg_amd.models.video_swin_transformer as vs
    cfg = dict(g.meta["swin_tiny"])
    bb = vs.VideoSwinTransformerBackbone(False, None, True, (0, 1, 2, 3), **cfg)
    _load(bb, g.meta["bb_shapes"], 10)
    bb.to(dev)
    x = synth.rand("bb_x", (5, 3, 64, 96)).to(dev).requires_grad_(True)
    out = bb(x, 5)
    loss = 0
    for k, v in out.items():
        close(v, g[f"bb_out{k}"], rtol, atol, f"backbone stage {k}")
        loss = loss + (v * synth.rand(f"bb_go{k}", v.shape).to(dev)).sum()
    grads = torch.autograd.grad(loss, [x] + list(bb.parameters()), allow_unused=True)
    close(grads[0], g["bb_gx"], rtol * 10, atol * 10, "backbone gx")
    for (k, _), gg in zip(bb.named_parameters(), grads[1:]):
        ref = g.meta["bb_grad_norms"][k]
        if ref is None:
            assert gg is None, k
        else:
            assert abs(gg.norm().item() - ref) <= 3e-3 * abs(ref) + 1e-5, (k, gg.norm().item(), ref)


def check_e2e_swin(g, dev, rtol=5e-4, atol=5e-5):
    import model_checks
    from ocpg_amd.models import build_model
    from ocpg_amd.util.misc import NestedTensor
    meta = g.meta
    args = cases.default_args(device=str(dev), video_swin_cfg=meta["swin_cfg"], **meta["cfg"])
    model, crit, _ = build_model(args)
    sd = synth.synth_state_dict(meta["float_shapes"], seed=meta["seed"])
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all("relative_position_index" in k for k in missing.missing_keys)
    assert set(model.state_dict().keys()) == set(meta["state_shapes"].keys())
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.to(dev), crit.to(dev)
    B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
    x, mask, targets = cases.e2e_inputs(B, T, H, W, meta["pad_sizes"], dev)
    model.train(), crit.train()
    out = model(NestedTensor(x, mask), model_checks.text_for(B, dev), targets)
    losses, *_ = crit(out, targets)
    total = sum(losses[k] * crit.weight_dict[k] for k in losses if k in crit.weight_dict)
    total.backward()
    assert torch.equal(torch.cat([i[0] for i in out["main_matcher_index"]]).cpu(), g["pad_main_idx"].flatten())
    assert torch.equal(torch.stack([torch.cat([i[0] for i in a]) for a in out["aux_matcher_index"]]).cpu(), g["pad_aux_idx"].flatten(1))
    for k in ("pred_logits", "pred_boxes", "pred_masks", "pred_masks_low"):
        a = model_checks.MASK_LOGIT_ATOL if "mask" in k else atol
        close(out[k], g[f"pad_{k}"], rtol, a, "e2e swin " + k)
    for k, v in meta["pad_losses"].items():
        assert abs(losses[k].item() - v) <= 5e-3 * abs(v) + 5e-4, (k, losses[k].item(), v)
    bad = []
    params = dict(model.named_parameters())
    for k, v in meta["pad_grad_norms"].items():
        if v is None:
            continue
        n = params[k].grad.norm().item()
        if abs(n - v) > 2e-2 * abs(v) + 1e-4:
            bad.append((k, n, v))
    assert not bad, bad[:6]


# norm-relative bounds of the 16-bit storage paths against the reference's fp32 vectors: (y, gx, parameter-gradient norms).  fp16 keeps 11
# mantissa bits, bf16 8: the bf16 bounds are the fp16 ones x 8.  Measured on MI355X (r4): fp16 y 4e-4, gx 9e-4; bf16 y 3e-3, gx 7e-3.
TOL16 = {torch.float16: (5e-3, 1e-2, 2e-2), torch.bfloat16: (2e-2, 4e-2, 6e-2)}


def check_window_attention_16bit(g, dev, dtype):
    """swin3d.npz's head_dim-32 case (clamped window (5,7,7) = 245 tokens, 3 heads, 8 windows of which 4 are shifted) through the
    matrix-core kernels (csrc/win_attn_mfma.hip serves 16-bit storage only) under autocast, against the reference's fp32 vectors."""
    import ocpg_amd.models.video_swin_transformer as vs
    wc = vs.WindowAttention3D(96, (8, 7, 7), 3, qkv_bias=True)
    _load(wc, g.meta["wc_shapes"], 11)
    wc.to(dev)
    blk = vs.SwinTransformerBlock3D(96, 3, window_size=(8, 7, 7), shift_size=(4, 3, 3))
    regionc = blk._plan(5, 14, 14, dev)[3]
    xw = synth.rand("wc_x", (8, 245, 96)).to(dev).requires_grad_(True)
    ty, tg, tn = TOL16[dtype]
    rel = lambda a, b: float((a.detach().float().cpu() - b).norm() / b.norm())    # noqa: E731
    seen = {}
    for tag, use_mask in (("nomask", False), ("mask", True)):
        with torch.autocast(dev.type, dtype=dtype):
            y = wc(xw, None, regionc if use_mask else None)
        assert y.dtype == dtype
        grads = torch.autograd.grad((y.float() * synth.rand("wc_go", y.shape).to(dev)).sum(), [xw] + list(wc.parameters()))
        seen[tag] = (rel(y, g[f"wc_{tag}_y"]), rel(grads[0], g[f"wc_{tag}_gx"]))
        assert seen[tag][0] <= ty and seen[tag][1] <= tg, (tag, seen)
        for (k, _), gg in zip(wc.named_parameters(), grads[1:]):
            ref = g.meta[f"wc_{tag}_grad_norms"][k]
            assert abs(gg.float().norm().item() - ref) <= tn * abs(ref) + 1e-5, (tag, k, gg.norm().item(), ref)
            if "bias_table" in k:
                assert rel(gg, g[f"wc_{tag}_gtable"]) <= tg, (tag, rel(gg, g[f"wc_{tag}_gtable"]))
    print("window attention %s vs fp32 vectors (rel y, rel gx): %s" % (dtype, seen))


def check_window_attention_n392(g, dev, dtype=torch.float32, rtol=5e-4, atol=5e-5):
    """Config #5's window: the full (8,7,7) = 392 tokens, head_dim 32, 4 shifted windows; fp32, or fp16 storage (the
    reference's --amp) against the same fp32 vectors."""
    import ocpg_amd.models.video_swin_transformer as vs
    wa = vs.WindowAttention3D(64, (8, 7, 7), 2, qkv_bias=True)
    _load(wa, g.meta["shapes"], 13)
    wa.to(dev)
    blk = vs.SwinTransformerBlock3D(64, 2, window_size=(8, 7, 7), shift_size=(4, 3, 3))
    region = blk._plan(8, 14, 14, dev)[3]                                     # region ids of the 4 shifted windows
    mask = vs.compute_mask(8, 14, 14, (8, 7, 7), (0, 3, 3), dev) if dev.type != "cuda" else None
    x = synth.rand("w392_x", (4, 392, 64)).to(dev).requires_grad_(True)
    go = synth.rand("w392_go", (4, 392, 64)).to(dev)
    for tag, use_mask in (("nomask", False), ("mask", True)):
        with torch.autocast(dev.type, dtype=dtype, enabled=dtype != torch.float32):
            y = wa(x, mask if use_mask else None, region if use_mask else None)
        grads = torch.autograd.grad((y.float() * go).sum(), [x] + list(wa.parameters()))
        if dtype == torch.float32:
            close(y, g[f"{tag}_y"], rtol, atol, f"n392 {tag} y")
            close(grads[0], g[f"{tag}_gx"], rtol * 10, atol * 10, f"n392 {tag} gx")
        else:       # 16-bit storage of q/k/v/out: norm-relative bounds
            rel = lambda a, b: float((a.detach().float().cpu() - b).norm() / b.norm())    # noqa: E731
            assert rel(y, g[f"{tag}_y"]) <= TOL16[dtype][0], rel(y, g[f"{tag}_y"])
            assert rel(grads[0], g[f"{tag}_gx"]) <= TOL16[dtype][1], rel(grads[0], g[f"{tag}_gx"])
            print("n392 %s %s (rel y, rel gx): %.2e %.2e" % (dtype, tag, rel(y, g[f"{tag}_y"]), rel(grads[0], g[f"{tag}_gx"])))
        tol = 2e-3 if dtype == torch.float32 else TOL16[dtype][2]
        for (k, _), gg in zip(wa.named_parameters(), grads[1:]):
            ref = g.meta[f"{tag}_grad_norms"][k]
            assert abs(gg.float().norm().item() - ref) <= tol * abs(ref) + 1e-5, (tag, k, gg.norm().item(), ref)
            if "bias_table" in k:
                if dtype == torch.float32:
                    close(gg, g[f"{tag}_gtable"], rtol * 10, atol * 50, f"n392 {tag} bias-table grad")
