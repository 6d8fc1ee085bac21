"""Shared body of the product-vs-golden model checks (run on CPU with the MSDA test double, on GPU with the HIP op)."""
import torch

import cases
import synth


MASK_LOGIT_ATOL = 1e-3   # BASELINE.json north_star: "mask logits within 1e-3 fp32"


def build_product(meta, device, **over):
    from ocpg_amd.models import build_model
    cfg = dict(meta["cfg"])
    cfg.update(over)
    args = cases.default_args(device=str(device), **cfg)
    model, crit, _ = build_model(args)
    sd = synth.synth_state_dict({k: v for k, v in meta["float_shapes"].items()}, seed=meta["seed"])
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and not missing.missing_keys, missing
    for m in model.modules():          # parity runs: all dropout off on both sides (see make_fixtures.build_tiny)
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return args, model.to(device), crit.to(device)


def text_for(B, device):
    from ocpg_amd.models.text_encoder.text_encoder import PrecomputedText
    f, s, m = cases.tiny_text(B)
    return PrecomputedText(f.to(device), s.to(device), m.to(device))


def to_channels_last(model):
    """bench.py's layout: every Conv2d in channels-last (the ResNet body then runs NHWC end to end: fused frozen-BN NHWC
    kernels, 1x1 convs as GEMMs)."""
    for m in model.modules():
        if isinstance(m, torch.nn.Conv2d):
            m.to(memory_format=torch.channels_last)
    return model


def run_train_step(g, tag, device, rtol, atol, channels_last=False, tag_masks=False):
    from ocpg_amd.util.misc import NestedTensor, tag_rect_mask
    meta = g.meta
    args, model, crit = build_product(meta, device)
    if channels_last:
        to_channels_last(model)
    B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
    x, mask, targets = cases.e2e_inputs(B, T, H, W, meta[f"{tag}_sizes"], device)
    if tag_masks:       # what the collate helpers do: declare every frame's valid extent (memoised mask-only tensors, and the
        #                 "no padding anywhere" shortcut of the transformer when all frames fill the map)
        tag_rect_mask(mask, [hw for hw in meta[f"{tag}_sizes"] for _ in range(T)])
    model.train(), crit.train()
    out = model(NestedTensor(x, mask), text_for(B, device), targets)
    losses, *_ = crit(out, targets)
    wd = crit.weight_dict
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    total.backward()
    res = {}
    # integer assignment: bit-exact
    assert torch.equal(torch.cat([i[0] for i in out["main_matcher_index"]]).cpu(), g[f"{tag}_main_idx"].flatten())
    aux = torch.stack([torch.cat([i[0] for i in a]) for a in out["aux_matcher_index"]]).cpu()
    assert torch.equal(aux, g[f"{tag}_aux_idx"].flatten(1))
    for k in ("pred_logits", "pred_boxes", "pred_masks", "pred_masks_low", "ls_features"):
        got, exp = out[k].detach().cpu().float(), g[f"{tag}_{k}"]
        if k == "ls_features":
            # channel 11 is dot(f, t) / (cos(f, t) + 1e-5) (ocpg.py:376): unbounded where the cosine crosses zero, and
            # dropped by the level-set loss (criterion.py:175).  Checked by quantile; channels 0..10 element-wise.
            rel = (got[:, :, 11] - exp[:, :, 11]).abs() / (exp[:, :, 11].abs() + 1e-3)
            assert (rel > 1e-3).float().mean().item() < 0.01
            got, exp = got[:, :, :11], exp[:, :, :11]
        err = (got - exp).abs().max().item()
        res[k] = err
        # mask logits: the north-star bound (<= 1e-3 absolute, fp32); everything else rtol/atol
        a = MASK_LOGIT_ATOL if k in ("pred_masks", "pred_masks_low") else atol
        assert torch.allclose(got, exp, rtol=rtol, atol=a), f"{tag} {k}: max abs err {err:.3e}"
    for i, a in enumerate(out["aux_outputs"]):
        assert torch.allclose(a["pred_masks"].detach().cpu().float(), g[f"{tag}_aux{i}_pred_masks"], rtol=rtol, atol=MASK_LOGIT_ATOL)
        assert torch.allclose(a["pred_boxes"].detach().cpu().float(), g[f"{tag}_aux{i}_pred_boxes"], rtol=rtol, atol=atol)
    ref_losses = meta[f"{tag}_losses"]
    assert set(losses) == set(ref_losses)
    for k, v in ref_losses.items():
        assert abs(losses[k].item() - v) <= 10 * rtol * abs(v) + 10 * atol, (k, losses[k].item(), v)
    assert abs(total.item() - g[f"{tag}_total"].item()) <= 10 * rtol * abs(g[f"{tag}_total"].item())
    params = dict(model.named_parameters())
    gn_ref = meta[f"{tag}_grad_norms"]
    bad = []
    for k, v in gn_ref.items():
        gr = params[k].grad
        if v is None:
            assert gr is None, k
            continue
        assert gr is not None, k
        n = gr.norm().item()
        if abs(n - v) > 50 * rtol * abs(v) + 50 * atol:
            bad.append((k, n, v))
    assert not bad, bad[:8]
    for key in g.keys():
        if key.startswith(f"{tag}_grad_"):
            name = key[len(f"{tag}_grad_"):]
            got, exp = params[name].grad.detach().cpu(), g[key]
            # L2-relative: a sampling point that sits within an ulp of a pixel boundary flips its bilinear cell under
            # any re-association upstream and moves single entries of the offset gradients by a few percent
            rel = (got - exp).norm().item() / (exp.norm().item() + 1e-12)
            assert rel <= 5e-2, (name, rel)
    res["total"] = total.item()
    return res


def run_eval(g, tag, device, rtol, atol):
    from ocpg_amd.util.misc import NestedTensor
    meta = g.meta
    args, model, crit = build_product(meta, device)
    B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
    x, mask, targets = cases.e2e_inputs(B, T, H, W, meta[f"{tag}_sizes"], device)
    model.eval()
    with torch.no_grad():
        out = model(NestedTensor(x, mask), text_for(B, device), targets)
    for k in ("pred_logits", "pred_boxes", "pred_masks", "reference_points"):
        got, exp = out[k].cpu().float(), g[f"{tag}_eval_{k}"]
        assert got.shape == exp.shape, (k, got.shape, exp.shape)
        a = MASK_LOGIT_ATOL if k == "pred_masks" else atol
        assert torch.allclose(got, exp, rtol=rtol, atol=a), f"{tag} eval {k}: {(got - exp).abs().max().item():.3e}"
