"""Pin the full-model CPU oracle (oracle/ocpg_ref.py) to the reference's golden end-to-end vectors."""
import pytest
import torch

import cases
import synth
from oracle import ocpg_ref
from model_checks import MASK_LOGIT_ATOL


def _setup(g, tag):
    meta = g.meta
    cfg = ocpg_ref.cfg_from_args(cases.default_args(**meta["cfg"]))
    P = synth.synth_state_dict(meta["float_shapes"], seed=meta["seed"])
    B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
    x, mask, targets = cases.e2e_inputs(B, T, H, W, meta[f"{tag}_sizes"])
    return meta, cfg, P, x, mask, cases.tiny_text(B), targets


@pytest.mark.parametrize("tag", ["nopad", "pad"])
def test_train_step(golden, tag):
    g = golden("e2e_tiny")
    meta, cfg, P, x, mask, text, targets = _setup(g, tag)
    watch = ["query_embed.weight", "transformer.level_embed", "class_embed.1.weight", "controller.layers.2.bias",
             "transformer.encoder.layers.0.self_attn.sampling_offsets.bias", "input_proj.3.0.bias",
             "backbone.0.body.layer2.0.conv1.weight", "mask_refine.out_conv.weight"]
    for k in watch:
        P[k].requires_grad_(True)
    out, losses, total = ocpg_ref.train_step_loss(P, cfg, x, mask, text, targets)
    assert torch.equal(out["main_idx"], g[f"{tag}_main_idx"].flatten())
    assert torch.equal(torch.stack(out["aux_idx"]), g[f"{tag}_aux_idx"].flatten(1))
    for k in ("pred_logits", "pred_boxes", "pred_masks", "pred_masks_low"):
        a = MASK_LOGIT_ATOL if "mask" in k else 2e-5
        assert torch.allclose(out[k], g[f"{tag}_{k}"], rtol=2e-4, atol=a), (k, (out[k] - g[f"{tag}_{k}"]).abs().max())
    assert torch.allclose(out["ls_features"][:, :, :11], g[f"{tag}_ls_features"][:, :, :11], rtol=2e-4, atol=2e-5)
    ref = meta[f"{tag}_losses"]
    assert set(ref) == set(losses)
    for k, v in ref.items():
        assert abs(losses[k].item() - v) <= 2e-3 * abs(v) + 2e-4, (k, losses[k].item(), v)
    assert abs(total.item() - g[f"{tag}_total"].item()) <= 2e-3 * abs(g[f"{tag}_total"].item())
    grads = torch.autograd.grad(total, [P[k] for k in watch])
    for k, gr in zip(watch, grads):
        exp = g[f"{tag}_grad_{k}"]
        assert (gr - exp).abs().max().item() <= 1e-2 * exp.abs().max().item() + 1e-5, k


def test_eval_tail(golden):
    g = golden("e2e_tiny")
    meta, cfg, P, x, mask, text, targets = _setup(g, "pad")
    with torch.no_grad():
        out = ocpg_ref.forward(P, cfg, x, mask, text, targets, train=False)
    for k in ("pred_logits", "pred_boxes", "pred_masks", "reference_points"):
        a = MASK_LOGIT_ATOL if k == "pred_masks" else 2e-5
        assert torch.allclose(out[k], g[f"pad_eval_{k}"], rtol=2e-4, atol=a), k


def test_fp64_referee_brackets_the_reference_vectors(golden):
    """The oracle evaluated in float64 (`ocpg_ref.real`) is the referee of the full-size GPU test.  On the tiny fixture it must agree
    with the reference's own (fp32) outputs as well as the fp32 oracle does -- same assignment, mask logits within the north-star
    bound -- and the fp32 oracle's distance from it is the fp32 round-off the GPU test then allows the product."""
    g = golden("e2e_tiny")
    meta, cfg, P, x, mask, text, targets = _setup(g, "nopad")
    d = torch.float64
    with torch.no_grad():
        o32, l32, t32 = ocpg_ref.train_step_loss(P, cfg, x, mask, text, targets)
        with ocpg_ref.real(d):
            o64, l64, t64 = ocpg_ref.train_step_loss(ocpg_ref.as_real(P, d), cfg, x.double(), mask, ocpg_ref.as_real(text, d),
                                                     ocpg_ref.as_real(targets, d))
    assert ocpg_ref.REAL == torch.float32
    assert o64["pred_masks"].dtype == d and torch.equal(o64["main_idx"], g["nopad_main_idx"].flatten())
    for k in ("pred_masks", "pred_masks_low"):
        ref = g[f"nopad_{k}"].double()
        e_ref, e_32 = (o64[k] - ref).abs().max().item(), (o64[k] - o32[k].double()).abs().max().item()
        assert e_ref <= MASK_LOGIT_ATOL and e_32 <= MASK_LOGIT_ATOL, (k, e_ref, e_32)
    assert abs(float(t64) - g["nopad_total"].item()) <= 2e-4 * abs(float(t64))
