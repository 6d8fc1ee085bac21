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
