"""smoke(): one tiny OCPG training step (forward + criterion + backward) on the GPU, checked against the CPU oracle."""
import os
import sys

import torch


def smoke_model(dev):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import cases
    import synth
    from oracle import ocpg_ref                      # the checker (test infrastructure), never the thing shipped
    from ocpg_amd.models import build_model
    from ocpg_amd.models.text_encoder.text_encoder import PrecomputedText
    from ocpg_amd.util.misc import NestedTensor

    cfg = dict(cases.TINY)
    args = cases.default_args(device=str(dev), **cfg)
    model, crit, _ = build_model(args)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if v.dtype.is_floating_point}
    P = synth.synth_state_dict(shapes, seed=1)
    model.load_state_dict(P, strict=False)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.to(dev).train(), crit.to(dev).train()
    B, T, H, W = 1, 2, 192, 224
    x, mask, targets = cases.e2e_inputs(B, T, H, W, [(H, W)], "cpu")
    f, s, m = cases.tiny_text(B)
    out = model(NestedTensor(x.to(dev), mask.to(dev)), PrecomputedText(f.to(dev), s.to(dev), m.to(dev)),
                [{k: v.to(dev) for k, v in t.items()} for t in targets])
    losses, *_ = crit(out, [{k: v.to(dev) for k, v in t.items()} for t in targets])
    total = sum(losses[k] * crit.weight_dict[k] for k in losses if k in crit.weight_dict)
    total.backward()
    ref_out, ref_losses, ref_total = ocpg_ref.train_step_loss(P, ocpg_ref.cfg_from_args(args), x, mask, (f, s, m), targets)
    err = (out["pred_masks"].detach().cpu() - ref_out["pred_masks"]).abs().max().item()
    assert err <= 1e-3, f"mask-logit max abs err {err}"
    assert torch.equal(torch.cat([i[0] for i in out["main_matcher_index"]]).cpu(), ref_out["main_idx"]), "matcher index mismatch"
    assert abs(total.item() - ref_total.item()) <= 2e-3 * abs(ref_total.item()), (total.item(), ref_total.item())
    gn = torch.norm(torch.stack([p.grad.norm() for p in model.parameters() if p.grad is not None])).item()
    assert gn == gn and gn > 0
    print(f"smoke: tiny OCPG step on {torch.cuda.get_device_name(0)}: mask-logit max|err| {err:.2e}, matcher indices equal, "
          f"loss {total.item():.4f} (oracle {ref_total.item():.4f}), grad norm {gn:.3f}")
