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


def test_matcher_criterion(golden, dev):
    mc.check_matcher_crit(golden("matcher_crit"), dev, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("tag", ["nopad", "pad"])
def test_train_step_matches_reference(golden, dev, tag):
    """fp32 parity mode: mask logits <= 1e-3 abs (north star), matcher indices bit-exact, 18 losses, all grad norms."""
    res = model_checks.run_train_step(golden("e2e_tiny"), tag, dev, rtol=1e-3, atol=1e-4)
    print(res)


@pytest.mark.parametrize("tag", ["nopad", "pad"])
def test_eval_tail_matches_reference(golden, dev, tag):
    model_checks.run_eval(golden("e2e_tiny"), tag, dev, rtol=1e-3, atol=1e-4)


# ---- Video-Swin backbone (BASELINE configs #4 / #5) ----------------------------------------------------------
def test_swin_window_attention_and_masks(golden, dev):
    import swin_checks as sc
    sc.check_window_attention(golden("swin3d"), dev, rtol=5e-4, atol=5e-5)


def test_swin_block_and_backbone(golden, dev):
    import swin_checks as sc
    sc.check_block_and_merging(golden("swin3d"), dev, rtol=5e-4, atol=1e-4)
    sc.check_backbone(golden("swin3d"), dev, rtol=1e-3, atol=2e-4)


def test_e2e_with_video_swin(golden, dev):
    import swin_checks as sc
    sc.check_e2e_swin(golden("e2e_swin"), dev, rtol=1e-3, atol=1e-4)


def test_fused_autocast_param_cast_equals_per_op_casts(golden, dev):
    """amp_cache (one fused cast of all parameters per forward) must give what autocast's own per-op casts give (same
    casts, same kernels; run-to-run the bf16 GEMM/conv kernels differ in the last bf16 bit, hence the norm-relative bound)."""
    import cases
    from ocpg_amd.models import amp_cache
    from ocpg_amd.util.misc import NestedTensor
    g = golden("e2e_tiny")
    meta = g.meta
    res = []
    for enabled in (True, False):
        amp_cache.ENABLED = enabled
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
            gn = {k: p.grad.float().norm().item() for k, p in model.named_parameters() if p.grad is not None}
            res.append((out["pred_masks"].detach().float().cpu(), total.item(), gn))
        finally:
            amp_cache.ENABLED = True
    (m1, t1, g1), (m0, t0, g0) = res
    assert (m1 - m0).norm() <= 1e-2 * m0.norm() and abs(t1 - t0) <= 1e-2 * abs(t0)
    assert set(g1) == set(g0)
    bad = [(k, g1[k], g0[k]) for k in g1 if abs(g1[k] - g0[k]) > 5e-2 * abs(g0[k]) + 1e-5]
    assert not bad, bad[:5]
