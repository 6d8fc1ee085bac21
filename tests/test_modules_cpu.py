"""Product modules vs the reference's golden vectors on CPU (host logic; MSDA op replaced by the test double)."""
import pytest
import torch

import module_checks as mc

CPU = torch.device("cpu")


@pytest.fixture()
def msda_double(monkeypatch):
    from oracle.msda import MSDAOracleFunction
    import ocpg_amd.models.ops.modules.ms_deform_attn as mod
    monkeypatch.setattr(mod, "MSDeformAttnFunction", MSDAOracleFunction)


def test_lfm(golden):
    mc.check_lfm(golden("lfm"), CPU)


def test_fusion(golden):
    mc.check_fusion(golden("fusion"), CPU)


def test_msda_module(golden, msda_double):
    mc.check_msda_module(golden("msda_module"), CPU)


def test_transformer(golden, msda_double):
    mc.check_transformer(golden("transformer"), CPU)


@pytest.mark.parametrize("name,check", [("fusion_d32", "check_fusion"), ("msda_module_d32", "check_msda_module"),
                                        ("transformer_d32", "check_transformer")])
def test_head_dim_32_fixtures_host_logic(golden, msda_double, name, check):
    """The head_dim-32 reference vectors (2 heads x 32) against the host logic on the CPU; the -m gpu twins of these run the
    production HIP kernels and forbid every library fallback."""
    getattr(mc, check)(golden(name), CPU)


def test_dynmask_mso(golden):
    mc.check_dynmask_mso(golden("dynmask_mso"), CPU)


def test_matcher_criterion(golden):
    mc.check_matcher_crit(golden("matcher_crit"), CPU)


def test_mask_memo_equals_uncached():
    """Everything memoised on the tagged padding mask (level masks, 2-D position encodings) equals the uncached
    evaluation on the same mask without the tag; a different valid extent is a different cache entry."""
    from ocpg_amd.models.position_encoding import PositionEmbeddingSine2D
    from ocpg_amd.util import misc
    clips = [torch.randn(2, 3, 40, 56), torch.randn(2, 3, 33, 64)]
    nt = misc.nested_tensor_from_videos_list(clips, size_divisibility=32)
    assert misc.mask_key(nt.mask) == ("rect", 64, 64, ((40, 56), (40, 56), (33, 64), (33, 64)))
    m = nt.mask.flatten(0, 1)
    m._ocpg_key = nt.mask._ocpg_key
    plain = m.clone()
    pe = PositionEmbeddingSine2D(16, normalize=True)
    for size in ((8, 8), (4, 4), (2, 2)):
        a, b = misc.resize_mask(m, size), misc.resize_mask(plain, size)
        assert torch.equal(a, b) and misc.mask_key(a) is not None and misc.mask_key(b) is None
        assert misc.resize_mask(m, size) is a                    # second call: the cached tensor
        assert torch.equal(pe(misc.NestedTensor(None, a)), pe(misc.NestedTensor(None, b)))
    other = misc.nested_tensor_from_videos_list([torch.randn(2, 3, 40, 48), torch.randn(2, 3, 33, 64)], size_divisibility=32)
    o = other.mask.flatten(0, 1)
    o._ocpg_key = other.mask._ocpg_key
    assert not torch.equal(misc.resize_mask(o, (8, 8)), misc.resize_mask(m, (8, 8)))


def test_inverse_sigmoid_matches_reference_formula():
    from ocpg_amd.util.misc import inverse_sigmoid
    x = torch.cat([torch.rand(1000), torch.tensor([1e-4, 0.5, 1 - 1e-4, 0.3, 0.999])]).requires_grad_(True)
    ref = torch.log(x.clamp(0, 1).clamp(min=1e-5) / (1 - x.clamp(0, 1)).clamp(min=1e-5))
    got = inverse_sigmoid(x)
    assert torch.equal(got, ref)
    g0, = torch.autograd.grad(ref.sum(), x)
    g1, = torch.autograd.grad(got.sum(), x)
    assert torch.allclose(g0, g1, rtol=1e-6, atol=0)
    ends = torch.tensor([0.0, 1.0, -0.5, 1.5])
    ref_e = torch.log(ends.clamp(0, 1).clamp(min=1e-5) / (1 - ends.clamp(0, 1)).clamp(min=1e-5))
    assert (inverse_sigmoid(ends) - ref_e).abs().max() <= 1.5e-3           # saturated ends only (see the docstring)
    assert (inverse_sigmoid(ends).sigmoid() - ref_e.sigmoid()).abs().max() <= 2e-8


def test_row_split_and_layout_helpers():
    """Host logic behind the GEMM-shaped gradients and the fused casts: the row-split divisor rule, the gradient/parameter
    layout test that routes a gradient to the one-launch cast, and the no-padding predicate of a tagged mask."""
    from ocpg_amd.models import amp_cache
    from ocpg_amd.util import misc
    for m in (2400, 9600, 38400, 51000, 153600, 4097, 12345, 8191, 100003):
        s = amp_cache._split_rows(m)
        assert s >= 1 and m % s == 0
        if s > 1:
            assert 768 <= m // s <= 3072
    assert amp_cache._split_rows(2400) == 1 and amp_cache._split_rows(9600) > 1
    gy, x = torch.randn(9600, 24), torch.randn(9600, 40)
    assert torch.allclose(amp_cache.weight_grad(gy, x), gy.t() @ x, rtol=1e-4, atol=1e-3)
    w = torch.empty(64, 32, 1, 1).to(memory_format=torch.channels_last)
    assert amp_cache._same_layout(torch.empty(64, 32, 1, 1), w.shape, w.stride())              # 1x1: same memory order
    w3 = torch.empty(64, 32, 3, 3).to(memory_format=torch.channels_last)
    assert not amp_cache._same_layout(torch.empty(64, 32, 3, 3), w3.shape, w3.stride())         # 3x3 NCHW grad vs NHWC weight
    assert amp_cache._same_layout(torch.empty(64, 32, 3, 3).to(memory_format=torch.channels_last), w3.shape, w3.stride())
    assert amp_cache._dense(w3) and amp_cache._dense(torch.empty(5, 7)) and not amp_cache._dense(torch.empty(5, 8)[:, ::2])
    full = ("rect", 64, 96, ((64, 96),) * 4)
    assert misc.fully_valid(full) and misc.fully_valid(("resized", full, (8, 12)))
    assert not misc.fully_valid(("rect", 64, 96, ((64, 96), (60, 96)))) and not misc.fully_valid(None)


def test_resampling_helpers_match_interpolate():
    """models/resample.py (pure tensor programs, device-independent): bilinear as two matrix products for both align_corners
    conventions, nearest xk with a block-sum backward -- values and gradients against F.interpolate."""
    import torch.nn.functional as F
    from ocpg_amd.models.resample import bilinear_resize, nearest_upsample
    torch.manual_seed(0)
    for align in (True, False):
        for shape, size in (((2, 3, 12, 20), (48, 80)), ((1, 2, 7, 5), (4, 9)), ((1, 1, 1, 3), (2, 5))):
            x = torch.randn(*shape, dtype=torch.float64).float()
            a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
            ya, yb = bilinear_resize(a, size, align), F.interpolate(b, size=size, mode="bilinear", align_corners=align)
            assert (ya - yb).abs().max().item() <= 5e-6
            go = torch.randn_like(yb)
            ga, = torch.autograd.grad((ya * go).sum(), a)
            gb, = torch.autograd.grad((yb * go).sum(), b)
            assert (ga - gb).abs().max().item() <= 2e-5
    x = torch.randn(3, 1, 6, 10)
    a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = nearest_upsample(a, 4), F.interpolate(b, scale_factor=4)
    assert torch.equal(ya, yb)
    go = torch.randn_like(yb)
    ga, = torch.autograd.grad((ya * go).sum(), a)
    gb, = torch.autograd.grad((yb * go).sum(), b)
    assert (ga - gb).abs().max().item() <= 1e-5


def test_level_embedding_function_matches_broadcast_adds():
    """deformable_transformer._LevelPos == cat_l(pos_l + level_embed[l]) (deformable_transformer.py:158-159) incl. the gradient."""
    from ocpg_amd.models.deformable_transformer import _LevelPos
    torch.manual_seed(1)
    sizes = (12, 6, 2, 1)
    pos = [torch.randn(3, n, 8) for n in sizes]
    le = torch.randn(4, 8, requires_grad=True)
    le2 = le.detach().clone().requires_grad_(True)
    out = _LevelPos.apply(torch.cat(pos, 1), le, sizes)
    ref = torch.cat([p + le2[l].view(1, 1, -1) for l, p in enumerate(pos)], 1)
    assert torch.allclose(out, ref, atol=1e-6)
    go = torch.randn_like(ref)
    g1, = torch.autograd.grad((out * go).sum(), le)
    g2, = torch.autograd.grad((ref * go).sum(), le2)
    assert torch.allclose(g1, g2, atol=1e-5)


def test_gemm_helpers_and_groupnorm_take_the_tensor_path_off_gpu():
    """gemm_func.mm / mm_tn (plan-cache GEMMs on the GPU) and the input-projection GroupNorm module (HIP on channels-last GPU maps):
    operands that are not eligible -- here: CPU tensors -- take torch's own ops with identical semantics; the module keeps
    nn.GroupNorm's parameters / state_dict keys (checkpoints of the reference load unchanged)."""
    from ocpg_amd.models.ops.functions import gemm_func
    from ocpg_amd.models.ops.functions.groupnorm_func import GroupNorm, eligible
    torch.manual_seed(0)
    a, b, bias = torch.randn(24, 7), torch.randn(7, 5), torch.randn(5)
    assert torch.allclose(gemm_func.mm(a, b), a @ b)
    assert torch.allclose(gemm_func.mm(a, b.t().contiguous(), True, bias), a @ b + bias, atol=1e-6)
    g, x = torch.randn(24, 6), torch.randn(24, 7)
    assert torch.allclose(gemm_func.mm_tn(g, x), g.t() @ x, atol=1e-5)
    assert torch.allclose(gemm_func.mm_tn(g, x, 4), g.t() @ x, atol=1e-5)
    gn, ref = GroupNorm(4, 32), torch.nn.GroupNorm(4, 32)
    assert list(gn.state_dict().keys()) == list(ref.state_dict().keys())
    ref.load_state_dict(gn.state_dict())
    m = torch.randn(2, 32, 5, 6).contiguous(memory_format=torch.channels_last)
    assert not eligible(m, gn)
    assert torch.equal(gn(m), ref(m))


def test_deferred_partials_are_finished_on_the_slow_cast_path():
    """amp_cache.defer_sum hands slice 0 of a row-split weight gradient to autograd and leaves the other slices to the fused gradient
    cast.  A gradient whose layout does not match the cast plan takes FusedCast.backward's per-tensor path: it must finish the sum
    there (slice 0 alone is 1 / splits of the gradient) and consume the registry entry."""
    from ocpg_amd.models import amp_cache
    torch.manual_seed(0)
    amp_cache._PARTIALS.clear()
    part = torch.randn(5, 4, 6)
    g = amp_cache.defer_sum(part)
    assert g.data_ptr() in amp_cache._PARTIALS
    assert torch.allclose(amp_cache._finish_partials(g), part.sum(0), atol=1e-6) and not amp_cache._PARTIALS
    # a dense, permuted view of slice 0 (a 3x3 weight gradient computed in channels-last order)
    part = torch.randn(3, 8, 3, 3, 4)                          # [S, Co, ky, kx, Ci]
    g = amp_cache.defer_sum(part).permute(0, 3, 1, 2)          # [Co, Ci, ky, kx] with channels-last strides
    out = amp_cache._finish_partials(g)
    assert out.stride() == g.stride() and torch.allclose(out, part.sum(0).permute(0, 3, 1, 2), atol=1e-6) and not amp_cache._PARTIALS
    # fp32 partials behind a low-precision view of slice 0 (bias column sums)
    part = torch.randn(7, 16)
    g = amp_cache.defer_sum(part, torch.bfloat16)
    assert g.dtype == torch.bfloat16 and g.shape == (16,)
    out = amp_cache._finish_partials(g)
    assert out.dtype == torch.float32 and torch.allclose(out, part.sum(0), atol=1e-6) and not amp_cache._PARTIALS
    # nothing registered: the gradient passes through untouched
    t = torch.randn(3, 3)
    assert amp_cache._finish_partials(t) is t
