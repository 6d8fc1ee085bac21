"""Fused frozen-BN (+ residual) (+ ReLU) HIP kernel vs the reference arithmetic (models/backbone.py:46-56 + add + ReLU)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x, w, b, rm, rv, skip, relu):
    """The reference's op order, in fp32 on the CPU."""
    scale = w.reshape(1, -1, 1, 1) * (rv.reshape(1, -1, 1, 1) + 1e-5).rsqrt()
    y = x * scale + (b.reshape(1, -1, 1, 1) - rm.reshape(1, -1, 1, 1) * scale)
    if skip is not None:
        y = y + skip
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["nhwc", "nchw"])
@pytest.mark.parametrize("shape", [(2, 64, 12, 20), (3, 256, 7, 5), (1, 24, 3, 3), (2, 2048, 12, 20)])
@pytest.mark.parametrize("with_skip,relu", [(False, True), (True, True), (False, False)])
def test_bn_act_matches_reference(dev, dtype, layout, shape, with_skip, relu):
    from ocpg_amd.models.backbone import FrozenBatchNorm2d
    g = torch.Generator().manual_seed(sum(shape))
    n, c, h, w = shape
    bn = FrozenBatchNorm2d(c)
    bn.weight.copy_(1 + 0.1 * torch.randn(c, generator=g)); bn.bias.copy_(0.1 * torch.randn(c, generator=g))
    bn.running_mean.copy_(0.1 * torch.randn(c, generator=g)); bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
    x = torch.randn(shape, generator=g).to(dtype)
    skip = torch.randn(shape, generator=g).to(dtype) if with_skip else None
    go = torch.randn(shape, generator=g).to(dtype)
    xr = x.float().clone().requires_grad_(True)
    sr = skip.float().clone().requires_grad_(True) if with_skip else None
    yr = _ref(xr, bn.weight, bn.bias, bn.running_mean, bn.running_var, sr, relu)
    yr.backward(go.float())
    fmt = torch.channels_last if layout == "nhwc" else torch.contiguous_format
    bn.to(dev)
    xd = x.to(dev).contiguous(memory_format=fmt).requires_grad_(True)
    sd = skip.to(dev).contiguous(memory_format=fmt).requires_grad_(True) if with_skip else None
    y = bn(xd, skip=sd, relu=relu)
    assert y.dtype == dtype and y.stride() == xd.stride()
    y.backward(go.to(dev).contiguous(memory_format=fmt))
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=1.6e-2, atol=1.6e-2)
    assert torch.allclose(y.detach().cpu().float(), yr.detach(), **tol)
    # gradient masks come from the (possibly bf16-rounded) output: compare away from the ReLU kink
    safe = (yr.detach().abs() > 0.05) if relu else torch.ones_like(yr, dtype=torch.bool)
    assert torch.allclose(xd.grad.cpu().float()[safe], xr.grad[safe], **tol)
    if with_skip:
        assert torch.allclose(sd.grad.cpu().float()[safe], sr.grad[safe], **tol)


def test_cache_follows_buffer_updates(dev):
    from ocpg_amd.models.backbone import FrozenBatchNorm2d
    bn = FrozenBatchNorm2d(8).to(dev)
    x = torch.randn(2, 8, 4, 4, device=dev)
    y0 = bn(x)
    bn.load_state_dict({"weight": torch.full((8,), 2.0), "bias": torch.ones(8), "running_mean": torch.zeros(8), "running_var": torch.ones(8)})
    y1 = bn(x)
    assert torch.allclose(y1, x * (2.0 / (1 + 1e-5) ** 0.5) + 1.0, rtol=1e-5, atol=1e-6) and not torch.allclose(y0, y1)
