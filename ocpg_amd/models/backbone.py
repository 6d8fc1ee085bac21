"""ResNet-50/101 spatio-temporal backbone (frames folded into the batch) with frozen BatchNorm.

Mirrors the reference's models/backbone.py (FrozenBatchNorm2d :20-56, BackboneBase :59-85, Backbone :88-100,
Joiner :103-121, build_backbone :124-131).  The reference takes the conv stack from torchvision
(backbone.py:94-96); torchvision is not a dependency here -- the body is restated with torchvision's state_dict
names (conv1, bn1, layerK.i.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}) so reference checkpoints load.
MI355X notes: frozen BN + residual add + ReLU are ONE hand-written HIP pass (csrc/bn_act.hip) with a cached per-channel
scale/shift (the reference runs 4 elementwise kernels per BN, plus add, plus ReLU; 104 BNs in ResNet-101).
"""
import os
from typing import List

import torch
import torch.nn.functional as F
from torch import nn

from ..util.misc import NestedTensor, mask_key, resize_mask
from . import amp_cache
from .ops.functions import conv_bn_func
from .ops.functions.bn_act_func import frozen_bn_act
from .position_encoding import build_position_encoding

_STAGES = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3)}


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine parameters (all four are buffers, eps = 1e-5)."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        state_dict.pop(prefix + "num_batches_tracked", None)
        super()._load_from_state_dict(state_dict, prefix, *a, **k)

    def _apply(self, fn, *a, **k):
        self._cache = None
        return super()._apply(fn, *a, **k)

    def scale_shift(self):
        """Per-channel (scale, shift) in fp32, computed once and cached: the four tensors are frozen buffers, so the
        reference's per-forward rsqrt/mul/sub chain (backbone.py:49-55) only has to run when they are (re)loaded."""
        key = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version, self.weight.device)
        if getattr(self, "_cache", None) is None or self._cache[0] != key:
            with torch.no_grad():
                scale = (self.weight.float() * (self.running_var.float() + 1e-5).rsqrt()).contiguous()
                shift = (self.bias.float() - self.running_mean.float() * scale).contiguous()
            self._cache = (key, scale, shift)
        return self._cache[1], self._cache[2]

    def forward(self, x, skip=None, relu=False):
        """act(x*scale + shift (+ skip)): one fused HIP pass on the GPU (csrc/bn_act.hip)."""
        scale, shift = self.scale_shift()
        if x.is_cuda:
            return frozen_bn_act(x, scale, shift, skip, relu)
        # host-side (CPU) execution exists only so the model's wiring can be unit-tested without a GPU
        y = x * scale.view(1, -1, 1, 1).to(x.dtype) + shift.view(1, -1, 1, 1).to(x.dtype)
        if skip is not None:
            y = y + skip
        return F.relu(y) if relu else y


FUSED_CONV_BN = os.environ.get("OCPG_FUSED_CONV_BN", "1") != "0"     # A/B switch
MFMA_CONV3X3 = os.environ.get("OCPG_MFMA_CONV3X3", "1") != "0"     # A/B switch: 3x3 conv + BN + ReLU by csrc/conv3x3_mfma.hip (bf16, >= 128 channels)
STRIDED_1X1 = os.environ.get("OCPG_STRIDED_1X1", "1") != "0"     # A/B switch: the stride-2 projection shortcuts as subsample + GEMM (else MIOpen)
FUSED_CONV3X3_BN = os.environ.get("OCPG_FUSED_CONV3X3_BN", "0") != "0"     # opt-in: 3x3 conv + BN + ReLU as im2col + epilogue GEMM (small maps); measured neutral in the step


def conv_bn_act(conv, bn, x, skip, relu):
    """act(bn(conv(x)) (+ skip)).  1x1/stride-1 convs of channels-last GPU maps: one fused autograd node (hipBLASLt GEMM
    through the plan cache + the frozen-BN HIP kernel in place, ops/functions/conv_bn_func.py); anything else: the two
    modules in sequence."""
    if STRIDED_1X1 and FUSED_CONV_BN and amp_cache.GEMM_1X1 and conv_bn_func.eligible_s2(x, conv):
        w = amp_cache.lookup(conv.weight)
        if w.dtype == x.dtype:
            scale, shift = bn.scale_shift()
            xs = conv_bn_func.subsample2(x)
            n, _, h, wd = xs.shape
            return conv_bn_func.conv1x1_bn_act(xs, w, scale, shift, skip, relu, amp_cache._split_rows(n * h * wd) if amp_cache.SPLIT_K else 1)
    if FUSED_CONV_BN and amp_cache.GEMM_1X1 and conv_bn_func.eligible(x, conv):
        w = amp_cache.lookup(conv.weight)
        if w.dtype == x.dtype:
            scale, shift = bn.scale_shift()
            n, _, h, wd = x.shape
            return conv_bn_func.conv1x1_bn_act(x, w, scale, shift, skip, relu, amp_cache._split_rows(n * h * wd) if amp_cache.SPLIT_K else 1)
    if MFMA_CONV3X3 and skip is None and conv_bn_func.eligible3x3_mfma(x, conv):
        w = amp_cache.lookup(conv.weight)
        if w.dtype == x.dtype:
            scale, shift = bn.scale_shift()
            s = conv.stride[0]
            rows = x.shape[0] * ((x.shape[2] - 1) // s + 1) * ((x.shape[3] - 1) // s + 1)
            return conv_bn_func.conv3x3_mfma_bn_act(x, w, scale, shift, relu, s, amp_cache._split_rows(rows) if amp_cache.SPLIT_K else 1)
    if FUSED_CONV3X3_BN and skip is None and conv_bn_func.eligible3x3(x, conv):
        w = amp_cache.lookup(conv.weight)
        if w.dtype == x.dtype:
            scale, shift = bn.scale_shift()
            s = conv.stride[0]
            rows = x.shape[0] * ((x.shape[2] - 1) // s + 1) * ((x.shape[3] - 1) // s + 1)
            return conv_bn_func.conv3x3_bn_act(x, w, scale, shift, relu, s, conv.dilation[0],
                                               amp_cache._split_rows(rows) if amp_cache.SPLIT_K else 1)
    return bn(conv(x), skip=skip, relu=relu)


class Bottleneck(nn.Module):
    def __init__(self, cin, width, stride, dilation, project):
        super().__init__()
        self.conv1 = amp_cache.Conv2d(cin, width, 1, bias=False)
        self.bn1 = FrozenBatchNorm2d(width)
        self.conv2 = amp_cache.Conv2d(width, width, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = FrozenBatchNorm2d(width)
        self.conv3 = amp_cache.Conv2d(width, width * 4, 1, bias=False)
        self.bn3 = FrozenBatchNorm2d(width * 4)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(amp_cache.Conv2d(cin, width * 4, 1, stride=stride, bias=False),
                                            FrozenBatchNorm2d(width * 4))

    def forward(self, x):
        y = conv_bn_act(self.conv1, self.bn1, x, None, True)
        y = conv_bn_act(self.conv2, self.bn2, y, None, True)
        skip = x if self.downsample is None else conv_bn_act(self.downsample[0], self.downsample[1], x, None, False)
        return conv_bn_act(self.conv3, self.bn3, y, skip, True)


class ResNetBody(nn.Module):
    """conv1/bn1/maxpool + layer1..4; returns the requested stage outputs keyed "0".."3"."""

    def __init__(self, name, dilation=False, return_interm_layers=True):
        super().__init__()
        depths = _STAGES[name]
        self.conv1 = amp_cache.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBatchNorm2d(64)
        cin, dil = 64, 1
        for i, (width, n) in enumerate(zip((64, 128, 256, 512), depths)):
            stride = 1 if i == 0 else 2
            first_dil = dil
            if i == 3 and dilation:
                dil, stride = dil * 2, 1
            blocks = [Bottleneck(cin, width, stride, first_dil, project=(stride != 1 or cin != width * 4))]
            cin = width * 4
            blocks += [Bottleneck(cin, width, 1, dil, project=False) for _ in range(n - 1)]
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.return_layers = {"layer1": "0", "layer2": "1", "layer3": "2", "layer4": "3"} if return_interm_layers else {"layer4": "0"}
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        conv_bn_func.reset_skip_tokens()           # pairing of a block's conv1 with its identity skip is per forward
        x = self.bn1(self.conv1(x), relu=True)
        x = F.max_pool2d(x, 3, stride=2, padding=1)
        out = {}
        for name in ("layer1", "layer2", "layer3", "layer4"):
            x = getattr(self, name)(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


class Backbone(nn.Module):
    """ResNet body; stem + layer1 frozen (backbone.py:63-65), everything frozen when lr_backbone == 0."""

    def __init__(self, name: str, train_backbone: bool, return_interm_layers: bool, dilation: bool):
        super().__init__()
        assert name in _STAGES, f"unsupported backbone {name} (number of channels are hard coded)"
        self.body = ResNetBody(name, dilation, return_interm_layers)
        for pname, p in self.body.named_parameters():
            if not train_backbone or not any(k in pname for k in ("layer2", "layer3", "layer4")):
                p.requires_grad_(False)
        if return_interm_layers:
            self.strides, self.num_channels = [4, 8, 16, 32], [256, 512, 1024, 2048]
        else:
            self.strides, self.num_channels = [32], [2048]
        if dilation:
            self.strides[-1] //= 2

    def forward(self, tensor_list: NestedTensor):
        m = tensor_list.mask
        assert m is not None
        out = {}
        for name, x in self.body(tensor_list.tensors).items():
            mask = resize_mask(m, x.shape[-2:])
            out[name] = NestedTensor(x, mask)
        return out


class Joiner(nn.Sequential):
    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)
        self.strides = backbone.strides
        self.num_channels = backbone.num_channels

    def forward(self, tensor_list: NestedTensor):
        # NB (reference quirk, backbone.py:111-112): the caller's NestedTensor is folded IN PLACE to [(b t), ...]
        tensor_list.tensors = tensor_list.tensors.flatten(0, 1)
        key = mask_key(tensor_list.mask)
        tensor_list.mask = tensor_list.mask.flatten(0, 1)
        if key is not None:
            tensor_list.mask._ocpg_key = key
        feats: List[NestedTensor] = []
        pos = []
        for _, x in self[0](tensor_list).items():
            feats.append(x)
            pos.append(self[1](x).to(x.tensors.dtype))
        return feats, pos


def build_backbone(args):
    position_embedding = build_position_encoding(args)
    backbone = Backbone(args.backbone, args.lr_backbone > 0, bool(args.masks), args.dilation)
    model = Joiner(backbone, position_embedding)
    model.num_channels = backbone.num_channels
    return model
