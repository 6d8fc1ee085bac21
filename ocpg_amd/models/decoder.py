"""MSO -- refine the 16-channel dynamic-conv patch masks with the stride-8 and stride-4 backbone features.

Reference: models/decoder.py:14-46 (two residual 3x3 conv pairs, bilinear x2 in between, 3x3 conv to 1 channel).
Unlike the reference this does not modify the caller's `pred_masks` in place.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import amp_cache, fallbacks
from .amp_cache import lookup
from .ops.functions.mso_func import bilinear_nhwc, compute_code, conv3x3_n16, res_block_n16
from .resample import bilinear_resize

MATMUL_BILINEAR = True       # A/B switch: bilinear resampling as two matrix products (no atomics backward)
NATIVE = os.environ.get("OCPG_MSO_NATIVE", "1") != "0"      # A/B switch: csrc/mso.hip ("0" = F.conv2d -> MIOpen)
_FLOATS = (torch.float32, torch.bfloat16, torch.float16)      # index = the C ABI's dtype code


def _nhwc(f):
    """[N, C, H, W] -> [N, H, W, C] contiguous: a view of a channels-last map (the backbone's), one copy otherwise."""
    v = f.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def _tap_major(w, cdt):
    """conv weight [co, ci, 3, 3] (or a slice of its input channels) -> [co, 9, ci] in the kernels' operand type"""
    w = w.permute(0, 2, 3, 1).reshape(w.shape[0], 9, w.shape[1]).to(_FLOATS[cdt])
    return w if w.is_contiguous() else w.contiguous()


def _bias32(b):
    b = lookup(b)           # under autocast the convolution adds the bias rounded to the autocast dtype
    return b if b.dtype == torch.float32 else b.float()


class MSO(nn.Module):
    def __init__(self, mask_dim=16, img_dim=(96, 192), out_dim=16):
        super().__init__()
        self.mask_dim, self.img_dim, self.out_dim = mask_dim, list(img_dim), out_dim
        self.conv1_1div8 = amp_cache.Conv2d(mask_dim + img_dim[1], mask_dim, kernel_size=3, padding=1)
        self.conv2_1div8 = amp_cache.Conv2d(mask_dim, mask_dim, kernel_size=3, padding=1)
        self.conv1_1div4 = amp_cache.Conv2d(mask_dim + img_dim[0], mask_dim, kernel_size=3, padding=1)
        self.conv2_1div4 = amp_cache.Conv2d(mask_dim, mask_dim, kernel_size=3, padding=1)
        self.out_conv = amp_cache.Conv2d(mask_dim, 1, kernel_size=3, padding=1)

    def native_ok(self, pm, f4, f8):
        """csrc/mso.hip serves GPU maps of any size with a 16-channel (<= 16, multiple of 4) mask path."""
        if not (NATIVE and pm.is_cuda):
            return False
        if self.mask_dim > 16 or self.mask_dim % 4 or pm.dtype not in _FLOATS or f4.dtype not in _FLOATS or f8.dtype not in _FLOATS:
            fallbacks.note("MSO", f"mask_dim {self.mask_dim} / dtypes {pm.dtype}, {f4.dtype}, {f8.dtype} not served by csrc/mso.hip", pm)
            return False
        if pm.shape[0] > 65535:        # the kernels put the image index on a 16-bit grid axis (ocpg_mso_conv3x3 returns -1002 beyond it)
            fallbacks.note("MSO", f"{pm.shape[0]} mask maps in one call exceed the kernel's grid axis", pm)
            return False
        return True

    def forward_native(self, pm, f4, f8):
        """pm [n * bt, 16, h, w] (n mask sets over the SAME bt feature maps, set-major) -> [n * bt, 1, 2h, 2w] fp32.
        conv(cat[relu(m), relu(f)]) = conv_m(relu(m)) + conv_f(relu(f)): the feature halves of conv1_1div8 / conv1_1div4 (512 -> 16 and
        256 -> 16, ~95 % of MSO's MACs) are evaluated once and added to every set inside the mask-half kernel's epilogue."""
        c = self.mask_dim
        cdt = compute_code()
        p = _nhwc(pm if pm.dtype == torch.float32 else pm.float())
        for f, conv1, conv2 in ((f8, self.conv1_1div8, self.conv2_1div8), (f4, self.conv1_1div4, self.conv2_1div4)):
            if tuple(p.shape[1:3]) != tuple(f.shape[-2:]):
                p = bilinear_nhwc(p, f.shape[-2:])
            w1 = lookup(conv1.weight)
            shared = conv3x3_n16(_nhwc(f), _tap_major(w1[:, c:], cdt), _bias32(conv1.bias), relu_in=True, cdt=cdt)          # [bt, H, W, 16]
            p = res_block_n16(p, shared, _tap_major(w1[:, :c], cdt), _tap_major(lookup(conv2.weight), cdt), _bias32(conv2.bias), cdt)
        out = conv3x3_n16(p, _tap_major(lookup(self.out_conv.weight), cdt), _bias32(self.out_conv.bias), cdt=cdt)             # [N, 2h, 2w, 1]
        return out.permute(0, 3, 1, 2)

    def forward(self, pred_masks, image_features):
        f4, f8 = (x.tensors for x in image_features)           # stride 4, stride 8
        assert pred_masks.shape[-1] == f8.shape[-1], "First size wrong."
        if self.native_ok(pred_masks, f4, f8):
            return self.forward_native(pred_masks, f4, f8)
        x = F.relu(torch.cat([pred_masks, f8.to(pred_masks.dtype)], dim=1))
        pred_masks = pred_masks + self.conv2_1div8(F.relu(self.conv1_1div8(x)))
        pred_masks = F.interpolate(pred_masks, size=f4.shape[-2:], mode="bilinear", align_corners=False)
        assert pred_masks.shape[-1] == f4.shape[-1], "Second size wrong."
        x = F.relu(torch.cat([pred_masks, f4.to(pred_masks.dtype)], dim=1))
        pred_masks = pred_masks + self.conv2_1div4(F.relu(self.conv1_1div4(x)))
        return self.out_conv(pred_masks)


def _conv_input(f, dt):
    """A backbone feature as the input of an autocast conv: the reference's torch.cat([pm, f], 1) promotes the half-precision
    feature to pm's fp32 and the conv casts it straight back -- a lossless round trip, skipped here (two full-map casts
    forward, two backward per level)."""
    if f.is_cuda and torch.is_autocast_enabled("cuda") and f.dtype == torch.get_autocast_dtype("cuda"):
        return f
    return f.to(dt)


def _mso_forward_multi(self, pred_masks_list, image_features, stacked=False):
    """Refine several mask sets (one per decoder layer) that share the SAME backbone features.

    conv(cat[relu(m), relu(f)]) = conv_m(relu(m)) + conv_f(relu(f)): the feature halves of conv1_1div8 / conv1_1div4
    (512->16 and 256->16 3x3 convs, ~95 % of MSO's MACs) do not depend on the layer, so they are evaluated once and
    added to every layer's mask half.  Same result as calling forward() per layer, up to fp32 summation order."""
    f4, f8 = (x.tensors for x in image_features)
    n = len(pred_masks_list)
    c = self.mask_dim
    pm = torch.cat(pred_masks_list, 0)
    dt = pm.dtype
    assert pm.shape[-1] == f8.shape[-1], "First size wrong."
    if self.native_ok(pm, f4, f8):
        out = self.forward_native(pm, f4, f8)
        return out if stacked else list(out.chunk(n, 0))
    w8 = lookup(self.conv1_1div8.weight)
    shared8 = F.conv2d(F.relu(_conv_input(f8, dt)), w8[:, c:], lookup(self.conv1_1div8.bias), padding=1)
    y = F.conv2d(F.relu(pm), w8[:, :c], None, padding=1) + shared8.repeat(n, 1, 1, 1)
    pm = pm + self.conv2_1div8(F.relu(y))
    pm = (bilinear_resize(pm, tuple(f4.shape[-2:]), False) if pm.is_cuda and MATMUL_BILINEAR
          else F.interpolate(pm, size=f4.shape[-2:], mode="bilinear", align_corners=False))
    assert pm.shape[-1] == f4.shape[-1], "Second size wrong."
    w4 = lookup(self.conv1_1div4.weight)
    shared4 = F.conv2d(F.relu(_conv_input(f4, dt)), w4[:, c:], lookup(self.conv1_1div4.bias), padding=1)
    y = F.conv2d(F.relu(pm), w4[:, :c], None, padding=1) + shared4.repeat(n, 1, 1, 1)
    pm = pm + self.conv2_1div4(F.relu(y))
    out = self.out_conv(pm)
    return out if stacked else list(out.chunk(n, 0))


MSO.forward_multi = _mso_forward_multi
