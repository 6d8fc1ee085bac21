"""MSO -- refine the 16-channel dynamic-conv patch masks with the stride-8 and stride-4 backbone features.

Reference: models/decoder.py:14-46 (two residual 3x3 conv pairs, bilinear x2 in between, 3x3 conv to 1 channel).
Unlike the reference this does not modify the caller's `pred_masks` in place.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import amp_cache
from .amp_cache import lookup
from .resample import bilinear_resize

MATMUL_BILINEAR = True       # A/B switch: bilinear resampling as two matrix products (no atomics backward)


class MSO(nn.Module):
    def __init__(self, mask_dim=16, img_dim=(96, 192), out_dim=16):
        super().__init__()
        self.mask_dim, self.img_dim, self.out_dim = mask_dim, list(img_dim), out_dim
        self.conv1_1div8 = amp_cache.Conv2d(mask_dim + img_dim[1], mask_dim, kernel_size=3, padding=1)
        self.conv2_1div8 = amp_cache.Conv2d(mask_dim, mask_dim, kernel_size=3, padding=1)
        self.conv1_1div4 = amp_cache.Conv2d(mask_dim + img_dim[0], mask_dim, kernel_size=3, padding=1)
        self.conv2_1div4 = amp_cache.Conv2d(mask_dim, mask_dim, kernel_size=3, padding=1)
        self.out_conv = amp_cache.Conv2d(mask_dim, 1, kernel_size=3, padding=1)

    def forward(self, pred_masks, image_features):
        f4, f8 = (x.tensors for x in image_features)           # stride 4, stride 8
        assert pred_masks.shape[-1] == f8.shape[-1], "First size wrong."
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
