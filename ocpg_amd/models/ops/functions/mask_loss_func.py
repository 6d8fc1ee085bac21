"""Autograd bindings of the fused mask-criterion kernels (csrc/levelset.hip, csrc/proj.hip): level-set loss and
box-projection loss for all decoder layers per call (reference models/segmentation.py:203-211,253-315)."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ...._lib import check, lib


def _st():
    return torch.cuda.current_stream().cuda_stream


class LevelSetLoss(Function):
    @staticmethod
    def forward(ctx, x, feats, box, C):
        """x [Lr,N,h,w] logits, feats [N,CF,h,w] (first C channels used), box [N,h,w] -> [Lr]."""
        x, feats, box = x.float().contiguous(), feats.float().contiguous(), box.float().contiguous()
        lr, n, h, w = x.shape
        cf = feats.shape[1]
        assert feats.shape == (n, cf, h, w) and box.shape == (n, h, w) and 0 < C <= cf
        sums = torch.empty((lr, n, (h * w + 1023) // 1024, 7 + 2 * C), dtype=torch.float32, device=x.device)
        coef = torch.empty((lr, n, 8 + 2 * C), dtype=torch.float32, device=x.device)
        loss = torch.empty((lr,), dtype=torch.float32, device=x.device)
        check(lib().ocpg_levelset_fwd_f32(x.data_ptr(), feats.data_ptr(), box.data_ptr(), lr, n, C, cf, h, w, sums.data_ptr(),
                                          coef.data_ptr(), loss.data_ptr(), _st()), "ocpg_levelset_fwd_f32")
        ctx.save_for_backward(x, feats, box, coef)
        ctx.C = C
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss):
        x, feats, box, coef = ctx.saved_tensors
        lr, n, h, w = x.shape
        gx = torch.empty_like(x)
        gfeat = torch.empty_like(feats) if ctx.needs_input_grad[1] else None
        check(lib().ocpg_levelset_bwd_f32(x.data_ptr(), feats.data_ptr(), box.data_ptr(), coef.data_ptr(), gloss.float().contiguous().data_ptr(),
                                          lr, n, ctx.C, feats.shape[1], h, w, gx.data_ptr(), None if gfeat is None else gfeat.data_ptr(), _st()),
              "ocpg_levelset_bwd_f32")
        return gx, gfeat, None, None


class ProjLoss(Function):
    @staticmethod
    def forward(ctx, x, tcmax, trmax, tcmean, trmean):
        """x [Lr,B,T,H,W] logits; targets [B,T,W] / [B,T,H] (see include/ocpg_hip.h) -> [Lr]."""
        x = x.float().contiguous()
        tcmax, trmax, tcmean, trmean = (t.float().contiguous() for t in (tcmax, trmax, tcmean, trmean))
        lr, b, t, h, w = x.shape
        assert tcmax.shape == tcmean.shape == (b, t, w) and trmax.shape == trmean.shape == (b, t, h)
        f = lr * b * t
        colstat = torch.empty((f, 3, w), dtype=torch.float32, device=x.device)
        rowstat = torch.empty((f, 3, h), dtype=torch.float32, device=x.device)
        iu = torch.empty((lr, b, 4, 2), dtype=torch.float32, device=x.device)
        loss = torch.empty((lr,), dtype=torch.float32, device=x.device)
        check(lib().ocpg_proj_fwd_f32(x.data_ptr(), tcmax.data_ptr(), trmax.data_ptr(), tcmean.data_ptr(), trmean.data_ptr(), lr, b, t, h, w,
                                      colstat.data_ptr(), rowstat.data_ptr(), iu.data_ptr(), loss.data_ptr(), _st()), "ocpg_proj_fwd_f32")
        ctx.save_for_backward(x, tcmax, trmax, tcmean, trmean, colstat, rowstat, iu)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss):
        x, tcmax, trmax, tcmean, trmean, colstat, rowstat, iu = ctx.saved_tensors
        lr, b, t, h, w = x.shape
        f = lr * b * t
        gc = torch.empty((f, 2, w), dtype=torch.float32, device=x.device)
        gr = torch.empty((f, 2, h), dtype=torch.float32, device=x.device)
        gx = torch.empty_like(x)
        check(lib().ocpg_proj_bwd_f32(x.data_ptr(), tcmax.data_ptr(), trmax.data_ptr(), tcmean.data_ptr(), trmean.data_ptr(), colstat.data_ptr(),
                                      rowstat.data_ptr(), iu.data_ptr(), gloss.float().contiguous().data_ptr(), lr, b, t, h, w, gc.data_ptr(),
                                      gr.data_ptr(), gx.data_ptr(), _st()), "ocpg_proj_bwd_f32")
        return gx, None, None, None, None


def levelset_loss(x, feats, box, C):
    return LevelSetLoss.apply(x, feats, box, C)


def proj_loss(x, region, weak):
    """x [Lr,B,T,H,W]; region, weak [B,T,H,W] (targets, no gradient)."""
    with torch.no_grad():
        tcmax, trmax, tcmean, trmean = region.amax(2), region.amax(3), weak.mean(2), weak.mean(3)
    return ProjLoss.apply(x, tcmax, trmax, tcmean, trmean)


class DetLosses(Function):
    """Focal classification + L1 + GIoU losses of the matched queries for all layers (csrc/det_loss.hip) -> [3, Lr]."""

    @staticmethod
    def forward(ctx, logits, boxes, src, valid, labels, tboxes, num_boxes, alpha, bad):
        logits, boxes = logits.float().contiguous(), boxes.float().contiguous()
        lr, b, t, q, k = logits.shape
        src = src.to(torch.int64).contiguous()
        valid, tboxes = valid.float().contiguous(), tboxes.float().contiguous()
        labels = None if labels is None else labels.to(torch.int64).contiguous()
        nb = num_boxes.float().reshape(1).contiguous()
        loss = torch.empty((3, lr), dtype=torch.float32, device=logits.device)
        check(lib().ocpg_det_loss_fwd_f32(logits.data_ptr(), boxes.data_ptr(), src.data_ptr(), valid.data_ptr(),
                                          None if labels is None else labels.data_ptr(), tboxes.data_ptr(), nb.data_ptr(), float(alpha), lr, b, t, q,
                                          k, loss.data_ptr(), None if bad is None else bad.data_ptr(), _st()), "ocpg_det_loss_fwd_f32")
        ctx.save_for_backward(logits, boxes, src, valid, tboxes, nb, *(() if labels is None else (labels,)))
        ctx.alpha = float(alpha)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss):
        logits, boxes, src, valid, tboxes, nb, *rest = ctx.saved_tensors
        labels = rest[0] if rest else None
        lr, b, t, q, k = logits.shape
        glogits, gboxes = torch.empty_like(logits), torch.empty_like(boxes)
        check(lib().ocpg_det_loss_bwd_f32(logits.data_ptr(), boxes.data_ptr(), src.data_ptr(), valid.data_ptr(),
                                          None if labels is None else labels.data_ptr(), tboxes.data_ptr(), nb.data_ptr(),
                                          gloss.float().contiguous().data_ptr(), ctx.alpha, lr, b, t, q, k, glogits.data_ptr(), gboxes.data_ptr(), _st()),
              "ocpg_det_loss_bwd_f32")
        return glogits, gboxes, None, None, None, None, None, None, None


def det_losses(logits, boxes, src, valid, labels, tboxes, num_boxes, alpha, bad=None):
    return DetLosses.apply(logits, boxes, src, valid, labels, tboxes, num_boxes, alpha, bad)


class MaskedCE(Function):
    """mean over the clip of BCE-with-logits(sigmoid(x) * w, m * w) for all layers (csrc/masked_ce.hip) -> [Lr]."""

    @staticmethod
    def forward(ctx, x, w, t):
        x, w, t = x.float().contiguous(), w.float().contiguous(), t.float().contiguous()
        lr = x.shape[0]
        per = x.numel() // lr
        assert w.numel() == per and t.numel() == per and per % 4 == 0
        part = torch.empty((lr, 512), dtype=torch.float32, device=x.device)
        check(lib().ocpg_masked_ce_fwd_f32(x.data_ptr(), w.data_ptr(), t.data_ptr(), lr, per, part.data_ptr(), _st()), "ocpg_masked_ce_fwd_f32")
        ctx.save_for_backward(x, w, t)
        return part.sum(1) / per

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss):
        x, w, t = ctx.saved_tensors
        lr = x.shape[0]
        gx = torch.empty_like(x)
        check(lib().ocpg_masked_ce_bwd_f32(x.data_ptr(), w.data_ptr(), t.data_ptr(), gloss.float().contiguous().data_ptr(), lr, x.numel() // lr,
                                           gx.data_ptr(), _st()), "ocpg_masked_ce_bwd_f32")
        return gx, None, None


def masked_ce(x, w, m):
    """x [Lr,B,T,H,W] logits; w, m [B,T,H,W] weight map and weak mask (no gradient)."""
    return MaskedCE.apply(x, w, m * w)
