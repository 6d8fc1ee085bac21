"""Dynamic mask head: autograd binding of ocpg_dynmask_fwd_f32 (csrc/dynmask.hip).

Forward is the fused HIP kernel (coordinate channels + both per-query 1x1 convs, one pass over the features).  The
backward is three batched GEMMs plus a handful of reductions on the saved layer-1 pre-activation -- every large
contraction of the backward (dW0 = dpre . feats^T, dfeat = W0^T . dpre, dh = W1^T . dout) is GEMM-shaped and goes to
hipBLASLt; no [b*t*q*(C+2), h, w] tensor is ever formed (models/ocpg.py:513-517 materialises 99 MB per call).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ...._lib import check, lib, stream_ptr

CH = 16


class DynamicMaskFunction(Function):
    @staticmethod
    def forward(ctx, feats, params, refpix, stride):
        """feats [BT,C,H,W]; params [BT*Q, NP]; refpix [BT*Q, 2] (input pixels) -> [BT*Q, 16, H, W]; all fp32."""
        if not feats.is_cuda:
            raise RuntimeError("DynamicMaskFunction: feats must be a GPU tensor: Not implemented on the CPU")
        feats, params, refpix = feats.float().contiguous(), params.float().contiguous(), refpix.float().contiguous()
        bt, c, h, w = feats.shape
        n = params.shape[0]
        q = n // bt
        assert n == bt * q and params.shape[1] == (c + 2) * CH + CH * CH + 2 * CH
        out = torch.empty((n, CH, h, w), dtype=torch.float32, device=feats.device)
        pre1 = torch.empty_like(out)
        with torch.cuda.device(feats.device):
            check(lib().ocpg_dynmask_fwd_f32(feats.data_ptr(), params.data_ptr(), refpix.data_ptr(), bt, q, c, h, w, int(stride),
                                             out.data_ptr(), pre1.data_ptr(), stream_ptr()), "ocpg_dynmask_fwd_f32")
        ctx.save_for_backward(feats, params, refpix, pre1)
        ctx.stride = int(stride)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        feats, params, refpix, pre1 = ctx.saved_tensors
        bt, c, h, w = feats.shape
        n, hw = params.shape[0], h * w
        q = n // bt
        s = ctx.stride
        dout = dout.float().reshape(n, CH, hw)
        pre = pre1.view(n, CH, hw)
        w0 = params[:, :(c + 2) * CH].view(n, CH, c + 2)
        w1 = params[:, (c + 2) * CH:(c + 2) * CH + CH * CH].view(n, CH, CH)
        hact = pre.clamp(min=0)
        dw1 = torch.bmm(dout, hact.transpose(1, 2))                                   # [n,16,16]
        db1 = dout.sum(-1)
        dpre = torch.bmm(w1.transpose(1, 2), dout) * (pre > 0)                        # [n,16,hw]
        db0 = dpre.sum(-1)                                                            # [n,16]
        xs = (torch.arange(w, device=feats.device, dtype=torch.float32) * s + s // 2).repeat(h)
        ys = (torch.arange(h, device=feats.device, dtype=torch.float32) * s + s // 2).repeat_interleave(w)
        mom = torch.matmul(dpre, torch.stack([xs, ys], 1))                            # [n,16,2]: sum_px dpre * (x, y)
        dwx = refpix[:, 0:1] * db0 - mom[..., 0]
        dwy = refpix[:, 1:2] * db0 - mom[..., 1]
        dref = torch.stack([(w0[..., c] * db0).sum(-1), (w0[..., c + 1] * db0).sum(-1)], -1) if ctx.needs_input_grad[2] else None
        f = feats.view(bt, c, hw)
        dpre_bt = dpre.view(bt, q * CH, hw)
        dw0 = torch.bmm(dpre_bt, f.transpose(1, 2)).view(n, CH, c)                    # [n,16,C]
        dfeat = None
        if ctx.needs_input_grad[0]:
            dfeat = torch.bmm(w0[..., :c].reshape(bt, q * CH, c).transpose(1, 2), dpre_bt).view(bt, c, h, w)
        dparams = torch.cat([torch.cat([dw0, dwx[..., None], dwy[..., None]], -1).flatten(1), dw1.flatten(1), db0, db1], 1)
        return dfeat, dparams, dref, None


def dynamic_mask(feats, params, refpix, stride):
    return DynamicMaskFunction.apply(feats, params, refpix, stride)
