"""Dynamic mask head: autograd binding of ocpg_dynmask_fwd_f32 (csrc/dynmask.hip).

Forward is the fused HIP kernel (coordinate channels + both per-query 1x1 convs, one pass over the features).  The
backward is two HIP launches (dpre and every small reduction; assembly of dparams) plus the two large contractions
(dW0 = dpre . feats^T, dfeat = W0^T . dpre) as hipBLASLt GEMMs; no [b*t*q*(C+2), h, w] tensor is ever formed
(models/ocpg.py:513-517 materialises 99 MB per call).
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
        """Two HIP launches (csrc/dynmask.hip: dpre + every small reduction; assembly of dparams / dref) around the two
        GEMM-shaped contractions dW0 = dpre . feats^T and dfeat = W0^T . dpre (hipBLASLt through ocpg_gemm, written
        straight into their destinations): 4 launches for all decoder layers, no intermediate of the reference's
        [1, b*t*q*(C+2), h, w] size (models/ocpg.py:513-517)."""
        feats, params, refpix, pre1 = ctx.saved_tensors
        bt, c, h, w = feats.shape
        n, hw = params.shape[0], h * w
        q = n // bt
        dev = feats.device
        dout = dout.float().contiguous()
        strips = (hw + 255) // 256
        dpre = torch.empty((n, CH, hw), dtype=torch.float32, device=dev)
        part = torch.empty((n, strips, 320), dtype=torch.float32, device=dev)
        w0d = torch.empty((n, CH, c), dtype=torch.float32, device=dev)
        dw0 = torch.empty((n, CH, c), dtype=torch.float32, device=dev)
        dparams = torch.empty_like(params)
        need_ref, need_feat = ctx.needs_input_grad[2], ctx.needs_input_grad[0]
        dref = torch.empty((n, 2), dtype=torch.float32, device=dev) if need_ref else None
        dfeat = torch.empty_like(feats) if need_feat else None
        L = lib()
        with torch.cuda.device(dev):
            st = stream_ptr()
            check(L.ocpg_dynmask_bwd_pre_f32(dout.data_ptr(), pre1.data_ptr(), params.data_ptr(), bt, q, c, h, w, ctx.stride, dpre.data_ptr(),
                                             part.data_ptr(), w0d.data_ptr(), st), "ocpg_dynmask_bwd_pre_f32")
            # dW0 [bt, q*16, C] = dpre [bt, q*16, hw] . feats^T (feats stored [C, hw] = B^T)
            check(L.ocpg_gemm(dpre.data_ptr(), feats.data_ptr(), dw0.data_ptr(), None, 0, 0, 0, 1, q * CH, c, hw, hw, hw, c, bt,
                              q * CH * hw, c * hw, q * CH * c, 1.0, 0.0, st), "ocpg_gemm (dynmask dW0)")
            if need_feat:
                # dfeat [bt, C, hw] = W0^T [C, q*16] . dpre [q*16, hw]  (W0 stored [q*16, C] = A^T)
                check(L.ocpg_gemm(w0d.data_ptr(), dpre.data_ptr(), dfeat.data_ptr(), None, 0, 0, 1, 0, c, hw, q * CH, c, hw, hw, bt,
                                  q * CH * c, q * CH * hw, c * hw, 1.0, 0.0, st), "ocpg_gemm (dynmask dfeat)")
            check(L.ocpg_dynmask_bwd_fin_f32(part.data_ptr(), params.data_ptr(), refpix.data_ptr(), dw0.data_ptr(), bt, q, c, h, w,
                                             dparams.data_ptr(), None if dref is None else dref.data_ptr(), st), "ocpg_dynmask_bwd_fin_f32")
        return dfeat, dparams, dref, None


def dynamic_mask(feats, params, refpix, stride):
    return DynamicMaskFunction.apply(feats, params, refpix, stride)
