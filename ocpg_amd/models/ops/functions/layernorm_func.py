"""Autograd binding of csrc/layernorm.hip: LayerNorm with low-precision input / output in one pass each way."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ...._lib import check, lib

_CODE = {torch.float32: 1, torch.bfloat16: 0, torch.float16: 2}


class LayerNormLP(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        c = x.shape[-1]
        x2 = x.reshape(-1, c)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        rows = x2.shape[0]
        y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        check(lib().ocpg_layernorm_fwd(x2.data_ptr(), _CODE[x2.dtype], weight.data_ptr(), bias.data_ptr(), rows, c, float(eps), y.data_ptr(),
                                       _CODE[out_dtype], mean.data_ptr(), rstd.data_ptr(), torch.cuda.current_stream().cuda_stream), "ocpg_layernorm_fwd")
        ctx.save_for_backward(x2, weight, mean, rstd)
        ctx.x_shape = x.shape
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x2, weight, mean, rstd = ctx.saved_tensors
        rows, c = x2.shape
        g2 = gy.reshape(rows, c)
        if g2.dtype not in _CODE:
            g2 = g2.float()
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        dx = torch.empty_like(x2)
        nb = lib().ocpg_layernorm_blocks(rows)
        part = torch.empty((2, nb, c), dtype=torch.float32, device=x2.device)
        check(lib().ocpg_layernorm_bwd(g2.data_ptr(), _CODE[g2.dtype], x2.data_ptr(), _CODE[x2.dtype], weight.data_ptr(), mean.data_ptr(),
                                       rstd.data_ptr(), rows, c, dx.data_ptr(), _CODE[dx.dtype], part[0].data_ptr(), part[1].data_ptr(),
                                       torch.cuda.current_stream().cuda_stream), "ocpg_layernorm_bwd")
        dgb = part.sum(1)                   # one reduction for both
        return dx.view(ctx.x_shape), dgb[0], dgb[1], None, None


class LayerNorm(torch.nn.LayerNorm):
    """nn.LayerNorm (same parameters / state_dict); on the GPU under bf16 / fp16 autocast, for a 1-D normalized_shape <= 1024 with fp32
    affine parameters, one HIP pass each way that reads x in its own dtype and writes the output in the autocast dtype (what the
    following Linear consumes) instead of cast + fp32 layer_norm + cast."""

    def forward(self, x):
        if (x.is_cuda and torch.is_autocast_enabled("cuda") and len(self.normalized_shape) == 1 and self.normalized_shape[0] <= 1024
                and self.elementwise_affine and self.bias is not None and x.dtype in _CODE and self.weight.dtype == torch.float32
                and x.numel() > 0):
            # fp32_out (set by the owner): the result stays fp32 -- a norm that STARTS a residual stream (Video-Swin's patch-embedding
            # norm), where autocast's layer_norm returns fp32 too; the kernel still reads x in its own dtype: one pass, no casts
            out = torch.float32 if getattr(self, "fp32_out", False) else torch.get_autocast_dtype("cuda")
            return LayerNormLP.apply(x, self.weight, self.bias, self.eps, out)
        return super().forward(x)


class StaticGather(Function):
    """table[idx] for an index that never changes (a registered buffer): the backward is a segmented sum over rows sorted by
    destination (csrc/layernorm.hip: seg_sum) instead of index_put(accumulate)."""

    @staticmethod
    def plan(idx, rows):
        with torch.no_grad():
            order = torch.argsort(idx, stable=True).contiguous()
            seg = torch.zeros(rows + 1, dtype=torch.int64, device=idx.device)
            seg[1:] = torch.bincount(idx, minlength=rows).cumsum(0)
            return order, seg

    @staticmethod
    def forward(ctx, table, idx, plan):
        ctx.plan, ctx.shape = plan, table.shape
        return table[idx]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        order, seg = ctx.plan
        t, h = ctx.shape
        g2 = g.float().contiguous()
        out = torch.empty((t, h), dtype=torch.float32, device=g.device)
        check(lib().ocpg_gather_rows_bwd(g2.data_ptr(), order.data_ptr(), seg.data_ptr(), t, h, out.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "ocpg_gather_rows_bwd")
        return out.to(g.dtype), None, None


class RelPosBias(Function):
    """(bias, bias_t) [H, N, N] fp32 = table[index[:N, :N]] as [H, N, N] and its transpose over the two token axes, one launch
    (csrc/layernorm.hip relpos_bias); backward: a segmented sum straight from the gradient of `bias` -- or from its transpose when that is
    what lies contiguous (the attention backward hands `ds.sum(0).transpose(1, 2)`)."""

    @staticmethod
    def forward(ctx, table, index2d, plan):
        n = index2d.shape[0]
        t, h = table.shape
        assert table.dtype == torch.float32 and table.is_contiguous() and index2d.dtype == torch.int64 and index2d.stride(1) == 1
        bias = torch.empty((h, n, n), dtype=torch.float32, device=table.device)
        bias_t = torch.empty_like(bias)
        check(lib().ocpg_relpos_bias_fwd(table.data_ptr(), index2d.data_ptr(), n, index2d.stride(0), h, bias.data_ptr(), bias_t.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "ocpg_relpos_bias_fwd")
        ctx.plan, ctx.shape, ctx.n = plan, (t, h), n
        ctx.mark_non_differentiable(bias_t)
        return bias, bias_t

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _gt):
        order, seg = ctx.plan
        t, h = ctx.shape
        g = g.float()
        if g.transpose(1, 2).is_contiguous() and not g.is_contiguous():
            src, transposed = g.transpose(1, 2), 1
        else:
            src, transposed = g.contiguous(), 0
        out = torch.empty((t, h), dtype=torch.float32, device=g.device)
        check(lib().ocpg_relpos_bias_bwd(src.data_ptr(), order.data_ptr(), seg.data_ptr(), t, h, ctx.n, transposed, out.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "ocpg_relpos_bias_bwd")
        return out, None, None


class PermuteGather(Function):
    """out[b, j] = x[b, fwd_idx[j]] (a slot whose index is outside [0, S) reads zeros); `bwd_idx` [S] is the inverse map (for every source
    row the slot that holds it, or -1): the backward is the same kernel with the roles swapped."""

    @staticmethod
    def forward(ctx, x, fwd_idx, bwd_idx):
        x = x.contiguous()
        b, s, c = x.shape
        m = fwd_idx.shape[0]
        out = torch.empty((b, m, c), dtype=x.dtype, device=x.device)
        check(lib().ocpg_gather_rows_pad(x.data_ptr(), fwd_idx.data_ptr(), b, s, m, c * x.element_size(), out.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "ocpg_gather_rows_pad")
        ctx.idx, ctx.s = (fwd_idx, bwd_idx), s
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        fwd_idx, bwd_idx = ctx.idx
        g = g.contiguous()
        b, m, c = g.shape
        gx = torch.empty((b, ctx.s, c), dtype=g.dtype, device=g.device)
        check(lib().ocpg_gather_rows_pad(g.data_ptr(), bwd_idx.data_ptr(), b, m, ctx.s, c * g.element_size(), gx.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "ocpg_gather_rows_pad")
        return gx, None, None


def permute_gather_ok(x):
    return x.is_cuda and x.dim() == 3 and (x.shape[-1] * x.element_size()) % 16 == 0
