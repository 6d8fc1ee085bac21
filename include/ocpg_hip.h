/* libocpg_hip.so -- C ABI of the MI355X (gfx950) kernels behind OCPG's per-clip hot path.
 *
 * Drop-in boundary #2 of SURVEY.md section 8b.  The reference binds its native op through the pybind11
 * module `MultiScaleDeformableAttention` (models/ops/src/vision.cpp:13-16) whose two entry points are
 * declared in models/ops/src/ms_deform_attn.h:20-61 and implemented for CUDA in
 * models/ops/src/cuda/ms_deform_attn_cuda.cu:20-80 (forward) and :83-152 (backward).  The functions below
 * are what a ctypes / pybind stub for that path binds instead.
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes, no torch types; every tensor contiguous, row-major, same device;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); launches are asynchronous.  The KERNEL entry points
 *     neither synchronise nor allocate.  ONE exception, the GEMM family (ocpg_gemm, ocpg_gemm_bn_act: csrc/gemm.hip): it keeps a
 *     per-device PLAN CACHE (mutex-protected) and, at the FIRST use of a (device, stream) pair / of a plan, hipMallocs a hipBLASLt
 *     workspace and -- with OCPG_GEMM_TUNE=1, the default -- scratch outputs to time the heuristic's candidates, synchronising on
 *     its own events while it does.  Consequences for callers: run one un-captured call per shape and stream before capturing a
 *     HIP graph (bench.py's warm-up steps on the capture stream); OCPG_GEMM_TUNE=0 pins the heuristic's first choice
 *     (run-to-run and rank-to-rank reproducible kernel selection);
 *   - return 0 on success, a negative hipError_t on a launch/runtime error, -1000-k for the k-th
 *     argument being invalid (null pointer / non-positive size);
 *   - outputs are caller-allocated.  Forward outputs are fully overwritten.  In the backward,
 *     grad_value is ACCUMULATED into (scatter-add): the caller zeroes it first, exactly as
 *     ms_deform_attn_cuda.cu:121 does with at::zeros_like; grad_loc / grad_attn are fully overwritten;
 *   - re-entrant and thread-safe; no global state except the GEMM plan cache above (and, in diagnostic builds only, the
 *     EXP_STAMPS counters).
 *
 * Tensor contract of MSDeformAttn (ms_deform_attn_cuda.cu:28-48):
 *   value        [N, S, M, D]          S = sum_l H_l*W_l
 *   shapes       [L, 2]  int64 (H, W)   device memory  (+ optional host copy, see below)
 *   level_start  [L]     int64          device memory
 *   loc          [N, Lq, M, L, P, 2]    (x, y) normalised to [0,1]
 *   attn         [N, Lq, M, L, P]
 *   out / grad_out [N, Lq, M*D]
 * Unlike the reference there is no im2col_step chunking and therefore no `N % im2col_step == 0`
 * restriction (ms_deform_attn_cuda.cu:50-52); any N works.
 */
#ifndef OCPG_HIP_H
#define OCPG_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* replaces ms_deform_attn_forward (ms_deform_attn.h:20-39 -> ms_deform_attn_cuda.cu:20-80), float32.
 * shapes_host: optional HOST copy of `shapes` (may be NULL).  When given, Lq == S (self-attention over the value's
 * own pixels, the encoder case: query q IS pixel q) and level_start is the exclusive prefix sum of H_l*W_l, the
 * column-tile kernels (LDS-staged sampling windows) are selected; results are identical either way up to fp32
 * summation order.  The host copy is what the reference's module already owns: ms_deform_attn.py:94 asserts on the
 * same tensor, which reads it back to the host on every call. */
int ocpg_msda_fwd_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                      const float* loc, const float* attn,
                      int N, int S, int M, int D, int L, int Lq, int P,
                      float* out, const int64_t* shapes_host, void* stream);
/* float64 variant (AT_DISPATCH_FLOATING_TYPES, ms_deform_attn_cuda.cu:64); used by the test.py protocol */
int ocpg_msda_fwd_f64(const double* value, const int64_t* shapes, const int64_t* level_start,
                      const double* loc, const double* attn,
                      int N, int S, int M, int D, int L, int Lq, int P,
                      double* out, void* stream);

/* replaces ms_deform_attn_backward (ms_deform_attn.h:41-61 -> ms_deform_attn_cuda.cu:83-152), float32.
 * shapes_host: as for the forward (may be NULL): selects the column-tile backward (gather kernel for grad_loc /
 * grad_attn + bin-and-sum scatter kernel for grad_value) when Lq == S. */
int ocpg_msda_bwd_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                      const float* loc, const float* attn, const float* grad_out,
                      int N, int S, int M, int D, int L, int Lq, int P,
                      float* grad_value, float* grad_loc, float* grad_attn,
                      const int64_t* shapes_host, void* stream);
/* The two independent halves of ocpg_msda_bwd_f32 on their own (same contract; return -2000 when the shape is not
 * served by the dedicated kernel -- call ocpg_msda_bwd_f32 then).  A caller that needs only some of the gradients
 * (ctx.needs_input_grad of ms_deform_attn_func.py:30-39), or that wants the two kernels timed separately, binds these:
 *   _value:   grad_value (+=) from loc / attn / grad_out alone (column-tile bin-and-sum scatter; needs shapes_host, Lq == S)
 *   _locattn: grad_loc and grad_attn (overwritten) -- gather-only row kernel, any Lq */
int ocpg_msda_bwd_value_f32(const float* loc, const float* attn, const float* grad_out,
                            int N, int S, int M, int D, int L, int Lq, int P,
                            float* grad_value, const int64_t* shapes_host, void* stream);
int ocpg_msda_bwd_locattn_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                              const float* loc, const float* attn, const float* grad_out,
                              int N, int S, int M, int D, int L, int Lq, int P,
                              float* grad_loc, float* grad_attn, void* stream);
/* ocpg_msda_bwd_value_f32 with PER-CALL PATH SELECTION (round 4; replaces the same reference kernel, cuh:301-403 via cu:83-152, for
 * grad_value).  Two kernel families serve the self-attention shape: the query-owned column scatter (fastest while the offsets stay
 * within ~5 pixels of the query, as at initialisation) and the output-tiled kernels (no halo atomics: ahead once training has spread
 * the offsets).  sel_state: 8 ints of device memory (8-byte aligned: slots 0 / 1 are one 64-bit word the reporting workgroups add to)
 * owned by the CALL SITE (one per MSDeformAttn module), zero-filled once and passed
 * to every call; the kernels keep in it the share of samples that missed the active path's locality assumption and the path the next
 * call takes (0 = column, 1 = tiled; slots 6 / 7: far and total samples of the last call, for diagnostics).  Both families are launched
 * on every call and the inactive one's workgroups return at once, so the choice needs no host round trip and survives HIP-graph replay.
 * NULL state, a shape one of the families does not serve, or a forced path (OCPG_MSDA_TILE / OCPG_MSDA_COL): as ocpg_msda_bwd_value_f32. */
/* FUSED FRONT END of the MSDeformAttn module (round 4).  Replaces, for self-attention calls with 2-d reference points, the elementwise
 * passes of models/ops/modules/ms_deform_attn.py:96-110 around the op: softmax over the L*P attention logits, `reference + offset`, and in
 * the backward the softmax gradient and the concatenation of the offset / logit gradients.
 *   qproj [N*Lq, 3*M*L*P]  the merged query projection: columns [0, 2*M*L*P) = sampling offsets (M, L, P, 2) ALREADY divided by (W_l, H_l)
 *                          (the caller folds that division into the projection's weight rows), columns [2*M*L*P, 3*M*L*P) = logits (M, L*P)
 *   ref   [N*Lq, L, 2]     reference points (x, y), normalised
 *   _fwd:       out [N, Lq, M*D] as ocpg_msda_fwd_f32, plus the sampling locations loc_out [N, Lq, M, L, P, 2] and the attention weights
 *               attn_out [N, Lq, M, L, P] it derived (the module returns them; the backward entry points below read them)
 *   _bwd_qproj: grad_qproj [N*Lq, 3*M*L*P] (overwritten) = [d offsets | d logits] from value / loc / attn / grad_out -- the gather half of
 *               the backward (ocpg_msda_bwd_locattn_f32) with the softmax backward in its epilogue; grad_value comes from
 *               ocpg_msda_bwd_value_f32 / _sel_f32 as before.
 * D = 32 and L*P = 16 only; -2000 otherwise (nothing launched: keep the unfused path). */
int ocpg_msda_fused_fwd_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                            const float* qproj, const float* ref,
                            int N, int S, int M, int D, int L, int Lq, int P,
                            float* out, float* loc_out, float* attn_out, void* stream);
int ocpg_msda_fused_bwd_qproj_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                  const float* loc, const float* attn, const float* grad_out,
                                  int N, int S, int M, int D, int L, int Lq, int P,
                                  float* grad_qproj, void* stream);
int ocpg_msda_bwd_value_sel_f32(const float* loc, const float* attn, const float* grad_out,
                                int N, int S, int M, int D, int L, int Lq, int P,
                                float* grad_value, const int64_t* shapes_host, int* sel_state, void* stream);
int ocpg_msda_bwd_f64(const double* value, const int64_t* shapes, const int64_t* level_start,
                      const double* loc, const double* attn, const double* grad_out,
                      int N, int S, int M, int D, int L, int Lq, int P,
                      double* grad_value, double* grad_loc, double* grad_attn, void* stream);

/* Fused frozen-BatchNorm affine (+ residual) (+ ReLU) over a feature map -- replaces the per-BN elementwise chain of
 * FrozenBatchNorm2d.forward (models/backbone.py:46-56: x*scale + bias with scale = w*rsqrt(var+1e-5)) followed by
 * torchvision Bottleneck's "out += identity" / ReLU.  scale/shift are the per-channel fp32 vectors [C].
 *   y[o,c,i] = act(x[o,c,i] * scale[c] + shift[c] (+ skip[o,c,i])),  element (o,c,i) at ((o*C + c)*inner + i)
 *   NHWC / channels_last: n_outer = N*H*W, inner = 1;   NCHW: n_outer = N, inner = H*W.
 * dtype: 0 = float32, 1 = bfloat16 (storage; arithmetic is fp32).  skip may be NULL.  y may alias x.
 * Backward needs only the saved OUTPUT y: g = relu ? (y > 0 ? gy : 0) : gy; gx = g*scale[c]; gskip = g.
 * gx or gskip may be NULL (not wanted); gskip may alias gy. */
int ocpg_bn_act_fwd(const void* x, const float* scale, const float* shift, const void* skip, void* y,
                    long long n_outer, int C, long long inner, int relu, int dtype, void* stream);
int ocpg_bn_act_bwd(const void* gy, const void* y, const float* scale, void* gx, void* gskip,
                    long long n_outer, int C, long long inner, int relu, int dtype, void* stream);

/* Fused 3-D (shifted-)window attention of Video-Swin -- replaces WindowAttention3D.forward's score / bias / mask /
 * softmax / PV chain (models/video_swin_transformer.py:138-169) and the shift-mask tensor of compute_mask (:316-329).
 *   qkv    [BW, N, 3, H, head_dim]   output of the qkv Linear (BW = batch * windows), head_dim must be 32
 *   bias   [H, N, N]  relative-position bias (query i, key j);  biasT [H, N, N] the same transposed (key j, query i)
 *   region [NW, N] int32 cyclic-shift region id of every window slot, or NULL (no shift); window = bw % NW;
 *          a (query, key) pair in different regions gets -100 added, exactly like the reference's mask
 *   out    [BW, N, H*head_dim];  lse [BW, H, N] fp32 log-sum-exp of every row (saved for the backward)
 * dtype: 0 float32, 1 bfloat16, 2 float16 (storage of qkv/out/dout/dqkv; arithmetic and bias/lse are fp32).
 * Backward: dqkv [BW, N, 3, H, head_dim] fully overwritten; Dbuf [BW, H, N] fp32 scratch;
 * dbiasT [H, N, N] fp32 is ACCUMULATED into (caller zeroes it; may be NULL when the bias needs no gradient). */
int ocpg_win_attn_fwd(const void* qkv, const float* biasT, const int* region, float scale, int BW, int NW, int N, int H,
                      int head_dim, void* out, float* lse, int dtype, void* stream);
int ocpg_win_attn_bwd(const void* qkv, const float* bias, const float* biasT, const int* region, float scale, int BW, int NW,
                      int N, int H, int head_dim, const void* out, const void* dout, const float* lse, void* dqkv, float* Dbuf,
                      float* dbiasT, int dtype, void* stream);

/* Matrix-core backward for bf16 / fp16 storage (csrc/win_attn_mfma.hip; the forward switches by itself): as ocpg_win_attn_bwd, but
 * the bias gradient leaves as dS [BW, H, N, N] (storage dtype, (key, query) order, fully written; NULL: not needed) for the caller to
 * sum over BW -- 2 x 232 MB of streaming traffic at Swin-T stage 1 instead of 464 MB of float atomics.  -2000: shape not served. */
int ocpg_win_attn_bwd_mfma(const void* qkv, const float* bias, const float* biasT, const int* region, float scale, int BW, int NW, int N,
                           int H, int head_dim, const void* out, const void* dout, const float* lse, void* dqkv, float* Dbuf, void* dS,
                           int dtype, void* stream);

/* Dynamic (per-query) mask head, forward -- replaces OCPG.dynamic_mask_with_coords + mask_heads_forward
 * (models/ocpg.py:475-549) for the reference's fixed head shape (2 layers, 16 channels, relative coordinates on).
 * Q counts the parameter sets per frame: the reference calls the head once per decoder layer on the SAME mask features
 * (ocpg.py:339-349); passing Q = layers x queries does all of them in one launch.
 *   feats  [BT, C, H, W] fp32 mask features;  params [BT*Q, (C+2)*16 + 16*16 + 16 + 16] controller outputs in the
 *   reference's order (parse_dynamic_params, ocpg.py:552-569: W0 [16,(C+2)] incl. the x,y coordinate columns, W1 [16,16],
 *   b0, b1);  refpix [BT*Q, 2] reference point in INPUT pixels (ref_xy * (img_w, img_h));  stride = mask_feat_stride (8)
 *   out    [BT*Q, 16, H, W];  pre1 [BT*Q, 16, H, W] = layer-1 pre-activation (kept for the backward; may be NULL). */
int ocpg_dynmask_fwd_f32(const float* feats, const float* params, const float* refpix, int BT, int Q, int C, int H, int W,
                         int stride, float* out, float* pre1, void* stream);

/* Backward of the dynamic mask head (autograd of models/ocpg.py:505-549 in the reference), all decoder layers per launch.
 * _pre: dout, pre1 [BT*Q,16,H,W] -> dpre [BT*Q,16,H,W] = (W1^T dout) * (pre1 > 0); part [BT*Q, ceil(HW/256), 320] per-strip
 *       partial sums (dW1 16x16 | db0 | db1 | sum dpre*x | sum dpre*y); w0d [BT*Q,16,C] = the feature columns of W0, dense.
 *       The two large contractions dW0 = dpre . feats^T and dfeat = W0^T . dpre are plain GEMMs on these buffers (ocpg_gemm).
 * _fin: part + dw0 [BT*Q,16,C] (the dW0 GEMM's result) -> dparams [BT*Q, (C+2)*16+16*16+32] fully written in the reference's
 *       parameter order (parse_dynamic_params, ocpg.py:552-569) and dref [BT*Q,2] (may be NULL). */
int ocpg_dynmask_bwd_pre_f32(const float* dout, const float* pre1, const float* params, int BT, int Q, int C, int H, int W,
                             int stride, float* dpre, float* part, float* w0d, void* stream);
int ocpg_dynmask_bwd_fin_f32(const float* part, const float* params, const float* refpix, const float* dw0, int BT, int Q, int C,
                             int H, int W, float* dparams, float* dref, void* stream);

/* 3x3 convolution of channels-last maps as one dense GEMM -- replaces the conv kernels behind nn.Conv2d(k=3) in the ResNet
 * body (torchvision Bottleneck.conv2 via models/backbone.py:86-117) and the neck (models/ocpg.py:118-126); the GEMM itself
 * is hipBLASLt's.  Geometry: kernel 3x3, padding == dil, stride in {1,2}; Ho = (H-1)/stride + 1, Wo likewise.
 *   im2col: x [N,H,W,C] -> cols [N*Ho*Wo, 9*C], column order (ky, kx, c) = the physical order of a channels-last weight
 *   col2im: dcols [N*Ho*Wo, 9*C] -> dx [N,H,W,C] (the adjoint: every dx element fully written, fp32 accumulation)
 * dtype: 0 fp32, 1 bf16, 2 fp16; C * sizeof(elem) must be a multiple of 16. */
int ocpg_im2col3x3_nhwc(const void* x, int N, int H, int W, int C, int stride, int dil, void* cols, int dtype, void* stream);
int ocpg_col2im3x3_nhwc(const void* dcols, int N, int H, int W, int C, int stride, int dil, void* dx, int dtype, void* stream);

/* 3x3 convolution (padding 1, stride 1 | 2) of channels-last bf16 maps as an implicit GEMM on the matrix cores
 * (csrc/conv3x3_mfma.hip: MFMA 32x32x16 bf16, fp32 accumulation, no im2col buffer) -- replaces the MIOpen kernels behind
 * torchvision Bottleneck.conv2 (models/backbone.py:86-117) and, in its epilogue, FrozenBatchNorm2d's affine + ReLU
 * (models/backbone.py:46-56).
 *   fwd:   x [N,H,W,Cin], w [Cout,3,3,Cin] (a channels-last weight as it lies in memory) -> y [N,Ho,Wo,Cout] =
 *          act(conv(x, w) * scale[co] + bias[co]); scale / bias fp32 or NULL; Cin % 32 == 0.
 *   dgrad: dy [N,Ho,Wo,Cout], wT [Cin,3,3,Cout] (channel axes swapped, taps NOT flipped) -> dx [N,H,W,Cin] fully written;
 *          Cout % 32 == 0.
 * -2000: geometry not served (the caller keeps its other path).  The weight gradient is a GEMM over the im2col matrix
 * (ocpg_im2col3x3_nhwc + ocpg_gemm). */
int ocpg_conv3x3_mfma_fwd(const void* x, const void* w, const float* scale, const float* bias, int relu, int N, int H, int W,
                          int Cin, int Cout, int stride, void* y, void* stream);
/* the same, and the patch (im2col) matrix of x written on the way: cols [N*Ho*Wo, 9*Cin] bf16 in (ky, kx, ci) column order (what
 * ocpg_im2col3x3_nhwc produces), or NULL -- the weight gradient of the convolution contracts the output gradient with it. */
int ocpg_conv3x3_mfma_fwd_cols(const void* x, const void* w, const float* scale, const float* bias, int relu, int N, int H, int W, int Cin,
                               int Cout, int stride, void* y, void* cols, void* stream);
int ocpg_conv3x3_mfma_dgrad(const void* dy, const void* wT, int N, int H, int W, int Cin, int Cout, int stride, void* dx,
                            void* stream);
/* ocpg_conv3x3_mfma_dgrad that ALSO applies the frozen-BN + ReLU backward of the layer in front (round 4; FrozenBatchNorm2d backbone.py:46-56
 * + ReLU of torchvision's Bottleneck between conv1 and conv2): dx[n,h,w,ci] = conv_transpose(dy)[n,h,w,ci] * scale[ci] where
 * mask_y[n,h,w,ci] > 0, else 0.  mask_y = that layer's post-ReLU output (the convolution's own input), bf16, laid out like dx; scale fp32
 * [Cin] or NULL (= 1).  Saves one ocpg_bn_act_bwd pass per bottleneck. */
int ocpg_conv3x3_mfma_dgrad_masked(const void* dy, const void* wT, const void* mask_y, const float* scale, int N, int H, int W, int Cin,
                                   int Cout, int stride, void* dx, void* stream);
/* The same from the convolution's OWN weight w [Cout, 3, 3, Cin] (round 4: no transposed copy per step; the kernel reads its weight
 * fragments with transposing LDS loads).  mask_y / scale as above (NULL: plain input gradient).  Cin % 8 != 0: -2000. */
int ocpg_conv3x3_mfma_dgrad_w(const void* dy, const void* w, const void* mask_y, const float* scale, int N, int H, int W, int Cin, int Cout,
                              int stride, void* dx, void* stream);
/* Weight gradient of the same convolution straight from the two channels-last maps (csrc/conv3x3_wgrad.hip; round 4: replaces
 * ocpg_im2col3x3_nhwc + a row-split ocpg_gemm): gz [N,Ho,Wo,Cout] bf16 (the gradient after the BN / ReLU backward), x [N,H,W,Cin] bf16 ->
 * part [S][Cout][3][3][Cin] bf16, S = ocpg_conv3x3_mfma_wgrad_splits(...) partial sums over ranges of output rows (the caller adds them:
 * gw = sum_z part[z]).  Cin % 8 or Cout % 8 != 0: -2000. */
/* The same forward / input gradient with the K chain split over the grid for launches of few tiles and long K (round 4: ResNet layer3 /
 * layer4 at 1-2 clips per step): `splits` = ocpg_conv3x3_mfma_body_splits(rows, GEMM columns, K channels) (1 = use the un-split entry);
 * part: fp32 scratch [splits][rows][columns]; the summing pass applies the epilogue (BN affine + ReLU; scale + mask of the layer in front). */
int ocpg_conv3x3_mfma_body_splits(long long M, int ncols, int kchannels);
int ocpg_conv3x3_mfma_fwd_bn_splitk(const void* x, const void* w, const float* scale, const float* shift, int relu, int N, int H, int W, int Cin,
                                    int Cout, int stride, int splits, float* part, void* y, void* stream);
int ocpg_conv3x3_mfma_dgrad_w_splitk(const void* dy, const void* w, const void* mask_y, const float* scale, int N, int H, int W, int Cin,
                                     int Cout, int stride, int splits, float* part, void* dx, void* stream);
int ocpg_conv3x3_mfma_wgrad_splits(int N, int H, int W, int Cin, int Cout, int stride);
int ocpg_conv3x3_mfma_wgrad(const void* gz, const void* x, int N, int H, int W, int Cin, int Cout, int stride, void* part, void* stream);

/* Split-K form of ocpg_conv3x3_mfma_fwd for convolutions with FEW output pixels and a LONG reduction (round 4): the neck's extra level
 * input_proj[3] = nn.Conv2d(2048, 256, 3, stride=2, padding=1) (models/ocpg.py:119-123; 600 output pixels at config #2, K = 18 432) --
 * MIOpen's split-K kernel until round 3.  blockIdx.z takes a contiguous range of 64-channel chunks; the tiles leave as fp32 partial
 * sums part[splits][N*Ho*Wo][Cout] (caller-provided scratch, fully written) and a second kernel adds them and the bias into
 * y [N,Ho,Wo,Cout] (out_dt 0 fp32 / 1 bf16).  ocpg_conv3x3_mfma_splits: the number of ranges to use for a shape (1 = the plain kernel
 * fills the chip: call ocpg_conv3x3_mfma_fwd).  cols (may be NULL): the patch matrix, as ocpg_conv3x3_mfma_fwd_cols writes it.  The input
 * gradient is ocpg_conv3x3_mfma_dgrad (its GEMM has N*H*W rows: no split needed). */
int ocpg_conv3x3_mfma_splits(int N, int H, int W, int Cin, int Cout, int stride);
int ocpg_conv3x3_mfma_fwd_splitk(const void* x, const void* w, const float* bias, int N, int H, int W, int Cin, int Cout, int stride,
                                 int splits, float* part, void* y, int out_dt, void* cols, void* stream);

/* Dense GEMM with a per-shape plan cache over hipBLASLt -- replaces the at::mm / at::addmm / at::bmm calls behind
 * nn.Linear and the 1x1 nn.Conv2d layers on the path (models/deformable_transformer.py:236-257,313-327 FFNs,
 * models/ops/modules/ms_deform_attn.py:63-66 projections, torchvision Bottleneck.conv1/conv3 via models/backbone.py) and
 * their gradients; same hipBLASLt kernels, ~1/3 of the host cost per call (the step is launch-bound).
 * Row-major:  C[M,N] = alpha * op(A) op(B) + beta * C (+ bias[N]);  op(A) = A [M,K] (lda) or A^T with A stored [K,M];
 * op(B) = B [K,N] (ldb) or B^T with B stored [N,K].  batch > 1: strided batches (element strides).  dtype / out_dtype:
 * 0 fp32, 1 bf16, 2 fp16 (fp32 accumulation).  State (handle, plans) is per device, the 64-MB workspace per (device, stream). */
int ocpg_gemm(const void* A, const void* B, void* C, const void* bias, int dtype, int out_dtype, int transA, int transB,
              long long M, long long N, long long K, long long lda, long long ldb, long long ldc, long long batch,
              long long strideA, long long strideB, long long strideC, float alpha, float beta, void* stream);
long long ocpg_gemm_plans(void);      /* number of cached plans (diagnostics) */
/* Plan selection: the first call of a plan that can be repeated without changing its result (beta == 0; the BN form with skip != D)
 * times hipBLASLt's ranked candidates on the caller's operands (eager calls only, never inside a stream capture) and keeps the
 * fastest one whose result agrees with the heuristic's single choice (what at::mm runs) to the rounding of the output type (8e-3 /
 * 1e-3 of max |C| for bf16 / fp16 outputs).  bf16 / fp16 plans only: fp32 plans always keep the heuristic's choice (its ranked fp32
 * list holds kernels that are 2e-3 off).  OCPG_GEMM_TUNE=0 keeps the heuristic's choice everywhere;
 * OCPG_GEMM_TUNE_FP32=1 (experiment, off by default) also times fp32 plans, validated to 1e-5 of max |C|.  ocpg_gemm_tuned returns the number of plans timed on the current device and, in *changed (may be NULL),
 * how many of them left the first choice; ocpg_gemm_tune_rejected the number of candidates dropped for a differing result. */
long long ocpg_gemm_tuned(long long* changed);
long long ocpg_gemm_tune_rejected(void);
/* Rank-consistent plan choices for data-parallel training (the role of the reference's main.py:62 DistributedDataParallel ranks running
 * one cuBLAS build: every rank must run the same GEMM kernels, or the slowest rank's choice sets the step).  The candidate timing above is
 * a per-process measurement, so under N > 1 ranks ONE rank times (ocpg_gemm_set_tuning(0) on the others before their first GEMM),
 * exports its choices as [key hash, candidate index] pairs, the caller broadcasts them, and every other rank imports them: they apply to
 * the plans that already exist and to those built later (the heuristic's ranked candidate list is identical on identical hardware).
 * set_tuning: 1 / 0 = candidate timing on / off in this process, -1 = OCPG_GEMM_TUNE decides.  export: returns the number of pairs
 * (buf may be NULL to ask for the count); import: 0 on success. */
void ocpg_gemm_set_tuning(int on);
long long ocpg_gemm_export_picks(long long* buf, long long cap_pairs);
int ocpg_gemm_import_picks(const long long* buf, long long n_pairs);
/* D[M,N] = act(scale[n] * (A W^T)[m,n] + shift[n] (+ skip[m,n])): the 1x1 conv + FrozenBatchNorm2d affine (+ identity) (+ ReLU) of a
 * Bottleneck (models/backbone.py:46-56 + torchvision's block) inside the GEMM epilogue (per-channel alpha vector, fp32 bias, ReLU,
 * beta = 1 on the skip operand).  A [M,K], W [N,K], skip / D [M,N] dense row-major, dtype 0/1/2; scale, shift fp32 [N].
 * Returns -1105 when hipBLASLt offers no kernel for the combination (fall back to ocpg_gemm + ocpg_bn_act_fwd). */
int ocpg_gemm_bn_act(const void* A, const void* W, void* D, const float* scale, const float* shift, const void* skip, int relu, int dtype,
                     long long M, long long N, long long K, void* stream);

/* Mask-criterion losses, all decoder layers per call (fp32; replace the elementwise/reduction chains of
 * models/segmentation.py:203-211,253-315 as called from models/criterion.py:141-178).
 *
 * Level-set loss: x [Lr,N,h,w] logits, feats [N,CF,h,w] (first C channels used, C <= 16), box [N,h,w] in {0,1}.
 *   fwd: sums [Lr,N,ceil(h*w/1024),7+2C] scratch (per-workgroup partials), coef [Lr,N,8+2C] kept for the backward; loss [Lr].
 *   bwd: gloss [Lr] upstream gradient -> gx [Lr,N,h,w], gfeat [N,CF,h,w] (may be NULL), both fully written. */
int ocpg_levelset_fwd_f32(const float* x, const float* feats, const float* box, int Lr, int N, int C, int CF, int h, int w, float* sums,
                          float* coef, float* loss, void* stream);
int ocpg_levelset_bwd_f32(const float* x, const float* feats, const float* box, const float* coef, const float* gloss, int Lr, int N,
                          int C, int CF, int h, int w, float* gx, float* gfeat, void* stream);
/* Box-projection loss with mean term: x [Lr,B,T,H,W] logits; targets tcmax/tcmean [B,T,W] (region.amax / weak.mean over H),
 * trmax/trmean [B,T,H] (over W).  fwd: colstat [Lr*B*T,3,W], rowstat [Lr*B*T,3,H], IU [Lr,B,4,2] scratch kept for the
 * backward; loss [Lr].  bwd: Gc [Lr*B*T,2,W], Gr [Lr*B*T,2,H] scratch; gx [Lr,B,T,H,W] fully written. */
int ocpg_proj_fwd_f32(const float* x, const float* tcmax, const float* trmax, const float* tcmean, const float* trmean, int Lr, int B,
                      int T, int H, int W, float* colstat, float* rowstat, float* IU, float* loss, void* stream);
int ocpg_proj_bwd_f32(const float* x, const float* tcmax, const float* trmax, const float* tcmean, const float* trmean,
                      const float* colstat, const float* rowstat, const float* IU, const float* gloss, int Lr, int B, int T, int H, int W,
                      float* Gc, float* Gr, float* gx, void* stream);

/* Matching cost [Lr,B,Q] of every query against the clip's single target -- replaces the cost construction of
 * HungarianMatcher.forward (models/matcher.py:74-160) for all decoder layers: wc * focal-class (mean over valid frames) +
 * wb * L1 + wg * (-GIoU) (means over frames) + wm * sigmoid-focal + wd * (-dice) (over the clip's pixels).
 * logits [Lr,B,T,Q,K], boxes [Lr,B,T,Q,4] cxcywh, contiguous; masks: logits with element strides (sl,sb,st,sq) for
 * (layer, clip, frame, query) and a contiguous [h,w] tail; gt [B,T,h,w]; tboxes [B,T,4]; valid [B,T] (float);
 * labels [B,T] int64 or NULL (class 0).  sums [Lr,B,Q,4] scratch.  bad: optional int32 counter, +1 per (layer,clip,query)
 * with a malformed box. */
int ocpg_matcher_cost_f32(const float* logits, const float* boxes, const float* masks, long long sl, long long sb, long long st,
                          long long sq, const float* gt, const float* tboxes, const float* valid, const long long* labels, int Lr,
                          int B, int T, int Q, int K, int h, int w, float wc, float wb, float wg, float wm, float wd, float* sums,
                          float* cost, int* bad, void* stream);

/* Transformer-layer glue (models/deformable_transformer.py:236-257,313-336), one HBM pass each way:
 *   y = LayerNorm(res + dropout(x)):  x [R,C] (x_dtype 0 fp32 / 1 bf16), res / y fp32, C % 4 == 0, C <= 2048; mean, rstd [R]
 *     are kept for the backward, which recomputes the dropout mask from (seed, offset) (Philox-4x32-10, counter = element/4);
 *     rng_base: NULL, or a DEVICE pointer to one 64-bit word that the kernel adds to `offset` when it runs -- a call captured
 *     into a HIP graph bakes `offset` into the node, so the caller keeps the per-step base in device memory and advances it
 *     between replays (every replay then draws fresh masks, as models/deformable_transformer.py:236-257 does every step);
 *     bwd: gx (x's dtype, may be NULL), gres (may be NULL) fully written; dgb_part [slots, 2, C] partial (dgamma, dbeta) sums,
 *     fully written, slots = ocpg_dropout_add_ln_bwd_slots(R); the caller sums over the slots.
 *   h = dropout(relu(a + bias)):  a, bias, h share dtype (0 fp32 / 1 bf16), h may alias a; bwd from h only:
 *     ga = gh / (1-p) where h > 0; dbias_part [slots, C] fp32 partial column sums, fully written, slots =
 *     ocpg_bias_relu_dropout_bwd_slots(R, C, dtype); the caller sums over the slots. */
int ocpg_dropout_add_ln_fwd(const void* x, const float* res, const float* gamma, const float* beta, long long R, int C, float eps, float p,
                            unsigned long long seed, unsigned long long offset, const unsigned long long* rng_base, int x_dtype, float* y,
                            float* mean, float* rstd, void* stream);
int ocpg_dropout_add_ln_bwd(const float* gy, const void* x, const float* res, const float* gamma, const float* mean, const float* rstd,
                            long long R, int C, float p, unsigned long long seed, unsigned long long offset, const unsigned long long* rng_base,
                            int x_dtype, void* gx, float* gres, float* dgb_part, void* stream);
long long ocpg_dropout_add_ln_bwd_slots(long long R);
int ocpg_bias_relu_dropout_fwd(const void* a, const void* bias, long long R, int C, float p, unsigned long long seed,
                               unsigned long long offset, const unsigned long long* rng_base, int dtype, void* h, void* stream);
int ocpg_bias_relu_dropout_bwd(const void* gh, const void* h, long long R, int C, float p, int dtype, void* ga, float* dbias_part,
                               void* stream);
long long ocpg_bias_relu_dropout_bwd_slots(long long R, int C, int dtype);

/* Dtype cast of MANY dense tensors in one launch -- replaces the per-parameter autocast casts and the per-parameter
 * gradient casts of an AMP step (torch.cuda.amp as used by engine.py:49-62).  Device tables (int64): srcs / dsts = element
 * pointers, numels, chunk_prefix [n+1] = prefix sum of ceil(numel / 2048); total_chunks = chunk_prefix[n].
 * dtype codes: 0 fp32, 1 bf16, 2 fp16; supported pairs: fp32 <-> bf16, fp32 <-> fp16. */
int ocpg_multi_cast(const long long* srcs, const long long* dsts, const long long* numels, const long long* chunk_prefix, int n,
                    long long total_chunks, int src_dtype, int dst_dtype, void* stream);
/* The same with a reduction on the way: tensor t = sum of splits[t] slices of numels[t] elements, strides[t] elements apart (the
 * row-split weight-gradient partial products of the fused conv nodes: replaces one at::sum launch per layer, ~90 per step in the
 * ResNet body of models/backbone.py, plus the cast).  src_dtype 0 fp32 / 1 bf16 / 2 fp16 -> fp32; fp32 accumulation. */
int ocpg_multi_cast_sum(const long long* srcs, const long long* dsts, const long long* numels, const long long* chunk_prefix,
                        const long long* splits, const long long* strides, int n, long long total_chunks, int src_dtype, int dst_dtype,
                        void* stream);
/* Partial column sums of x [R, C] (dtype 0 fp32 / 1 bf16 / 2 fp16) -> part [ocpg_colsum_blocks(R), C] fp32, fully written: the
 * bias gradient `grad_output.sum(0)` of nn.Linear / a 1x1 nn.Conv2d over many rows (models/deformable_transformer.py:236-257 FFNs,
 * models/modules.py:12-16 LFM convolutions) as ONE launch; ocpg_multi_cast_sum finishes the sum over the blocks. */
long long ocpg_colsum_blocks(long long R);
int ocpg_colsum_partials(const void* x, long long R, int C, int dtype, float* part, void* stream);

/* Gradient clipping + AdamW over MANY fp32 tensors (device tables of pointers / element counts / 2048-element chunk prefixes, as
 * ocpg_multi_cast) -- replaces torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW.step of engine.py:100-106 / main.py:76-99:
 *   ocpg_grad_norm_clip: partials [total_chunks] (scratch) -> norm_and_coef[0] = total 2-norm of all gradients,
 *                        norm_and_coef[1] = min(1, max_norm / (norm + 1e-6)) (1 when max_norm <= 0); two launches;
 *   ocpg_adamw_step:     p, exp_avg, exp_avg_sq updated in one launch; the gradient is multiplied by norm_and_coef[1] on the fly (NULL:
 *                        no clipping); lr / weight_decay: per-tensor fp32 device arrays (the optimizer's groups); step >= 1 is the
 *                        step count AFTER this update (bias corrections).  betas / eps are doubles: 1 - beta is formed in double, as
 *                        torch's Python scalars are (1.f - 0.999f is 4.7e-5 off 0.001).  Arithmetic = torch's single-tensor AdamW. */
int ocpg_grad_norm_clip(const long long* grads, const long long* numels, const long long* chunk_prefix, int n, long long total_chunks,
                        float max_norm, float* partials, float* norm_and_coef, void* stream);
int ocpg_adamw_step(const long long* params, const long long* grads, const long long* exp_avg, const long long* exp_avg_sq,
                    const long long* numels, const long long* chunk_prefix, const float* lr, const float* weight_decay, int n,
                    long long total_chunks, const float* norm_and_coef, double beta1, double beta2, double eps, long long step, void* stream);
/* The same under torch.amp.GradScaler (engine.py:98-106, --amp with fp16: scaler.unscale_ + clip_grad_norm_ + scaler.step, whose
 * found_inf.item() stalls the host every step): the gradients still carry `grad_scale[0]` (device scalar; NULL: already unscaled).
 * amp_state (fp32[6], device): [0] norm of the unscaled gradients, [1] coefficient applied to the scaled gradients = clip / scale,
 * [2] 1 when the step is skipped (non-finite norm, or found_inf[0] != 0 when found_inf is given), [3] number of steps TAKEN (in/out:
 * skipped steps do not advance the bias corrections, as GradScaler skips optimizer.step() as a whole), [4], [5] bias corrections of
 * that count.  ocpg_adamw_step_amp returns without touching p / exp_avg / exp_avg_sq when [2] is set. */
int ocpg_grad_norm_clip_amp(const long long* grads, const long long* numels, const long long* chunk_prefix, int n, long long total_chunks,
                            float max_norm, float* partials, const float* grad_scale, const float* found_inf, double beta1, double beta2,
                            float* amp_state, void* stream);
int ocpg_adamw_step_amp(const long long* params, const long long* grads, const long long* exp_avg, const long long* exp_avg_sq,
                        const long long* numels, const long long* chunk_prefix, const float* lr, const float* weight_decay, int n,
                        long long total_chunks, const float* amp_state, double beta1, double beta2, double eps, void* stream);

/* Classification (sigmoid focal, alpha < 0 disables the alpha weighting) + L1 + GIoU losses of the matched queries, all layers
 * per launch -- replaces SetCriterion.loss_labels / loss_boxes (models/criterion.py:46-107; sigmoid_focal_loss
 * models/segmentation.py:134-160; util/box_ops.py:45-85).  logits [Lr,B,T,Q,K], boxes [Lr,B,T,Q,4] (cxcywh), src [Lr,B] int64
 * matched query, valid [B,T] float, labels [B,T] int64 or NULL, tboxes [B,T,4], num_boxes: device scalar.
 * loss / gloss [3, Lr] = (ce, bbox, giou);  bwd writes glogits and gboxes fully.  bad: optional malformed-box counter. */
int ocpg_det_loss_fwd_f32(const float* logits, const float* boxes, const long long* src, const float* valid, const long long* labels,
                          const float* tboxes, const float* num_boxes, float alpha, int Lr, int B, int T, int Q, int K, float* loss, int* bad,
                          void* stream);
int ocpg_det_loss_bwd_f32(const float* logits, const float* boxes, const long long* src, const float* valid, const long long* labels,
                          const float* tboxes, const float* num_boxes, const float* gloss, float alpha, int Lr, int B, int T, int Q, int K,
                          float* glogits, float* gboxes, void* stream);

/* Spectral gate of LFMResizeAdaptive (models/modules.py:44-50): out [N,2C,h,w] = [Re, Im](X * (1 - coef[n] * high)) for a complex64
 * spectrum X [N,C,h,w] (interleaved re/im), coef [N], high [h*w]; bwd: dX complex [N,C,h,w] fully written, part
 * [N, C * ceil(hw/256)] partial sums with dcoef[n] = sum(part[n]). */
int ocpg_spectral_gate_fwd(const void* X, const float* coef, const float* high, int N, int C, int hw, float* out, void* stream);
/* Channels-last forms of the same block (round 2): the [Re || Im] side as [N, hw, 2C] in the 1x1 convs' compute dtype (0 fp32 /
 * 1 bf16 / 2 fp16), the complex side as [N, C, hw]; 64 x 64 tile transposes through LDS.  c2p = the gate forward (coef, high)
 * or the plain split (coef NULL: the backward of torch.complex(yr, yi), models/modules.py:52-53); p2c = the gate backward
 * (Xs = the saved spectrum: dcoef partials into part [N, ceil(C/64) * ceil(hw/64)]) or the plain merge (Xs, coef NULL). */
int ocpg_spectral_c2p(const void* X, const float* coef, const float* high, int N, int C, int hw, void* pair, int dtype, void* stream);
int ocpg_spectral_p2c(const void* pair, const void* Xs, const float* coef, const float* high, int N, int C, int hw, void* out, float* part,
                      int dtype, void* stream);
int ocpg_spectral_gate_bwd(const float* gout, const void* X, const float* coef, const float* high, int N, int C, int hw, void* dX, float* part,
                           void* stream);

/* Heat-map-weighted cross entropy (masked_ce_loss, models/segmentation.py:177-201, via criterion.py:128-139), all layers:
 * x [Lr, per_layer] logits, w / t [per_layer] = weight map and weighted target of the clip (shared by the layers),
 * per_layer % 4 == 0.  fwd: part [Lr, 512] with loss[l] = sum(part[l]) / per_layer;  bwd: gx [Lr, per_layer] fully written. */
int ocpg_masked_ce_fwd_f32(const float* x, const float* w, const float* t, int Lr, long long per_layer, float* part, void* stream);
int ocpg_masked_ce_bwd_f32(const float* x, const float* w, const float* t, const float* gloss, int Lr, long long per_layer, float* gx,
                           void* stream);

/* Multi-head attention against a SHORT key sequence (Lk <= 32, head_dim 32, H <= 8 with 256 % H == 0) -- replaces the
 * softmax(q k^T * scale + key padding) v core of nn.MultiheadAttention as called by VisionLanguageFusionModule.forward
 * (models/segmentation.py:103-113) and by the decoder layer's self-attention (models/deformable_transformer.py:323-326); the
 * projections around it stay GEMMs.  q [Lq, B, H*32], k / v [Lk, B, H*32], out / dout / dq like q: row (l, b) of tensor X
 * starts at X + (l * B + b) * ldX elements (the projections' own layout, any row stride).  key_pad [B, Lk] bytes, non-zero
 * = ignore this key, or NULL.  pdrop: dropout on the attention weights (0 = off), mask from Philox(seed, offset) keyed on
 * (row, head, key) -- the backward regenerates it.  lse [Lq, B, H] fp32 kept for the backward.  dk / dv [Lk, B, H*32] fp32 are
 * ACCUMULATED into (caller zeroes them).  dtype 0 fp32 / 1 bf16 / 2 fp16 (storage; fp32 arithmetic).
 * Returns -2000 when the shape is not served (the caller keeps its generic attention path). */
int ocpg_attn_smallk_fwd(const void* q, long long ldq, const void* k, long long ldk, const void* v, long long ldv,
                         const unsigned char* key_pad, float scale, int Lq, int B, int H, int hd, int Lk, float pdrop,
                         unsigned long long seed, unsigned long long offset, const unsigned long long* rng_base, void* out,
                         long long ldo, float* lse, int dtype, void* stream);
int ocpg_attn_smallk_bwd(const void* q, long long ldq, const void* k, long long ldk, const void* v, long long ldv,
                         const unsigned char* key_pad, const void* dout, long long ldo, const float* lse, float scale, int Lq,
                         int B, int H, int hd, int Lk, float pdrop, unsigned long long seed, unsigned long long offset,
                         const unsigned long long* rng_base, void* dq, long long lddq, float* dk, float* dv, int dtype, void* stream);

/* LFM coefficient branch (models/modules.py:17-19,36-39: `self.fc(self.pool(self.laplace(x)))` with laplace a 3x3 VALID conv): the
 * spatial mean of a convolution is linear in the input, mean conv(x)[co] = b[co] + sum w[co,ci,ky,kx] m[ci,ky,kx] with m the mean
 * of x[ci] over the (h-2)x(w-2) window at offset (ky,kx).  x [planes, h, w] fp32 contiguous -> out [planes, 9] window means
 * (as_bf16 != 0: every element is rounded to bf16 first, as the autocast convolution would see it); bwd: gm [planes, 9] ->
 * dx [planes, h, w] fully written. */
/* The LFM block's 2-D Fourier transforms on channels-last maps (csrc/lfm_dft.hip) -- replaces torch.fft.fft2 / ifft2 + the gate / cat /
 * complex / .real / residual chain of LFMResizeAdaptive.forward (models/modules.py:44-56) and their backward:
 *   ocpg_lfm_spectrum_fwd: x real fp32 [N,H,W,C] -> pair [N,H,W,2C] (dtype code pair_dt; channels [0,C) = Re, [C,2C) = Im) =
 *       norm * fft2(x)[n,c,u,v] * (1 - coef[n] * high[u*W+v])   (coef / high NULL: no gate);
 *   ocpg_lfm_spectrum_inv: pair -> out real fp32 [N,H,W,C] = norm * Re(sum_{u,v} pair_c[u,v] * gate * exp(+2 pi i (uy/H + vx/W)))
 *       (+ residual, same layout as out, or NULL); coef_part (or NULL): [N, (W/2+1) * ceil(C/64)] partial sums of
 *       d/dcoef = -sum high * (pair_re * S_re + pair_im * S_im), S = z_saved / gate (z_saved = the forward's gated pair, pair_dt).
 * tmp: scratch of N*H*(W/2+1)*C float2.  Lengths must factor as L1 * L2 with both <= 16 and be <= 128 (ocpg_lfm_dft_supported;
 * ocpg_lfm_dft_split(L) = L1 * 256 + L2, 0 when not served; -2000 from the transforms: the caller keeps the library FFT).
 * tw_h / tw_w: the float2 tables of the length: [0, L) exp(-2 pi i m / L); then the L1-point DFT matrix exp(-2 pi i j k / L1) at
 * [k * L1 + j]; then the L2-point one -- L + L1^2 + L2^2 entries. */
int ocpg_lfm_dft_supported(int H, int W);
int ocpg_lfm_dft_split(int L);
int ocpg_lfm_spectrum_fwd(const float* x, const float* coef, const float* high, int N, int H, int W, int C, const void* tw_h, const void* tw_w,
                          float norm, void* tmp, void* pair, int pair_dt, void* stream);
int ocpg_lfm_spectrum_inv(const void* pair, int pair_dt, const float* coef, const float* high, const void* z_saved, float* coef_part, int N, int H,
                          int W, int C, const void* tw_h, const void* tw_w, float norm, void* tmp, const float* residual, float* out, void* stream);
int ocpg_window_means3x3_fwd(const float* x, long long planes, int h, int w, int as_bf16, float* out, void* stream);
int ocpg_window_means3x3_bwd(const float* gm, long long planes, int h, int w, float* dx, void* stream);
/* The same on a channels-last map x [N, h, w, C] fp32 (round 4): part [N][ocpg_window_sums3x3_cl_bands(h)][C * 9] = window SUMS of a band of
 * rows (the caller adds the bands and divides by (h-2)(w-2)); bwd_cl: dx [N, h, w, C] = the means' gradient (+ addend, dx's layout, or NULL). */
int ocpg_window_sums3x3_cl_bands(int h);
int ocpg_window_sums3x3_cl(const float* x, int N, int h, int w, int C, int as_bf16, float* part, void* stream);
int ocpg_window_means3x3_bwd_cl(const float* gm, int N, int h, int w, int C, const float* addend, float* dx, void* stream);

/* MSO mask refinement (reference models/decoder.py:14-46; called at models/ocpg.py:375-390 once per decoder layer): all of its
 * convolutions are 3x3 / padding 1 with <= 16 output channels per group, on CHANNELS-LAST maps (csrc/mso.hip).
 *   ocpg_mso_conv3x3   out[n,y,x,o] = sum_{tap,c} w[o][tap][c] * act(in[n,y+dy,x+dx,c]) (+ bias[o]) (+ addend[n % NA,y,x,o]), kept
 *                      where mask[n,y,x,o] > 0 (mask may be NULL), (+ residual[n,y,x,o]).  in [NB,H,W,C] (in_dt), act = ReLU when
 *                      relu_in; w [co_total][9][C] (w_dt; tap = 3 ky + kx); out [NB,H,W,co_total] (out_dt), any co_total (groups
 *                      of 16 output channels).  Forward convolutions AND their input gradients (the same convolution with flipped,
 *                      transposed weights; mask = the input of the ReLU in front of the forward convolution).
 *   ocpg_mso_wgrad     part[band][o][tap][c] = sum over the band's pixels of g[n,y,x,o] * act(x[n,y+dy,x+dx,c]); x [NB,H,W,C]
 *                      (x_dt), g [NB,H,W,co] fp32, co <= 16; bands = NB * ceil(H / rows_per_band), rows_per_band from
 *                      ocpg_mso_wgrad_rows(); the caller sums the bands (the weight gradient).  part_bias (may be NULL)
 *                      [bands][16]: the band's sum of g per output channel (the bias gradient's partial sums).  accumulate != 0:
 *                      part is [co][9][C] and part_bias [16], both ZEROED BY THE CALLER; every band adds its product with float
 *                      atomics (one launch, no reduction pass; the summation order is then not reproducible bit for bit).
 *   ocpg_bilinear_nhwc_fwd / _bwd   F.interpolate(mode="bilinear", align_corners=False, size=(HO, WO)) of a channels-last fp32
 *                      map [NB,H,W,C] (C % 4 == 0) and its input gradient (gather form, fully written, no atomics).
 * compute_dt: 1 / 2 = bf16 / fp16 operands with fp32 accumulation (the autocast convolution), 0 = fp32 operands. */
int ocpg_mso_conv3x3(const void* in, int in_dt, int relu_in, const void* w, int w_dt, const float* bias, const float* addend, int NA,
                     const void* mask, int mask_dt, const float* residual, void* out, int out_dt, int NB, int H, int W, int C,
                     int co_total, int compute_dt, void* stream);
int ocpg_mso_wgrad_rows(int NB, int H, int C, int compute_dt);
int ocpg_mso_wgrad(const void* x, int x_dt, int relu_in, const float* g, float* part, float* part_bias, int accumulate, int NB, int H, int W,
                   int C, int co, int rows_per_band, int compute_dt, void* stream);
int ocpg_bilinear_nhwc_fwd(const float* in, int NB, int H, int W, int C, int HO, int WO, float* out, void* stream);
int ocpg_bilinear_nhwc_bwd(const float* gout, int NB, int H, int W, int C, int HO, int WO, float* gin, void* stream);

/* nn.Linear over FEW rows (decoder layers, box / class heads, controller, LFM coefficient MLPs: models/deformable_transformer.py:
 * 313-336, models/ocpg.py:83-110) under autocast, one launch forward and ONE launch backward (csrc/small_linear.hip):
 *   fwd: y [R, Cout] bf16 = act(x [R, Cin] (fp32 when x_f32 != 0, else bf16) . w [Cout, Cin]^T (bf16) + b [Cout] (bf16 or NULL)),
 *        act = ReLU when relu != 0 (the MLPs' `F.relu(layer(x))`, models/ocpg.py:53-58)
 *   bwd: gx [R, Cin] in x's dtype (NULL: not needed), gw [Cout, Cin] bf16, gb [Cout] bf16 (NULL: no bias) from gy [R, Cout];
 *        y_relu = the forward's output when it applied the ReLU (gy is masked where y <= 0), else NULL
 * Cin must be a multiple of 64 and R <= 4096; otherwise -2000 (the caller keeps its GEMM path). */
int ocpg_small_linear_fwd(const void* x, int x_f32, const void* w, const void* b, int R, int Cin, int Cout, int relu, void* y, void* stream);
int ocpg_small_linear_bwd(const void* gy, int gy_f32, const void* x, int x_f32, const void* w, const void* y_relu, int R, int Cin, int Cout,
                          void* gx, void* gw, void* gb, void* stream);
/* The same for the fp32 islands (MSDeformAttn's projections over the decoder's few query rows run with autocast disabled, reference
 * models/deformable_transformer.py:329-332): x / w / b / y and every gradient fp32, exact fp32 products on the matrix cores
 * (csrc/small_linear_f32.hip).  gx NULL: not needed; gb NULL: no bias.  Cin % 64 != 0 or R > 4096: -2000. */
int ocpg_small_linear_f32_fwd(const float* x, const float* w, const float* b, int R, int Cin, int Cout, float* y, void* stream);
int ocpg_small_linear_f32_bwd(const float* gy, const float* x, const float* w, int R, int Cin, int Cout, float* gx, float* gw, float* gb, void* stream);

/* nn.LayerNorm over the last axis of a token matrix with low-precision input / output (Video-Swin blocks: norm1, norm2,
 * PatchMerging.norm -- models/video_swin_transformer.py:194,201,225): x [rows, C] with storage code x_f32 (1 fp32, 0 bf16, 2 fp16),
 * gamma / beta fp32 [C], y with storage code y_f32, mean / rstd [rows] fp32 kept for the backward; bwd: dx with code dx_f32, part_g / part_b
 * [ocpg_layernorm_blocks(rows), C] fp32 per-workgroup partial sums of dgamma / dbeta (the caller sums them).  C <= 1024, else -2000. */
int ocpg_layernorm_blocks(long long rows);
int ocpg_layernorm_fwd(const void* x, int x_f32, const float* gamma, const float* beta, long long rows, int C, float eps, void* y, int y_f32,
                       float* mean, float* rstd, void* stream);
int ocpg_layernorm_bwd(const void* gy, int gy_f32, const void* x, int x_f32, const float* gamma, const float* mean, const float* rstd,
                       long long rows, int C, void* dx, int dx_f32, float* part_g, float* part_b, void* stream);

/* nn.GroupNorm behind the input projections (models/ocpg.py:108-119: Conv2d -> GroupNorm(32, hidden) per level) between the layouts
 * its neighbours use (csrc/groupnorm.hip): x [N, HW, C] channels-last with storage code x_dtype (1 fp32, 0 bf16, 2 fp16) as the projection
 * GEMM writes it, y [N, C, HW] fp32 planes as the LFM's FFTs read them; gamma / beta fp32 [C]; mean / rstd [N, G] fp32 kept for the backward.
 *   bwd: gy [N, C, HW] fp32 -> dx [N, HW, C] in x's dtype; part [N, 2, C] fp32 = per-frame partial sums of dgamma ([:, 0]) and dbeta ([:, 1]),
 *        summed over N by the caller.
 * Groups of exactly 8 channels (C == 8 G, the reference's 256 / 32), else -2000.  Large maps (C == 256, HW >= 1500) are tiled over pixels
 * and need scratch: work = ocpg_groupnorm_cl_work(N, HW, C, G) floats (0: not needed, work may be NULL); the forward's and the backward's
 * scratch are independent (nothing is carried between the two calls through it). */
long long ocpg_groupnorm_cl_work(long long N, int HW, int C, int G);
int ocpg_groupnorm_cl_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, long long N, int HW, int C, int G, float eps,
                          float* y, float* mean, float* rstd, float* work, void* stream);
int ocpg_groupnorm_cl_bwd(const float* gy, const void* x, int x_dtype, const float* gamma, const float* mean, const float* rstd, long long N,
                          int HW, int C, int G, void* dx, float* part, float* work, void* stream);
/* The same with y / gy channels-last as well ([N, HW, C] fp32; round 4: the LFM's own transforms, csrc/lfm_dft.hip, read that). */
int ocpg_groupnorm_cl2cl_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, long long N, int HW, int C, int G, float eps,
                             float* y, float* mean, float* rstd, float* work, void* stream);
int ocpg_groupnorm_cl2cl_bwd(const float* gy, const void* x, int x_dtype, const float* gamma, const float* mean, const float* rstd, long long N,
                             int HW, int C, int G, void* dx, float* part, float* work, void* stream);

/* Backward of table[idx] ([T, H] -> [M, H]) for a STATIC index (the relative-position-bias lookup, models/video_swin_transformer.py:
 * 112-114,151-153): g [M, H] fp32, order [M] int64 = argsort(idx), seg [T + 1] int64 = CSR offsets of every table row's segment in
 * `order`; out [T, H] fully written.  Segmented sum, no atomics (autograd's index_put(accumulate): ~40 colliding atomics per address). */
int ocpg_gather_rows_bwd(const float* g, const long long* order, const long long* seg, int T, int H, float* out, void* stream);
/* Video-Swin's relative-position bias in the two layouts the window-attention kernels read, straight from the table (round 4) --
 * replaces `relative_position_bias_table[relative_position_index[:N, :N].reshape(-1)].reshape(N, N, -1).permute(2, 0, 1).contiguous()`
 * (models/video_swin_transformer.py:151-153) and the transposed copy: bias[h][a][b] = table[idx[a * ldi + b]][h] (idx: the int64 index
 * buffer, row stride ldi >= N), bias_t = its transpose over (a, b); both fp32 [H, N, N].  bwd: dtable [T, H] from the gradient of
 * `bias` stored [H][N N] (transposed = 0) or from its transpose (transposed != 0: the attention backward's dS sum as it lies);
 * order / seg: positions p = a N + b sorted by idx, segment offsets per table row (T + 1). */
int ocpg_relpos_bias_fwd(const float* table, const long long* idx, int N, long long ldi, int H, float* bias, float* bias_t, void* stream);
int ocpg_relpos_bias_bwd(const float* g, const long long* order, const long long* seg, int T, int H, int N, int transposed, float* out,
                         void* stream);

/* Row gather with padding slots: out [B, M, row] = idx[j] in [0, S) ? x [B, S, row][:, idx[j]] : 0; rows are row_bytes bytes (multiple of
 * 16, any dtype).  Video-Swin's pad + cyclic shift + window partition and its reverse (models/video_swin_transformer.py:171-199) are
 * two such gathers that are each other's backward. */
int ocpg_gather_rows_pad(const void* x, const long long* idx, long long B, long long S, long long M, long long row_bytes, void* out,
                         void* stream);

/* Whole-step HIP-graph capture support (no reference counterpart: the reference launches eagerly).  Replaces every memset node of
 * a captured, not yet instantiated hipGraph_t by a kernel node with the same destination, value, extent and edges: with the HIP
 * runtime of ROCm 7.x a captured hipMemsetAsync writes a stale pattern from the second launch of the instantiated graph on
 * (csrc/graph_util.hip).  *n_replaced (may be NULL) receives the number of nodes rewritten. */
int ocpg_graph_replace_memsets(void* hip_graph, int* n_replaced);
/* Topology of a captured graph: out[9] = nodes, edges, roots, max out-degree, max in-degree, kernel / memset / memcpy / other node
 * counts (a one-stream capture is a chain: roots 1, degrees <= 1). */
int ocpg_graph_stats(void* hip_graph, long long* out9);
/* Memcpy nodes of a captured graph: out[4 i ..] = kind (hipMemcpyKind), bytes, source, destination; returns their number (at most
 * cap are written).  Host-to-device nodes re-read their host source on every replay. */
int ocpg_graph_memcpy_nodes(void* hip_graph, long long* out, int cap);

/* library / build identification: returns e.g. "ocpg_hip gfx950 r1" */
const char* ocpg_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* OCPG_HIP_H */
