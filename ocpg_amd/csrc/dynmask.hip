// Dynamic (per-query) mask head, forward: fused relative-coordinate build + two per-query 1x1 convs (C+2 -> 16 -> 16).
//
// Reference (models/ocpg.py:475-549): repeat the [b,t,C,h,w] mask features per query, append 2 relative-coordinate
// channels (ref_xy * img_size - pixel centre, raw input pixels), reshape to [1, b*t*q*(C+2), h, w] (99 MB at
// config #2) and run two grouped F.conv2d with the controller's weights, ReLU in between -- once per decoder layer.
// Here: ONE launch for all decoder layers (the caller passes the Q = layers x queries parameter sets of a frame together;
// the features [BT, C, H, W] are shared).  A workgroup owns one query and a strip of pixels of its frame.  The query's
// parameters (17.7 KB) are staged into LDS once, transposed to input-channel-major, and read back as broadcast
// `ds_read_b128` (the 16 output weights of an input channel = 4 reads, every lane the same address -> conflict free).
// Each lane carries PX pixels x 16 output channels in registers as 8 float2, so the inner loop is pure `v_pk_fma_f32`
// (weights pair x broadcast feature); one LDS read feeds 4*PX FMAs and the kernel is bound by the fp32 VALU:
// FLOPs per query-pixel = 2*16*(C+2) + 2*16*16; HBM bytes = features in (L2-resident across the queries of a frame)
// + 2 x 16 channels out.  The coordinate channels are two FMAs from closed form; layer 2 (16x16) runs on the 16
// activations already in registers; nothing but the result and the layer-1 pre-activation (kept for the backward) is
// written.
// Parameter layout per query (parse_dynamic_params, ocpg.py:552-569): [16 x (C+2)] W0 (row o: C feature weights, then
// w_x, w_y), [16 x 16] W1, [16] b0, [16] b1.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"

namespace {

constexpr int CH = 16;      // dynamic_mask_channels: fixed by the reference (pixel_shuffle(.., 4), literal c=16 at ocpg.py:528)
constexpr int NT = 128;     // threads per workgroup
#ifndef DYNMASK_PX
#define DYNMASK_PX 2        // pixels per lane in the large-map kernel (measured: 2/4 waves 130 us, 4/2 157 us, 8/2 152 us)
#endif
#ifndef DYNMASK_WPE
#define DYNMASK_WPE 4       // waves per SIMD the register allocator is held to
#endif

typedef float v2f __attribute__((ext_vector_type(2)));

template <int PX>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(DYNMASK_WPE, 8))) void dynmask_fwd(
    const float* __restrict__ feats, const float* __restrict__ params, const float* __restrict__ refpix, int Q, int C, int H, int W,
    int stride, float* __restrict__ out, float* __restrict__ pre1) {
  extern __shared__ float4 smem4[];
  float* smem = reinterpret_cast<float*>(smem4);
  const int HW = H * W;
  const int bt = blockIdx.z;
  const long long n = (long long)bt * Q + blockIdx.y;
  const int NW0 = (C + 2) * CH;
  const int NP = NW0 + CH * CH + 2 * CH;
  // LDS: W0 transposed to [C+2][16] (input channel major: the 16 outputs of one input channel are 4 aligned float4),
  // W1 transposed to [16 in][16 out], b0 [16], b1 [16]
  float* w0t = smem;
  float* w1t = smem + (C + 2) * CH;
  float* b0 = w1t + CH * CH;
  float* b1 = b0 + CH;
  {
    const float* p = params + n * NP;
    for (int i = threadIdx.x; i < NW0; i += NT) w0t[i] = p[(i & 15) * (C + 2) + (i >> 4)];       // i = k*16 + o
    for (int i = threadIdx.x; i < CH * CH; i += NT) w1t[i] = p[NW0 + (i & 15) * CH + (i >> 4)];  // i = o*16 + o2
    if (threadIdx.x < 2 * CH) b0[threadIdx.x] = p[NW0 + CH * CH + threadIdx.x];
  }
  __syncthreads();

  const int px0 = blockIdx.x * (NT * PX) + threadIdx.x;
  int pxc[PX];
#pragma unroll
  for (int j = 0; j < PX; ++j) {
    const int px = px0 + j * NT;
    pxc[j] = px < HW ? px : HW - 1;
  }
  const float* fb = feats + (long long)bt * C * HW;
  v2f acc[PX][CH / 2];
#pragma unroll
  for (int j = 0; j < PX; ++j)
#pragma unroll
    for (int o = 0; o < CH / 2; ++o) acc[j][o] = (v2f)(0.f);

  // KC input channels per step; the next step's features are requested before this step's FMAs (register double buffer)
  constexpr int KC = 4;
  float fn[KC][PX];
#pragma unroll
  for (int kk = 0; kk < KC; ++kk)
#pragma unroll
    for (int j = 0; j < PX; ++j) fn[kk][j] = fb[(long long)(kk < C ? kk : C - 1) * HW + pxc[j]];
  for (int k0 = 0; k0 < C; k0 += KC) {
    float f[KC][PX];
#pragma unroll
    for (int kk = 0; kk < KC; ++kk)
#pragma unroll
      for (int j = 0; j < PX; ++j) f[kk][j] = fn[kk][j];
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      const int kn = k0 + KC + kk < C ? k0 + KC + kk : C - 1;          // clamped: the tail re-reads the last channel, unused
#pragma unroll
      for (int j = 0; j < PX; ++j) fn[kk][j] = fb[(long long)kn * HW + pxc[j]];
    }
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      if (k0 + kk < C) {                                                 // uniform; only the last step of a C % KC != 0 map
        const float4* wr = reinterpret_cast<const float4*>(w0t + (k0 + kk) * CH);
#pragma unroll
        for (int o4 = 0; o4 < CH / 4; ++o4) {
          const float4 wv = wr[o4];
          const v2f wa = {wv.x, wv.y}, wb = {wv.z, wv.w};
#pragma unroll
          for (int j = 0; j < PX; ++j) {
            const v2f ff = (v2f)(f[kk][j]);
            acc[j][2 * o4] = __builtin_elementwise_fma(wa, ff, acc[j][2 * o4]);
            acc[j][2 * o4 + 1] = __builtin_elementwise_fma(wb, ff, acc[j][2 * o4 + 1]);
          }
        }
      }
    }
  }

  const float rx = refpix[2 * n], ry = refpix[2 * n + 1];
  const v2f* wx2 = reinterpret_cast<const v2f*>(w0t + C * CH);
  const v2f* wy2 = reinterpret_cast<const v2f*>(w0t + (C + 1) * CH);
  const v2f* b02 = reinterpret_cast<const v2f*>(b0);
  const v2f* b12 = reinterpret_cast<const v2f*>(b1);
#pragma unroll
  for (int j = 0; j < PX; ++j) {
    asm volatile("" ::: "memory");      // keep each pixel's epilogue (and its LDS reads) separate: no cross-pixel hoisting/spills
    const int px = px0 + j * NT;
    const bool live = px < HW;
    const float relx = rx - (float)((pxc[j] % W) * stride + stride / 2);
    const float rely = ry - (float)((pxc[j] / W) * stride + stride / 2);
    float hbuf[CH];
#pragma unroll
    for (int o = 0; o < CH / 2; ++o) {
      const v2f v = acc[j][o] + wx2[o] * (v2f)(relx) + wy2[o] * (v2f)(rely) + b02[o];
      if (live && pre1) {
        pre1[(n * CH + 2 * o) * HW + px] = v.x;
        pre1[(n * CH + 2 * o + 1) * HW + px] = v.y;
      }
      hbuf[2 * o] = v.x > 0.f ? v.x : 0.f;
      hbuf[2 * o + 1] = v.y > 0.f ? v.y : 0.f;
    }
    v2f r[CH / 2];
#pragma unroll
    for (int o2 = 0; o2 < CH / 2; ++o2) r[o2] = b12[o2];
#pragma unroll
    for (int o = 0; o < CH; ++o) {
      const v2f* wrow = reinterpret_cast<const v2f*>(w1t + o * CH);
      const v2f hh = (v2f)(hbuf[o]);
#pragma unroll
      for (int o2 = 0; o2 < CH / 2; ++o2) r[o2] = __builtin_elementwise_fma(wrow[o2], hh, r[o2]);
    }
    if (live) {
#pragma unroll
      for (int o2 = 0; o2 < CH / 2; ++o2) {
        out[(n * CH + 2 * o2) * HW + px] = r[o2].x;
        out[(n * CH + 2 * o2 + 1) * HW + px] = r[o2].y;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Backward, everything that is not a large contraction, in two launches for all decoder layers:
//   dynmask_bwd_pre:  per (query n, strip of 256 pixels): dh = W1^T dout, dpre = dh * (pre1 > 0) -> dpre [n,16,hw] (operand of
//                     the two GEMMs dW0 = dpre f^T and dfeat = W0^T dpre, which stay hipBLASLt's), and the strip's partial
//                     sums of dW1 = dout relu(pre1)^T (16x16), db1 = sum dout, db0 = sum dpre, sum dpre * (x, y)  -> part
//                     [n, strips, 320]; strip 0 also packs the feature columns of W0 densely (A operand of the dfeat GEMM).
//   dynmask_bwd_fin:  per query: strip sums -> the tail of dparams (dW1, db0, db1), the coordinate columns dwx / dwy =
//                     ref * db0 - sum dpre * (x, y), dref = (W0[:,C] . db0, W0[:,C+1] . db0), and the placement of the dW0 GEMM's
//                     result into the [16, C+2] rows of dparams.
// The reference has no counterpart kernel: autograd differentiates the repeat / cat / two grouped conv2d of
// models/ocpg.py:505-549 (a [1, b*t*q*(C+2), h, w] tensor per decoder layer).
constexpr int BT_ = 256;      // pixels (= threads) per strip
constexpr int TS = 257;       // LDS row stride of a [16][256] tile (floats): row reads by 16 lanes hit 16 different banks

__global__ __launch_bounds__(BT_) void dynmask_bwd_pre(const float* __restrict__ dout, const float* __restrict__ pre1,
                                                       const float* __restrict__ params, int C, int HW, int W, int stride, int strips,
                                                       float* __restrict__ dpre, float* __restrict__ part, float* __restrict__ w0d) {
  __shared__ float w1[CH * CH];
  __shared__ float td[CH * TS], th[CH * TS], tp[CH * TS];      // dout, relu(pre1), dpre tiles of the strip
  const long long n = blockIdx.y;
  const int NW0 = (C + 2) * CH, NP = NW0 + CH * CH + 2 * CH;
  const float* p = params + n * NP;
  if (threadIdx.x < CH * CH) w1[threadIdx.x] = p[NW0 + threadIdx.x];          // W1[o2][o] at o2*16 + o
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < CH * C; i += BT_) w0d[n * CH * C + i] = p[(i / C) * (C + 2) + i % C];
  __syncthreads();
  const int px = blockIdx.x * BT_ + threadIdx.x;
  const bool live = px < HW;
  float go[CH], pr[CH];
#pragma unroll
  for (int o = 0; o < CH; ++o) {
    go[o] = live ? dout[(n * CH + o) * HW + px] : 0.f;
    pr[o] = live ? pre1[(n * CH + o) * HW + px] : 0.f;
  }
#pragma unroll
  for (int o = 0; o < CH; ++o) {
    float dh = 0.f;
#pragma unroll
    for (int o2 = 0; o2 < CH; ++o2) dh += w1[o2 * CH + o] * go[o2];
    const float dp = pr[o] > 0.f ? dh : 0.f;
    if (live) dpre[(n * CH + o) * HW + px] = dp;
    td[o * TS + threadIdx.x] = go[o];
    th[o * TS + threadIdx.x] = pr[o] > 0.f ? pr[o] : 0.f;
    tp[o * TS + threadIdx.x] = dp;
  }
  __syncthreads();
  // thread (a, o): dW1[a][o]; the a == 0 / 1 / 2 / 3 rows of threads also reduce db0 / db1 / sum dpre*x / sum dpre*y of channel o
  const int a = threadIdx.x / CH, o = threadIdx.x % CH;
  float acc = 0.f, extra = 0.f;
  const int px0 = blockIdx.x * BT_;
  for (int i = 0; i < BT_; ++i) {
    acc += td[a * TS + i] * th[o * TS + i];
    if (a < 4) {
      const int pxi = min(px0 + i, HW - 1);
      const float v = a == 1 ? td[o * TS + i] : tp[o * TS + i];
      const float wgt = a < 2 ? 1.f : a == 2 ? (float)((pxi % W) * stride + stride / 2) : (float)((pxi / W) * stride + stride / 2);
      extra += v * wgt;
    }
  }
  float* out = part + (n * strips + blockIdx.x) * 320;
  out[threadIdx.x] = acc;
  if (a < 4) out[256 + a * CH + o] = extra;        // [db0 | db1 | momx | momy] x 16
}

__global__ __launch_bounds__(320) void dynmask_bwd_fin(const float* __restrict__ part, const float* __restrict__ params,
                                                       const float* __restrict__ refpix, const float* __restrict__ dw0, int C, int strips,
                                                       float* __restrict__ dparams, float* __restrict__ dref) {
  __shared__ float s[320];
  const long long n = blockIdx.x;
  const int NW0 = (C + 2) * CH, NP = NW0 + CH * CH + 2 * CH;
  float acc = 0.f;
  for (int k = 0; k < strips; ++k) acc += part[(n * strips + k) * 320 + threadIdx.x];
  s[threadIdx.x] = acc;
  __syncthreads();
  float* dp = dparams + n * NP;
  const float* p = params + n * NP;
  const int t = threadIdx.x;
  if (t < 256) dp[NW0 + t] = s[t];                                        // dW1
  else if (t < 272) dp[NW0 + 256 + (t - 256)] = s[t];                     // db0
  else if (t < 288) dp[NW0 + 272 + (t - 272)] = s[t];                     // db1
  else if (t < 304) {                                                     // coordinate column x of W0 row o
    const int o = t - 288;
    dp[o * (C + 2) + C] = refpix[2 * n] * s[256 + o] - s[288 + o];
  } else {
    const int o = t - 304;
    dp[o * (C + 2) + C + 1] = refpix[2 * n + 1] * s[256 + o] - s[304 + o];
  }
  if (dref && t < 2) {
    float r = 0.f;
    for (int o = 0; o < CH; ++o) r += p[o * (C + 2) + C + t] * s[256 + o];
    dref[2 * n + t] = r;
  }
  for (int i = t; i < CH * C; i += 320) dp[(i / C) * (C + 2) + i % C] = dw0[n * CH * C + i];
}

}  // namespace

extern "C" int ocpg_dynmask_bwd_pre_f32(const float* dout, const float* pre1, const float* params, int BT, int Q, int C, int H, int W,
                                        int stride, float* dpre, float* part, float* w0d, void* stream) {
  if (BT < 0 || Q <= 0 || C <= 0 || H <= 0 || W <= 0) return -1006;
  if ((long long)BT * Q > 65535) return -1008;
  if (BT == 0) return 0;
  if (!dout) return -1001;
  if (!pre1) return -1002;
  if (!params) return -1003;
  if (!dpre || !part || !w0d) return -1010;
  const int HW = H * W, strips = (HW + BT_ - 1) / BT_;
  dynmask_bwd_pre<<<dim3(strips, BT * Q), BT_, 0, (hipStream_t)stream>>>(dout, pre1, params, C, HW, W, stride, strips, dpre, part, w0d);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ocpg_dynmask_bwd_fin_f32(const float* part, const float* params, const float* refpix, const float* dw0, int BT, int Q, int C,
                                        int H, int W, float* dparams, float* dref, void* stream) {
  if (BT < 0 || Q <= 0 || C <= 0 || H <= 0 || W <= 0) return -1006;
  if (BT == 0) return 0;
  if (!part || !params || !refpix || !dw0) return -1001;
  if (!dparams) return -1010;
  const int strips = (H * W + BT_ - 1) / BT_;
  dynmask_bwd_fin<<<BT * Q, 320, 0, (hipStream_t)stream>>>(part, params, refpix, dw0, C, strips, dparams, dref);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ocpg_dynmask_fwd_f32(const float* feats, const float* params, const float* refpix, int BT, int Q, int C, int H, int W,
                                    int stride, float* out, float* pre1, void* stream) {
  if (BT < 0 || Q <= 0 || C <= 0 || H <= 0 || W <= 0) return -1006;
  if (Q > 65535 || BT > 65535) return -1008;
  if (BT == 0) return 0;
  if (!feats) return -1001;
  if (!params) return -1002;
  if (!refpix) return -1003;
  if (!out) return -1010;
  const size_t lds = (size_t)(CH * (C + 2) + CH * CH + 2 * CH) * sizeof(float);
  if (lds > 64 * 1024) return -1009;
  hipStream_t st = (hipStream_t)stream;
  const int HW = H * W;
  if (HW >= 4 * NT * DYNMASK_PX) {
    const unsigned gx = (unsigned)((HW + NT * DYNMASK_PX - 1) / (NT * DYNMASK_PX));
    dynmask_fwd<DYNMASK_PX><<<dim3(gx, Q, BT), NT, lds, st>>>(feats, params, refpix, Q, C, H, W, stride, out, pre1);
  } else {
    const unsigned gx = (unsigned)((HW + NT - 1) / NT);
    dynmask_fwd<1><<<dim3(gx, Q, BT), NT, lds, st>>>(feats, params, refpix, Q, C, H, W, stride, out, pre1);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
