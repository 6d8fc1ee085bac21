// Dynamic (per-query) mask head, forward: fused relative-coordinate build + two per-query 1x1 convs (C+2 -> 16 -> 16).
//
// Reference (models/ocpg.py:475-549): repeat the [b,t,C,h,w] mask features per query, append 2 relative-coordinate
// channels (ref_xy * img_size - pixel centre, raw input pixels), reshape to [1, b*t*q*(C+2), h, w] (99 MB at
// config #2) and run two grouped F.conv2d with the controller's weights, ReLU in between.
// Here: one thread per pixel of a frame, all QT queries of the frame at once.  The features of the pixel are read
// ONCE (coalesced, channel after channel); the per-query weights are wave-uniform, so they arrive through scalar
// loads and every FMA is `v_fmac acc, s_weight, v_feature`; the coordinate channels are two extra FMAs from closed
// form; layer 2 (16x16) runs on the 16 activations that are already in registers; nothing but the result (and the
// layer-1 pre-activation, kept for the backward) is written.  HBM traffic = features in + 16 channels out per query.
// Parameter layout per query (parse_dynamic_params, ocpg.py:552-569): [16 x (C+2)] W0 (row o: C feature weights, then
// w_x, w_y), [16 x 16] W1, [16] b0, [16] b1.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"

namespace {

constexpr int CH = 16;      // dynamic_mask_channels: fixed by the reference (pixel_shuffle(.., 4), literal c=16 at ocpg.py:528)
constexpr int KC = 8;       // feature channels per scalar-load batch

template <int QT>
__global__ __launch_bounds__(256) void dynmask_fwd(const float* __restrict__ feats, const float* __restrict__ params,
                                                   const float* __restrict__ refpix, int Q, int C, int H, int W, int stride,
                                                   float* __restrict__ out, float* __restrict__ pre1) {
  const int HW = H * W;
  const int bt = blockIdx.z;
  const int q0 = blockIdx.y * QT;
  const int px = blockIdx.x * 256 + threadIdx.x;
  const bool live = px < HW;
  const int pxc = live ? px : HW - 1;
  const int NP = (C + 2) * CH + CH * CH + 2 * CH;
  const float* fb = feats + (long long)bt * C * HW + pxc;
  float acc[QT][CH];
#pragma unroll
  for (int q = 0; q < QT; ++q)
#pragma unroll
    for (int o = 0; o < CH; ++o) acc[q][o] = 0.f;
  for (int c0 = 0; c0 < C; c0 += KC) {
    float f[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) f[k] = (c0 + k < C) ? fb[(long long)(c0 + k) * HW] : 0.f;
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      if (q0 + q < Q) {      // uniform
        const float* w = params + (long long)(bt * Q + q0 + q) * NP + c0;     // wave-uniform address -> scalar loads
#pragma unroll
        for (int o = 0; o < CH; ++o) {
#pragma unroll
          for (int k = 0; k < KC; ++k)
            if (c0 + k < C) acc[q][o] += w[o * (C + 2) + k] * f[k];
        }
      }
    }
  }
  const float xs = (float)((pxc % W) * stride + stride / 2), ys = (float)((pxc / W) * stride + stride / 2);
#pragma unroll
  for (int q = 0; q < QT; ++q) {
    if (q0 + q >= Q) continue;
    const long long n = (long long)bt * Q + q0 + q;
    const float* pw = params + n * NP;
    const float relx = refpix[2 * n] - xs, rely = refpix[2 * n + 1] - ys;
    const float* w1 = pw + (C + 2) * CH;
    const float* b0 = w1 + CH * CH;
    const float* b1 = b0 + CH;
    float hbuf[CH];
#pragma unroll
    for (int o = 0; o < CH; ++o) {
      const float v = acc[q][o] + pw[o * (C + 2) + C] * relx + pw[o * (C + 2) + C + 1] * rely + b0[o];
      if (live && pre1) pre1[(n * CH + o) * HW + px] = v;
      hbuf[o] = v > 0.f ? v : 0.f;
    }
#pragma unroll
    for (int o2 = 0; o2 < CH; ++o2) {
      float r = b1[o2];
#pragma unroll
      for (int o = 0; o < CH; ++o) r += w1[o2 * CH + o] * hbuf[o];
      if (live) out[(n * CH + o2) * HW + px] = r;
    }
  }
}

}  // namespace

extern "C" int ocpg_dynmask_fwd_f32(const float* feats, const float* params, const float* refpix, int BT, int Q, int C, int H, int W,
                                    int stride, float* out, float* pre1, void* stream) {
  if (BT < 0 || Q <= 0 || C <= 0 || H <= 0 || W <= 0) return -1006;
  if (BT == 0) return 0;
  if (!feats) return -1001;
  if (!params) return -1002;
  if (!refpix) return -1003;
  if (!out) return -1010;
  hipStream_t st = (hipStream_t)stream;
  const int HW = H * W;
  const unsigned gx = (unsigned)((HW + 255) / 256);
  if (Q % 5 == 0 || Q > 4) {
    dynmask_fwd<5><<<dim3(gx, (Q + 4) / 5, BT), 256, 0, st>>>(feats, params, refpix, Q, C, H, W, stride, out, pre1);
  } else if (Q > 2) {
    dynmask_fwd<4><<<dim3(gx, (Q + 3) / 4, BT), 256, 0, st>>>(feats, params, refpix, Q, C, H, W, stride, out, pre1);
  } else if (Q == 2) {
    dynmask_fwd<2><<<dim3(gx, 1, BT), 256, 0, st>>>(feats, params, refpix, Q, C, H, W, stride, out, pre1);
  } else {
    dynmask_fwd<1><<<dim3(gx, 1, BT), 256, 0, st>>>(feats, params, refpix, Q, C, H, W, stride, out, pre1);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
