// Matching cost of every query against the clip's single target, for all decoder layers in one launch.
//
// Reference: HungarianMatcher.forward (models/matcher.py:74-171): per clip (Python loop over B, syncing on the valid
// flags) focal class cost averaged over the valid frames, L1 and -GIoU box costs averaged over the frames, sigmoid-focal
// and -dice mask costs over all pixels of the clip, weighted sum, argmin over the queries.  As tensor ops that is ~140
// small kernels per step (all layers stacked); here the mask terms of every (layer, clip, query) are reduced by up to 64
// workgroups each (the masks are the only sizeable operand, streamed once) and a finishing kernel adds the per-frame
// class / box terms and forms the weighted total.
// HBM-bound: T*h*w*4 bytes per (layer, clip, query) + the shared target mask (L2-resident across the queries).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"
#include "fill.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

struct Box { float x0, y0, x1, y1; };
__device__ __forceinline__ Box to_xyxy(const float* b) { return {b[0] - 0.5f * b[2], b[1] - 0.5f * b[3], b[0] + 0.5f * b[2], b[1] + 0.5f * b[3]}; }
__device__ __forceinline__ bool well_formed(const Box& b) { return b.x1 >= b.x0 && b.y1 >= b.y0; }     // false for NaN

// GIoU with the reference's +1e-6 smoothing (util/box_ops.py:45-85)
__device__ __forceinline__ float giou(const Box& a, const Box& b) {
  const float area_a = (a.x1 - a.x0) * (a.y1 - a.y0), area_b = (b.x1 - b.x0) * (b.y1 - b.y0);
  const float iw = fmaxf(fminf(a.x1, b.x1) - fmaxf(a.x0, b.x0), 0.f), ih = fmaxf(fminf(a.y1, b.y1) - fmaxf(a.y0, b.y0), 0.f);
  const float inter = iw * ih, uni = area_a + area_b - inter;
  const float iou = (inter + 1e-6f) / (uni + 1e-6f);
  const float hw = fmaxf(fmaxf(a.x1, b.x1) - fminf(a.x0, b.x0), 0.f), hh = fmaxf(fmaxf(a.y1, b.y1) - fminf(a.y0, b.y0), 0.f);
  const float hull = hw * hh;
  return iou - ((hull - uni) + 1e-6f) / (hull + 1e-6f);
}

// partial sums of the mask terms: sums [Lr,B,Q,4] = focal sum, sum p*g, sum p, sum g; blockIdx.x splits the clip's pixels
__global__ __launch_bounds__(256) void matcher_mask_sums(const float* __restrict__ masks, long long sl, long long sb, long long st,
                                                         long long sq, const float* __restrict__ gt, int B, int T, int Q, int hw,
                                                         float* __restrict__ sums) {
  const int q = blockIdx.y, b = blockIdx.z % B, l = blockIdx.z / B;
  const float alpha = 0.25f;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const long long total = (long long)T * hw;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int t = (int)(e / hw), i = (int)(e % hw);
    const float xv = masks[l * sl + b * sb + t * st + q * sq + i], gv = gt[((long long)b * T + t) * hw + i];
    const float p = 1.f / (1.f + __expf(-xv));
    const float ce = fmaxf(xv, 0.f) - xv * gv + log1pf(__expf(-fabsf(xv)));
    const float pt = p * gv + (1.f - p) * (1.f - gv);
    const float om = 1.f - pt;
    acc[0] += (alpha * gv + (1.f - alpha) * (1.f - gv)) * ce * om * om;
    acc[1] += p * gv;
    acc[2] += p;
    acc[3] += gv;
  }
  __shared__ float red[4][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float s = wave_sum(acc[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int i = threadIdx.x;
    atomicAdd(sums + (((long long)l * B + b) * Q + q) * 4 + i, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
  }
}

// one lane per (layer, clip, query): class / box terms over the frames + the weighted total
__global__ __launch_bounds__(64) void matcher_final(const float* __restrict__ logits, const float* __restrict__ boxes,
                                                    const float* __restrict__ sums, const float* __restrict__ tboxes,
                                                    const float* __restrict__ valid, const long long* __restrict__ labels, int Lr, int B,
                                                    int T, int Q, int K, int hw, float wc, float wb, float wg, float wm, float wd,
                                                    float* __restrict__ cost, int* __restrict__ bad) {
  const int idx = blockIdx.x * 64 + threadIdx.x;
  if (idx >= Lr * B * Q) return;
  const int q = idx % Q, b = (idx / Q) % B, l = idx / (Q * B);
  const float alpha = 0.25f;
  const float* s = sums + (long long)idx * 4;
  const float cost_mask = s[0] / (float)((long long)T * hw);
  const float cost_dice = -((2.f * s[1] + 1.f) / (s[2] + s[3] + 1.f));
  float cls = 0.f, nvalid = 0.f, l1 = 0.f, gsum = 0.f;
  bool malformed = false;
  for (int t = 0; t < T; ++t) {
    const long long row = (((long long)l * B + b) * T + t) * Q + q;
    const int k = labels ? (int)labels[b * T + t] : 0;
    const float prob = 1.f / (1.f + expf(-logits[row * K + k]));
    const float neg = (1.f - alpha) * (prob * prob) * (-logf(1.f - prob + 1e-8f));
    const float pos = alpha * ((1.f - prob) * (1.f - prob)) * (-logf(prob + 1e-8f));
    const float v = valid[b * T + t];
    cls += (pos - neg) * v;
    nvalid += v;
    const float* pb = boxes + row * 4;
    const float* tb = tboxes + ((long long)b * T + t) * 4;
    l1 += fabsf(pb[0] - tb[0]) + fabsf(pb[1] - tb[1]) + fabsf(pb[2] - tb[2]) + fabsf(pb[3] - tb[3]);
    const Box a = to_xyxy(pb), c = to_xyxy(tb);
    malformed |= !well_formed(a) || !well_formed(c);
    gsum += giou(a, c);
  }
  cost[idx] = wc * (cls / nvalid) + wb * (l1 / (float)T) + wg * (-gsum / (float)T) + wm * cost_mask + wd * cost_dice;
  if (malformed && bad) atomicAdd(bad, 1);
}

}  // namespace

extern "C" int ocpg_matcher_cost_f32(const float* logits, const float* boxes, const float* masks, long long sl, long long sb, long long st,
                                     long long sq, const float* gt, const float* tboxes, const float* valid, const long long* labels,
                                     int Lr, int B, int T, int Q, int K, int h, int w, float wc, float wb, float wg, float wm, float wd,
                                     float* sums, float* cost, int* bad, void* stream) {
  if (Lr <= 0 || B <= 0 || T <= 0 || Q <= 0 || K <= 0 || h <= 0 || w <= 0) return -1006;
  if (B > 65535 || Lr > 65535) return -1007;
  if (!logits) return -1001;
  if (!boxes) return -1002;
  if (!masks) return -1003;
  if (!gt || !tboxes || !valid) return -1004;
  if (!sums || !cost) return -1010;
  if ((long long)B * Lr > 65535 || Q > 65535) return -1007;
  hipStream_t s_ = (hipStream_t)stream;
  hipError_t e = ocpg_fill::zero_async(sums, sizeof(float) * 4 * (size_t)Lr * B * Q, s_);
  if (e != hipSuccess) return -(int)e;
  const long long total = (long long)T * h * w;
  const unsigned split = (unsigned)((total + 256 * 16 - 1) / (256 * 16) < 64 ? (total + 256 * 16 - 1) / (256 * 16) : 64);   // >= 16 px per lane
  matcher_mask_sums<<<dim3(split < 1 ? 1 : split, Q, B * Lr), 256, 0, s_>>>(masks, sl, sb, st, sq, gt, B, T, Q, h * w, sums);
  matcher_final<<<(Lr * B * Q + 63) / 64, 64, 0, s_>>>(logits, boxes, sums, tboxes, valid, labels, Lr, B, T, Q, K, h * w, wc, wb, wg, wm, wd,
                                                       cost, bad);
  e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
