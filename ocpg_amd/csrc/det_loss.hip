// Classification (sigmoid focal), L1 and GIoU box losses of the matched queries, all decoder layers per launch, fwd + bwd.
//
// Reference: SetCriterion.loss_labels / loss_boxes (models/criterion.py:46-107) with sigmoid_focal_loss
// (models/segmentation.py:134-160) and util/box_ops.generalized_box_iou (:45-85): ~70 tiny kernels forward and ~140 backward
// on [Lr,B,T,q,*] tensors of a few hundred elements -- pure launch overhead.  Here: one small workgroup per layer.
//   loss_ce[l]   = sum_{b,t,q,k} a_t * bce(x, t) * (1 - p_t)^2 / num_boxes,   t = [q == src[l,b]] * valid[b,t] (* [k == label])
//   loss_bbox[l] = sum_{b,t} |box_sel - tgt|_1 / num_boxes;   loss_giou[l] = sum_{b,t} (1 - GIoU(xyxy(box_sel), xyxy(tgt))) / num_boxes
// The GIoU gradient w.r.t. (cx, cy, w, h) is evaluated in forward-mode dual arithmetic (4 tangents) -- no hand-derived formula.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"

namespace {

struct Dual {                 // value + d/d(cx, cy, w, h)
  float v, d[4];
};
__device__ __forceinline__ Dual cst(float v) { return {v, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ Dual operator+(const Dual& a, const Dual& b) { return {a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2], a.d[3] + b.d[3]}}; }
__device__ __forceinline__ Dual operator-(const Dual& a, const Dual& b) { return {a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2], a.d[3] - b.d[3]}}; }
__device__ __forceinline__ Dual operator*(const Dual& a, const Dual& b) {
  return {a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2], a.d[3] * b.v + a.v * b.d[3]}};
}
__device__ __forceinline__ Dual operator/(const Dual& a, const Dual& b) {
  const float q = a.v / b.v, ib = 1.f / b.v;
  return {q, {(a.d[0] - q * b.d[0]) * ib, (a.d[1] - q * b.d[1]) * ib, (a.d[2] - q * b.d[2]) * ib, (a.d[3] - q * b.d[3]) * ib}};
}
__device__ __forceinline__ Dual dmin(const Dual& a, const Dual& b) { return a.v <= b.v ? a : b; }
__device__ __forceinline__ Dual dmax(const Dual& a, const Dual& b) { return a.v >= b.v ? a : b; }
__device__ __forceinline__ Dual relu0(const Dual& a) { return a.v > 0.f ? a : cst(0.f); }      // clamp(min=0)

// GIoU of the predicted box (cx,cy,w,h as duals) against a constant target box; also reports malformed boxes
__device__ __forceinline__ Dual giou_dual(const float* pb, const float* tb, bool& malformed) {
  const Dual cx = {pb[0], {1.f, 0.f, 0.f, 0.f}}, cy = {pb[1], {0.f, 1.f, 0.f, 0.f}}, w = {pb[2], {0.f, 0.f, 1.f, 0.f}}, h = {pb[3], {0.f, 0.f, 0.f, 1.f}};
  const Dual half = cst(0.5f);
  const Dual ax0 = cx - half * w, ay0 = cy - half * h, ax1 = cx + half * w, ay1 = cy + half * h;
  const Dual bx0 = cst(tb[0] - 0.5f * tb[2]), by0 = cst(tb[1] - 0.5f * tb[3]), bx1 = cst(tb[0] + 0.5f * tb[2]), by1 = cst(tb[1] + 0.5f * tb[3]);
  malformed = !(ax1.v >= ax0.v && ay1.v >= ay0.v) || !(bx1.v >= bx0.v && by1.v >= by0.v);
  const Dual area_a = (ax1 - ax0) * (ay1 - ay0), area_b = (bx1 - bx0) * (by1 - by0);
  const Dual inter = relu0(dmin(ax1, bx1) - dmax(ax0, bx0)) * relu0(dmin(ay1, by1) - dmax(ay0, by0));
  const Dual uni = area_a + area_b - inter;
  const Dual eps = cst(1e-6f);
  const Dual iou = (inter + eps) / (uni + eps);
  const Dual hull = relu0(dmax(ax1, bx1) - dmin(ax0, bx0)) * relu0(dmax(ay1, by1) - dmin(ay0, by0));
  return iou - ((hull - uni) + eps) / (hull + eps);
}

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// focal term and its derivative for a binary target
__device__ __forceinline__ void focal(float x, float t, float alpha, float& loss, float& dx) {
  const float p = 1.f / (1.f + expf(-x));
  const float ce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
  const float pt = p * t + (1.f - p) * (1.f - t);
  const float om = 1.f - pt;
  const float at = alpha >= 0.f ? alpha * t + (1.f - alpha) * (1.f - t) : 1.f;
  loss = at * ce * om * om;
  dx = at * ((p - t) * om * om - 2.f * ce * om * (2.f * t - 1.f) * p * (1.f - p));
}

// one workgroup per layer; loss [3, Lr]
__global__ __launch_bounds__(256) void det_loss_fwd(const float* __restrict__ logits, const float* __restrict__ boxes,
                                                    const long long* __restrict__ src, const float* __restrict__ valid,
                                                    const long long* __restrict__ labels, const float* __restrict__ tboxes,
                                                    const float* __restrict__ num_boxes, float alpha, int Lr, int B, int T, int Q, int K,
                                                    float* __restrict__ loss, int* __restrict__ bad) {
  __shared__ float red[4];
  const int l = blockIdx.x;
  const float inv_nb = 1.f / num_boxes[0];
  float ce = 0.f, l1 = 0.f, gi = 0.f;
  bool malformed = false;
  const int n_el = B * T * Q * K;
  for (int i = threadIdx.x; i < n_el; i += 256) {
    const int k = i % K, q = (i / K) % Q, t = (i / (K * Q)) % T, b = i / (K * Q * T);
    const bool hit = q == (int)src[l * B + b] && valid[b * T + t] > 0.f && (labels ? k == (int)labels[b * T + t] : true);
    float f, dx;
    focal(logits[(long long)l * n_el + i], hit ? 1.f : 0.f, alpha, f, dx);
    ce += f;
  }
  for (int i = threadIdx.x; i < B * T; i += 256) {
    const int b = i / T;
    const float* pb = boxes + ((((long long)l * B + b) * T + i % T) * Q + (int)src[l * B + b]) * 4;
    const float* tb = tboxes + (long long)i * 4;
    l1 += fabsf(pb[0] - tb[0]) + fabsf(pb[1] - tb[1]) + fabsf(pb[2] - tb[2]) + fabsf(pb[3] - tb[3]);
    bool m;
    gi += 1.f - giou_dual(pb, tb, m).v;
    malformed |= m;
  }
  ce = block_sum(ce, red);
  l1 = block_sum(l1, red);
  gi = block_sum(gi, red);
  if (threadIdx.x == 0) { loss[l] = ce * inv_nb; loss[Lr + l] = l1 * inv_nb; loss[2 * Lr + l] = gi * inv_nb; }
  if (malformed && bad) atomicAdd(bad, 1);
}

// one lane per (l, b, t, q): glogits [Lr,B,T,Q,K], gboxes [Lr,B,T,Q,4]; gloss [3, Lr]
__global__ __launch_bounds__(256) void det_loss_bwd(const float* __restrict__ logits, const float* __restrict__ boxes,
                                                    const long long* __restrict__ src, const float* __restrict__ valid,
                                                    const long long* __restrict__ labels, const float* __restrict__ tboxes,
                                                    const float* __restrict__ num_boxes, const float* __restrict__ gloss, float alpha, int Lr,
                                                    int B, int T, int Q, int K, float* __restrict__ glogits, float* __restrict__ gboxes) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)Lr * B * T * Q) return;
  const int q = (int)(idx % Q), t = (int)((idx / Q) % T), b = (int)((idx / (Q * T)) % B), l = (int)(idx / ((long long)Q * T * B));
  const float inv_nb = 1.f / num_boxes[0];
  const bool matched = q == (int)src[l * B + b];
  const float gce = gloss[l] * inv_nb;
  for (int k = 0; k < K; ++k) {
    const bool hit = matched && valid[b * T + t] > 0.f && (labels ? k == (int)labels[b * T + t] : true);
    float f, dx;
    focal(logits[idx * K + k], hit ? 1.f : 0.f, alpha, f, dx);
    glogits[idx * K + k] = gce * dx;
  }
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  if (matched) {
    const float* pb = boxes + idx * 4;
    const float* tb = tboxes + ((long long)b * T + t) * 4;
    const float gl1 = gloss[Lr + l] * inv_nb, ggi = gloss[2 * Lr + l] * inv_nb;
    bool m;
    const Dual gd = giou_dual(pb, tb, m);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = pb[c] - tb[c];
      g[c] = gl1 * ((d > 0.f) - (d < 0.f)) - ggi * gd.d[c];
    }
  }
  *reinterpret_cast<float4*>(gboxes + idx * 4) = make_float4(g[0], g[1], g[2], g[3]);
}

}  // namespace

extern "C" {

int ocpg_det_loss_fwd_f32(const float* logits, const float* boxes, const long long* src, const float* valid, const long long* labels,
                          const float* tboxes, const float* num_boxes, float alpha, int Lr, int B, int T, int Q, int K, float* loss, int* bad,
                          void* stream) {
  if (Lr <= 0 || B <= 0 || T <= 0 || Q <= 0 || K <= 0) return -1006;
  if (!logits || !boxes || !src || !valid || !tboxes || !num_boxes) return -1001;
  if (!loss) return -1010;
  det_loss_fwd<<<Lr, 256, 0, (hipStream_t)stream>>>(logits, boxes, src, valid, labels, tboxes, num_boxes, alpha, Lr, B, T, Q, K, loss, bad);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

int ocpg_det_loss_bwd_f32(const float* logits, const float* boxes, const long long* src, const float* valid, const long long* labels,
                          const float* tboxes, const float* num_boxes, const float* gloss, float alpha, int Lr, int B, int T, int Q, int K,
                          float* glogits, float* gboxes, void* stream) {
  if (Lr <= 0 || B <= 0 || T <= 0 || Q <= 0 || K <= 0) return -1006;
  if (!logits || !boxes || !src || !valid || !tboxes || !num_boxes || !gloss) return -1001;
  if (!glogits || !gboxes) return -1010;
  const long long n = (long long)Lr * B * T * Q;
  det_loss_bwd<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(logits, boxes, src, valid, labels, tboxes, num_boxes, gloss, alpha,
                                                                            Lr, B, T, Q, K, glogits, gboxes);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // extern "C"
