// Heat-map-weighted cross entropy of the weakly-supervised mask criterion, all decoder layers per launch, fwd + bwd.
//
// Reference: masked_ce_loss (models/segmentation.py:177-201) as called from SetCriterion.loss_masks
// (models/criterion.py:128-139): BCE-with-logits applied to sigmoid(x) * w against m * w (the reference's double
// squashing), mean over the clip's pixels.  As tensor ops: 5 elementwise/reduction kernels forward, 5 backward, per
// resolution, over the [Lr,B,T,H,W] logits (39 MB at full resolution).  Here one streaming pass each way:
//   u = sigmoid(x) * w;  t = m * w;  loss_l = mean( max(u,0) - u t + log1p(exp(-|u|)) )
//   dx = g_l / count * (sigmoid(u) - t) * w * s (1 - s)
// w, t: [F, HW] target maps shared by the layers (F = B*T).  Per-workgroup partial sums (no same-address atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"

namespace {

constexpr int NBLK = 512;       // workgroups per layer (grid-stride over the layer's elements)

__global__ __launch_bounds__(256) void mce_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ t,
                                               long long per_layer, float* __restrict__ part) {
  __shared__ float red[4];
  const int l = blockIdx.y;
  const float* xl = x + (long long)l * per_layer;
  float acc = 0.f;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < per_layer; i += (long long)gridDim.x * 1024) {
    const float4 xv = *reinterpret_cast<const float4*>(xl + i), wv = *reinterpret_cast<const float4*>(w + i), tv = *reinterpret_cast<const float4*>(t + i);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ws[4] = {wv.x, wv.y, wv.z, wv.w}, ts[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float u = ws[j] / (1.f + __expf(-xs[j]));
      acc += fmaxf(u, 0.f) - u * ts[j] + log1pf(__expf(-fabsf(u)));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[l * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void mce_bwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ t,
                                               const float* __restrict__ gloss, long long per_layer, float* __restrict__ gx) {
  const int l = blockIdx.y;
  const float* xl = x + (long long)l * per_layer;
  float* gl = gx + (long long)l * per_layer;
  const float g = gloss[l] / (float)per_layer;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < per_layer; i += (long long)gridDim.x * 1024) {
    const float4 xv = *reinterpret_cast<const float4*>(xl + i), wv = *reinterpret_cast<const float4*>(w + i), tv = *reinterpret_cast<const float4*>(t + i);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ws[4] = {wv.x, wv.y, wv.z, wv.w}, ts[4] = {tv.x, tv.y, tv.z, tv.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float s = 1.f / (1.f + __expf(-xs[j]));
      const float u = s * ws[j];
      o[j] = g * (1.f / (1.f + __expf(-u)) - ts[j]) * ws[j] * s * (1.f - s);
    }
    *reinterpret_cast<float4*>(gl + i) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

}  // namespace

extern "C" {

/* x [Lr, per_layer], w / t [per_layer] (per_layer % 4 == 0); part [Lr, 512]: loss[l] = sum(part[l]) / per_layer */
int ocpg_masked_ce_fwd_f32(const float* x, const float* w, const float* t, int Lr, long long per_layer, float* part, void* stream) {
  if (Lr <= 0 || per_layer <= 0 || per_layer % 4 != 0 || Lr > 65535) return -1006;
  if (!x || !w || !t) return -1001;
  if (!part) return -1010;
  mce_fwd<<<dim3(NBLK, Lr), 256, 0, (hipStream_t)stream>>>(x, w, t, per_layer, part);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

int ocpg_masked_ce_bwd_f32(const float* x, const float* w, const float* t, const float* gloss, int Lr, long long per_layer, float* gx,
                           void* stream) {
  if (Lr <= 0 || per_layer <= 0 || per_layer % 4 != 0 || Lr > 65535) return -1006;
  if (!x || !w || !t || !gloss) return -1001;
  if (!gx) return -1010;
  mce_bwd<<<dim3(NBLK, Lr), 256, 0, (hipStream_t)stream>>>(x, w, t, gloss, per_layer, gx);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // extern "C"
