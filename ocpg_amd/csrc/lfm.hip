// LFM block (reference models/modules.py:9-61), the coefficient branch: coef = fc(mean_{y,x} laplace(x)) where laplace is a 3x3
// VALID convolution.  The spatial mean of a convolution is linear in the input:
//     mean_{y,x} conv(x)[co] = bias[co] + sum_{ci,ky,kx} w[co,ci,ky,kx] * m[ci,ky,kx],
//     m[ci,ky,kx] = mean of x[ci] over the (h-2) x (w-2) window whose top-left corner is (ky, kx),
// so the 45-GFLOP convolution (and its 90-GFLOP backward) at the finest level collapses into nine window means per plane
// (this file) and a [B, 9 C] x [9 C, C] matrix product.  HBM-bound: one read of x forward, one write of dx backward.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float round_bf16(float v) { return __bfloat162float(__float2bfloat16(v)); }

// one workgroup per (n, c) plane of a contiguous [planes, h, w] fp32 tensor -> out[plane][9] window MEANS
__global__ __launch_bounds__(NT) void window_means3x3_fwd(const float* __restrict__ x, int h, int w, int as_bf16, float* __restrict__ out) {
  const long long plane = blockIdx.x;
  const float* p = x + plane * h * w;
  float acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.f;
  const int hw = h * w;
  for (int i = threadIdx.x; i < hw; i += NT) {
    const int y = i / w, xx = i - y * w;
    float v = p[i];
    if (as_bf16) v = round_bf16(v);              // autocast: the convolution would have seen the bf16-rounded input
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const bool rok = y >= ky && y <= h - 3 + ky;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] += (rok && xx >= kx && xx <= w - 3 + kx) ? v : 0.f;
    }
  }
  __shared__ float red[NT / 64][9];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    float s = acc[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < NT / 64; ++wv) s += red[wv][threadIdx.x];
    out[plane * 9 + threadIdx.x] = s / (float)((h - 2) * (w - 2));
  }
}

// dx[plane][y][x] = sum over the windows that contain (y, x) of gm[plane][k] / ((h-2)(w-2))
__global__ __launch_bounds__(NT) void window_means3x3_bwd(const float* __restrict__ gm, int h, int w, float* __restrict__ dx) {
  const long long plane = blockIdx.x;
  __shared__ float g[9];
  if (threadIdx.x < 9) g[threadIdx.x] = gm[plane * 9 + threadIdx.x] / (float)((h - 2) * (w - 2));
  __syncthreads();
  float* p = dx + plane * h * w;
  const int hw = h * w;
  for (int i = threadIdx.x; i < hw; i += NT) {
    const int y = i / w, xx = i - y * w;
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const bool rok = y >= ky && y <= h - 3 + ky;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) s += (rok && xx >= kx && xx <= w - 3 + kx) ? g[ky * 3 + kx] : 0.f;
    }
    p[i] = s;
  }
}

}  // namespace

extern "C" int ocpg_window_means3x3_fwd(const float* x, long long planes, int h, int w, int as_bf16, float* out, void* stream) {
  if (planes < 0 || h < 3 || w < 3) return -1002;
  if (planes == 0) return 0;
  if (!x) return -1001;
  if (!out) return -1006;
  if (planes > 0x7fffffffLL) return -1002;
  window_means3x3_fwd<<<(unsigned)planes, NT, 0, (hipStream_t)stream>>>(x, h, w, as_bf16, out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ocpg_window_means3x3_bwd(const float* gm, long long planes, int h, int w, float* dx, void* stream) {
  if (planes < 0 || h < 3 || w < 3) return -1002;
  if (planes == 0) return 0;
  if (!gm) return -1001;
  if (!dx) return -1005;
  if (planes > 0x7fffffffLL) return -1002;
  window_means3x3_bwd<<<(unsigned)planes, NT, 0, (hipStream_t)stream>>>(gm, h, w, dx);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
