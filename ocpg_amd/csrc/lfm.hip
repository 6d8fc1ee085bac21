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

// ---- the same on a channels-last map [N, h, w, C] (round 4: GroupNorm and the LFM's own transforms keep the map channels-last): a lane
// owns a channel, a workgroup 64 channels x one band of rows of one frame; part [N][bands][C * 9] window SUMS, added up by the caller.
constexpr int CLU = 8;       // loads in flight per lane

__global__ __launch_bounds__(NT) void window_sums3x3_cl(const float* __restrict__ x, int h, int w, int C, int as_bf16, int rows_per_band,
                                                       float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (scalar pixel index arithmetic)
  const int n = blockIdx.z, band = blockIdx.y, c = blockIdx.x * 64 + lane;
  const bool ok = c < C;
  const int y0 = band * rows_per_band, y1 = min(h, y0 + rows_per_band);
  const int npx = (y1 - y0) * w;
  const float* p = x + ((long long)n * h + y0) * w * C + (ok ? c : 0);
  float acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.f;
  for (int i0 = wave; i0 < npx; i0 += (NT / 64) * CLU) {
    float v[CLU];
#pragma unroll
    for (int u = 0; u < CLU; ++u) {
      const int i = i0 + u * (NT / 64);
      v[u] = i < npx ? p[(long long)i * C] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < CLU; ++u) {
      const int i = i0 + u * (NT / 64);
      if (i >= npx) break;
      const int y = y0 + i / w, xx = i % w;
      const float t = as_bf16 ? round_bf16(v[u]) : v[u];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const bool rok = y >= ky && y <= h - 3 + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] += (rok && xx >= kx && xx <= w - 3 + kx) ? t : 0.f;
      }
    }
  }
  __shared__ float red[NT / 64][9][64];
#pragma unroll
  for (int k = 0; k < 9; ++k) red[wave][k][lane] = acc[k];
  __syncthreads();
  if (wave == 0 && ok) {
    float* o = part + (((long long)n * gridDim.y + band) * C + c) * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = (red[0][k][lane] + red[1][k][lane]) + (red[2][k][lane] + red[3][k][lane]);
  }
}

// dx [N, h, w, C] = sum over the windows containing the pixel of gm[n][c][k] / ((h-2)(w-2)) (+ addend, same layout, or NULL)
__global__ __launch_bounds__(NT) void window_means3x3_bwd_cl(const float* __restrict__ gm, int h, int w, int C, const float* __restrict__ addend,
                                                            float* __restrict__ dx) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = blockIdx.z, c = blockIdx.x * 64 + lane;
  if (c >= C) return;
  const float inv = 1.f / (float)((h - 2) * (w - 2));
  float g[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) g[k] = gm[((long long)n * C + c) * 9 + k] * inv;
  const int hw = h * w;
  for (int i = blockIdx.y * (NT / 64) + wave; i < hw; i += gridDim.y * (NT / 64)) {
    const int y = i / w, xx = i - y * w;
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const bool rok = y >= ky && y <= h - 3 + ky;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) s += (rok && xx >= kx && xx <= w - 3 + kx) ? g[ky * 3 + kx] : 0.f;
    }
    const long long o = ((long long)n * hw + i) * C + c;
    dx[o] = addend ? s + addend[o] : s;
  }
}

}  // namespace

extern "C" int ocpg_window_sums3x3_cl_bands(int h) { return h >= 16 ? 8 : (h >= 6 ? 2 : 1); }

extern "C" int ocpg_window_sums3x3_cl(const float* x, int N, int h, int w, int C, int as_bf16, float* part, void* stream) {
  if (N < 0 || h < 3 || w < 3 || C < 1) return -1003;
  if (N == 0) return 0;
  if (!x) return -1001;
  if (!part) return -1007;
  if (N > 65535) return -1002;
  const int bands = ocpg_window_sums3x3_cl_bands(h), rows = (h + bands - 1) / bands;
  window_sums3x3_cl<<<dim3((unsigned)((C + 63) / 64), (unsigned)bands, (unsigned)N), NT, 0, (hipStream_t)stream>>>(x, h, w, C, as_bf16, rows, part);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ocpg_window_means3x3_bwd_cl(const float* gm, int N, int h, int w, int C, const float* addend, float* dx, void* stream) {
  if (N < 0 || h < 3 || w < 3 || C < 1) return -1003;
  if (N == 0) return 0;
  if (!gm) return -1001;
  if (!dx) return -1007;
  if (N > 65535) return -1002;
  const int chunks = min(64, max(1, h * w / 64));
  window_means3x3_bwd_cl<<<dim3((unsigned)((C + 63) / 64), (unsigned)chunks, (unsigned)N), NT, 0, (hipStream_t)stream>>>(gm, h, w, C, addend, dx);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

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
