// GroupNorm(32, 256) behind the input projections (models/ocpg.py:108-119: Conv2d 1x1 -> GroupNorm per level, and the stride-2 3x3
// extra level), on the layouts the neighbours actually use: the projection GEMM writes a channels-last map [N, HW, C] in the autocast
// dtype, the LFM that follows runs fp32 FFTs over planes [N, C, HW].  ATen's path between the two is a cast, a layout copy, a
// moments kernel and an apply kernel forward, and five kernels + copy + cast backward.  Here: one launch each way.
//   fwd: y [N, C, HW] fp32 = (x - mean_g) * rstd_g * gamma_c + beta_c, statistics over the 8 channels x HW pixels of a group (biased
//        variance, two passes: mean, then centred squares); mean / rstd [N, G] kept
//   bwd: dx [N, HW, C] in x's dtype = rstd * (gy gamma - xhat * mean_g(gy gamma xhat) - mean_g(gy gamma));
//        part [N, 2, C] = per-frame partial sums of dgamma (gy xhat) / dbeta (gy); the caller sums them over N
// One workgroup per (frame, group).  A lane owns pixels: its group's 8 channels of a pixel are ONE 16-byte (bf16 / fp16) or 32-byte
// (fp32) load from the channels-last map, and for a fixed channel the 64 lanes of a wave touch 64 consecutive floats of the plane.
// The map of one frame (<= 1.8 MB) stays in L2 between the passes.  HBM-bound; algorithmic bytes = read x once + write y once.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/ocpg_hip.h"

namespace {

constexpr int NT = 256, D = 8;

// storage codes: 1 fp32, 0 bf16, 2 fp16
template <int DT> __device__ __forceinline__ void load8(const void* base, long long elem, float (&v)[D]) {
  if (DT == 1) {
    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem);
    const float4 a = p[0], b = p[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + elem);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (DT == 0) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
      } else {
        const __half2 h = *reinterpret_cast<const __half2*>(&w[i]);
        v[2 * i] = __low2float(h);
        v[2 * i + 1] = __high2float(h);
      }
    }
  }
}

template <int DT> __device__ __forceinline__ void store8(void* base, long long elem, const float (&v)[D]) {
  if (DT == 1) {
    float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + elem);
    p[0] = make_float4(v[0], v[1], v[2], v[3]);
    p[1] = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (DT == 0) {
        const __hip_bfloat16 lo = __float2bfloat16(v[2 * i]), hi = __float2bfloat16(v[2 * i + 1]);
        w[i] = (uint32_t) * reinterpret_cast<const uint16_t*>(&lo) | ((uint32_t) * reinterpret_cast<const uint16_t*>(&hi) << 16);
      } else {
        const __half2 h = __floats2half2_rn(v[2 * i], v[2 * i + 1]);
        w[i] = *reinterpret_cast<const uint32_t*>(&h);
      }
    }
    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(base) + elem) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum of K per-thread values over the workgroup; every thread gets every total
template <int K> __device__ __forceinline__ void block_sum(float (&v)[K], float (*red)[16]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
  __syncthreads();            // the previous use of red is over
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[wave][k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
}

template <int DT>
__global__ __launch_bounds__(NT) void gn_fwd(const void* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, int HW,
                                             int C, int G, float eps, float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int ycl) {
  __shared__ float red[NT / 64][16];
  const int g = blockIdx.x % G;
  const long long n = blockIdx.x / G;
  const long long x0 = n * HW * C + g * D;              // element offset of pixel 0, first channel of the group
  const float inv = 1.f / ((float)HW * D);
  float v[D];

  float s[1] = {0.f};
  for (int p = threadIdx.x; p < HW; p += NT) {
    load8<DT>(x, x0 + (long long)p * C, v);
#pragma unroll
    for (int j = 0; j < D; ++j) s[0] += v[j];
  }
  block_sum<1>(s, red);
  const float mu = s[0] * inv;

  float q[1] = {0.f};
  for (int p = threadIdx.x; p < HW; p += NT) {
    load8<DT>(x, x0 + (long long)p * C, v);
#pragma unroll
    for (int j = 0; j < D; ++j) q[0] += (v[j] - mu) * (v[j] - mu);
  }
  block_sum<1>(q, red);
  const float rs = rsqrtf(q[0] * inv + eps);

  float a[D], b[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    a[j] = rs * gamma[g * D + j];
    b[j] = beta[g * D + j] - mu * a[j];
  }
  float* yp = y + (n * C + g * D) * HW;
  for (int p = threadIdx.x; p < HW; p += NT) {
    load8<DT>(x, x0 + (long long)p * C, v);
    if (ycl) {                                          // y channels-last too (round 4: the LFM's own transforms read that)
      float o[D];
#pragma unroll
      for (int j = 0; j < D; ++j) o[j] = v[j] * a[j] + b[j];
      store8<1>(y, x0 + (long long)p * C, o);
    } else {
#pragma unroll
      for (int j = 0; j < D; ++j) yp[(long long)j * HW + p] = v[j] * a[j] + b[j];
    }
  }
  if (threadIdx.x == 0) {
    mean[n * G + g] = mu;
    rstd[n * G + g] = rs;
  }
}

template <int DT>
__global__ __launch_bounds__(NT) void gn_bwd(const float* __restrict__ gy, const void* __restrict__ x, const float* __restrict__ gamma,
                                             const float* __restrict__ mean, const float* __restrict__ rstd, int HW, int C, int G,
                                             void* __restrict__ dx, float* __restrict__ part, int ycl) {
  __shared__ float red[NT / 64][16];
  const int g = blockIdx.x % G;
  const long long n = blockIdx.x / G;
  const long long x0 = n * HW * C + g * D;
  const float mu = mean[n * G + g], rs = rstd[n * G + g];
  const float* gp = gy + (n * C + g * D) * HW;
  float gm[D], v[D];
#pragma unroll
  for (int j = 0; j < D; ++j) gm[j] = gamma[g * D + j];

  float acc[2 * D];                 // [0, D): sum gy xhat   [D, 2D): sum gy      (per channel)
#pragma unroll
  for (int j = 0; j < 2 * D; ++j) acc[j] = 0.f;
  for (int p = threadIdx.x; p < HW; p += NT) {
    load8<DT>(x, x0 + (long long)p * C, v);
    float gvv[D];
    if (ycl) load8<1>(gy, x0 + (long long)p * C, gvv);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float gv = ycl ? gvv[j] : gp[(long long)j * HW + p];
      acc[j] += gv * (v[j] - mu) * rs;
      acc[D + j] += gv;
    }
  }
  block_sum<2 * D>(acc, red);
  if (threadIdx.x < 2 * D) {
    const int j = threadIdx.x % D, which = threadIdx.x / D;
    float val = 0.f;
#pragma unroll
    for (int k = 0; k < 2 * D; ++k) val = k == (int)threadIdx.x ? acc[k] : val;
    part[(n * 2 + which) * C + g * D + j] = val;
  }
  float ca = 0.f, cb = 0.f;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    ca += gm[j] * acc[j];
    cb += gm[j] * acc[D + j];
  }
  const float inv = 1.f / ((float)HW * D);
  ca *= inv;
  cb *= inv;
  for (int p = threadIdx.x; p < HW; p += NT) {
    load8<DT>(x, x0 + (long long)p * C, v);
    float o[D], gvv[D];
    if (ycl) load8<1>(gy, x0 + (long long)p * C, gvv);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float xh = (v[j] - mu) * rs;
      o[j] = rs * ((ycl ? gvv[j] : gp[(long long)j * HW + p]) * gm[j] - ca * xh - cb);
    }
    store8<DT>(dx, x0 + (long long)p * C, o);
  }
}

// ---- large maps (level 0: 3 600 pixels x 256 channels per frame) --------------------------------------------------------------------------
// With one workgroup per (frame, group) every 128-byte line of the channels-last map is pulled into eight different CUs' L1s (each uses
// 16 bytes of it), three times: measured 81 us forward / 130 us backward at N = 10.  For big maps the work is tiled over pixels instead:
// a workgroup owns 64 pixels x all 256 channels, reads / writes the channels-last side in whole 512-byte pixel rows (lane = 16-byte
// octet), the plane side in 256-byte runs (lane = pixel), and turns one into the other through a 64 KB LDS tile whose rows are rotated by
// two banks per group so that both access patterns are conflict-free.  Statistics and the affine-gradient sums go through per-tile
// partials (combined with Chan's update for the variance), which costs a second launch each way.
constexpr int TP = 64, TC = 256, TG = 32;            // tile pixels, channels (== C), groups

__device__ __forceinline__ int tidx(int ch, int px) { return ch * TP + ((px + 2 * (ch >> 3)) & (TP - 1)); }

// sum over the 8 threads that share an octet (lane ^ 32, 4 waves); result valid in every thread; red: [4][32][K] floats
template <int K> __device__ __forceinline__ void octet_sum(float (&v)[K], float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, oct = threadIdx.x & 31;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += __shfl_xor(v[k], 32, 64);
  __syncthreads();
  if (lane < 32) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[(wave * 32 + oct) * K + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = red[(0 * 32 + oct) * K + k] + red[(1 * 32 + oct) * K + k] + red[(2 * 32 + oct) * K + k] + red[(3 * 32 + oct) * K + k];
}

// the thread's 8 pixel rows of the tile (octet `oct`), all loads issued back to back: rows beyond the map are clamped to the last valid
// pixel (the caller masks them) -- a load inside a branch would wait for its own data before the next one is issued
template <int DT>
__device__ __forceinline__ void load_rows(const void* __restrict__ x, long long n, int HW, int px0, int npx, int oct, int sub, float (&v)[8][D]) {
#pragma unroll
  for (int it = 0; it < 8; ++it) load8<DT>(x, (n * HW + px0 + min(it * 8 + sub, npx - 1)) * TC + oct * D, v[it]);
}

// stat [N, K, G, 2] = (mean, centred sum of squares) of every group over the tile's valid pixels
template <int DT>
__global__ __launch_bounds__(NT) void gn_tile_stats(const void* __restrict__ x, int HW, float* __restrict__ stat) {
  __shared__ float red[4 * 32];
  const int tile = blockIdx.x, K = gridDim.x, oct = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const long long n = blockIdx.y;
  const int px0 = tile * TP, npx = min(TP, HW - px0);
  float v[8][D];
  load_rows<DT>(x, n, HW, px0, npx, oct, sub, v);
  float s[1] = {0.f};
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    if (it * 8 + sub < npx) {
#pragma unroll
      for (int j = 0; j < D; ++j) s[0] += v[it][j];
    }
  }
  octet_sum<1>(s, red);
  const float mu = s[0] / (float)(npx * D);
  float q[1] = {0.f};
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    if (it * 8 + sub < npx) {
#pragma unroll
      for (int j = 0; j < D; ++j) q[0] += (v[it][j] - mu) * (v[it][j] - mu);
    }
  }
  octet_sum<1>(q, red);
  if (threadIdx.x < 32) {
    float* o = stat + ((n * K + tile) * TG + oct) * 2;
    o[0] = mu;
    o[1] = q[0];
  }
}

// combine the tile statistics of frame n into sm[0..31] = mean, sm[32..63] = rstd (Chan's update: total M2 = sum of the tiles' M2 + the
// spread of their means).  Thread (group = t & 31, slice = t >> 5) takes every 8th tile; `red` = 256 floats of scratch.
__device__ __forceinline__ void combine_stats(const float* __restrict__ stat, long long n, int K, int HW, float eps, float* sm, float* red) {
  const int g = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const float* st = stat + (n * K * TG + g) * 2;
  float m = 0.f;
#pragma unroll 8
  for (int k = slice; k < K; k += 8) m += st[(long long)k * TG * 2] * (float)(min(TP, HW - k * TP) * D);
  red[threadIdx.x] = m;
  __syncthreads();
  m = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) m += red[i * 32 + g];
  m /= (float)HW * D;
  __syncthreads();
  float m2 = 0.f;
#pragma unroll 8
  for (int k = slice; k < K; k += 8) {
    const float d = st[(long long)k * TG * 2] - m;
    m2 += st[(long long)k * TG * 2 + 1] + d * d * (float)(min(TP, HW - k * TP) * D);
  }
  red[threadIdx.x] = m2;
  __syncthreads();
  if (threadIdx.x < 32) {
    m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) m2 += red[i * 32 + g];
    sm[g] = m;
    sm[32 + g] = rsqrtf(m2 / ((float)HW * D) + eps);
  }
}

template <int DT>
__global__ __launch_bounds__(NT) void gn_tile_apply(const void* __restrict__ x, const float* __restrict__ stat, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, int HW, float eps, float* __restrict__ y,
                                                    float* __restrict__ mean, float* __restrict__ rstd, int ycl) {
  extern __shared__ __attribute__((aligned(16))) float lds[];          // [TC * TP] tile + 64 statistics
  float* sm = lds + TC * TP;
  const int tile = blockIdx.x, K = gridDim.x, oct = threadIdx.x & 31, sub = threadIdx.x >> 5, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long n = blockIdx.y;
  const int px0 = tile * TP, npx = min(TP, HW - px0);
  float v[8][D];
  load_rows<DT>(x, n, HW, px0, npx, oct, sub, v);          // in flight while the statistics are combined
  combine_stats(stat, n, K, HW, eps, sm, lds);
  __syncthreads();
  if (tile == 0 && threadIdx.x < 32) {
    mean[n * TG + threadIdx.x] = sm[threadIdx.x];
    rstd[n * TG + threadIdx.x] = sm[32 + threadIdx.x];
  }
  float a[D], b[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    a[j] = sm[32 + oct] * gamma[oct * D + j];
    b[j] = beta[oct * D + j] - sm[oct] * a[j];
  }
  if (ycl) {                                  // channels-last out: the thread's own pixel rows, no transpose
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int pl = it * 8 + sub;
      float o[D];
#pragma unroll
      for (int j = 0; j < D; ++j) o[j] = v[it][j] * a[j] + b[j];
      if (pl < npx) store8<1>(y, (n * HW + px0 + pl) * TC + oct * D, o);
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int pl = it * 8 + sub;              // rows beyond the map hold a copy of the last pixel: written, never read out
#pragma unroll
    for (int j = 0; j < D; ++j) lds[tidx(oct * D + j, pl)] = v[it][j] * a[j] + b[j];
  }
  __syncthreads();
  if (lane < npx) {
    float* yp = y + n * TC * HW + px0 + lane;
#pragma unroll 16
    for (int ch = wave; ch < TC; ch += NT / 64) yp[(long long)ch * HW] = lds[tidx(ch, lane)];
  }
}

// the tile of gy planes -> LDS (zero beyond the map); 16 loads in flight per lane
__device__ __forceinline__ void load_plane_tile(const float* __restrict__ gy, long long n, int HW, int px0, int npx, float* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* gp = gy + n * TC * HW + px0 + min(lane, npx - 1);
#pragma unroll
  for (int c0 = 0; c0 < TC; c0 += 64) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = gp[(long long)(c0 + wave + 4 * i) * HW];
#pragma unroll
    for (int i = 0; i < 16; ++i) lds[tidx(c0 + wave + 4 * i, lane)] = lane < npx ? r[i] : 0.f;
  }
}

// the same LDS image from a channels-last gy (rows beyond the map zero)
__device__ __forceinline__ void load_cl_tile(const float* __restrict__ gy, long long n, int HW, int px0, int npx, int oct, int sub, float* lds) {
  float g[8][D];
  load_rows<1>(gy, n, HW, px0, npx, oct, sub, g);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int pl = it * 8 + sub;
#pragma unroll
    for (int j = 0; j < D; ++j) lds[tidx(oct * D + j, pl)] = pl < npx ? g[it][j] : 0.f;
  }
}

// P [N, K, 2, C]: per tile, per channel: sum gy xhat, sum gy
template <int DT>
__global__ __launch_bounds__(NT) void gn_tile_bwd_sums(const float* __restrict__ gy, const void* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, int HW, float* __restrict__ P, int ycl) {
  extern __shared__ __attribute__((aligned(16))) float lds[];          // [TC * TP] tile, later reused as [4][32][16] reduction scratch
  const int tile = blockIdx.x, K = gridDim.x, oct = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const long long n = blockIdx.y;
  const int px0 = tile * TP, npx = min(TP, HW - px0);
  float v[8][D];
  load_rows<DT>(x, n, HW, px0, npx, oct, sub, v);
  if (ycl) load_cl_tile(gy, n, HW, px0, npx, oct, sub, lds);
  else load_plane_tile(gy, n, HW, px0, npx, lds);
  const float mu = mean[n * TG + oct], rs = rstd[n * TG + oct];
  __syncthreads();
  float acc[2 * D];
#pragma unroll
  for (int j = 0; j < 2 * D; ++j) acc[j] = 0.f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int pl = it * 8 + sub;              // beyond the map the gy tile is zero: the clamped copy of x adds nothing
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float gv = lds[tidx(oct * D + j, pl)];
      acc[j] += gv * (v[it][j] - mu) * rs;
      acc[D + j] += gv;
    }
  }
  octet_sum<2 * D>(acc, lds);                 // its first barrier ends the tile reads
  // thread t = channel t: its octet's totals sit in the scratch of any wave-row; re-read them from LDS in channel order
  __syncthreads();
  {
    const int c = threadIdx.x, o = c >> 3, j = c & 7;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      s1 += lds[(w * 32 + o) * 2 * D + j];
      s2 += lds[(w * 32 + o) * 2 * D + D + j];
    }
    float* out = P + (n * K + tile) * 2 * TC;
    out[c] = s1;
    out[TC + c] = s2;
  }
}

template <int DT>
__global__ __launch_bounds__(NT) void gn_tile_bwd_apply(const float* __restrict__ gy, const void* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ P,
                                                        int HW, void* __restrict__ dx, float* __restrict__ part, int ycl) {
  extern __shared__ __attribute__((aligned(16))) float lds[];          // [TC * TP] tile + 64 group coefficients
  float* coef = lds + TC * TP;
  const int tile = blockIdx.x, K = gridDim.x, oct = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const long long n = blockIdx.y;
  const int px0 = tile * TP, npx = min(TP, HW - px0);
  {
    const int c = threadIdx.x;
    const float* p = P + n * K * 2 * TC + c;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 16
    for (int k = 0; k < K; ++k) {
      s1 += p[(long long)k * 2 * TC];
      s2 += p[(long long)k * 2 * TC + TC];
    }
    if (tile == 0) {
      part[(n * 2 + 0) * TC + c] = s1;
      part[(n * 2 + 1) * TC + c] = s2;
    }
    float ca = gamma[c] * s1, cb = gamma[c] * s2;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      ca += __shfl_xor(ca, o, 64);
      cb += __shfl_xor(cb, o, 64);
    }
    if ((c & 7) == 0) {
      coef[c >> 3] = ca / ((float)HW * D);
      coef[32 + (c >> 3)] = cb / ((float)HW * D);
    }
  }
  float v[8][D];
  load_rows<DT>(x, n, HW, px0, npx, oct, sub, v);
  if (ycl) load_cl_tile(gy, n, HW, px0, npx, oct, sub, lds);
  else load_plane_tile(gy, n, HW, px0, npx, lds);
  const float mu = mean[n * TG + oct], rs = rstd[n * TG + oct];
  float gm[D];
#pragma unroll
  for (int j = 0; j < D; ++j) gm[j] = gamma[oct * D + j];
  __syncthreads();
  const float ca = coef[oct], cb = coef[32 + oct];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int pl = it * 8 + sub;
    float o[D];
#pragma unroll
    for (int j = 0; j < D; ++j) o[j] = rs * (lds[tidx(oct * D + j, pl)] * gm[j] - ca * (v[it][j] - mu) * rs - cb);
    if (pl < npx) store8<DT>(dx, (n * HW + px0 + pl) * TC + oct * D, o);
  }
}

constexpr size_t TILE_LDS = (size_t)(TC * TP + 64) * sizeof(float);

int tile_min() {
  static const int v = [] {
    const char* e = getenv("OCPG_GN_TILE_MIN");
    return e && *e ? atoi(e) : 1500;
  }();
  return v;
}
bool tiled(long long N, int HW, int C, int G) { return C == TC && G == TG && HW >= tile_min() && N < 65536; }

template <typename F> void allow_lds(F kernel) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE_LDS);
}

bool shape_ok(long long N, int HW, int C, int G, int dt) {
  return N > 0 && HW > 0 && G > 0 && C == G * D && N * G < (1ll << 31) && dt >= 0 && dt <= 2;
}

}  // namespace

#define GN_DISPATCH(KERNEL, ...)                                                    \
  do {                                                                              \
    if (x_dtype == 1) hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__);                   \
    else if (x_dtype == 2) hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__);              \
    else hipLaunchKernelGGL(KERNEL<0>, __VA_ARGS__);                                \
  } while (0)
#define GN_ALLOW_LDS(KERNEL)                                                        \
  do {                                                                              \
    if (x_dtype == 1) allow_lds(KERNEL<1>);                                         \
    else if (x_dtype == 2) allow_lds(KERNEL<2>);                                    \
    else allow_lds(KERNEL<0>);                                                      \
  } while (0)

extern "C" long long ocpg_groupnorm_cl_work(long long N, int HW, int C, int G) {
  if (!shape_ok(N, HW, C, G, 0) || !tiled(N, HW, C, G)) return 0;
  return N * ((HW + TP - 1) / TP) * 2 * TC;
}

static int gn_forward(const void* x, int x_dtype, const float* gamma, const float* beta, long long N, int HW, int C, int G, float eps,
                      float* y, float* mean, float* rstd, float* work, void* stream, int ycl) {
  if (!x || !gamma || !beta || !y || !mean || !rstd) return -1;
  if (!shape_ok(N, HW, C, G, x_dtype)) return -2000;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (tiled(N, HW, C, G)) {
    if (!work) return -1;
    const dim3 grid((unsigned)((HW + TP - 1) / TP), (unsigned)N), block(NT);
    GN_DISPATCH(gn_tile_stats, grid, block, 0, s, x, HW, work);
    GN_ALLOW_LDS(gn_tile_apply);
    GN_DISPATCH(gn_tile_apply, grid, block, TILE_LDS, s, x, work, gamma, beta, HW, eps, y, mean, rstd, ycl);
  } else {
    const dim3 grid((unsigned)(N * G)), block(NT);
    GN_DISPATCH(gn_fwd, grid, block, 0, s, x, gamma, beta, HW, C, G, eps, y, mean, rstd, ycl);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ocpg_groupnorm_cl_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, long long N, int HW, int C, int G, float eps,
                                     float* y, float* mean, float* rstd, float* work, void* stream) {
  return gn_forward(x, x_dtype, gamma, beta, N, HW, C, G, eps, y, mean, rstd, work, stream, 0);
}
extern "C" int ocpg_groupnorm_cl2cl_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, long long N, int HW, int C, int G, float eps,
                                        float* y, float* mean, float* rstd, float* work, void* stream) {
  return gn_forward(x, x_dtype, gamma, beta, N, HW, C, G, eps, y, mean, rstd, work, stream, 1);
}

static int gn_backward(const float* gy, const void* x, int x_dtype, const float* gamma, const float* mean, const float* rstd, long long N,
                       int HW, int C, int G, void* dx, float* part, float* work, void* stream, int ycl) {
  if (!gy || !x || !gamma || !mean || !rstd || !dx || !part) return -1;
  if (!shape_ok(N, HW, C, G, x_dtype)) return -2000;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (tiled(N, HW, C, G)) {
    if (!work) return -1;
    const dim3 grid((unsigned)((HW + TP - 1) / TP), (unsigned)N), block(NT);
    GN_ALLOW_LDS(gn_tile_bwd_sums);
    GN_DISPATCH(gn_tile_bwd_sums, grid, block, TILE_LDS, s, gy, x, mean, rstd, HW, work, ycl);
    GN_ALLOW_LDS(gn_tile_bwd_apply);
    GN_DISPATCH(gn_tile_bwd_apply, grid, block, TILE_LDS, s, gy, x, gamma, mean, rstd, work, HW, dx, part, ycl);
  } else {
    const dim3 grid((unsigned)(N * G)), block(NT);
    GN_DISPATCH(gn_bwd, grid, block, 0, s, gy, x, gamma, mean, rstd, HW, C, G, dx, part, ycl);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ocpg_groupnorm_cl_bwd(const float* gy, const void* x, int x_dtype, const float* gamma, const float* mean, const float* rstd, long long N,
                                     int HW, int C, int G, void* dx, float* part, float* work, void* stream) {
  return gn_backward(gy, x, x_dtype, gamma, mean, rstd, N, HW, C, G, dx, part, work, stream, 0);
}
extern "C" int ocpg_groupnorm_cl2cl_bwd(const float* gy, const void* x, int x_dtype, const float* gamma, const float* mean, const float* rstd, long long N,
                                        int HW, int C, int G, void* dx, float* part, float* work, void* stream) {
  return gn_backward(gy, x, x_dtype, gamma, mean, rstd, N, HW, C, G, dx, part, work, stream, 1);
}
