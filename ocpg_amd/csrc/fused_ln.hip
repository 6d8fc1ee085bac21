// Transformer-layer glue, fused: y = LayerNorm(res + dropout(x))  and  h = dropout(relu(a + bias)), forward and backward.
//
// Reference: every (de)encoder layer of models/deformable_transformer.py does `norm(src + dropout(sublayer))` three
// (two) times and `dropout(relu(linear1(src)))` once (:236-257, :313-336) as separate dropout / add / LayerNorm / ReLU
// kernels: on the encoder's [N*5100, 256] and [N*5100, 1024] activations that is 3-4 HBM passes per expression forward
// and more backward.  Here each expression is ONE pass forward and ONE backward:
//  * the dropout mask is never stored: a counter-based generator (Philox-4x32-10, keyed by (seed, call offset), counter =
//    element index / 4) is re-evaluated in the backward;
//  * LayerNorm: one wave per row (C <= 2048, C % 4 == 0), the row stays in registers between the mean, the variance
//    (two-pass, as ATen's kernel) and the normalisation; backward recomputes z = res + dropout(x) from its inputs, so
//    nothing but (mean, rstd) is saved; d(gamma), d(beta) are accumulated per lane over a persistent grid and flushed
//    with one atomic per column per workgroup;
//  * bias+ReLU+dropout: backward needs only the OUTPUT (h > 0 <=> active and kept), which the next GEMM keeps anyway.
// HBM-bound streaming kernels: 16 bytes per lane per access.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"
#include "philox.h"

namespace {

using ocpg_dev::philox;

// keep flags of the 4 elements starting at linear index idx4 * 4; thr = p * 2^32 (drop when rnd < thr)
__device__ __forceinline__ void keep4(uint64_t seed, uint64_t offset, uint64_t idx4, uint32_t thr, float scale, float (&k)[4]) {
  if (thr == 0u) { k[0] = k[1] = k[2] = k[3] = 1.f; return; }
  const uint4 r = philox(make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)),
                         make_uint4((uint32_t)idx4, (uint32_t)(idx4 >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)));
  k[0] = r.x >= thr ? scale : 0.f; k[1] = r.y >= thr ? scale : 0.f; k[2] = r.z >= thr ? scale : 0.f; k[3] = r.w >= thr ? scale : 0.f;
}

template <typename T> struct IO;
template <> struct IO<float> {
  static __device__ __forceinline__ void load4(const float* p, float (&f)[4]) { const float4 v = *reinterpret_cast<const float4*>(p); f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
  static __device__ __forceinline__ void store4(float* p, const float (&f)[4]) { *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct IO<__hip_bfloat16> {
  static __device__ __forceinline__ void load4(const __hip_bfloat16* p, float (&f)[4]) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u); f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void store4(__hip_bfloat16* p, const float (&f)[4]) {
    uint32_t w[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const __hip_bfloat16 lo = __float2bfloat16(f[2 * i]), hi = __float2bfloat16(f[2 * i + 1]);
      w[i] = (uint32_t)(*reinterpret_cast<const uint16_t*>(&lo)) | ((uint32_t)(*reinterpret_cast<const uint16_t*>(&hi)) << 16);
    }
    *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
  }
};

template <> struct IO<__half> {          // fp16 storage (round 4: the reference's --amp dtype, BASELINE config #5)
  static __device__ __forceinline__ void load4(const __half* p, float (&f)[4]) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    const __half2 a = *reinterpret_cast<const __half2*>(&v.x), b = *reinterpret_cast<const __half2*>(&v.y);
    const float2 fa = __half22float2(a), fb = __half22float2(b);
    f[0] = fa.x; f[1] = fa.y; f[2] = fb.x; f[3] = fb.y;
  }
  static __device__ __forceinline__ void store4(__half* p, const float (&f)[4]) {
    const __half2 a = __floats2half2_rn(f[0], f[1]), b = __floats2half2_rn(f[2], f[3]);
    uint2 v;
    v.x = *reinterpret_cast<const uint32_t*>(&a);
    v.y = *reinterpret_cast<const uint32_t*>(&b);
    *reinterpret_cast<uint2*>(p) = v;
  }
};

__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int NCH_MAX = 8;      // float4 chunks per lane: C <= 64 * 4 * NCH_MAX = 2048

// y = LN(res + dropout(x)) ; one wave per row
template <typename XT, int NCH>
__global__ __launch_bounds__(256) void dal_fwd(const XT* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, long long R, int C, float eps, uint32_t thr, float scale,
                                               uint64_t seed, uint64_t offset0, const uint64_t* __restrict__ rng_base,
                                               float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd) {
  const uint64_t offset = offset0 + (rng_base ? *rng_base : 0ull);      // graph replays: the step's base lives in device memory
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  const int nch = C / 4;
  float z[NCH][4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      float xv[4], rv[4], k[4];
      IO<XT>::load4(x + row * C + 4 * ch, xv);
      IO<float>::load4(res + row * C + 4 * ch, rv);
      keep4(seed, offset, (uint64_t)(row * C) / 4 + ch, thr, scale, k);
#pragma unroll
      for (int j = 0; j < 4; ++j) { z[i][j] = rv[j] + xv[j] * k[j]; s += z[i][j]; }
    }
  }
  const float mu = wave_sum_all(s) / (float)C;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (lane + 64 * i < nch)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = z[i][j] - mu; v += d * d; }
  const float rs = rsqrtf(wave_sum_all(v) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      float g[4], b[4], o[4];
      IO<float>::load4(gamma + 4 * ch, g);
      IO<float>::load4(beta + 4 * ch, b);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (z[i][j] - mu) * rs * g[j] + b[j];
      IO<float>::store4(y + row * C + 4 * ch, o);
    }
  }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// backward: persistent waves over the rows; gx (XT), gres (fp32), dgamma / dbeta accumulated (caller zeroes them)
template <typename XT, int NCH>
__global__ __launch_bounds__(256) void dal_bwd(const float* __restrict__ gy, const XT* __restrict__ x, const float* __restrict__ res,
                                               const float* __restrict__ gamma, const float* __restrict__ mean,
                                               const float* __restrict__ rstd, long long R, int C, uint32_t thr, float scale, uint64_t seed,
                                               uint64_t offset0, const uint64_t* __restrict__ rng_base, XT* __restrict__ gx,
                                               float* __restrict__ gres, float* __restrict__ dgamma) {
  const uint64_t offset = offset0 + (rng_base ? *rng_base : 0ull);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = C / 4;
  float dg[NCH][4], db[NCH][4], g[NCH][4];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dg[i][j] = db[i][j] = 0.f;
    if (lane + 64 * i < nch) IO<float>::load4(gamma + 4 * (lane + 64 * i), g[i]);
  }
  for (long long row = (long long)blockIdx.x * 4 + wave; row < R; row += (long long)gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    float xh[NCH][4], a[NCH][4], kk[NCH][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        float xv[4], rv[4], dy[4];
        IO<XT>::load4(x + row * C + 4 * ch, xv);
        IO<float>::load4(res + row * C + 4 * ch, rv);
        IO<float>::load4(gy + row * C + 4 * ch, dy);
        keep4(seed, offset, (uint64_t)(row * C) / 4 + ch, thr, scale, kk[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[i][j] = (rv[j] + xv[j] * kk[i][j] - mu) * rs;
          a[i][j] = dy[j] * g[i][j];
          s1 += a[i][j];
          s2 += a[i][j] * xh[i][j];
          dg[i][j] += dy[j] * xh[i][j];
          db[i][j] += dy[j];
        }
      }
    }
    s1 = wave_sum_all(s1) / (float)C;
    s2 = wave_sum_all(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        float gz[4], gxv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { gz[j] = rs * (a[i][j] - s1 - xh[i][j] * s2); gxv[j] = gz[j] * kk[i][j]; }
        if (gres) IO<float>::store4(gres + row * C + 4 * ch, gz);
        if (gx) IO<XT>::store4(gx + row * C + 4 * ch, gxv);
      }
    }
  }
  // flush d(gamma), d(beta): reduce the 4 waves of the workgroup through LDS, then ONE plain store per column into this
  // workgroup's row of the partial buffer dgb_part [gridDim.x, 2, C] (the caller sums the rows; same-address float atomics
  // from 1024 workgroups serialise)
  __shared__ float red[2][4][64 * 4];
  float* part = dgamma + (long long)blockIdx.x * 2 * C;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (64 * i >= nch) break;               // uniform
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][wave][lane * 4 + j] = dg[i][j]; red[1][wave][lane * 4 + j] = db[i][j]; }
    __syncthreads();
    const int col = threadIdx.x;            // 256 threads <-> the 256 columns of chunk group i
    const int ch = col / 4 + 64 * i;
    if (ch < nch) {
      part[64 * 4 * i + col] = red[0][0][col] + red[0][1][col] + red[0][2][col] + red[0][3][col];
      part[C + 64 * 4 * i + col] = red[1][0][col] + red[1][1][col] + red[1][2][col] + red[1][3][col];
    }
  }
}

// 16-byte lane accesses: V groups of 4 columns per lane (fp32: V = 1; bf16: V = 2 when C % 8 == 0)
template <typename T, int V> struct Wide {
  static __device__ __forceinline__ void load(const T* p, float (&f)[V][4]) {
#pragma unroll
    for (int v = 0; v < V; ++v) IO<T>::load4(p + 4 * v, f[v]);
  }
  static __device__ __forceinline__ void store(T* p, const float (&f)[V][4]) {
#pragma unroll
    for (int v = 0; v < V; ++v) IO<T>::store4(p + 4 * v, f[v]);
  }
};
template <> struct Wide<__hip_bfloat16, 2> {
  static __device__ __forceinline__ void load(const __hip_bfloat16* p, float (&f)[2][4]) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i / 2][2 * (i % 2)] = __uint_as_float(w[i] << 16); f[i / 2][2 * (i % 2) + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void store(__hip_bfloat16* p, const float (&f)[2][4]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __hip_bfloat16 lo = __float2bfloat16(f[i / 2][2 * (i % 2)]), hi = __float2bfloat16(f[i / 2][2 * (i % 2) + 1]);
      w[i] = (uint32_t)(*reinterpret_cast<const uint16_t*>(&lo)) | ((uint32_t)(*reinterpret_cast<const uint16_t*>(&hi)) << 16);
    }
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// h = dropout(relu(a + bias)) ; lane = 4*V consecutive columns (the generator's counter stays "element index / 4")
template <typename T, int V>
__global__ __launch_bounds__(256) void brd_fwd(const T* __restrict__ a, const T* __restrict__ bias, long long totalv, int C, uint32_t thr,
                                               float scale, uint64_t seed, uint64_t offset0, const uint64_t* __restrict__ rng_base,
                                               T* __restrict__ h) {
  const uint64_t offset = offset0 + (rng_base ? *rng_base : 0ull);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < totalv; i += (long long)gridDim.x * 256) {
    const int col = (int)((i * 4 * V) % C);
    float av[V][4], bv[V][4], o[V][4];
    Wide<T, V>::load(a + i * 4 * V, av);
    Wide<T, V>::load(bias + col, bv);
#pragma unroll
    for (int v = 0; v < V; ++v) {
      float k[4];
      keep4(seed, offset, (uint64_t)i * V + v, thr, scale, k);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[v][j] = fmaxf(av[v][j] + bv[v][j], 0.f) * k[j];
    }
    Wide<T, V>::store(h + i * 4 * V, o);
  }
}

// ga = (h > 0) ? gh * scale : 0 ; dbias_part[slot, c] = sum over the rows of that slot of ga[r, c]  (the caller sums the slots:
// 1024 workgroups x 1024 columns of same-address float atomics cost 3x the streaming time, measured).  A lane owns its
// columns for all the rows its slot visits (rows strided by the grid).
template <typename T, int V>
__global__ __launch_bounds__(256) void brd_bwd(const T* __restrict__ gh, const T* __restrict__ h, long long R, int C, float scale,
                                               T* __restrict__ ga, float* __restrict__ dbias) {
  const int lpr = C / (4 * V);                                  // lanes per row
  const int rpi = lpr >= 256 ? 1 : 256 / lpr;                   // rows per sweep of the workgroup (narrow matrices: several)
  const int ro = lpr >= 256 ? 0 : threadIdx.x / lpr;
  const int g0 = lpr >= 256 ? threadIdx.x : threadIdx.x % lpr;
  if (ro >= rpi) return;
  for (int g = g0; g < lpr; g += 256) {
    const int col = g * 4 * V;
    float acc[V][4];
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[v][j] = 0.f;
    for (long long r = (long long)blockIdx.x * rpi + ro; r < R; r += (long long)gridDim.x * rpi) {
      float gv[V][4], hv[V][4], o[V][4];
      Wide<T, V>::load(gh + r * C + col, gv);
      Wide<T, V>::load(h + r * C + col, hv);
#pragma unroll
      for (int v = 0; v < V; ++v)
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[v][j] = hv[v][j] > 0.f ? gv[v][j] * scale : 0.f; acc[v][j] += o[v][j]; }
      Wide<T, V>::store(ga + r * C + col, o);
    }
    // partial column sums of this (workgroup, row slot): plain stores, every element of dbias_part written exactly once
    float* o = dbias + ((long long)blockIdx.x * rpi + ro) * C + col;
#pragma unroll
    for (int v = 0; v < V; ++v) *reinterpret_cast<float4*>(o + 4 * v) = make_float4(acc[v][0], acc[v][1], acc[v][2], acc[v][3]);
  }
}

inline uint32_t threshold(float p) {
  if (p <= 0.f) return 0u;
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
}

}  // namespace

extern "C" {

int ocpg_dropout_add_ln_fwd(const void* x, const float* res, const float* gamma, const float* beta, long long R, int C, float eps, float p,
                            unsigned long long seed, unsigned long long offset, const unsigned long long* rng_base, int x_dtype, float* y,
                            float* mean, float* rstd, void* stream) {
  if (R < 0 || C <= 0 || C % 4 != 0 || C > 64 * 4 * NCH_MAX) return -1006;
  if (p < 0.f || p >= 1.f) return -1007;
  if (R == 0) return 0;
  if (!x || !res || !gamma || !beta) return -1001;
  if (!y || !mean || !rstd) return -1010;
  const uint32_t thr = threshold(p);
  const float scale = 1.f / (1.f - p);
  const unsigned grid = (unsigned)((R + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  if (x_dtype < 0 || x_dtype > 2) return -1008;
  const int nc = (C + 255) / 256;
#define DAL_FWD(XT_, N_) dal_fwd<XT_, N_><<<grid, 256, 0, st>>>((const XT_*)x, res, gamma, beta, R, C, eps, thr, scale, seed, offset, (const uint64_t*)rng_base, y, mean, rstd)
#define DAL_FWD_T(XT_) do { if (nc <= 1) DAL_FWD(XT_, 1); else if (nc <= 2) DAL_FWD(XT_, 2); else if (nc <= 4) DAL_FWD(XT_, 4); else DAL_FWD(XT_, 8); } while (0)
  if (x_dtype == 0) DAL_FWD_T(float); else if (x_dtype == 1) DAL_FWD_T(__hip_bfloat16); else DAL_FWD_T(__half);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

long long ocpg_dropout_add_ln_bwd_slots(long long R) {
  const long long want = (R + 3) / 4;
  return want < 1 ? 1 : (want < 1024 ? want : 1024);
}

int ocpg_dropout_add_ln_bwd(const float* gy, const void* x, const float* res, const float* gamma, const float* mean, const float* rstd,
                            long long R, int C, float p, unsigned long long seed, unsigned long long offset, const unsigned long long* rng_base,
                            int x_dtype, void* gx, float* gres, float* dgb_part, void* stream) {
  if (R < 0 || C <= 0 || C % 4 != 0 || C > 64 * 4 * NCH_MAX) return -1006;
  if (p < 0.f || p >= 1.f) return -1007;
  if (R == 0) return 0;
  if (!gy || !x || !res || !gamma || !mean || !rstd) return -1001;
  if (!dgb_part) return -1010;
  const uint32_t thr = threshold(p);
  const float scale = 1.f / (1.f - p);
  const unsigned grid = (unsigned)ocpg_dropout_add_ln_bwd_slots(R);
  hipStream_t st = (hipStream_t)stream;
  if (x_dtype < 0 || x_dtype > 2) return -1008;
  const int nc = (C + 255) / 256;
#define DAL_BWD(XT_, N_) dal_bwd<XT_, N_><<<grid, 256, 0, st>>>(gy, (const XT_*)x, res, gamma, mean, rstd, R, C, thr, scale, seed, offset, (const uint64_t*)rng_base, (XT_*)gx, gres, dgb_part)
#define DAL_BWD_T(XT_) do { if (nc <= 1) DAL_BWD(XT_, 1); else if (nc <= 2) DAL_BWD(XT_, 2); else if (nc <= 4) DAL_BWD(XT_, 4); else DAL_BWD(XT_, 8); } while (0)
  if (x_dtype == 0) DAL_BWD_T(float); else if (x_dtype == 1) DAL_BWD_T(__hip_bfloat16); else DAL_BWD_T(__half);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

int ocpg_bias_relu_dropout_fwd(const void* a, const void* bias, long long R, int C, float p, unsigned long long seed,
                               unsigned long long offset, const unsigned long long* rng_base, int dtype, void* h, void* stream) {
  if (R < 0 || C <= 0 || C % 4 != 0) return -1006;
  if (p < 0.f || p >= 1.f) return -1007;
  if (R == 0) return 0;
  if (!a || !bias) return -1001;
  if (!h) return -1010;
  const uint32_t thr = threshold(p);
  const float scale = 1.f / (1.f - p);
  const int V = (dtype != 0 && C % 8 == 0) ? 2 : 1;
  const long long totalv = R * C / (4 * V);
  const long long want = (totalv + 255) / 256;
  const unsigned grid = (unsigned)(want < 256 * 32 ? want : 256 * 32);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) brd_fwd<float, 1><<<grid, 256, 0, st>>>((const float*)a, (const float*)bias, totalv, C, thr, scale, seed, offset, (const uint64_t*)rng_base, (float*)h);
  else if (dtype == 1 && V == 2) brd_fwd<__hip_bfloat16, 2><<<grid, 256, 0, st>>>((const __hip_bfloat16*)a, (const __hip_bfloat16*)bias, totalv, C, thr, scale, seed, offset, (const uint64_t*)rng_base, (__hip_bfloat16*)h);
  else if (dtype == 1) brd_fwd<__hip_bfloat16, 1><<<grid, 256, 0, st>>>((const __hip_bfloat16*)a, (const __hip_bfloat16*)bias, totalv, C, thr, scale, seed, offset, (const uint64_t*)rng_base, (__hip_bfloat16*)h);
  else if (dtype == 2 && V == 2) brd_fwd<__half, 2><<<grid, 256, 0, st>>>((const __half*)a, (const __half*)bias, totalv, C, thr, scale, seed, offset, (const uint64_t*)rng_base, (__half*)h);
  else if (dtype == 2) brd_fwd<__half, 1><<<grid, 256, 0, st>>>((const __half*)a, (const __half*)bias, totalv, C, thr, scale, seed, offset, (const uint64_t*)rng_base, (__half*)h);
  else return -1008;
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

static int brd_rows_per_sweep(int C, int dtype) {
  const int V = (dtype != 0 && C % 8 == 0) ? 2 : 1;
  const int lpr = C / (4 * V);
  return lpr >= 256 ? 1 : 256 / lpr;
}
static unsigned brd_grid(long long R, int C, int dtype) {
  const long long want = (R + brd_rows_per_sweep(C, dtype) - 1) / brd_rows_per_sweep(C, dtype);
  return (unsigned)(want < 1024 ? want : 1024);
}

long long ocpg_bias_relu_dropout_bwd_slots(long long R, int C, int dtype) {
  if (R <= 0 || C <= 0 || C % 4 != 0) return 0;
  return (long long)brd_grid(R, C, dtype) * brd_rows_per_sweep(C, dtype);
}

int ocpg_bias_relu_dropout_bwd(const void* gh, const void* h, long long R, int C, float p, int dtype, void* ga, float* dbias, void* stream) {
  if (R < 0 || C <= 0 || C % 4 != 0) return -1006;
  if (p < 0.f || p >= 1.f) return -1007;
  if (R == 0) return 0;
  if (!gh || !h) return -1001;
  if (!ga || !dbias) return -1010;
  const float scale = 1.f / (1.f - p);
  hipStream_t st = (hipStream_t)stream;
  const int V = (dtype != 0 && C % 8 == 0) ? 2 : 1;
  const unsigned grid = brd_grid(R, C, dtype);
  if (dtype == 0) brd_bwd<float, 1><<<grid, 256, 0, st>>>((const float*)gh, (const float*)h, R, C, scale, (float*)ga, dbias);
  else if (dtype == 1 && V == 2) brd_bwd<__hip_bfloat16, 2><<<grid, 256, 0, st>>>((const __hip_bfloat16*)gh, (const __hip_bfloat16*)h, R, C, scale, (__hip_bfloat16*)ga, dbias);
  else if (dtype == 1) brd_bwd<__hip_bfloat16, 1><<<grid, 256, 0, st>>>((const __hip_bfloat16*)gh, (const __hip_bfloat16*)h, R, C, scale, (__hip_bfloat16*)ga, dbias);
  else if (dtype == 2 && V == 2) brd_bwd<__half, 2><<<grid, 256, 0, st>>>((const __half*)gh, (const __half*)h, R, C, scale, (__half*)ga, dbias);
  else if (dtype == 2) brd_bwd<__half, 1><<<grid, 256, 0, st>>>((const __half*)gh, (const __half*)h, R, C, scale, (__half*)ga, dbias);
  else return -1008;
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // extern "C"
