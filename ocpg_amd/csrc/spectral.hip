// Spectral gate of the LFM block: z = [Re, Im](fft2(x) * (1 - coef_n * gauss)) laid out as the [N, 2C, h, w] input of the
// following 1x1 conv, forward and backward.
//
// Reference: LFMResizeAdaptive.forward (models/modules.py:44-50): `high * coef`, `1 - ...`, complex multiply, `.real`, `.imag`,
// `cat` = 6 elementwise kernels forward over the [N,C,h,w] spectrum and ~14 backward (slice/select scatter, complex product
// rule, broadcast reductions), 8 LFM calls per step.  Here one HBM pass each way:
//   fwd: G = 1 - coef[n] * high[p];  out[n, c, p] = Re X * G;  out[n, C + c, p] = Im X * G
//   bwd: dX = (g_re + i g_im) * G;   dcoef[n] = - sum_{c,p} high[p] * (Re X * g_re + Im X * g_im)   (per-workgroup partials)
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"

namespace {

__global__ __launch_bounds__(256) void gate_fwd(const float2* __restrict__ X, const float* __restrict__ coef, const float* __restrict__ high,
                                                int C, int hw, float* __restrict__ out) {
  const int n = blockIdx.z, c = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  const float g = 1.f - coef[n] * high[p];
  const float2 x = X[((long long)n * C + c) * hw + p];
  out[((long long)n * 2 * C + c) * hw + p] = x.x * g;
  out[((long long)n * 2 * C + C + c) * hw + p] = x.y * g;
}

// part [N, C * gridDim.x]: partial sums for dcoef
__global__ __launch_bounds__(256) void gate_bwd(const float* __restrict__ gout, const float2* __restrict__ X, const float* __restrict__ coef,
                                                const float* __restrict__ high, int C, int hw, float2* __restrict__ dX,
                                                float* __restrict__ part) {
  __shared__ float red[4];
  const int n = blockIdx.z, c = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  if (p < hw) {
    const float h = high[p];
    const float g = 1.f - coef[n] * h;
    const float gr = gout[((long long)n * 2 * C + c) * hw + p], gi = gout[((long long)n * 2 * C + C + c) * hw + p];
    const float2 x = X[((long long)n * C + c) * hw + p];
    dX[((long long)n * C + c) * hw + p] = make_float2(gr * g, gi * g);
    acc = -h * (x.x * gr + x.y * gi);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[((long long)n * C + c) * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}


// ---- channels-last variants: the [Re || Im] side lives as [N, hw, 2C] (the operand layout of the 1x1 convs as GEMMs, in
// their compute dtype), the complex side as [N, C, hw] (what rocFFT wants).  Both directions are tile transposes through LDS:
// 64 channels x 64 pixels per workgroup, pixel-major reads / channel-major writes (or the reverse), all accesses coalesced.
//   c2p (complex -> pair):  pair[n, p, c] = Re X[n, c, p] * G,  pair[n, p, C + c] = Im X * G      (G = 1 - coef[n] high[p], or 1)
//   p2c (pair -> complex):  X'[n, c, p] = (pair[n, p, c] + i pair[n, p, C + c]) * G               (+ optional dcoef partials)
constexpr int TC = 64, TP = 64;

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return (float)*p; }
template <> __device__ __forceinline__ float ldf<__hip_bfloat16>(const __hip_bfloat16* p) { return __bfloat162float(*p); }
template <> __device__ __forceinline__ float ldf<__half>(const __half* p) { return __half2float(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v) { *p = (T)v; }
template <> __device__ __forceinline__ void stf<__hip_bfloat16>(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }
template <> __device__ __forceinline__ void stf<__half>(__half* p, float v) { *p = __float2half(v); }

template <typename T>
__global__ __launch_bounds__(256) void c2p_kernel(const float2* __restrict__ X, const float* __restrict__ coef, const float* __restrict__ high,
                                                  int C, int hw, T* __restrict__ pair) {
  __shared__ float2 tile[TC][TP + 1];
  const int n = blockIdx.z, c0 = blockIdx.y * TC, p0 = blockIdx.x * TP;
  const int lo = threadIdx.x & 63, hi = threadIdx.x >> 6;
  {
    const int p = p0 + lo;
    const float g = (p < hw && coef) ? 1.f - coef[n] * high[p] : 1.f;
#pragma unroll
    for (int r = 0; r < TC / 4; ++r) {
      const int c = c0 + r * 4 + hi;
      float2 v = make_float2(0.f, 0.f);
      if (c < C && p < hw) v = X[((long long)n * C + c) * hw + p];
      tile[r * 4 + hi][lo] = make_float2(v.x * g, v.y * g);
    }
  }
  __syncthreads();
  const int c = c0 + lo;
#pragma unroll
  for (int r = 0; r < TP / 4; ++r) {
    const int p = p0 + r * 4 + hi;
    if (c < C && p < hw) {
      const float2 v = tile[lo][r * 4 + hi];
      T* o = pair + ((long long)n * hw + p) * 2 * C + c;
      stf<T>(o, v.x);
      stf<T>(o + C, v.y);
    }
  }
}

// part [N, gridDim.y * gridDim.x]: per-workgroup partial sums of dcoef (only when Xs != nullptr)
template <typename T>
__global__ __launch_bounds__(256) void p2c_kernel(const T* __restrict__ pair, const float2* __restrict__ Xs, const float* __restrict__ coef,
                                                  const float* __restrict__ high, int C, int hw, float2* __restrict__ out,
                                                  float* __restrict__ part) {
  __shared__ float2 tile[TP][TC + 1];
  __shared__ float red[4];
  const int n = blockIdx.z, c0 = blockIdx.y * TC, p0 = blockIdx.x * TP;
  const int lo = threadIdx.x & 63, hi = threadIdx.x >> 6;
  {
    const int c = c0 + lo;
#pragma unroll
    for (int r = 0; r < TP / 4; ++r) {
      const int p = p0 + r * 4 + hi;
      float2 v = make_float2(0.f, 0.f);
      if (c < C && p < hw) {
        const T* i = pair + ((long long)n * hw + p) * 2 * C + c;
        v = make_float2(ldf<T>(i), ldf<T>(i + C));
      }
      tile[r * 4 + hi][lo] = v;
    }
  }
  __syncthreads();
  const int p = p0 + lo;
  const float h = (p < hw && coef) ? high[p] : 0.f;
  const float g = (p < hw && coef) ? 1.f - coef[n] * h : 1.f;
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < TC / 4; ++r) {
    const int c = c0 + r * 4 + hi;
    if (c < C && p < hw) {
      const float2 v = tile[lo][r * 4 + hi];
      const long long o = ((long long)n * C + c) * hw + p;
      out[o] = make_float2(v.x * g, v.y * g);
      if (Xs) {
        const float2 x = Xs[o];
        acc -= h * (x.x * v.x + x.y * v.y);
      }
    }
  }
  if (part) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lo == 0) red[hi] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(long long)n * gridDim.y * gridDim.x + blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
}

}  // namespace

extern "C" {

int ocpg_spectral_gate_fwd(const void* X, const float* coef, const float* high, int N, int C, int hw, float* out, void* stream) {
  if (N <= 0 || C <= 0 || hw <= 0 || N > 65535 || C > 65535) return -1006;
  if (!X || !coef || !high) return -1001;
  if (!out) return -1010;
  gate_fwd<<<dim3((hw + 255) / 256, C, N), 256, 0, (hipStream_t)stream>>>((const float2*)X, coef, high, C, hw, out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

/* part: [N, C * ceil(hw / 256)] fully written; dcoef[n] = part[n].sum() */
int ocpg_spectral_gate_bwd(const float* gout, const void* X, const float* coef, const float* high, int N, int C, int hw, void* dX, float* part,
                           void* stream) {
  if (N <= 0 || C <= 0 || hw <= 0 || N > 65535 || C > 65535) return -1006;
  if (!gout || !X || !coef || !high) return -1001;
  if (!dX || !part) return -1010;
  gate_bwd<<<dim3((hw + 255) / 256, C, N), 256, 0, (hipStream_t)stream>>>(gout, (const float2*)X, coef, high, C, hw, (float2*)dX, part);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}


/* Channels-last forms (the operand layout of the 1x1 convs as GEMMs).  dtype of the pair side: 0 fp32 / 1 bf16 / 2 fp16.
 * c2p: X complex64 [N,C,hw] -> pair [N,hw,2C] = [Re || Im](X * G), G = 1 - coef[n] * high[p] (coef == NULL: G = 1).
 * p2c: pair [N,hw,2C] -> out complex64 [N,C,hw] = (pair_re + i pair_im) * G; with Xs != NULL also the partial sums of
 *      dcoef[n] = -sum high[p] (Re Xs * pair_re + Im Xs * pair_im) into part [N, ceil(C/64) * ceil(hw/64)]. */
int ocpg_spectral_c2p(const void* X, const float* coef, const float* high, int N, int C, int hw, void* pair, int dtype, void* stream) {
  if (N <= 0 || C <= 0 || hw <= 0 || N > 65535) return -1006;
  if (!X || (coef && !high)) return -1001;
  if (!pair) return -1007;
  const dim3 grid((hw + TP - 1) / TP, (C + TC - 1) / TC, N);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) c2p_kernel<float><<<grid, 256, 0, st>>>((const float2*)X, coef, high, C, hw, (float*)pair);
  else if (dtype == 1) c2p_kernel<__hip_bfloat16><<<grid, 256, 0, st>>>((const float2*)X, coef, high, C, hw, (__hip_bfloat16*)pair);
  else if (dtype == 2) c2p_kernel<__half><<<grid, 256, 0, st>>>((const float2*)X, coef, high, C, hw, (__half*)pair);
  else return -1008;
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

int ocpg_spectral_p2c(const void* pair, const void* Xs, const float* coef, const float* high, int N, int C, int hw, void* out, float* part,
                      int dtype, void* stream) {
  if (N <= 0 || C <= 0 || hw <= 0 || N > 65535) return -1006;
  if (!pair || (coef && !high) || (Xs && !coef)) return -1001;
  if (!out || (Xs && !part)) return -1008;
  const dim3 grid((hw + TP - 1) / TP, (C + TC - 1) / TC, N);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) p2c_kernel<float><<<grid, 256, 0, st>>>((const float*)pair, (const float2*)Xs, coef, high, C, hw, (float2*)out, part);
  else if (dtype == 1) p2c_kernel<__hip_bfloat16><<<grid, 256, 0, st>>>((const __hip_bfloat16*)pair, (const float2*)Xs, coef, high, C, hw, (float2*)out, part);
  else if (dtype == 2) p2c_kernel<__half><<<grid, 256, 0, st>>>((const __half*)pair, (const float2*)Xs, coef, high, C, hw, (float2*)out, part);
  else return -1010;
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // extern "C"
