// Spectral gate of the LFM block: z = [Re, Im](fft2(x) * (1 - coef_n * gauss)) laid out as the [N, 2C, h, w] input of the
// following 1x1 conv, forward and backward.
//
// Reference: LFMResizeAdaptive.forward (models/modules.py:44-50): `high * coef`, `1 - ...`, complex multiply, `.real`, `.imag`,
// `cat` = 6 elementwise kernels forward over the [N,C,h,w] spectrum and ~14 backward (slice/select scatter, complex product
// rule, broadcast reductions), 8 LFM calls per step.  Here one HBM pass each way:
//   fwd: G = 1 - coef[n] * high[p];  out[n, c, p] = Re X * G;  out[n, C + c, p] = Im X * G
//   bwd: dX = (g_re + i g_im) * G;   dcoef[n] = - sum_{c,p} high[p] * (Re X * g_re + Im X * g_im)   (per-workgroup partials)
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

}  // extern "C"
