// Fused frozen-BatchNorm affine (+ residual) (+ ReLU) epilogue for the ResNet body -- one HBM pass instead of 3-4.
//
// Reference arithmetic: FrozenBatchNorm2d.forward (models/backbone.py:46-56): y = x * scale[c] + shift[c] with
// scale = w * rsqrt(var + 1e-5), shift = b - mean * scale, followed in torchvision's Bottleneck by ReLU, or by
// "+ identity" then ReLU.  The reference runs that as 4 + 1 (+1) elementwise kernels per BN; 104 BNs in ResNet-101.
// Here: y = act(x * scale[c] + shift[c] (+ skip)), 16 bytes per lane, fp32 math, bf16 or fp32 storage;
// backward from the saved OUTPUT only: g = relu ? (y > 0 ? gy : 0) : gy;  gx = g * scale[c];  gskip = g.
// Layout: element (o, c, i) lives at ((o * C + c) * inner + i):  NHWC (channels_last): inner = 1, o = pixel;
// NCHW: inner = H*W, o = image.  HBM-bound streaming kernel: 2-3 tensors in, 1-2 out, nothing re-read.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"

namespace {

template <typename T>
struct Vec;
template <>
struct Vec<float> {
  static constexpr int N = 4;
  using V = float4;
  static __device__ __forceinline__ void load(const float* p, float (&f)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&f)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  }
  static __device__ __forceinline__ float get(const float* p) { return *p; }
  static __device__ __forceinline__ void put(float* p, float v) { *p = v; }
};
template <>
struct Vec<__hip_bfloat16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const __hip_bfloat16* p, float (&f)[8]) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(__hip_bfloat16* p, const float (&f)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __hip_bfloat16 lo = __float2bfloat16(f[2 * i]), hi = __float2bfloat16(f[2 * i + 1]);   // RNE, NaN-preserving
      w[i] = (uint32_t)(*reinterpret_cast<const uint16_t*>(&lo)) | ((uint32_t)(*reinterpret_cast<const uint16_t*>(&hi)) << 16);
    }
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  static __device__ __forceinline__ float get(const __hip_bfloat16* p) { return __bfloat162float(*p); }
  static __device__ __forceinline__ void put(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }
};

// MODE 0: inner == 1 (NHWC), C % N == 0: a vector spans N consecutive channels.
// MODE 1: inner % N == 0 (NCHW): a vector lies inside one channel plane.
// MODE 2: scalar fallback.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_act_fwd(const T* __restrict__ x, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, const T* __restrict__ skip, T* __restrict__ y,
                                                  long long total, int C, long long inner, int relu) {
  constexpr int N = (MODE == 2) ? 1 : Vec<T>::N;
  const long long nvec = total / N;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long long)gridDim.x * blockDim.x) {
    const long long e = v * N;
    if constexpr (MODE == 2) {
      const int c = (int)((e / inner) % C);
      float r = Vec<T>::get(x + e) * scale[c] + shift[c];
      if (skip) r += Vec<T>::get(skip + e);
      if (relu) r = r > 0.f ? r : 0.f;
      Vec<T>::put(y + e, r);
    } else {
      float f[N], s[N];
      Vec<T>::load(x + e, f);
      if constexpr (MODE == 0) {
        const int c0 = (int)(e % C);
#pragma unroll
        for (int i = 0; i < N; ++i) f[i] = f[i] * scale[c0 + i] + shift[c0 + i];
      } else {
        const int c = (int)((e / inner) % C);
        const float sc = scale[c], sh = shift[c];
#pragma unroll
        for (int i = 0; i < N; ++i) f[i] = f[i] * sc + sh;
      }
      if (skip) {
        Vec<T>::load(skip + e, s);
#pragma unroll
        for (int i = 0; i < N; ++i) f[i] += s[i];
      }
      if (relu) {
#pragma unroll
        for (int i = 0; i < N; ++i) f[i] = f[i] > 0.f ? f[i] : 0.f;
      }
      Vec<T>::store(y + e, f);
    }
  }
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_act_bwd(const T* __restrict__ gy, const T* __restrict__ y, const float* __restrict__ scale,
                                                  T* __restrict__ gx, T* __restrict__ gskip, long long total, int C, long long inner,
                                                  int relu) {
  constexpr int N = (MODE == 2) ? 1 : Vec<T>::N;
  const long long nvec = total / N;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long long)gridDim.x * blockDim.x) {
    const long long e = v * N;
    if constexpr (MODE == 2) {
      const int c = (int)((e / inner) % C);
      float g = Vec<T>::get(gy + e);
      if (relu && !(Vec<T>::get(y + e) > 0.f)) g = 0.f;
      if (gskip) Vec<T>::put(gskip + e, g);
      if (gx) Vec<T>::put(gx + e, g * scale[c]);
    } else {
      float g[N], o[N];
      Vec<T>::load(gy + e, g);
      if (relu) {
        Vec<T>::load(y + e, o);
#pragma unroll
        for (int i = 0; i < N; ++i) g[i] = o[i] > 0.f ? g[i] : 0.f;
      }
      if (gskip) Vec<T>::store(gskip + e, g);
      if (gx) {
        if constexpr (MODE == 0) {
          const int c0 = (int)(e % C);
#pragma unroll
          for (int i = 0; i < N; ++i) g[i] *= scale[c0 + i];
        } else {
          const float sc = scale[(int)((e / inner) % C)];
#pragma unroll
          for (int i = 0; i < N; ++i) g[i] *= sc;
        }
        Vec<T>::store(gx + e, g);
      }
    }
  }
}

inline unsigned grid_for(long long nvec) {
  const long long blocks = (nvec + 255) / 256;
  return (unsigned)(blocks < 1 ? 1 : (blocks > 256 * 8 ? 256 * 8 : blocks));   // <= 8 blocks per CU, grid-stride the rest
}

template <typename T>
int pick_mode(long long total, int C, long long inner) {
  const int N = Vec<T>::N;
  if (inner == 1 && C % N == 0) return 0;
  if (inner % N == 0) return 1;
  return 2;
}

template <typename T>
int launch_fwd(const void* x, const float* scale, const float* shift, const void* skip, void* y, long long n_outer, int C,
               long long inner, int relu, hipStream_t st) {
  const long long total = n_outer * C * inner;
  if (total == 0) return 0;
  const int mode = pick_mode<T>(total, C, inner);
  const long long nvec = mode == 2 ? total : total / Vec<T>::N;
  const unsigned grid = grid_for(nvec);
  const T* xp = (const T*)x; const T* sp = (const T*)skip; T* yp = (T*)y;
  if (mode == 0) bn_act_fwd<T, 0><<<grid, 256, 0, st>>>(xp, scale, shift, sp, yp, total, C, inner, relu);
  else if (mode == 1) bn_act_fwd<T, 1><<<grid, 256, 0, st>>>(xp, scale, shift, sp, yp, total, C, inner, relu);
  else bn_act_fwd<T, 2><<<grid, 256, 0, st>>>(xp, scale, shift, sp, yp, total, C, inner, relu);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

template <typename T>
int launch_bwd(const void* gy, const void* y, const float* scale, void* gx, void* gskip, long long n_outer, int C, long long inner,
               int relu, hipStream_t st) {
  const long long total = n_outer * C * inner;
  if (total == 0) return 0;
  const int mode = pick_mode<T>(total, C, inner);
  const long long nvec = mode == 2 ? total : total / Vec<T>::N;
  const unsigned grid = grid_for(nvec);
  const T* gp = (const T*)gy; const T* yp = (const T*)y; T* gxp = (T*)gx; T* gsp = (T*)gskip;
  if (mode == 0) bn_act_bwd<T, 0><<<grid, 256, 0, st>>>(gp, yp, scale, gxp, gsp, total, C, inner, relu);
  else if (mode == 1) bn_act_bwd<T, 1><<<grid, 256, 0, st>>>(gp, yp, scale, gxp, gsp, total, C, inner, relu);
  else bn_act_bwd<T, 2><<<grid, 256, 0, st>>>(gp, yp, scale, gxp, gsp, total, C, inner, relu);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace

extern "C" {

int ocpg_bn_act_fwd(const void* x, const float* scale, const float* shift, const void* skip, void* y, long long n_outer, int C,
                    long long inner, int relu, int dtype, void* stream) {
  if (n_outer < 0 || C <= 0 || inner <= 0) return -1006;
  if (n_outer * C * inner == 0) return 0;
  if (!x) return -1001;
  if (!scale) return -1002;
  if (!shift) return -1003;
  if (!y) return -1005;
  if (dtype == 0) return launch_fwd<float>(x, scale, shift, skip, y, n_outer, C, inner, relu, (hipStream_t)stream);
  if (dtype == 1) return launch_fwd<__hip_bfloat16>(x, scale, shift, skip, y, n_outer, C, inner, relu, (hipStream_t)stream);
  return -1010;
}

int ocpg_bn_act_bwd(const void* gy, const void* y, const float* scale, void* gx, void* gskip, long long n_outer, int C,
                    long long inner, int relu, int dtype, void* stream) {
  if (n_outer < 0 || C <= 0 || inner <= 0) return -1006;
  if (n_outer * C * inner == 0) return 0;
  if (!gy) return -1001;
  if (relu && !y) return -1002;
  if (!scale) return -1003;
  if (dtype == 0) return launch_bwd<float>(gy, y, scale, gx, gskip, n_outer, C, inner, relu, (hipStream_t)stream);
  if (dtype == 1) return launch_bwd<__hip_bfloat16>(gy, y, scale, gx, gskip, n_outer, C, inner, relu, (hipStream_t)stream);
  return -1010;
}

}  // extern "C"
