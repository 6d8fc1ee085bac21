// 3x3 convolution of channels-last (NHWC) maps as ONE dense GEMM: patch gather (im2col) and its adjoint (col2im).
//
// The ResNet body's 3x3 convs (torchvision Bottleneck.conv2 behind models/backbone.py:86-117) and the neck's 3x3 convs
// (models/ocpg.py:118-126) run at batch = 10 frames; MIOpen's best kernels for those shapes reach ~5 % of the bf16 MFMA
// peak on MI355X (layer3: 90 us forward, 166-187 us backward for 11.3 / 22.6 GFLOP).  With 288 GB of HBM the 9x patch
// matrix is affordable (44 MB for a layer3 conv), and [N*Ho*Wo, 9*C] x [9*C, Cout] is a plain large-K GEMM for
// hipBLASLt.  These two kernels are the HBM-bound glue:
//   im2col3x3: cols[m, (ky,kx,c)] = x[n, yo*s + (ky-1)*d, xo*s + (kx-1)*d, c]  (0 outside), m = (n, yo, xo)
//   col2im3x3: dx[n, y, x, c] = sum over the taps (ky,kx) and outputs m that read pixel (y, x)  of dcols[m, (ky,kx,c)]
// (padding == dilation d, stride s in {1, 2}: the only 3x3 geometry in the model).  The tap order (ky, kx, c) is the
// physical order of a channels-last weight [Cout, Cin, 3, 3], so the GEMM's second operand is a free view.
// One lane moves 16 bytes; every global access is a full 16-byte vector, consecutive lanes -> consecutive addresses.
// im2col: 1 read + 9 writes per input byte (reads hit L2: each pixel is re-read by its 9 taps); col2im: 9 reads + 1 write.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"

namespace {

__global__ __launch_bounds__(256) void im2col3x3(const uint4* __restrict__ x, uint4* __restrict__ cols, int H, int W, int C16, int Ho,
                                                 int Wo, int stride, int dil, long long total) {
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % C16);
    const long long r = idx / C16;
    const int t = (int)(r % 9);
    const long long m = r / 9;
    const int xo = (int)(m % Wo);
    const long long q = m / Wo;
    const int yo = (int)(q % Ho);
    const long long n = q / Ho;
    const int y = yo * stride + (t / 3 - 1) * dil;
    const int xx = xo * stride + (t % 3 - 1) * dil;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (y >= 0 && y < H && xx >= 0 && xx < W) v = x[((n * H + y) * W + xx) * C16 + c];
    cols[idx] = v;
  }
}

template <typename T>
struct Pack;
template <>
struct Pack<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void add(const uint4& v, float (&a)[4]) {
    a[0] += __uint_as_float(v.x); a[1] += __uint_as_float(v.y); a[2] += __uint_as_float(v.z); a[3] += __uint_as_float(v.w);
  }
  static __device__ __forceinline__ uint4 pack(const float (&a)[4]) {
    return make_uint4(__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3]));
  }
};
template <>
struct Pack<__hip_bfloat16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void add(const uint4& v, float (&a)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[2 * i] += __uint_as_float(w[i] << 16);
      a[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ uint4 pack(const float (&a)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __hip_bfloat16 lo = __float2bfloat16(a[2 * i]), hi = __float2bfloat16(a[2 * i + 1]);
      w[i] = (uint32_t)(*reinterpret_cast<const uint16_t*>(&lo)) | ((uint32_t)(*reinterpret_cast<const uint16_t*>(&hi)) << 16);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};
template <>
struct Pack<__half> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void add(const uint4& v, float (&a)[8]) {
    const __half2* h = reinterpret_cast<const __half2*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float2 f = __half22float2(h[i]);
      a[2 * i] += f.x;
      a[2 * i + 1] += f.y;
    }
  }
  static __device__ __forceinline__ uint4 pack(const float (&a)[8]) {
    uint4 v;
    __half2* h = reinterpret_cast<__half2*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = __floats2half2_rn(a[2 * i], a[2 * i + 1]);
    return v;
  }
};

template <typename T>
__global__ __launch_bounds__(256) void col2im3x3(const uint4* __restrict__ dcols, uint4* __restrict__ dx, int H, int W, int C16, int Ho,
                                                 int Wo, int stride, int dil, long long total) {
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % C16);
    const long long p = idx / C16;
    const int xx = (int)(p % W);
    const long long q = p / W;
    const int y = (int)(q % H);
    const long long n = q / H;
    float acc[Pack<T>::N];
#pragma unroll
    for (int i = 0; i < Pack<T>::N; ++i) acc[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ny = y - (t / 3 - 1) * dil, nx = xx - (t % 3 - 1) * dil;      // = yo*stride, xo*stride of the reader
      if (ny < 0 || nx < 0 || ny % stride != 0 || nx % stride != 0) continue;
      const int yo = ny / stride, xo = nx / stride;
      if (yo >= Ho || xo >= Wo) continue;
      const long long m = (n * Ho + yo) * Wo + xo;
      Pack<T>::add(dcols[(m * 9 + t) * C16 + c], acc);
    }
    dx[idx] = Pack<T>::pack(acc);
  }
}

inline unsigned grid_for(long long total) {
  const long long b = (total + 255) / 256;
  return (unsigned)(b < 256LL * 64 ? (b < 1 ? 1 : b) : 256LL * 64);
}

inline int check_geom(int N, int H, int W, int C, int stride, int dil, int dtype) {
  if (N < 0 || H <= 0 || W <= 0 || C <= 0 || dil <= 0) return -1006;
  if (stride != 1 && stride != 2) return -1007;
  if (dtype < 0 || dtype > 2) return -1010;
  const int es = dtype == 0 ? 4 : 2;
  if ((C * es) % 16 != 0) return -1008;
  return 0;
}

}  // namespace

extern "C" {

int ocpg_im2col3x3_nhwc(const void* x, int N, int H, int W, int C, int stride, int dil, void* cols, int dtype, void* stream) {
  const int rc = check_geom(N, H, W, C, stride, dil, dtype);
  if (rc) return rc;
  if (N == 0) return 0;
  if (!x) return -1001;
  if (!cols) return -1002;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int C16 = C * (dtype == 0 ? 4 : 2) / 16;
  const long long total = (long long)N * Ho * Wo * 9 * C16;
  im2col3x3<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(static_cast<const uint4*>(x), static_cast<uint4*>(cols), H, W, C16, Ho, Wo,
                                                              stride, dil, total);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

int ocpg_col2im3x3_nhwc(const void* dcols, int N, int H, int W, int C, int stride, int dil, void* dx, int dtype, void* stream) {
  const int rc = check_geom(N, H, W, C, stride, dil, dtype);
  if (rc) return rc;
  if (N == 0) return 0;
  if (!dcols) return -1001;
  if (!dx) return -1002;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int C16 = C * (dtype == 0 ? 4 : 2) / 16;
  const long long total = (long long)N * H * W * C16;
  const unsigned g = grid_for(total);
  hipStream_t st = (hipStream_t)stream;
  const uint4* src = static_cast<const uint4*>(dcols);
  uint4* dst = static_cast<uint4*>(dx);
  if (dtype == 0) col2im3x3<float><<<g, 256, 0, st>>>(src, dst, H, W, C16, Ho, Wo, stride, dil, total);
  else if (dtype == 1) col2im3x3<__hip_bfloat16><<<g, 256, 0, st>>>(src, dst, H, W, C16, Ho, Wo, stride, dil, total);
  else col2im3x3<__half><<<g, 256, 0, st>>>(src, dst, H, W, C16, Ho, Wo, stride, dil, total);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // extern "C"
