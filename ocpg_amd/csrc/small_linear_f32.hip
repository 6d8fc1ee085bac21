// Linear layers over FEW rows in fp32 -- the fp32 islands of the decoder (round 4): MSDeformAttn's sampling_offsets /
// attention_weights / output_proj run with autocast disabled (reference models/deformable_transformer.py:329-332) on the decoder's
// 50 query rows, so they missed the bf16 one-launch kernels of csrc/small_linear.hip and ran as addmm forward (14 us each for
// 3 MFLOP) and mm + mm + sum backward: 48 library launches, 0.63 ms per step.  Same structure as csrc/small_linear.hip with fp32
// operands end to end (exact fp32 products, fp32 accumulation: v_mfma_f32_32x32x2_f32):
//   forward :  y[r, co]  = sum_ci x[r, ci] w[co, ci] + b[co]
//   backward:  gx[r, ci] = sum_co gy[r, co] w[co, ci];  gw[co, ci] = sum_r gy[r, co] x[r, ci];  gb[co] = sum_r gy[r, co]   (ONE launch)
// 64 x 64 x 64 tiles, 4 waves (2 x 2 blocks of 32 x 32); operands whose reduction axis is not contiguous in memory are transposed
// while they are staged into LDS.  Cin must be a multiple of 64 (else -2000: the caller keeps the library path).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int T = 64, LROW = T + 1, NT = 256, PF = 4;

struct Seg { float4 v[4]; };

// 16 consecutive floats of row r0 + (tid >> 2) starting at column c0 + (tid & 3) * 16 of a row-major [rows, cols] matrix, zeros outside
__device__ __forceinline__ Seg load_seg(const float* __restrict__ src, long long ld, int r0, int c0, int rows, int cols) {
  const int row = r0 + (threadIdx.x >> 2), col = c0 + (threadIdx.x & 3) * 16;
  Seg s;
  if (row < rows && col + 16 <= cols && ((ld | col) & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    const float4* q = reinterpret_cast<const float4*>(src + row * ld + col);
#pragma unroll
    for (int u = 0; u < 4; ++u) s.v[u] = q[u];
  } else {
    float t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = (row < rows && col + u < cols) ? src[row * ld + col + u] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) s.v[u] = make_float4(t[4 * u], t[4 * u + 1], t[4 * u + 2], t[4 * u + 3]);
  }
  return s;
}

__device__ __forceinline__ void commit(float* dst, const Seg& s, bool transposed) {
  const int row = threadIdx.x >> 2, seg = threadIdx.x & 3;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float t[4] = {s.v[u].x, s.v[u].y, s.v[u].z, s.v[u].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = seg * 16 + u * 4 + e;
      if (transposed) dst[col * LROW + row] = t[e];
      else dst[row * LROW + col] = t[e];
    }
  }
}

// one K step (64) of the 64 x 64 tile product; As / Bs are [tile row (m or n)][k]
__device__ __forceinline__ f32x16 tile_mma(const float* As, const float* Bs, f32x16 acc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave & 1, wn = wave >> 1, fr = lane & 31, fh = lane >> 5;
#pragma unroll 8
  for (int kk = 0; kk < T / 2; ++kk)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(wm * 32 + fr) * LROW + kk * 2 + fh], Bs[(wn * 32 + fr) * LROW + kk * 2 + fh], acc, 0, 0, 0);
  return acc;
}

template <typename LA, typename LB>
__device__ __forceinline__ f32x16 k_loop(int nk, float* As, float* Bs, bool ta, bool tb, LA load_a, LB load_b, f32x16 acc) {
  for (int kb = 0; kb < nk; kb += PF) {
    Seg sa[PF], sb[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j)
      if (kb + j < nk) { sa[j] = load_a((kb + j) * T); sb[j] = load_b((kb + j) * T); }
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      if (kb + j < nk) {
        commit(As, sa[j], ta);
        commit(Bs, sb[j], tb);
        __syncthreads();
        acc = tile_mma(As, Bs, acc);
        __syncthreads();
      }
    }
  }
  return acc;
}

__device__ __forceinline__ void store_tile(float* C, long long ld, int m0, int n0, int M, int N, f32x16 acc, const float* bias) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave & 1, wn = wave >> 1;
  const int col = n0 + wn * 32 + (lane & 31);
  if (col >= N) return;
  const float bv = bias ? bias[col] : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    if (row < M) C[row * ld + col] = acc[i] + bv;
  }
}

__global__ __launch_bounds__(NT) void sl32_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, int R, int Cin,
                                               int Cout, float* __restrict__ y) {
  __shared__ float As[T * LROW], Bs[T * LROW];
  const int n0 = blockIdx.x * T, m0 = blockIdx.y * T;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = k_loop(Cin / T, As, Bs, false, false, [&](int k0) { return load_seg(x, Cin, m0, k0, R, Cin); },
               [&](int k0) { return load_seg(w, Cin, n0, k0, Cout, Cin); }, acc);
  store_tile(y, Cout, m0, n0, R, Cout, acc, b);
}

// blocks [0, n_dx): gx tiles (M = R, N = Cin, K = Cout);  then gw tiles (M = Cout, N = Cin, K = R) + gb from the n-tile 0 column
__global__ __launch_bounds__(NT) void sl32_bwd(const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ w, int R, int Cin,
                                               int Cout, int n_dx, float* __restrict__ gx, float* __restrict__ gw, float* __restrict__ gb) {
  __shared__ float As[T * LROW], Bs[T * LROW];
  const int ntn = Cin / T;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if ((int)blockIdx.x < n_dx) {
    const int m0 = (blockIdx.x / ntn) * T, n0 = (blockIdx.x % ntn) * T;
    // A[m = r][k = co] = gy;  B[n = ci][k = co] = w[co][ci]
    acc = k_loop((Cout + T - 1) / T, As, Bs, false, true, [&](int k0) { return load_seg(gy, Cout, m0, k0, R, Cout); },
                 [&](int k0) { return load_seg(w, Cin, k0, n0, Cout, Cin); }, acc);
    store_tile(gx, Cin, m0, n0, R, Cin, acc, nullptr);
  } else {
    const int t = blockIdx.x - n_dx;
    const int m0 = (t / ntn) * T, n0 = (t % ntn) * T;                   // m = co, n = ci
    // A[m = co][k = r] = gy[r][co];  B[n = ci][k = r] = x[r][ci]
    acc = k_loop((R + T - 1) / T, As, Bs, true, true, [&](int k0) { return load_seg(gy, Cout, k0, m0, R, Cout); },
                 [&](int k0) { return load_seg(x, Cin, k0, n0, R, Cin); }, acc);
    store_tile(gw, Cin, m0, n0, Cout, Cin, acc, nullptr);
    if (gb && (t % ntn) == 0) {                                         // 64 columns x 4 row phases, fixed summation order
      __syncthreads();
      float* red = As;
      const int cq = threadIdx.x & 63, rq = threadIdx.x >> 6, co = m0 + cq;
      float s = 0.f;
      if (co < Cout)
        for (int r = rq; r < R; r += 4) s += gy[(long long)r * Cout + co];
      red[rq * 64 + cq] = s;
      __syncthreads();
      if (threadIdx.x < T && co < Cout) gb[co] = (red[cq] + red[64 + cq]) + (red[128 + cq] + red[192 + cq]);
    }
  }
}

inline int status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace

extern "C" {

int ocpg_small_linear_f32_fwd(const float* x, const float* w, const float* b, int R, int Cin, int Cout, float* y, void* stream) {
  if (R < 0 || Cin <= 0 || Cout <= 0) return -1004;
  if (Cin % T != 0 || R > 4096) return -2000;
  if (R == 0) return 0;
  if (!x) return -1001;
  if (!w) return -1002;
  if (!y) return -1007;
  sl32_fwd<<<dim3((Cout + T - 1) / T, (R + T - 1) / T), NT, 0, (hipStream_t)stream>>>(x, w, b, R, Cin, Cout, y);
  return status();
}

int ocpg_small_linear_f32_bwd(const float* gy, const float* x, const float* w, int R, int Cin, int Cout, float* gx, float* gw, float* gb, void* stream) {
  if (R < 0 || Cin <= 0 || Cout <= 0) return -1004;
  if (Cin % T != 0 || R > 4096) return -2000;
  if (!gy) return -1001;
  if (!x) return -1002;
  if (!w) return -1003;
  if (!gw) return -1008;
  const int n_dx = gx ? ((R + T - 1) / T) * (Cin / T) : 0;
  const int n_dw = ((Cout + T - 1) / T) * (Cin / T);
  sl32_bwd<<<n_dx + n_dw, NT, 0, (hipStream_t)stream>>>(gy, x, w, R, Cin, Cout, n_dx, gx, gw, gb);
  return status();
}

}  // extern "C"
