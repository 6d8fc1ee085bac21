// Linear layers over FEW rows (the decoder: 50 query rows; heads, controller, LFM coefficient MLPs: 10..200 rows) in one launch
// forward and ONE launch backward.
//
// Reference: every nn.Linear of the decoder layers, the box / class heads and the controller (models/deformable_transformer.py:
// 313-336, models/ocpg.py:83-110,325-349).  Under autocast each of them is a cast of the input + addmm forward and two mm + a bias
// reduction + a cast of the input gradient backward: 6 launches whose work is a few MFLOP -- on MI355X a graph kernel node costs
// ~5 us of GPU timeline even when empty, so the step pays the launches, not the math (~50 such layers per step).
//   forward :  y[r, co]  = sum_ci x[r, ci] w[co, ci] + b[co]                      (x fp32 or bf16, w / b / y bf16)
//   backward:  gx[r, ci] = sum_co gy[r, co] w[co, ci]                             (gx in x's dtype)
//              gw[co, ci] = sum_r gy[r, co] x[r, ci],   gb[co] = sum_r gy[r, co]   (bf16, like autograd's gradients of bf16 copies)
// 64 x 64 x 64 tiles of v_mfma_f32_32x32x16_bf16, 4 waves (2 x 2); operands whose reduction axis is not contiguous in memory
// (w for gx; gy and x for gw) are transposed while they are staged into LDS.  Cin must be a multiple of 64; anything else returns
// -2000 and the caller keeps the library path.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int T = 64, LROW = T + 8, NT = 256;      // tile edge, LDS row (144 B), threads

__device__ __forceinline__ short f2b(float v) { return (short)__bfloat16_as_ushort(__float2bfloat16(v)); }
__device__ __forceinline__ float b2f(short b) { return __uint_as_float(((unsigned)(unsigned short)b) << 16); }

// element (row, col) of a [rows, cols] matrix stored as fp32 (is_f32) or bf16, 0 outside
__device__ __forceinline__ short ld_el(const void* p, int is_f32, long long ld, int row, int col, int rows, int cols) {
  if (row >= rows || col >= cols) return 0;
  return is_f32 ? f2b(reinterpret_cast<const float*>(p)[row * ld + col]) : reinterpret_cast<const short*>(p)[row * ld + col];
}

// 16 consecutive elements of row `row` starting at column `col` (fp32 or bf16 storage) as bf16 bits, zeros outside the matrix;
// 16-byte loads when the run is inside and aligned
__device__ __forceinline__ void ld16_raw(short (&v)[16], const void* p, int is_f32, long long ld, int row, int col, int rows, int cols) {
  if (row < rows && col + 16 <= cols && ((ld | col) & 7) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {     // (a view with a storage offset may be unaligned)
    if (is_f32) {
      const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + row * ld + col);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 f = q[u];
        v[4 * u] = f2b(f.x); v[4 * u + 1] = f2b(f.y); v[4 * u + 2] = f2b(f.z); v[4 * u + 3] = f2b(f.w);
      }
    } else {
      const bf16x8* q = reinterpret_cast<const bf16x8*>(reinterpret_cast<const short*>(p) + row * ld + col);
      const bf16x8 a = q[0], b = q[1];
#pragma unroll
      for (int u = 0; u < 8; ++u) { v[u] = a[u]; v[8 + u] = b[u]; }
    }
  } else {
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = ld_el(p, is_f32, ld, row, col + u, rows, cols);
  }
}

// the same with an optional ReLU mask: `mask` = the bf16 output y of a fused ReLU (same shape as p): gy is zeroed where y <= 0
__device__ __forceinline__ void ld16(short (&v)[16], const void* p, int is_f32, long long ld, int row, int col, int rows, int cols,
                                     const short* mask = nullptr) {
  ld16_raw(v, p, is_f32, ld, row, col, rows, cols);
  if (mask) {
    short m[16];
    ld16_raw(m, mask, 0, ld, row, col, rows, cols);
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = b2f(m[u]) > 0.f ? v[u] : (short)0;
  }
}

// A thread's share of one staged T x T tile: 16 consecutive elements of tile row threadIdx.x >> 2 (as bf16 bits)
struct Seg {
  bf16x8 a, b;
};

__device__ __forceinline__ Seg load_seg(const void* src, int is_f32, long long ld, int r0, int c0, int rows, int cols, const short* mask = nullptr) {
  const int row = threadIdx.x >> 2, seg = threadIdx.x & 3;
  short v[16];
  ld16(v, src, is_f32, ld, r0 + row, c0 + seg * 16, rows, cols, mask);
  Seg s;
#pragma unroll
  for (int u = 0; u < 8; ++u) { s.a[u] = v[u]; s.b[u] = v[8 + u]; }
  return s;
}

// park it as [tile row][tile col] (reduction axis = columns of the source)
__device__ __forceinline__ void commit_plain(short* dst, const Seg& s) {
  const int row = threadIdx.x >> 2, seg = threadIdx.x & 3;
  *reinterpret_cast<bf16x8*>(dst + row * LROW + seg * 16) = s.a;
  *reinterpret_cast<bf16x8*>(dst + row * LROW + seg * 16 + 8) = s.b;
}

// park it TRANSPOSED: LDS [tile col][tile row] (reduction axis = rows of the source)
__device__ __forceinline__ void commit_transposed(short* dst, const Seg& s) {
  const int row = threadIdx.x >> 2, seg = threadIdx.x & 3;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    dst[(seg * 16 + u) * LROW + row] = s.a[u];
    dst[(seg * 16 + 8 + u) * LROW + row] = s.b[u];
  }
}

// one K step (64) of the 64 x 64 tile product: wave (wm, wn) owns the 32 x 32 block
__device__ __forceinline__ f32x16 tile_mma(const short* As, const short* Bs, f32x16 acc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave & 1, wn = wave >> 1, fr = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < T / 16; ++kk) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(As + (wm * 32 + fr) * LROW + kk * 16 + fh * 8);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bs + (wn * 32 + fr) * LROW + kk * 16 + fh * 8);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  return acc;
}

// The K loop: the global loads of PF consecutive K steps are issued back to back, then the steps are parked and multiplied one by
// one -- these launches are a chain of (load latency -> barrier -> 4 MFMAs) per step and nothing else runs on the CU, so the latency
// is paid once per PF steps instead of once per step (Cin = 256: 1 instead of 4; the FFN's 1024: 4 instead of 16).
constexpr int PF = 4;
template <bool TA, bool TB, typename LA, typename LB>
__device__ __forceinline__ f32x16 k_loop(int nk, short* As, short* Bs, LA load_a, LB load_b, f32x16 acc) {
  for (int kb = 0; kb < nk; kb += PF) {
    Seg sa[PF], sb[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j)
      if (kb + j < nk) { sa[j] = load_a((kb + j) * T); sb[j] = load_b((kb + j) * T); }
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      if (kb + j < nk) {
        if (TA) commit_transposed(As, sa[j]); else commit_plain(As, sa[j]);
        if (TB) commit_transposed(Bs, sb[j]); else commit_plain(Bs, sb[j]);
        __syncthreads();
        acc = tile_mma(As, Bs, acc);
        __syncthreads();
      }
    }
  }
  return acc;
}

// store the wave's 32 x 32 block of C[m0.., n0..] (row-major [M, N], fp32 or bf16), + optional per-column bias
__device__ __forceinline__ void store_tile(void* C, int out_f32, long long ld, int m0, int n0, int M, int N, f32x16 acc, const short* bias,
                                           int relu = 0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave & 1, wn = wave >> 1;
  const int col = n0 + wn * 32 + (lane & 31);
  if (col >= N) return;
  const float bv = bias ? b2f(bias[col]) : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    if (row < M) {
      float v = acc[i] + bv;
      if (relu) v = fmaxf(v, 0.f);
      if (out_f32) reinterpret_cast<float*>(C)[row * ld + col] = v;
      else reinterpret_cast<short*>(C)[row * ld + col] = f2b(v);
    }
  }
}

__global__ __launch_bounds__(NT) void sl_fwd(const void* __restrict__ x, int x_f32, const short* __restrict__ w, const short* __restrict__ b,
                                             int R, int Cin, int Cout, int relu, short* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) short As[T * LROW], Bs[T * LROW];
  const int n0 = blockIdx.x * T, m0 = blockIdx.y * T;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = k_loop<false, false>((Cin + T - 1) / T, As, Bs, [&](int k0) { return load_seg(x, x_f32, Cin, m0, k0, R, Cin); },
                             [&](int k0) { return load_seg(w, 0, Cin, n0, k0, Cout, Cin); }, acc);
  store_tile(y, 0, Cout, m0, n0, R, Cout, acc, b, relu);
}

// blocks [0, n_dx): gx tiles (M = R, N = Cin, K = Cout);  blocks [n_dx, ..): gw tiles (M = Cout, N = Cin, K = R) + gb from the n-tile 0 column
__global__ __launch_bounds__(NT) void sl_bwd(const void* __restrict__ gy, int gy_f32, const void* __restrict__ x, int x_f32,
                                             const short* __restrict__ w, const short* __restrict__ ymask, int R, int Cin, int Cout, int n_dx,
                                             int need_gx, void* __restrict__ gx, short* __restrict__ gw, short* __restrict__ gb) {
  __shared__ __attribute__((aligned(16))) short As[T * LROW], Bs[T * LROW];
  const int ntn = (Cin + T - 1) / T;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if ((int)blockIdx.x < n_dx) {
    if (!need_gx) return;
    const int m0 = (blockIdx.x / ntn) * T, n0 = (blockIdx.x % ntn) * T;
    // A[m = r][k = co] = gy;  B[n = ci][k = co] = w[co][ci]
    acc = k_loop<false, true>((Cout + T - 1) / T, As, Bs, [&](int k0) { return load_seg(gy, gy_f32, Cout, m0, k0, R, Cout, ymask); },
                              [&](int k0) { return load_seg(w, 0, Cin, k0, n0, Cout, Cin); }, acc);
    store_tile(gx, x_f32, Cin, m0, n0, R, Cin, acc, nullptr);
  } else {
    const int t = blockIdx.x - n_dx;
    const int m0 = (t / ntn) * T, n0 = (t % ntn) * T;                   // m = co, n = ci
    const bool do_bias = gb && (t % ntn) == 0;
    // A[m = co][k = r] = gy[r][co];  B[n = ci][k = r] = x[r][ci]
    acc = k_loop<true, true>((R + T - 1) / T, As, Bs, [&](int k0) { return load_seg(gy, gy_f32, Cout, k0, m0, R, Cout, ymask); },
                             [&](int k0) { return load_seg(x, x_f32, Cin, k0, n0, R, Cin); }, acc);
    store_tile(gw, 0, Cin, m0, n0, Cout, Cin, acc, nullptr);
    if (do_bias) {
      // bias gradient = column sum of gy in fp32 from the ORIGINAL values (ATen reduces the fp32 gradient before the cast; summing
      // the bf16-rounded staged tile lost precision for heads with many rows -- ADVICE r2).  All 256 threads: 64 columns x 4 row
      // phases, 8 independent loads in flight per thread (a serial loop by 64 threads measured 110 us at 600 rows), LDS reduce.
      __syncthreads();
      float* red = reinterpret_cast<float*>(As);                        // [4][64] partial sums
      const int cq = threadIdx.x & 63, rq = threadIdx.x >> 6;
      const int co = m0 + cq;
      float bsum = 0.f;
      if (co < Cout) {
        for (int r0 = rq; r0 < R; r0 += 32) {
          float g[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int r = r0 + 4 * u;
            const long long i = (long long)min(r, R - 1) * Cout + co;
            float v = gy_f32 ? reinterpret_cast<const float*>(gy)[i] : b2f(reinterpret_cast<const short*>(gy)[i]);
            if (ymask && !(b2f(ymask[i]) > 0.f)) v = 0.f;
            g[u] = r < R ? v : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) bsum += g[u];
        }
      }
      red[rq * 64 + cq] = bsum;
      __syncthreads();
      if (threadIdx.x < T && co < Cout) gb[co] = f2b(red[cq] + red[64 + cq] + red[128 + cq] + red[192 + cq]);
    }
  }
}

inline int status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace

extern "C" {

/* y [R, Cout] bf16 = act(x [R, Cin] (fp32: x_f32 != 0, else bf16) . w[Cout, Cin]^T (bf16) + b [Cout] (bf16 or NULL)), act = ReLU when
 * relu != 0.  -2000: not served. */
int ocpg_small_linear_fwd(const void* x, int x_f32, const void* w, const void* b, int R, int Cin, int Cout, int relu, void* y, void* stream) {
  if (R < 0 || Cin <= 0 || Cout <= 0) return -1005;
  if (Cin % T != 0 || R > 4096) return -2000;
  if (R == 0) return 0;
  if (!x) return -1001;
  if (!w) return -1003;
  if (!y) return -1008;
  sl_fwd<<<dim3((Cout + T - 1) / T, (R + T - 1) / T), NT, 0, (hipStream_t)stream>>>(x, x_f32, (const short*)w, (const short*)b, R, Cin, Cout,
                                                                                    relu, (short*)y);
  return status();
}

/* gx [R, Cin] (x's dtype; NULL: not needed), gw [Cout, Cin] bf16, gb [Cout] bf16 (NULL: no bias) from gy [R, Cout] (fp32 / bf16);
 * y_relu = the forward's output when it applied the ReLU (gy is masked where y <= 0), else NULL. */
int ocpg_small_linear_bwd(const void* gy, int gy_f32, const void* x, int x_f32, const void* w, const void* y_relu, int R, int Cin, int Cout,
                          void* gx, void* gw, void* gb, void* stream) {
  if (R < 0 || Cin <= 0 || Cout <= 0) return -1006;
  if (Cin % T != 0 || R > 4096) return -2000;
  if (!gy) return -1001;
  if (!x) return -1003;
  if (!w) return -1005;
  if (!gw) return -1010;
  const int n_dx = gx ? ((R + T - 1) / T) * ((Cin + T - 1) / T) : 0;
  const int n_dw = ((Cout + T - 1) / T) * ((Cin + T - 1) / T);
  sl_bwd<<<n_dx + n_dw, NT, 0, (hipStream_t)stream>>>(gy, gy_f32, x, x_f32, (const short*)w, (const short*)y_relu, R, Cin, Cout, n_dx,
                                                      gx != nullptr, gx, (short*)gw, (short*)gb);
  return status();
}

}  // extern "C"
