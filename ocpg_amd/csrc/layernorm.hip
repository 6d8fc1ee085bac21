// LayerNorm over the last axis for token matrices, low-precision in / out (Video-Swin blocks: norm1, norm2, PatchMerging.norm --
// models/video_swin_transformer.py:194,201,225).  Under autocast torch runs layer_norm in fp32: a cast of the input, the fp32
// kernel, and a cast of the output at the next Linear (and the mirror image backward) -- three passes over [tokens, C] each way.
// Here one pass each way: x in fp32 or bf16, statistics and arithmetic in fp32, y written in the dtype the next Linear consumes.
//   fwd: y = (x - mean) * rstd * gamma + beta;  mean / rstd [rows] fp32 kept for the backward
//   bwd: dx = rstd * (g gamma - mean_c(g gamma) - xhat * mean_c(g gamma xhat));  dgamma / dbeta through per-workgroup partial rows
// One wave per row, C <= 1024 (16 elements per lane); HBM-bound.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"

namespace {

constexpr int NT = 256, MAXE = 16;

// storage codes (the *_f32 arguments of the entry points): 1 fp32, 0 bf16, 2 fp16
__device__ __forceinline__ float ldv(const void* p, int code, long long i) {
  if (code == 1) return reinterpret_cast<const float*>(p)[i];
  if (code == 2) return __half2float(reinterpret_cast<const __half*>(p)[i]);
  return __bfloat162float(reinterpret_cast<const __hip_bfloat16*>(p)[i]);
}
__device__ __forceinline__ void stv(void* p, int code, long long i, float v) {
  if (code == 1) reinterpret_cast<float*>(p)[i] = v;
  else if (code == 2) reinterpret_cast<__half*>(p)[i] = __float2half(v);
  else reinterpret_cast<__hip_bfloat16*>(p)[i] = __float2bfloat16(v);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(NT) void ln_fwd(const void* __restrict__ x, int x_f32, const float* __restrict__ gamma, const float* __restrict__ beta,
                                             long long rows, int C, float eps, void* __restrict__ y, int y_f32, float* __restrict__ mean,
                                             float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long long row = (long long)blockIdx.x * (NT / 64) + wave; row < rows; row += (long long)gridDim.x * (NT / 64)) {
    float v[MAXE];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int c = lane + 64 * e;
      v[e] = c < C ? ldv(x, x_f32, row * C + c) : 0.f;
      s += v[e];
    }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int c = lane + 64 * e;
      const float d = c < C ? v[e] - mu : 0.f;
      q += d * d;
    }
    const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int c = lane + 64 * e;
      if (c < C) stv(y, y_f32, row * C + c, (v[e] - mu) * rs * gamma[c] + beta[c]);
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

// part_g / part_b [gridDim.x, C]: per-workgroup partial sums of dgamma / dbeta
__global__ __launch_bounds__(NT) void ln_bwd(const void* __restrict__ gy, int gy_f32, const void* __restrict__ x, int x_f32,
                                             const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                                             long long rows, int C, void* __restrict__ dx, int dx_f32, float* __restrict__ part_g,
                                             float* __restrict__ part_b) {
  __shared__ float red[2][NT / 64][64 * MAXE];      // 2 x 4 x 1024 floats = 32 KB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[MAXE], ab[MAXE], gm[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    ag[e] = 0.f; ab[e] = 0.f;
    const int c = lane + 64 * e;
    gm[e] = c < C ? gamma[c] : 0.f;
  }
  for (long long row = (long long)blockIdx.x * (NT / 64) + wave; row < rows; row += (long long)gridDim.x * (NT / 64)) {
    const float mu = mean[row], rs = rstd[row];
    float g[MAXE], xh[MAXE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int c = lane + 64 * e;
      const bool ok = c < C;
      g[e] = ok ? ldv(gy, gy_f32, row * C + c) : 0.f;
      xh[e] = ok ? (ldv(x, x_f32, row * C + c) - mu) * rs : 0.f;
      const float gg = g[e] * gm[e];
      s1 += gg;
      s2 += gg * xh[e];
      ag[e] += g[e] * xh[e];
      ab[e] += g[e];
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int c = lane + 64 * e;
      if (c < C) stv(dx, dx_f32, row * C + c, rs * (g[e] * gm[e] - s1 - xh[e] * s2));
    }
  }
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    red[0][wave][lane + 64 * e] = ag[e];
    red[1][wave][lane + 64 * e] = ab[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += NT) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { sg += red[0][w][c]; sb += red[1][w][c]; }
    part_g[(long long)blockIdx.x * C + c] = sg;
    part_b[(long long)blockIdx.x * C + c] = sb;
  }
}

inline int status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace

extern "C" {

int ocpg_layernorm_blocks(long long rows) {
  const long long b = (rows + NT / 64 - 1) / (NT / 64);
  return (int)(b < 1 ? 1 : b > 1024 ? 1024 : b);
}

int ocpg_layernorm_fwd(const void* x, int x_f32, const float* gamma, const float* beta, long long rows, int C, float eps, void* y, int y_f32,
                       float* mean, float* rstd, void* stream) {
  if (rows < 0 || C <= 0) return -1005;
  if (C > 64 * MAXE) return -2000;
  if (rows == 0) return 0;
  if (!x || !gamma || !beta) return -1001;
  if (!y || !mean || !rstd) return -1008;
  ln_fwd<<<ocpg_layernorm_blocks(rows), NT, 0, (hipStream_t)stream>>>(x, x_f32, gamma, beta, rows, C, eps, y, y_f32, mean, rstd);
  return status();
}

int ocpg_layernorm_bwd(const void* gy, int gy_f32, const void* x, int x_f32, const float* gamma, const float* mean, const float* rstd,
                       long long rows, int C, void* dx, int dx_f32, float* part_g, float* part_b, void* stream) {
  if (rows <= 0 || C <= 0) return -1008;
  if (C > 64 * MAXE) return -2000;
  if (!gy || !x || !gamma || !mean || !rstd) return -1001;
  if (!dx || !part_g || !part_b) return -1010;
  ln_bwd<<<ocpg_layernorm_blocks(rows), NT, 0, (hipStream_t)stream>>>(gy, gy_f32, x, x_f32, gamma, mean, rstd, rows, C, dx, dx_f32, part_g, part_b);
  return status();
}

}  // extern "C"

// ---- gradient of a row gather with a STATIC index (the relative-position-bias lookup of WindowAttention3D,
// models/video_swin_transformer.py:112-114,151-153): table [T, H] -> table[idx] [M, H].  autograd's backward is
// index_put(accumulate): M = N^2 = 60 025 float atomics into 1 521 rows (~40 collisions per address, 161 us per block).  The
// index is a buffer, so the rows are sorted by destination ONCE (order [M], seg [T + 1] = CSR offsets) and the backward is a
// segmented sum: one thread per (table row, head), no atomics.
namespace {

__global__ __launch_bounds__(256) void seg_sum(const float* __restrict__ g, const long long* __restrict__ order, const long long* __restrict__ seg,
                                               int T, int H, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * H) return;
  const int t = i / H, h = i - t * H;
  float s = 0.f;
  for (long long p = seg[t]; p < seg[t + 1]; ++p) s += g[order[p] * H + h];
  out[i] = s;
}

}  // namespace

extern "C" int ocpg_gather_rows_bwd(const float* g, const long long* order, const long long* seg, int T, int H, float* out, void* stream) {
  if (T <= 0 || H <= 0) return -1004;
  if (!g || !order || !seg) return -1001;
  if (!out) return -1006;
  seg_sum<<<(T * H + 255) / 256, 256, 0, (hipStream_t)stream>>>(g, order, seg, T, H, out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ---- row gather with padding slots: out[b, j, :] = idx[j] < S ? x[b, idx[j], :] : 0, rows moved as 16-byte segments (any dtype).
// Video-Swin's pad + cyclic shift + window partition (and its reverse) is one such gather (models/video_swin_transformer.py:
// 171-199 does pad, roll, view/permute, and the mirror image); because every token sits in exactly one window slot, the backward
// of the partition gather IS the reverse gather and vice versa -- no index_add atomics, no appended zero row.
namespace {

__global__ __launch_bounds__(256) void gather_rows_pad(const uint4* __restrict__ x, const long long* __restrict__ idx, long long S, long long M,
                                                       int segs, uint4* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over M * segs of one batch element
  if (i >= M * segs) return;
  const long long j = i / segs;
  const int sg = (int)(i - j * segs);
  const long long b = blockIdx.y, src = idx[j];
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (src >= 0 && src < S) v = x[(b * S + src) * segs + sg];
  out[(b * M + j) * segs + sg] = v;
}

}  // namespace

extern "C" int ocpg_gather_rows_pad(const void* x, const long long* idx, long long B, long long S, long long M, long long row_bytes, void* out,
                                    void* stream) {
  if (B < 0 || S < 0 || M < 0 || row_bytes <= 0 || row_bytes % 16 != 0 || B > 65535) return -1005;
  if (B == 0 || M == 0) return 0;
  if (!x || !idx) return -1001;
  if (!out) return -1007;
  const int segs = (int)(row_bytes / 16);
  const dim3 grid((unsigned)((M * segs + 255) / 256), (unsigned)B);
  gather_rows_pad<<<grid, 256, 0, (hipStream_t)stream>>>((const uint4*)x, idx, S, M, segs, (uint4*)out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
