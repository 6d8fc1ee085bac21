// LayerNorm over the last axis for token matrices, low-precision in / out (Video-Swin blocks: norm1, norm2, PatchMerging.norm --
// models/video_swin_transformer.py:194,201,225).  Under autocast torch runs layer_norm in fp32: a cast of the input, the fp32
// kernel, and a cast of the output at the next Linear (and the mirror image backward) -- three passes over [tokens, C] each way.
// Here one pass each way: x in fp32 or bf16, statistics and arithmetic in fp32, y written in the dtype the next Linear consumes.
//   fwd: y = (x - mean) * rstd * gamma + beta;  mean / rstd [rows] fp32 kept for the backward
//   bwd: dx = rstd * (g gamma - mean_c(g gamma) - xhat * mean_c(g gamma xhat));  dgamma / dbeta through per-workgroup partial rows
// One wave per row (2 / 4 / 8 / 16 contiguous channels per lane, vector loads, up to four rows in flight), C <= 1024; HBM-bound.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

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

// ---- a lane's contiguous chunk of E elements (c = lane * E .. + E - 1): vector loads / stores in the storage type ----------------
template <int E> __device__ __forceinline__ void ld_chunk(const void* p, int code, long long i, float (&v)[E]) {
  if (code == 1) {
    const float* q = reinterpret_cast<const float*>(p) + i;
    if constexpr (E >= 4) {
#pragma unroll
      for (int k = 0; k < E / 4; ++k) {
        const float4 t = reinterpret_cast<const float4*>(q)[k];
        v[4 * k] = t.x, v[4 * k + 1] = t.y, v[4 * k + 2] = t.z, v[4 * k + 3] = t.w;
      }
    } else {
      const float2 t = *reinterpret_cast<const float2*>(q);
      v[0] = t.x, v[1] = t.y;
    }
    return;
  }
  const unsigned short* q = reinterpret_cast<const unsigned short*>(p) + i;
  unsigned w[E / 2];
  if constexpr (E >= 8) {
#pragma unroll
    for (int k = 0; k < E / 8; ++k) {
      const uint4 t = reinterpret_cast<const uint4*>(q)[k];
      w[4 * k] = t.x, w[4 * k + 1] = t.y, w[4 * k + 2] = t.z, w[4 * k + 3] = t.w;
    }
  } else if constexpr (E == 4) {
    const uint2 t = *reinterpret_cast<const uint2*>(q);
    w[0] = t.x, w[1] = t.y;
  } else {
    w[0] = *reinterpret_cast<const unsigned*>(q);
  }
#pragma unroll
  for (int k = 0; k < E / 2; ++k) {
    const unsigned short lo = (unsigned short)(w[k] & 0xffffu), hi = (unsigned short)(w[k] >> 16);
    if (code == 2) {
      v[2 * k] = __half2float(__ushort_as_half(lo));
      v[2 * k + 1] = __half2float(__ushort_as_half(hi));
    } else {
      v[2 * k] = __uint_as_float((unsigned)lo << 16);
      v[2 * k + 1] = __uint_as_float((unsigned)hi << 16);
    }
  }
}

template <int E> __device__ __forceinline__ void st_chunk(void* p, int code, long long i, const float (&v)[E]) {
  if (code == 1) {
    float* q = reinterpret_cast<float*>(p) + i;
    if constexpr (E >= 4) {
#pragma unroll
      for (int k = 0; k < E / 4; ++k) reinterpret_cast<float4*>(q)[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    } else {
      *reinterpret_cast<float2*>(q) = make_float2(v[0], v[1]);
    }
    return;
  }
  unsigned w[E / 2];
#pragma unroll
  for (int k = 0; k < E / 2; ++k) {
    unsigned short lo, hi;
    if (code == 2) {
      lo = __half_as_ushort(__float2half(v[2 * k]));
      hi = __half_as_ushort(__float2half(v[2 * k + 1]));
    } else {
      lo = __bfloat16_as_ushort(__float2bfloat16(v[2 * k]));
      hi = __bfloat16_as_ushort(__float2bfloat16(v[2 * k + 1]));
    }
    w[k] = (unsigned)lo | ((unsigned)hi << 16);
  }
  unsigned short* q = reinterpret_cast<unsigned short*>(p) + i;
  if constexpr (E >= 8) {
#pragma unroll
    for (int k = 0; k < E / 8; ++k) reinterpret_cast<uint4*>(q)[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
  } else if constexpr (E == 4) {
    *reinterpret_cast<uint2*>(q) = make_uint2(w[0], w[1]);
  } else {
    *reinterpret_cast<unsigned*>(q) = w[0];
  }
}

// One wave per row, a lane owns the E consecutive channels lane * E .. (C % E == 0, C <= 64 E), R rows in flight per wave: all loads of
// the R rows are issued before the first reduction (a wave that took one row at a time with 2-byte loads ran at 0.9 TB/s on the
// [102 720, 128] fp16 maps of Swin-B's first stage: each row was a load -> 12 shuffles -> store latency chain).
template <int E, int R>
__global__ __launch_bounds__(NT) void ln_fwd(const void* __restrict__ x, int x_f32, const float* __restrict__ gamma, const float* __restrict__ beta,
                                             long long rows, int C, float eps, void* __restrict__ y, int y_f32, float* __restrict__ mean,
                                             float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = lane * E;
  const bool act = c0 < C;
  float gm[E], bt[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { gm[e] = act ? gamma[c0 + e] : 0.f; bt[e] = act ? beta[c0 + e] : 0.f; }
  const float inv = 1.f / (float)C;
  for (long long rb = ((long long)blockIdx.x * (NT / 64) + wave) * R; rb < rows; rb += (long long)gridDim.x * (NT / 64) * R) {
    float v[R][E];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long row = rb + r < rows ? rb + r : rows - 1;
      if (act) ld_chunk<E>(x, x_f32, row * C + c0, v[r]);
      else
#pragma unroll
        for (int e = 0; e < E; ++e) v[r][e] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) s += v[r][e];
      const float mu = wave_sum(s) * inv;
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float d = act ? v[r][e] - mu : 0.f;
        q += d * d;
      }
      const float rs = rsqrtf(wave_sum(q) * inv + eps);
      if (rb + r < rows) {
        if (act) {
          float o[E];
#pragma unroll
          for (int e = 0; e < E; ++e) o[e] = (v[r][e] - mu) * rs * gm[e] + bt[e];
          st_chunk<E>(y, y_f32, (rb + r) * C + c0, o);
        }
        if (lane == 0) { mean[rb + r] = mu; rstd[rb + r] = rs; }
      }
    }
  }
}

// part_g / part_b [gridDim.x, C]: per-workgroup partial sums of dgamma / dbeta
template <int E, int R>
__global__ __launch_bounds__(NT) void ln_bwd(const void* __restrict__ gy, int gy_f32, const void* __restrict__ x, int x_f32,
                                             const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                                             long long rows, int C, void* __restrict__ dx, int dx_f32, float* __restrict__ part_g,
                                             float* __restrict__ part_b) {
  __shared__ float red[2][NT / 64][64 * E];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = lane * E;
  const bool act = c0 < C;
  float ag[E], ab[E], gm[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    ag[e] = 0.f; ab[e] = 0.f;
    gm[e] = act ? gamma[c0 + e] : 0.f;
  }
  const float inv = 1.f / (float)C;
  for (long long rb = ((long long)blockIdx.x * (NT / 64) + wave) * R; rb < rows; rb += (long long)gridDim.x * (NT / 64) * R) {
    float g[R][E], xh[R][E], mu[R], rs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool ok = rb + r < rows;
      const long long row = ok ? rb + r : rows - 1;
      mu[r] = mean[row];
      rs[r] = rstd[row];
      if (act && ok) {
        ld_chunk<E>(gy, gy_f32, row * C + c0, g[r]);
        ld_chunk<E>(x, x_f32, row * C + c0, xh[r]);
      } else {
#pragma unroll
        for (int e = 0; e < E; ++e) { g[r][e] = 0.f; xh[r][e] = mu[r]; }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        xh[r][e] = (xh[r][e] - mu[r]) * rs[r];
        const float gg = g[r][e] * gm[e];
        s1 += gg;
        s2 += gg * xh[r][e];
        ag[e] += g[r][e] * xh[r][e];
        ab[e] += g[r][e];
      }
      s1 = wave_sum(s1) * inv;
      s2 = wave_sum(s2) * inv;
      if (act && rb + r < rows) {
        float o[E];
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = rs[r] * (g[r][e] * gm[e] - s1 - xh[r][e] * s2);
        st_chunk<E>(dx, dx_f32, (rb + r) * C + c0, o);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    red[0][wave][c0 + e] = ag[e];
    red[1][wave][c0 + e] = ab[e];
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

// elements per lane for C channels: the smallest of 2 / 4 / 8 / 16 with 64 E >= C and C % E == 0 (0: not served)
inline int chunk_for(int C) {
  for (int e = 2; e <= MAXE; e *= 2)
    if (64 * e >= C && C % e == 0) return e;
  return 0;
}

inline int status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace

extern "C" {

int ocpg_layernorm_blocks(long long rows) {
  // (the backward leaves one partial row of dgamma / dbeta per workgroup for the caller to sum: the cap is also the height of that sum)
  static const long long cap = [] { const char* e = getenv("OCPG_LN_BLOCKS"); const long long v = e && *e ? atoll(e) : 1024; return v < 1 ? 1 : v; }();
  const long long b = (rows + NT / 64 - 1) / (NT / 64);
  return (int)(b < 1 ? 1 : b > cap ? cap : b);
}

int ocpg_layernorm_fwd(const void* x, int x_f32, const float* gamma, const float* beta, long long rows, int C, float eps, void* y, int y_f32,
                       float* mean, float* rstd, void* stream) {
  if (rows < 0 || C <= 0) return -1005;
  const int e = chunk_for(C);
  if (C > 64 * MAXE || !e) return -2000;
  if (rows == 0) return 0;
  if (!x || !gamma || !beta) return -1001;
  if (!y || !mean || !rstd) return -1008;
  const unsigned nb = (unsigned)ocpg_layernorm_blocks(rows);
  hipStream_t st = (hipStream_t)stream;
  if (e == 2) ln_fwd<2, 4><<<nb, NT, 0, st>>>(x, x_f32, gamma, beta, rows, C, eps, y, y_f32, mean, rstd);
  else if (e == 4) ln_fwd<4, 4><<<nb, NT, 0, st>>>(x, x_f32, gamma, beta, rows, C, eps, y, y_f32, mean, rstd);
  else if (e == 8) ln_fwd<8, 2><<<nb, NT, 0, st>>>(x, x_f32, gamma, beta, rows, C, eps, y, y_f32, mean, rstd);
  else ln_fwd<16, 1><<<nb, NT, 0, st>>>(x, x_f32, gamma, beta, rows, C, eps, y, y_f32, mean, rstd);
  return status();
}

int ocpg_layernorm_bwd(const void* gy, int gy_f32, const void* x, int x_f32, const float* gamma, const float* mean, const float* rstd,
                       long long rows, int C, void* dx, int dx_f32, float* part_g, float* part_b, void* stream) {
  if (rows <= 0 || C <= 0) return -1008;
  const int e = chunk_for(C);
  if (C > 64 * MAXE || !e) return -2000;
  if (!gy || !x || !gamma || !mean || !rstd) return -1001;
  if (!dx || !part_g || !part_b) return -1010;
  const unsigned nb = (unsigned)ocpg_layernorm_blocks(rows);
  hipStream_t st = (hipStream_t)stream;
  if (e == 2) ln_bwd<2, 4><<<nb, NT, 0, st>>>(gy, gy_f32, x, x_f32, gamma, mean, rstd, rows, C, dx, dx_f32, part_g, part_b);
  else if (e == 4) ln_bwd<4, 4><<<nb, NT, 0, st>>>(gy, gy_f32, x, x_f32, gamma, mean, rstd, rows, C, dx, dx_f32, part_g, part_b);
  else if (e == 8) ln_bwd<8, 2><<<nb, NT, 0, st>>>(gy, gy_f32, x, x_f32, gamma, mean, rstd, rows, C, dx, dx_f32, part_g, part_b);
  else ln_bwd<16, 1><<<nb, NT, 0, st>>>(gy, gy_f32, x, x_f32, gamma, mean, rstd, rows, C, dx, dx_f32, part_g, part_b);
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
  // a segment is ~40 rows, each a dependent pair (order[p] -> g[...]): eight pairs in flight instead of one (98 -> the latency of five trips)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long long p = seg[t];
  const long long pe = seg[t + 1];
  for (; p + 8 <= pe; p += 8) {
    long long o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = order[p + k];
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = g[o[k] * H + h];
    s0 += v[0] + v[4], s1 += v[1] + v[5], s2 += v[2] + v[6], s3 += v[3] + v[7];
  }
  for (; p < pe; ++p) s0 += g[order[p] * H + h];
  out[i] = (s0 + s1) + (s2 + s3);
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

// ---- Video-Swin's relative-position bias (models/video_swin_transformer.py:151-153: table[index[:N, :N]].reshape(N, N, H).permute(2, 0, 1))
// in the two layouts the attention kernels read, from the table directly (round 4): per block the chain was a gather to [N N, H], a
// permuting copy to [H, N, N], a second copy for the transpose (and a transposing copy of the gradient on the way back): 100-200 us per
// block at N = 392, 2.4 ms per config-#5 step.
//   bias[h][a][b] = table[idx[a][b]][h]      bias_t[h][a][b] = table[idx[b][a]][h] = bias[h][b][a]          (all writes coalesced along b)
namespace {

__global__ __launch_bounds__(256) void relpos_bias(const float* __restrict__ table, const long long* __restrict__ idx, int N, long long ldi, int H,
                                                   float* __restrict__ bias, float* __restrict__ bias_t) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int P = N * N;
  if (q >= P) return;
  const int a = q / N, b = q - a * N;
  const float* r1 = table + idx[a * ldi + b] * H;
  const float* r2 = table + idx[b * ldi + a] * H;
  for (int h = 0; h < H; ++h) {
    bias[(long long)h * P + q] = r1[h];
    bias_t[(long long)h * P + q] = r2[h];
  }
}

// dtable[t][h] = sum over the positions p = a N + b with idx[a][b] == t of g[h][pos(p)]: g is the gradient of `bias` stored [H][N N]
// (pos(p) = p), or -- transposed != 0 -- of its transpose, i.e. the attention backward's own dS sum (pos(p) = b N + a: no transposing copy)
__global__ __launch_bounds__(256) void relpos_bias_bwd(const float* __restrict__ g, const long long* __restrict__ order, const long long* __restrict__ seg,
                                                       int T, int H, int N, int transposed, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * H) return;
  const int t = i / H, h = i - t * H;
  const float* gh = g + (long long)h * N * N;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long long p = seg[t];
  const long long pe = seg[t + 1];
  auto pos = [&](long long o) { return transposed ? (o % N) * N + o / N : o; };
  for (; p + 8 <= pe; p += 8) {
    long long o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = order[p + k];
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = gh[pos(o[k])];
    s0 += v[0] + v[4], s1 += v[1] + v[5], s2 += v[2] + v[6], s3 += v[3] + v[7];
  }
  for (; p < pe; ++p) s0 += gh[pos(order[p])];
  out[i] = (s0 + s1) + (s2 + s3);
}

}  // namespace

extern "C" int ocpg_relpos_bias_fwd(const float* table, const long long* idx, int N, long long ldi, int H, float* bias, float* bias_t, void* stream) {
  if (N <= 0 || H <= 0 || ldi < N) return -1003;
  if (!table || !idx) return -1001;
  if (!bias || !bias_t) return -1006;
  relpos_bias<<<(N * N + 255) / 256, 256, 0, (hipStream_t)stream>>>(table, idx, N, ldi, H, bias, bias_t);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ocpg_relpos_bias_bwd(const float* g, const long long* order, const long long* seg, int T, int H, int N, int transposed, float* out,
                                    void* stream) {
  if (T <= 0 || H <= 0 || N <= 0) return -1004;
  if (!g || !order || !seg) return -1001;
  if (!out) return -1008;
  relpos_bias_bwd<<<(T * H + 255) / 256, 256, 0, (hipStream_t)stream>>>(g, order, seg, T, H, N, transposed, out);
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
