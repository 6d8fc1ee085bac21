// 3-D (shifted-)window attention for Video-Swin on the matrix cores (bf16 / fp16 storage, head_dim 32) -- forward and backward.
//
// Same contract as csrc/win_attn.hip (reference models/video_swin_transformer.py:138-169); that file's thread-per-query kernels do
// 4 N^2 32 FLOP per (window, head) on the vector ALUs: 14.8 GFLOP per call at Swin-T stage 1 = 94 us at the fp32 vector peak, and
// they measured 291 / 383 / 383 us per call (25 ms of the 75 ms Swin-T step).  Here the two products run on
// v_mfma_f32_32x32x16_{bf16,f16}; what remains on the vector ALUs is the softmax.
//
// One workgroup = one (window, head), 4 waves; a wave owns 32-query tiles.  Everything is computed TRANSPOSED so that a lane owns
// one query COLUMN:
//   S^T[key, q]  = K[key, :] . Q[q, :]            A = K rows (LDS, d contiguous), B = Q rows (registers, loaded once per tile)
//   accumulator layout of a 32x32 tile: lane = (q = lane & 31, h = lane >> 5), register i <-> key (i & 3) + 8 (i >> 2) + 4 h
//   -> the row statistics of query q (max, sum) are a reduction over the lane's own 16 registers plus ONE exchange with lane ^ 32.
//   O^T[d, q]   += V^T[d, key] P^T[key, q]        B = the lane's own P registers, 8 per K-slab: slab s uses registers 8s..8s+7,
//   i.e. keys sigma_s(h, j) = 16 s + 4 h + (j & 3) + 8 (j >> 2); the A operand V^T is read from LDS in the SAME key order (two
//   8-byte reads per lane), so the probabilities never leave the registers.
// P is rounded to the storage dtype before the second product and q * scale before the first, as the reference's autocast
// matmuls see them.  Masking: -100 between different shift regions (int compare), -inf for the padding keys of the last tile.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "win_attn_mfma.h"

namespace {

constexpr int HD = 32;
constexpr int KROW = HD + 8;        // LDS row of K / Q / dO tiles in elements (80 B: conflict-free 16-B reads of 16 rows)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T> struct MM;
template <> struct MM<__hip_bfloat16> {
  typedef short v8 __attribute__((ext_vector_type(8)));
  typedef short v4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ short bits(float v) { return (short)__bfloat16_as_ushort(__float2bfloat16(v)); }
  static __device__ __forceinline__ float val(short b) { return __uint_as_float(((unsigned)(unsigned short)b) << 16); }
};
template <> struct MM<__half> {
  typedef _Float16 v8 __attribute__((ext_vector_type(8)));
  typedef _Float16 v4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ _Float16 bits(float v) { return (_Float16)v; }
  static __device__ __forceinline__ float val(_Float16 b) { return (float)b; }
};

__device__ __forceinline__ int key_of(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }
__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }

// rows `which` (0 q, 1 k, 2 v) of one (window, head): [N][32] -> LDS [Np][KROW] in the storage dtype (rows >= N zero), times mul
template <typename T, typename E>
__device__ __forceinline__ void stage_rows(E* dst, const T* qkv_bw, int N, int Np, int H, int h, int which, float mul) {
  typedef MM<T> M;
  for (int e = threadIdx.x; e < Np * 4; e += blockDim.x) {
    const int j = e >> 2, seg = e & 3;
    typename M::v8 v = {};
    if (j < N) {
      v = *reinterpret_cast<const typename M::v8*>(qkv_bw + (((long long)j * 3 + which) * H + h) * HD + seg * 8);
      if (mul != 1.f) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = M::bits(M::val(v[u]) * mul);
      }
    }
    *reinterpret_cast<typename M::v8*>(dst + j * KROW + seg * 8) = v;
  }
}

// rows of a [*, N, H*32] tensor (out / dout) of one (window, head) -> LDS [Np][KROW]
template <typename T, typename E>
__device__ __forceinline__ void stage_out_rows(E* dst, const T* t_bw, int N, int Np, int H, int h) {
  typedef MM<T> M;
  for (int e = threadIdx.x; e < Np * 4; e += blockDim.x) {
    const int j = e >> 2, seg = e & 3;
    typename M::v8 v = {};
    if (j < N) v = *reinterpret_cast<const typename M::v8*>(t_bw + (long long)j * H * HD + h * HD + seg * 8);
    *reinterpret_cast<typename M::v8*>(dst + j * KROW + seg * 8) = v;
  }
}

// A operand of the second-type products (O^T += V^T P^T, dq^T += K^T dS^T, dv^T += dO^T P, dk^T += Q^T dS): row d = lane & 31 of the
// TRANSPOSE of a row-major LDS tile [key][KROW], at the 8 positions sigma_s(h, .) = 16 s + 4 h + {0..3, 8..11} of the 32-wide tile at
// t0.  Round 4: read straight from the row-major tile with gfx950's transposing LDS read (ds_read_b64_tr_b16: per 16-lane group a
// 4-row x 16-column block, delivered column-major: lane i of the group gets column i of the 4 rows; lane 4q + p supplies row q,
// columns 4p..4p+3) -- the separate [32][Np + 8] transposed copies are gone, and with them a third of the LDS of the backward kernels:
// two workgroups per CU at N = 392 instead of one (one wave per SIMD had every latency exposed).
__device__ __forceinline__ MM<__hip_bfloat16>::v4 tr_read4(const short* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) MM<__hip_bfloat16>::v4*)p);
}
__device__ __forceinline__ MM<__half>::v4 tr_read4(const _Float16* p) {
  const MM<__hip_bfloat16>::v4 raw = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) MM<__hip_bfloat16>::v4*)p);
  MM<__half>::v4 v;
  __builtin_memcpy(&v, &raw, sizeof(v));          // the same 64 bits, typed as halves
  return v;
}
template <typename T, typename E>
__device__ __forceinline__ typename MM<T>::v8 tr_frag(const E* tile, int t0, int s, int lane) {
  typedef MM<T> M;
  const int g = lane >> 4, li = lane & 15;
  const E* p = tile + (t0 + 16 * s + 4 * (g >> 1) + (li >> 2)) * KROW + 16 * (g & 1) + 4 * (li & 3);
  const typename M::v4 lo = tr_read4(p), hi = tr_read4(p + 8 * KROW);
  typename M::v8 a;
  a[0] = lo[0]; a[1] = lo[1]; a[2] = lo[2]; a[3] = lo[3]; a[4] = hi[0]; a[5] = hi[1]; a[6] = hi[2]; a[7] = hi[3];
  return a;
}

constexpr float kLog2e = 1.4426950408889634f;
// exp(x - m) as ONE fused multiply-add and the hardware exp2: x, m natural-log units, mL = m * log2(e)
__device__ __forceinline__ float exp_sub(float x, float mL) { return __builtin_amdgcn_exp2f(__builtin_fmaf(x, kLog2e, -mL)); }

// region ids of this window -> LDS; returns (workgroup-uniform) whether the window spans more than one shift region: only then the
// -100 mask exists at all (interior windows of a shifted block and every window of an unshifted block skip its 3 ops + 1 LDS read per score)
__device__ __forceinline__ bool stage_regions(int* reg_s, const int* reg_w, int N, int Np) {
  int differs = 0;
  const int r0 = reg_w ? reg_w[0] : 0;
  for (int j = threadIdx.x; j < Np; j += blockDim.x) {
    const int v = (reg_w && j < N) ? reg_w[j] : r0;
    reg_s[j] = v;
    differs |= v != r0;
  }
  return __syncthreads_or(differs) != 0;
}

template <typename T>
__global__ __launch_bounds__(256) void k_fwd(const T* __restrict__ qkv, const float* __restrict__ biasT, const int* __restrict__ region,
                                             float scale, int NW, int N, int Np, int H, T* __restrict__ out, float* __restrict__ lse) {
  typedef MM<T> M;
  typedef decltype(M::bits(0.f)) E;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  E* Ks = reinterpret_cast<E*>(smem);                       // [Np][KROW]
  E* Vs = Ks + (size_t)Np * KROW;                           // [Np][KROW] (read transposed: tr_frag)
  int* reg_s = reinterpret_cast<int*>(Vs + (size_t)Np * KROW);
  const int bw = blockIdx.x / H, hh = blockIdx.x % H;
  const T* base = qkv + (long long)bw * N * 3 * H * HD;
  stage_rows<T, E>(Ks, base, N, Np, H, hh, 1, 1.f);
  stage_rows<T, E>(Vs, base, N, Np, H, hh, 2, 1.f);
  const bool shifted = stage_regions(reg_s, region ? region + (long long)(bw % NW) * N : nullptr, N, Np);     // (ends with a barrier)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const float* bT = biasT + (long long)hh * N * N;
  for (int qt = wave; qt < Np / 32; qt += 4) {
    const int q = qt * 32 + r;
    const bool qok = q < N;
    typename M::v8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      typename M::v8 v = {};
      if (qok) {
        v = *reinterpret_cast<const typename M::v8*>(base + (((long long)q * 3 + 0) * H + hh) * HD + 16 * s + 8 * h);
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = M::bits(M::val(v[u]) * scale);
      }
      qf[s] = v;
    }
    const int rq = reg_s[qok ? q : 0];
    const int qc = qok ? q : N - 1;
    float m = -INFINITY, l = 0.f;
    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    // the bias of the NEXT key tile is fetched while this one is soft-maxed; loads are unconditional on clamped indices (a load
    // inside a branch gets its own s_waitcnt: 16 serialised L2 round trips per tile, measured 13.8k cycles per tile)
    float bn[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bn[i] = bT[(long long)min(key_of(i, h), N - 1) * N + qc];
    for (int kt = 0; kt < Np; kt += 32) {
      float bc[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) bc[i] = bn[i];
      if (kt + 32 < Np) {
#pragma unroll
        for (int i = 0; i < 16; ++i) bn[i] = bT[(long long)min(kt + 32 + key_of(i, h), N - 1) * N + qc];
      }
      f32x16 sa;
#pragma unroll
      for (int i = 0; i < 16; ++i) sa[i] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s)
        sa = M::mfma(*reinterpret_cast<const typename M::v8*>(Ks + (kt + r) * KROW + 16 * s + 8 * h), qf[s], sa);
#pragma unroll
      for (int i = 0; i < 16; ++i) sa[i] += bc[i];
      if (shifted) {                     // (workgroup-uniform) only windows that span two shift regions carry the -100 mask
#pragma unroll
        for (int i = 0; i < 16; ++i) sa[i] += (reg_s[kt + key_of(i, h)] != rq) ? -100.f : 0.f;
      }
      if (kt + 32 > N) {                 // (uniform) only the last key tile holds padding keys
#pragma unroll
        for (int i = 0; i < 16; ++i) sa[i] = kt + key_of(i, h) < N ? sa[i] : -INFINITY;
      }
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sa[i]);
      mx = fmaxf(mx, xhalf(mx));
      const float mn = fmaxf(m, mx), mnL = mn * kLog2e;
      const float corr = exp_sub(m, mnL);
      l *= corr;
      typename M::v8 pf[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        o[i] *= corr;
        const float p = exp_sub(sa[i], mnL);
        l += p;
        pf[i >> 3][i & 7] = M::bits(p);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) o = M::mfma(tr_frag<T, E>(Vs, kt, s, lane), pf[s], o);
      m = mn;
    }
    l += xhalf(l);
    const float inv = 1.f / l;
    if (qok) {
      T* orow = out + ((long long)bw * N + q) * H * HD + hh * HD;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typename M::v4 w;
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = M::bits(o[4 * g + u] * inv);
        *reinterpret_cast<typename M::v4*>(orow + 8 * g + 4 * h) = w;       // registers 4g..4g+3 <-> d = 8g + 4h + (0..3)
      }
      if (h == 0) lse[((long long)bw * H + hh) * N + q] = m + __logf(l);
    }
  }
}

// ---- backward, pass 1: a wave owns a 32-query tile (transposed layout as in the forward).
//   D_q = dO_q . O_q;   P^T = exp(S^T + bias + mask - lse_q);   dP^T[key, q] = V[key, :] . dO[q, :];   dS^T = P^T (dP^T - D_q)
//   dq^T[d, q] = scale * sum_key K^T[d, key] dS^T[key, q]      (A = K^T read transposed from the row-major K tile, B = the lane's own dS registers)
//   dS^T is also written (storage dtype, [BW, H, N(key), N(q)]) for the bias gradient: summing that tensor over the windows costs
//   2 x 232 MB of streaming traffic at Swin-T stage 1, the per-element float atomics of the vector-ALU kernel 464 MB of atomic adds.
template <typename T>
__global__ __launch_bounds__(256) void k_bwd_q(const T* __restrict__ qkv, const float* __restrict__ biasT, const int* __restrict__ region,
                                               float scale, int NW, int N, int Np, int H, const T* __restrict__ out,
                                               const T* __restrict__ dout, const float* __restrict__ lse, T* __restrict__ dqkv,
                                               float* __restrict__ Dbuf, T* __restrict__ dS) {
  typedef MM<T> M;
  typedef decltype(M::bits(0.f)) E;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  E* Ks = reinterpret_cast<E*>(smem);                       // [Np][KROW] (row reads for S^T, transposed reads for dq^T)
  E* Vs = Ks + (size_t)Np * KROW;                           // [Np][KROW]
  int* reg_s = reinterpret_cast<int*>(Vs + (size_t)Np * KROW);
  const int bw = blockIdx.x / H, hh = blockIdx.x % H;
  const T* base = qkv + (long long)bw * N * 3 * H * HD;
  stage_rows<T, E>(Ks, base, N, Np, H, hh, 1, 1.f);
  stage_rows<T, E>(Vs, base, N, Np, H, hh, 2, 1.f);
  const bool shifted = stage_regions(reg_s, region ? region + (long long)(bw % NW) * N : nullptr, N, Np);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const float* bT = biasT + (long long)hh * N * N;
  E* dSw = dS ? reinterpret_cast<E*>(dS) + ((long long)bw * H + hh) * N * N : nullptr;
  for (int qt = wave; qt < Np / 32; qt += 4) {
    const int q = qt * 32 + r;
    const bool qok = q < N;
    const int qc = qok ? q : N - 1;
    typename M::v8 qf[2], gf[2];
    float Dq = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      typename M::v8 v = *reinterpret_cast<const typename M::v8*>(base + (((long long)qc * 3 + 0) * H + hh) * HD + 16 * s + 8 * h);
      const typename M::v8 g = *reinterpret_cast<const typename M::v8*>(dout + ((long long)bw * N + qc) * H * HD + hh * HD + 16 * s + 8 * h);
      const typename M::v8 ov = *reinterpret_cast<const typename M::v8*>(out + ((long long)bw * N + qc) * H * HD + hh * HD + 16 * s + 8 * h);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = M::bits(M::val(v[u]) * scale);
        Dq += M::val(g[u]) * M::val(ov[u]);
      }
      qf[s] = v;
      gf[s] = g;
    }
    Dq += xhalf(Dq);
    const float lqL = lse[((long long)bw * H + hh) * N + qc] * kLog2e;
    const int rq = reg_s[qc];
    f32x16 dq;
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[i] = 0.f;
    float bn[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bn[i] = bT[(long long)min(key_of(i, h), N - 1) * N + qc];
    for (int kt = 0; kt < Np; kt += 32) {
      float bc[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) bc[i] = bn[i];
      if (kt + 32 < Np) {
#pragma unroll
        for (int i = 0; i < 16; ++i) bn[i] = bT[(long long)min(kt + 32 + key_of(i, h), N - 1) * N + qc];
      }
      f32x16 sa, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { sa[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        sa = M::mfma(*reinterpret_cast<const typename M::v8*>(Ks + (kt + r) * KROW + 16 * s + 8 * h), qf[s], sa);
        dp = M::mfma(*reinterpret_cast<const typename M::v8*>(Vs + (kt + r) * KROW + 16 * s + 8 * h), gf[s], dp);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) sa[i] += bc[i];
      if (shifted) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sa[i] += (reg_s[kt + key_of(i, h)] != rq) ? -100.f : 0.f;
      }
      if (kt + 32 > N) {                 // padding keys: P = exp2(-inf) = 0
#pragma unroll
        for (int i = 0; i < 16; ++i) sa[i] = kt + key_of(i, h) < N ? sa[i] : -INFINITY;
      }
      typename M::v8 df[2];
      const bool tail = kt + 32 > N;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kt + key_of(i, h);
        const float p = exp_sub(sa[i], lqL);
        const float ds = p * (dp[i] - Dq);
        const E dsb = M::bits(ds);
        df[i >> 3][i & 7] = dsb;
        if (dSw && qok && (!tail || key < N)) dSw[(long long)key * N + q] = dsb;       // lanes = consecutive q: 64 contiguous bytes per key
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) dq = M::mfma(tr_frag<T, E>(Ks, kt, s, lane), df[s], dq);
    }
    if (qok) {
      T* drow = dqkv + (((long long)(bw * (long long)N + q) * 3 + 0) * H + hh) * HD;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typename M::v4 w;
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = M::bits(dq[4 * g + u] * scale);
        *reinterpret_cast<typename M::v4*>(drow + 8 * g + 4 * h) = w;
      }
      if (h == 0) Dbuf[((long long)bw * H + hh) * N + q] = Dq;
    }
  }
}

// ---- backward, pass 2: a wave owns a 32-KEY tile; S is computed un-transposed (lane = key column, registers = 16 query rows):
//   S[q, key] = (scale Q)[q, :] . K[key, :]   (A = scaled Q rows from LDS, B = the lane's K row in registers)
//   dP[q, key] = dO[q, :] . V[key, :]         (A = dO rows from LDS, B = the lane's V row)
//   dv^T[d, key] = sum_q dO^T[d, q] P[q, key],   dk^T[d, key] = sum_q (scale Q)^T[d, q] dS[q, key]   (A: the same LDS tiles, read transposed)
template <typename T>
__global__ __launch_bounds__(256) void k_bwd_kv(const T* __restrict__ qkv, const float* __restrict__ bias, const int* __restrict__ region,
                                                float scale, int NW, int N, int Np, int H, const T* __restrict__ dout,
                                                const float* __restrict__ lse, const float* __restrict__ Dbuf, T* __restrict__ dqkv) {
  typedef MM<T> M;
  typedef decltype(M::bits(0.f)) E;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  E* Qs = reinterpret_cast<E*>(smem);                       // [Np][KROW] scaled q
  E* Gs = Qs + (size_t)Np * KROW;                           // [Np][KROW] dO
  float* Ls = reinterpret_cast<float*>(Gs + (size_t)Np * KROW);       // [Np] lse * log2(e) (+inf on padding rows: P = 0)
  float* Ds = Ls + Np;
  int* reg_s = reinterpret_cast<int*>(Ds + Np);
  const int bw = blockIdx.x / H, hh = blockIdx.x % H;
  const T* base = qkv + (long long)bw * N * 3 * H * HD;
  const T* gbase = dout + (long long)bw * N * H * HD;
  stage_rows<T, E>(Qs, base, N, Np, H, hh, 0, scale);
  stage_out_rows<T, E>(Gs, gbase, N, Np, H, hh);
  for (int j = threadIdx.x; j < Np; j += blockDim.x) {
    Ls[j] = j < N ? lse[((long long)bw * H + hh) * N + j] * kLog2e : INFINITY;
    Ds[j] = j < N ? Dbuf[((long long)bw * H + hh) * N + j] : 0.f;
  }
  const bool shifted = stage_regions(reg_s, region ? region + (long long)(bw % NW) * N : nullptr, N, Np);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const float* b = bias + (long long)hh * N * N;
  for (int kt = wave; kt < Np / 32; kt += 4) {
    const int key = kt * 32 + r;
    const bool kok = key < N;
    const int kc = kok ? key : N - 1;
    typename M::v8 kf[2], vf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      kf[s] = *reinterpret_cast<const typename M::v8*>(base + (((long long)kc * 3 + 1) * H + hh) * HD + 16 * s + 8 * h);
      vf[s] = *reinterpret_cast<const typename M::v8*>(base + (((long long)kc * 3 + 2) * H + hh) * HD + 16 * s + 8 * h);
    }
    const int rk = reg_s[kc];
    f32x16 dk, dv;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[i] = 0.f; dv[i] = 0.f; }
    float bn[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bn[i] = b[(long long)min(key_of(i, h), N - 1) * N + kc];
    for (int qt = 0; qt < Np; qt += 32) {
      float bc[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) bc[i] = bn[i];
      if (qt + 32 < Np) {
#pragma unroll
        for (int i = 0; i < 16; ++i) bn[i] = b[(long long)min(qt + 32 + key_of(i, h), N - 1) * N + kc];
      }
      f32x16 sa, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { sa[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        sa = M::mfma(*reinterpret_cast<const typename M::v8*>(Qs + (qt + r) * KROW + 16 * s + 8 * h), kf[s], sa);
        dp = M::mfma(*reinterpret_cast<const typename M::v8*>(Gs + (qt + r) * KROW + 16 * s + 8 * h), vf[s], dp);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) sa[i] += bc[i];
      if (shifted) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sa[i] += (reg_s[qt + key_of(i, h)] != rk) ? -100.f : 0.f;
      }
      typename M::v8 pf[2], df[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int q = qt + key_of(i, h);                 // the same register <-> row map, rows are queries here
        const float p = exp_sub(sa[i], Ls[q]);           // padding rows: exp2(-inf) = 0
        pf[i >> 3][i & 7] = M::bits(p);
        df[i >> 3][i & 7] = M::bits(p * (dp[i] - Ds[q]));
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        dv = M::mfma(tr_frag<T, E>(Gs, qt, s, lane), pf[s], dv);
        dk = M::mfma(tr_frag<T, E>(Qs, qt, s, lane), df[s], dk);
      }
    }
    if (kok) {
      T* dkrow = dqkv + (((long long)(bw * (long long)N + key) * 3 + 1) * H + hh) * HD;
      T* dvrow = dqkv + (((long long)(bw * (long long)N + key) * 3 + 2) * H + hh) * HD;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typename M::v4 wk, wv;
#pragma unroll
        for (int u = 0; u < 4; ++u) { wk[u] = M::bits(dk[4 * g + u]); wv[u] = M::bits(dv[4 * g + u]); }
        *reinterpret_cast<typename M::v4*>(dkrow + 8 * g + 4 * h) = wk;
        *reinterpret_cast<typename M::v4*>(dvrow + 8 * g + 4 * h) = wv;
      }
    }
  }
}

inline int status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

inline size_t fwd_lds(int Np, size_t esz) { return (size_t)2 * Np * KROW * esz + (size_t)Np * sizeof(int); }

template <typename T>
int fwd_t(const void* qkv, const float* biasT, const int* region, float scale, int BW, int NW, int N, int H, void* out, float* lse,
          hipStream_t st) {
  const int Np = (N + 31) / 32 * 32;
  const size_t lds = fwd_lds(Np, 2);
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(int)e;
  }
  k_fwd<T><<<BW * H, 256, lds, st>>>((const T*)qkv, biasT, region, scale, NW, N, Np, H, (T*)out, lse);
  return status();
}

inline size_t bwd_q_lds(int Np, size_t esz) { return (size_t)2 * Np * KROW * esz + (size_t)Np * sizeof(int); }
inline size_t bwd_kv_lds(int Np, size_t esz) { return (size_t)2 * Np * KROW * esz + (size_t)Np * (2 * sizeof(float) + sizeof(int)); }

template <typename T>
int bwd_t(const void* qkv, const float* bias, const float* biasT, const int* region, float scale, int BW, int NW, int N, int H, const void* out,
          const void* dout, const float* lse, void* dqkv, float* Dbuf, void* dS, hipStream_t st) {
  const int Np = (N + 31) / 32 * 32;
  const size_t l1 = bwd_q_lds(Np, 2), l2 = bwd_kv_lds(Np, 2);
  if (l1 > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd_q<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1);
    if (e != hipSuccess) return -(int)e;
  }
  if (l2 > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd_kv<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
    if (e != hipSuccess) return -(int)e;
  }
  k_bwd_q<T><<<BW * H, 256, l1, st>>>((const T*)qkv, biasT, region, scale, NW, N, Np, H, (const T*)out, (const T*)dout, lse, (T*)dqkv, Dbuf,
                                      (T*)dS);
  if (int e = status()) return e;
  k_bwd_kv<T><<<BW * H, 256, l2, st>>>((const T*)qkv, bias, region, scale, NW, N, Np, H, (const T*)dout, lse, Dbuf, (T*)dqkv);
  return status();
}

}  // namespace

namespace ocpg_win_mfma {

bool supported(int N, int head_dim, int dtype) {
  if (head_dim != HD || (dtype != 1 && dtype != 2) || N < 1) return false;
  const int Np = (N + 31) / 32 * 32;
  return fwd_lds(Np, 2) <= 150 * 1024 && bwd_q_lds(Np, 2) <= 150 * 1024 && bwd_kv_lds(Np, 2) <= 150 * 1024;
}

int bwd(const void* qkv, const float* bias, const float* biasT, const int* region, float scale, int BW, int NW, int N, int H, const void* out,
        const void* dout, const float* lse, void* dqkv, float* Dbuf, void* dS, int dtype, hipStream_t st) {
  return dtype == 1 ? bwd_t<__hip_bfloat16>(qkv, bias, biasT, region, scale, BW, NW, N, H, out, dout, lse, dqkv, Dbuf, dS, st)
                    : bwd_t<__half>(qkv, bias, biasT, region, scale, BW, NW, N, H, out, dout, lse, dqkv, Dbuf, dS, st);
}

int fwd(const void* qkv, const float* biasT, const int* region, float scale, int BW, int NW, int N, int H, void* out, float* lse, int dtype,
        hipStream_t st) {
  return dtype == 1 ? fwd_t<__hip_bfloat16>(qkv, biasT, region, scale, BW, NW, N, H, out, lse, st)
                    : fwd_t<__half>(qkv, biasT, region, scale, BW, NW, N, H, out, lse, st);
}

}  // namespace ocpg_win_mfma
