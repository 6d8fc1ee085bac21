// Fused 3-D (shifted-)window attention for Video-Swin on MI355X -- forward and backward, head_dim = 32.
//
// Reference arithmetic (models/video_swin_transformer.py:138-169): per window and head,
//     attn = softmax( (q * scale) @ k^T + relative_position_bias[h] + shift_mask[window] ),   out = attn @ v
// with the shift mask = -100 between tokens of different cyclic-shift regions (compute_mask :316-329).
// The reference materialises attn [windows, heads, N, N] in HBM (232 MB per block at Swin-T stage 1) and runs five
// kernels over it.  Here one workgroup owns one (window, head): K and V live in LDS as fp32, every thread owns one
// query row (q and the output accumulator stay in registers), scores are produced, biased, masked, soft-maxed
// (online, blocks of 8 keys) and consumed without ever leaving the CU.  The shift mask is never a tensor: it is the
// comparison of two int region ids.  Arithmetic is fp32 on the vector ALUs (exact reference numerics; this shape,
// N = 245/392 with d = 32, is nowhere near needing MFMA: the whole Swin-T forward attention is ~0.03 TFLOP).
// Backward recomputes the probabilities from the saved log-sum-exp (no N x N tensor is ever stored):
//   pass 1, thread per QUERY:  D_i = dO_i . O_i,  dq_i = scale * sum_j dS_ij k_j,  dBias[h,i,j] += dS_ij (atomics)
//   pass 2, thread per KEY:    dv_j = sum_i P_ij dO_i,  dk_j = sum_i dS_ij (scale q_i)
// Layouts: qkv [BW, N, 3, H, 32] (the qkv Linear's output), out [BW, N, H*32], lse / D [BW, H, N],
// bias [H, N, N] (i, j) and biasT [H, N, N] (j, i) so that both passes read it coalesced, region [NW, N] int32 or NULL.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "../../include/ocpg_hip.h"
#include "win_attn_mfma.h"

namespace {

constexpr int HD = 32;          // head dim of every Video-Swin variant (96/3, 128/4, ...)
constexpr int KB = 8;           // keys per online-softmax block

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <> __device__ __forceinline__ float to_f<__half>(__half v) { return __half2float(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __hip_bfloat16 from_f<__hip_bfloat16>(float v) { return __float2bfloat16(v); }
template <> __device__ __forceinline__ __half from_f<__half>(float v) { return __float2half(v); }

// stage rows `which` (0 q, 1 k, 2 v) of one (window, head) into LDS as fp32 [N][32]
template <typename T>
__device__ __forceinline__ void stage(float* dst, const T* qkv_bw, int N, int H, int h, int which, float mul) {
  for (int e = threadIdx.x; e < N * HD; e += blockDim.x) {
    const int j = e / HD, d = e % HD;
    dst[e] = to_f<T>(qkv_bw[(((long long)j * 3 + which) * H + h) * HD + d]) * mul;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void win_attn_fwd(const T* __restrict__ qkv, const float* __restrict__ biasT,
                                                    const int* __restrict__ region, float scale, int NW, int N, int H,
                                                    T* __restrict__ out, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;                 // [N][32]
  float* Vs = smem + (size_t)N * HD;
  int* reg_s = reinterpret_cast<int*>(Vs + (size_t)N * HD);   // [N]
  const int bw = blockIdx.x / H, h = blockIdx.x % H;
  const T* base = qkv + (long long)bw * N * 3 * H * HD;
  stage<T>(Ks, base, N, H, h, 1, 1.f);
  stage<T>(Vs, base, N, H, h, 2, 1.f);
  const int* reg_w = region ? region + (long long)(bw % NW) * N : nullptr;
  if (reg_w)
    for (int j = threadIdx.x; j < N; j += blockDim.x) reg_s[j] = reg_w[j];
  __syncthreads();
  const float* bT = biasT + (long long)h * N * N;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    float q[HD], o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      q[d] = to_f<T>(base[(((long long)i * 3 + 0) * H + h) * HD + d]) * scale;
      o[d] = 0.f;
    }
    const int ri = reg_w ? reg_s[i] : 0;
    float m = -INFINITY, l = 0.f;
    for (int j0 = 0; j0 < N; j0 += KB) {
      float s[KB];
      float bm = -INFINITY;
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int j = j0 + u;
        if (j < N) {
          const float4* kr = reinterpret_cast<const float4*>(Ks + j * HD);
          float acc = 0.f;
#pragma unroll
          for (int d4 = 0; d4 < HD / 4; ++d4) {
            const float4 kv = kr[d4];
            acc += q[4 * d4] * kv.x + q[4 * d4 + 1] * kv.y + q[4 * d4 + 2] * kv.z + q[4 * d4 + 3] * kv.w;
          }
          acc += bT[(long long)j * N + i];
          if (reg_w && reg_s[j] != ri) acc += -100.f;
          s[u] = acc;
          bm = fmaxf(bm, acc);
        } else {
          s[u] = -INFINITY;
        }
      }
      const float mn = fmaxf(m, bm);
      const float corr = __expf(m - mn);          // exp(-inf) = 0 on the first block
      l *= corr;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] *= corr;
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int j = j0 + u;
        if (j < N) {
          const float p = __expf(s[u] - mn);
          l += p;
          const float4* vr = reinterpret_cast<const float4*>(Vs + j * HD);
#pragma unroll
          for (int d4 = 0; d4 < HD / 4; ++d4) {
            const float4 vv = vr[d4];
            o[4 * d4] += p * vv.x; o[4 * d4 + 1] += p * vv.y; o[4 * d4 + 2] += p * vv.z; o[4 * d4 + 3] += p * vv.w;
          }
        }
      }
      m = mn;
    }
    const float inv = 1.f / l;
    T* orow = out + ((long long)bw * N + i) * H * HD + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) orow[d] = from_f<T>(o[d] * inv);
    lse[((long long)bw * H + h) * N + i] = m + __logf(l);
  }
}

// pass 1: thread per query.  dqkv[.., 0, h, :] = dq;  Dbuf = rowsum(dO * O);  dbiasT[h, j, i] += dS_ij.
template <typename T>
__global__ __launch_bounds__(256) void win_attn_bwd_q(const T* __restrict__ qkv, const float* __restrict__ biasT,
                                                      const int* __restrict__ region, float scale, int NW, int N, int H,
                                                      const T* __restrict__ out, const T* __restrict__ dout,
                                                      const float* __restrict__ lse, T* __restrict__ dqkv,
                                                      float* __restrict__ Dbuf, float* __restrict__ dbiasT) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  float* Vs = smem + (size_t)N * HD;
  int* reg_s = reinterpret_cast<int*>(Vs + (size_t)N * HD);
  const int bw = blockIdx.x / H, h = blockIdx.x % H;
  const T* base = qkv + (long long)bw * N * 3 * H * HD;
  stage<T>(Ks, base, N, H, h, 1, 1.f);
  stage<T>(Vs, base, N, H, h, 2, 1.f);
  const int* reg_w = region ? region + (long long)(bw % NW) * N : nullptr;
  if (reg_w)
    for (int j = threadIdx.x; j < N; j += blockDim.x) reg_s[j] = reg_w[j];
  __syncthreads();
  const float* bT = biasT + (long long)h * N * N;
  float* dbT = dbiasT ? dbiasT + (long long)h * N * N : nullptr;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    float q[HD], go[HD], dq[HD];
    float Di = 0.f;
    const T* orow = out + ((long long)bw * N + i) * H * HD + h * HD;
    const T* grow = dout + ((long long)bw * N + i) * H * HD + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      q[d] = to_f<T>(base[(((long long)i * 3 + 0) * H + h) * HD + d]) * scale;
      go[d] = to_f<T>(grow[d]);
      Di += go[d] * to_f<T>(orow[d]);
      dq[d] = 0.f;
    }
    const float li = lse[((long long)bw * H + h) * N + i];
    const int ri = reg_w ? reg_s[i] : 0;
    for (int j = 0; j < N; ++j) {
      const float4* kr = reinterpret_cast<const float4*>(Ks + j * HD);
      const float4* vr = reinterpret_cast<const float4*>(Vs + j * HD);
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < HD / 4; ++d4) {
        const float4 kv = kr[d4], vv = vr[d4];
        s += q[4 * d4] * kv.x + q[4 * d4 + 1] * kv.y + q[4 * d4 + 2] * kv.z + q[4 * d4 + 3] * kv.w;
        dp += go[4 * d4] * vv.x + go[4 * d4 + 1] * vv.y + go[4 * d4 + 2] * vv.z + go[4 * d4 + 3] * vv.w;
      }
      s += bT[(long long)j * N + i];
      if (reg_w && reg_s[j] != ri) s += -100.f;
      const float p = __expf(s - li);
      const float ds = p * (dp - Di);
      if (dbT) atomicAdd(dbT + (long long)j * N + i, ds);        // lanes = consecutive i: contiguous 256-B segments
#pragma unroll
      for (int d4 = 0; d4 < HD / 4; ++d4) {
        const float4 kv = kr[d4];
        dq[4 * d4] += ds * kv.x; dq[4 * d4 + 1] += ds * kv.y; dq[4 * d4 + 2] += ds * kv.z; dq[4 * d4 + 3] += ds * kv.w;
      }
    }
    T* dqrow = dqkv + (((long long)(bw * (long long)N + i) * 3 + 0) * H + h) * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) dqrow[d] = from_f<T>(dq[d] * scale);
    Dbuf[((long long)bw * H + h) * N + i] = Di;
  }
}

// pass 2: thread per key.  dqkv[.., 1, h, :] = dk, dqkv[.., 2, h, :] = dv.
template <typename T>
__global__ __launch_bounds__(256) void win_attn_bwd_kv(const T* __restrict__ qkv, const float* __restrict__ bias,
                                                       const int* __restrict__ region, float scale, int NW, int N, int H,
                                                       const T* __restrict__ dout, const float* __restrict__ lse,
                                                       const float* __restrict__ Dbuf, T* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;                                    // scaled q [N][32]
  float* Gs = smem + (size_t)N * HD;                   // dO      [N][32]
  float* Ls = Gs + (size_t)N * HD;                     // lse [N]
  float* Ds = Ls + N;                                  // D   [N]
  int* reg_s = reinterpret_cast<int*>(Ds + N);
  const int bw = blockIdx.x / H, h = blockIdx.x % H;
  const T* base = qkv + (long long)bw * N * 3 * H * HD;
  stage<T>(Qs, base, N, H, h, 0, scale);
  for (int e = threadIdx.x; e < N * HD; e += blockDim.x)
    Gs[e] = to_f<T>(dout[((long long)bw * N + e / HD) * H * HD + h * HD + e % HD]);
  const int* reg_w = region ? region + (long long)(bw % NW) * N : nullptr;
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    Ls[j] = lse[((long long)bw * H + h) * N + j];
    Ds[j] = Dbuf[((long long)bw * H + h) * N + j];
    if (reg_w) reg_s[j] = reg_w[j];
  }
  __syncthreads();
  const float* b = bias + (long long)h * N * N;
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    float k[HD], v[HD], dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      k[d] = to_f<T>(base[(((long long)j * 3 + 1) * H + h) * HD + d]);
      v[d] = to_f<T>(base[(((long long)j * 3 + 2) * H + h) * HD + d]);
      dk[d] = 0.f;
      dv[d] = 0.f;
    }
    const int rj = reg_w ? reg_s[j] : 0;
    for (int i = 0; i < N; ++i) {
      const float4* qr = reinterpret_cast<const float4*>(Qs + i * HD);
      const float4* gr = reinterpret_cast<const float4*>(Gs + i * HD);
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < HD / 4; ++d4) {
        const float4 qv = qr[d4], gv = gr[d4];
        s += qv.x * k[4 * d4] + qv.y * k[4 * d4 + 1] + qv.z * k[4 * d4 + 2] + qv.w * k[4 * d4 + 3];
        dp += gv.x * v[4 * d4] + gv.y * v[4 * d4 + 1] + gv.z * v[4 * d4 + 2] + gv.w * v[4 * d4 + 3];
      }
      s += b[(long long)i * N + j];
      if (reg_w && reg_s[i] != rj) s += -100.f;
      const float p = __expf(s - Ls[i]);
      const float ds = p * (dp - Ds[i]);
#pragma unroll
      for (int d4 = 0; d4 < HD / 4; ++d4) {
        const float4 qv = qr[d4], gv = gr[d4];
        dv[4 * d4] += p * gv.x; dv[4 * d4 + 1] += p * gv.y; dv[4 * d4 + 2] += p * gv.z; dv[4 * d4 + 3] += p * gv.w;
        dk[4 * d4] += ds * qv.x; dk[4 * d4 + 1] += ds * qv.y; dk[4 * d4 + 2] += ds * qv.z; dk[4 * d4 + 3] += ds * qv.w;
      }
    }
    T* dkrow = dqkv + (((long long)(bw * (long long)N + j) * 3 + 1) * H + h) * HD;
    T* dvrow = dqkv + (((long long)(bw * (long long)N + j) * 3 + 2) * H + h) * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      dkrow[d] = from_f<T>(dk[d]);
      dvrow[d] = from_f<T>(dv[d]);
    }
  }
}

inline int status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

template <typename T>
int fwd(const void* qkv, const float* biasT, const int* region, float scale, int BW, int NW, int N, int H, void* out, float* lse,
        hipStream_t st) {
  const size_t lds = (size_t)2 * N * HD * sizeof(float) + (size_t)N * sizeof(int);
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(win_attn_fwd<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(int)e;
  }
  win_attn_fwd<T><<<BW * H, 256, lds, st>>>((const T*)qkv, biasT, region, scale, NW, N, H, (T*)out, lse);
  return status();
}

template <typename T>
int bwd(const void* qkv, const float* bias, const float* biasT, const int* region, float scale, int BW, int NW, int N, int H,
        const void* out, const void* dout, const float* lse, void* dqkv, float* Dbuf, float* dbiasT, hipStream_t st) {
  const size_t lds1 = (size_t)2 * N * HD * sizeof(float) + (size_t)N * sizeof(int);
  const size_t lds2 = (size_t)2 * N * HD * sizeof(float) + (size_t)3 * N * sizeof(float);
  if (lds1 > 64 * 1024 || lds2 > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(win_attn_bwd_q<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    if (e != hipSuccess) return -(int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(win_attn_bwd_kv<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    if (e != hipSuccess) return -(int)e;
  }
  win_attn_bwd_q<T><<<BW * H, 256, lds1, st>>>((const T*)qkv, biasT, region, scale, NW, N, H, (const T*)out, (const T*)dout, lse,
                                               (T*)dqkv, Dbuf, dbiasT);
  int e = status();
  if (e) return e;
  win_attn_bwd_kv<T><<<BW * H, 256, lds2, st>>>((const T*)qkv, bias, region, scale, NW, N, H, (const T*)dout, lse, Dbuf, (T*)dqkv);
  return status();
}

inline int check_dims(int BW, int NW, int N, int H, int hd) {
  if (BW < 0 || NW <= 0 || N <= 0 || H <= 0) return -1006;
  if (hd != HD) return -1007;
  if ((size_t)2 * N * HD * sizeof(float) + (size_t)4 * N * sizeof(float) > 160 * 1024) return -1008;   // K/V (or Q/dO) must fit the CU's LDS
  if (NW > 0 && BW % NW) return -1009;
  return 0;
}

inline bool mfma_enabled() {      // A/B switch, read per call: OCPG_WIN_ATTN_MFMA=0 keeps the fp32 vector-ALU kernels for bf16 / fp16 too
  const char* e = std::getenv("OCPG_WIN_ATTN_MFMA");
  return !(e && e[0] == '0');
}

}  // namespace

extern "C" {

int ocpg_win_attn_fwd(const void* qkv, const float* biasT, const int* region, float scale, int BW, int NW, int N, int H,
                      int head_dim, void* out, float* lse, int dtype, void* stream) {
  if (int e = check_dims(BW, NW, N, H, head_dim)) return e;
  if (BW == 0) return 0;
  if (!qkv) return -1001;
  if (!biasT) return -1002;
  if (!out) return -1010;
  if (!lse) return -1011;
  hipStream_t st = (hipStream_t)stream;
  if (mfma_enabled() && ocpg_win_mfma::supported(N, head_dim, dtype))
    return ocpg_win_mfma::fwd(qkv, biasT, region, scale, BW, NW, N, H, out, lse, dtype, st);
  switch (dtype) {
    case 0: return fwd<float>(qkv, biasT, region, scale, BW, NW, N, H, out, lse, st);
    case 1: return fwd<__hip_bfloat16>(qkv, biasT, region, scale, BW, NW, N, H, out, lse, st);
    case 2: return fwd<__half>(qkv, biasT, region, scale, BW, NW, N, H, out, lse, st);
  }
  return -1012;
}

int ocpg_win_attn_bwd(const void* qkv, const float* bias, const float* biasT, const int* region, float scale, int BW, int NW,
                      int N, int H, int head_dim, const void* out, const void* dout, const float* lse, void* dqkv, float* Dbuf,
                      float* dbiasT, int dtype, void* stream) {
  if (int e = check_dims(BW, NW, N, H, head_dim)) return e;
  if (BW == 0) return 0;
  if (!qkv) return -1001;
  if (!bias) return -1002;
  if (!biasT) return -1003;
  if (!out) return -1011;
  if (!dout) return -1012;
  if (!lse) return -1013;
  if (!dqkv) return -1014;
  if (!Dbuf) return -1015;
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case 0: return bwd<float>(qkv, bias, biasT, region, scale, BW, NW, N, H, out, dout, lse, dqkv, Dbuf, dbiasT, st);
    case 1: return bwd<__hip_bfloat16>(qkv, bias, biasT, region, scale, BW, NW, N, H, out, dout, lse, dqkv, Dbuf, dbiasT, st);
    case 2: return bwd<__half>(qkv, bias, biasT, region, scale, BW, NW, N, H, out, dout, lse, dqkv, Dbuf, dbiasT, st);
  }
  return -1017;
}

/* Matrix-core backward (csrc/win_attn_mfma.hip): same arguments, but the bias gradient leaves as dS [BW, H, N, N] in the storage dtype
 * ((key, query) order, fully written; NULL when the bias needs no gradient) for the caller to sum over BW.  Returns -2000 when the shape /
 * dtype is not served (or OCPG_WIN_ATTN_MFMA=0): the caller then uses ocpg_win_attn_bwd. */
int ocpg_win_attn_bwd_mfma(const void* qkv, const float* bias, const float* biasT, const int* region, float scale, int BW, int NW, int N,
                           int H, int head_dim, const void* out, const void* dout, const float* lse, void* dqkv, float* Dbuf, void* dS,
                           int dtype, void* stream) {
  if (int e = check_dims(BW, NW, N, H, head_dim)) return e;
  if (!mfma_enabled() || !ocpg_win_mfma::supported(N, head_dim, dtype)) return -2000;
  if (BW == 0) return 0;
  if (!qkv) return -1001;
  if (!bias) return -1002;
  if (!biasT) return -1003;
  if (!out) return -1011;
  if (!dout) return -1012;
  if (!lse) return -1013;
  if (!dqkv) return -1014;
  if (!Dbuf) return -1015;
  return ocpg_win_mfma::bwd(qkv, bias, biasT, region, scale, BW, NW, N, H, out, dout, lse, dqkv, Dbuf, dS, dtype, (hipStream_t)stream);
}

}  // extern "C"
