// Multi-head attention against a SHORT key sequence (<= 32 keys), forward and backward -- the attention core of
//   * VisionLanguageFusionModule (models/segmentation.py:95-113): 19 200 x B visual tokens of a level attend to the <= ~20
//     text tokens of their clip, 8 heads x 32 channels (nn.MultiheadAttention's softmax(q k^T / sqrt(d)) v with a key padding
//     mask); round 1 ran this shape on AOTriton's flash-attention kernels (230 us per backward call);
//   * the decoder layer's self-attention over the 5 queries of a frame (models/deformable_transformer.py:323-326), with
//     nn.MultiheadAttention's dropout on the attention weights in training.
// The projections before and after stay GEMMs; q / k / v are consumed in the layout the projections leave them in
// ([L, B, H*32] rows with an arbitrary row stride: no permute / contiguous copies).
//
// Mapping.  A workgroup = 256 / H query tokens x H heads of ONE batch element; its K and V (<= 32 x H x 32 fp32, head slices
// padded to 36 floats so that the H slices a wave reads together fall on disjoint LDS banks) are staged in LDS once.
// A thread owns one (token, head): q and the output accumulator live in registers (32 + 32), keys are visited once with an
// online softmax -- no score array, any Lk.  Backward (same mapping) recomputes the probabilities from the saved
// log-sum-exp, writes dq, and leaves (p~, ds) of its (token, head) in LDS; the threads then regroup as (head, channel) and
// add the workgroup's tokens into dK / dV in registers (compile-time key count), flushed with one fp32 atomic per
// (key, channel) per workgroup.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"
#include "philox.h"

namespace {

constexpr int HD = 32;        // head dimension (every BASELINE configuration: 256 / 8)
constexpr int HS = 36;        // LDS stride of a head slice (floats)
constexpr int NT = 256;

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static __device__ __forceinline__ float to(float v) { return v; }
  static __device__ __forceinline__ float from(float v) { return v; }
};
template <> struct Cvt<__hip_bfloat16> {
  static __device__ __forceinline__ float to(__hip_bfloat16 v) { return __bfloat162float(v); }
  static __device__ __forceinline__ __hip_bfloat16 from(float v) { return __float2bfloat16(v); }
};
template <> struct Cvt<__half> {
  static __device__ __forceinline__ float to(__half v) { return __half2float(v); }
  static __device__ __forceinline__ __half from(float v) { return __float2half(v); }
};

template <typename T>
__device__ __forceinline__ void load_row(const T* p, float (&f)[HD]) {
#pragma unroll
  for (int d = 0; d < HD; ++d) f[d] = Cvt<T>::to(p[d]);
}

// keep-scale of attention weight (row, j): 1/(1-p) or 0; thr = p * 2^32 (0 = no dropout)
__device__ __forceinline__ float keep_scale(uint64_t seed, uint64_t offset, uint64_t row, int j, uint32_t thr, float inv_keep) {
  if (thr == 0u) return 1.f;
  const uint64_t idx = row * 32u + (uint64_t)j;
  const uint4 r = ocpg_dev::philox(make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)),
                                   make_uint4((uint32_t)(idx >> 2), (uint32_t)(idx >> 34), (uint32_t)offset, (uint32_t)(offset >> 32)));
  const uint32_t v = (idx & 3) == 0 ? r.x : (idx & 3) == 1 ? r.y : (idx & 3) == 2 ? r.z : r.w;
  return v >= thr ? inv_keep : 0.f;
}

// stage K and V of batch element b into LDS (fp32, head slices padded); bias[j] = 0 or -inf (key padding)
template <typename T>
__device__ __forceinline__ void stage_kv(const T* __restrict__ k, long long ldk, const T* __restrict__ v, long long ldv,
                                         const unsigned char* __restrict__ pad, int b, int B, int H, int Lk, float* ks, float* vs,
                                         float* bias) {
  const int C = H * HD;
  for (int i = threadIdx.x; i < Lk * C; i += NT) {
    const int j = i / C, c = i - j * C;
    const int at = (j * H + c / HD) * HS + c % HD;
    ks[at] = Cvt<T>::to(k[((long long)j * B + b) * ldk + c]);
    vs[at] = Cvt<T>::to(v[((long long)j * B + b) * ldv + c]);
  }
  if (threadIdx.x < Lk) bias[threadIdx.x] = (pad && pad[(long long)b * Lk + threadIdx.x]) ? -INFINITY : 0.f;
}

template <typename T>
__global__ __launch_bounds__(NT) void attn_smallk_fwd(const T* __restrict__ q, long long ldq, const T* __restrict__ k, long long ldk,
                                                      const T* __restrict__ v, long long ldv, const unsigned char* __restrict__ pad,
                                                      float scale, int Lq, int B, int H, int Lk, float pdrop, uint64_t seed,
                                                      uint64_t offset0, const uint64_t* __restrict__ rng_base, T* __restrict__ out,
                                                      long long ldo, float* __restrict__ lse) {
  const uint64_t offset = offset0 + (rng_base ? *rng_base : 0ull);      // graph replays: the step's base lives in device memory
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ks = smem;
  float* vs = ks + Lk * H * HS;
  float* bias = vs + Lk * H * HS;
  const int b = blockIdx.y;
  stage_kv<T>(k, ldk, v, ldv, pad, b, B, H, Lk, ks, vs, bias);
  __syncthreads();
  const int tok_per = NT / H;
  const int tl = threadIdx.x / H, h = threadIdx.x % H;
  const int tok = blockIdx.x * tok_per + tl;
  if (tl >= tok_per || tok >= Lq) return;
  const long long row = (long long)tok * B + b;
  float qr[HD], acc[HD];
  load_row<T>(q + row * ldq + h * HD, qr);
#pragma unroll
  for (int d = 0; d < HD; ++d) { qr[d] *= scale; acc[d] = 0.f; }
  const uint32_t thr = pdrop > 0.f ? (uint32_t)fminf(pdrop * 4294967296.f, 4294967040.f) : 0u;
  const float inv_keep = pdrop > 0.f ? 1.f / (1.f - pdrop) : 1.f;
  const uint64_t rowh = (uint64_t)row * H + h;
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j < Lk; ++j) {
    const float4* kj = reinterpret_cast<const float4*>(ks + (j * H + h) * HS);
    float s = bias[j];
#pragma unroll
    for (int d4 = 0; d4 < HD / 4; ++d4) {
      const float4 kk = kj[d4];
      s += qr[4 * d4] * kk.x + qr[4 * d4 + 1] * kk.y + qr[4 * d4 + 2] * kk.z + qr[4 * d4 + 3] * kk.w;
    }
    const float mn = fmaxf(m, s);
    // masked keys (s = -inf) contribute nothing, also while every key so far was masked (m = mn = -inf: no inf - inf)
    const float corr = m == -INFINITY ? 0.f : __expf(m - mn), p = s == -INFINITY ? 0.f : __expf(s - mn);
    l = l * corr + p;
    const float pk = p * keep_scale(seed, offset, rowh, j, thr, inv_keep);
    const float4* vj = reinterpret_cast<const float4*>(vs + (j * H + h) * HS);
#pragma unroll
    for (int d4 = 0; d4 < HD / 4; ++d4) {
      const float4 vv = vj[d4];
      acc[4 * d4] = acc[4 * d4] * corr + pk * vv.x;
      acc[4 * d4 + 1] = acc[4 * d4 + 1] * corr + pk * vv.y;
      acc[4 * d4 + 2] = acc[4 * d4 + 2] * corr + pk * vv.z;
      acc[4 * d4 + 3] = acc[4 * d4 + 3] * corr + pk * vv.w;
    }
    m = mn;
  }
  const float il = 1.f / l;
  T* o = out + row * ldo + h * HD;
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = Cvt<T>::from(acc[d] * il);
  lse[rowh] = m + __logf(l);
}

// Backward.  LKP = compile-time bound on Lk (8 / 16 / 32) for the register accumulators of phase 2.
template <typename T, int LKP>
__global__ __launch_bounds__(NT) void attn_smallk_bwd(const T* __restrict__ q, long long ldq, const T* __restrict__ k, long long ldk,
                                                      const T* __restrict__ v, long long ldv, const unsigned char* __restrict__ pad,
                                                      const T* __restrict__ dout, long long ldo, const float* __restrict__ lse, float scale,
                                                      int Lq, int B, int H, int Lk, float pdrop, uint64_t seed, uint64_t offset0,
                                                      const uint64_t* __restrict__ rng_base, int groups_per_block, T* __restrict__ dq,
                                                      long long lddq, float* __restrict__ dk, float* __restrict__ dv) {
  const uint64_t offset = offset0 + (rng_base ? *rng_base : 0ull);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tok_per = NT / H;
  float* ks = smem;
  float* vs = ks + Lk * H * HS;
  float* bias = vs + Lk * H * HS;
  float* ps = bias + 32;                         // [tok_per][H][LKP]  dropped probabilities p~
  float* dss = ps + tok_per * H * LKP;           // [tok_per][H][LKP]  ds * scale
  const int b = blockIdx.y;
  stage_kv<T>(k, ldk, v, ldv, pad, b, B, H, Lk, ks, vs, bias);
  __syncthreads();
  const int tl = threadIdx.x / H, h = threadIdx.x % H;
  const int C = H * HD;
  const uint32_t thr = pdrop > 0.f ? (uint32_t)fminf(pdrop * 4294967296.f, 4294967040.f) : 0u;
  const float inv_keep = pdrop > 0.f ? 1.f / (1.f - pdrop) : 1.f;
  // phase-2 identity of this thread: one channel of one head
  const int c2 = threadIdx.x % C, part = threadIdx.x / C, nparts = NT / C > 0 ? NT / C : 1;     // C <= 256
  const int h2 = c2 / HD;
  float accv[LKP], acck[LKP];
#pragma unroll
  for (int j = 0; j < LKP; ++j) accv[j] = acck[j] = 0.f;
  for (int g = 0; g < groups_per_block; ++g) {
    const int tok0 = (blockIdx.x * groups_per_block + g) * tok_per;
    if (tok0 >= Lq) break;                                              // uniform
    const int tok = tok0 + tl;
    if (tl < tok_per) {
      float* pr = ps + (tl * H + h) * LKP;
      float* dr = dss + (tl * H + h) * LKP;
      if (tok < Lq) {
        const long long row = (long long)tok * B + b;
        const uint64_t rowh = (uint64_t)row * H + h;
        float qr[HD], go[HD];
        load_row<T>(q + row * ldq + h * HD, qr);
        load_row<T>(dout + row * ldo + h * HD, go);
        const float ls = lse[rowh];
        float Dsum = 0.f;
        for (int j = 0; j < Lk; ++j) {                                   // pass 1: p, dp -> D = sum_j p_j dp_j
          const float4* kj = reinterpret_cast<const float4*>(ks + (j * H + h) * HS);
          const float4* vj = reinterpret_cast<const float4*>(vs + (j * H + h) * HS);
          float s = 0.f, dpt = 0.f;
#pragma unroll
          for (int d4 = 0; d4 < HD / 4; ++d4) {
            const float4 kk = kj[d4], vv = vj[d4];
            s += qr[4 * d4] * kk.x + qr[4 * d4 + 1] * kk.y + qr[4 * d4 + 2] * kk.z + qr[4 * d4 + 3] * kk.w;
            dpt += go[4 * d4] * vv.x + go[4 * d4 + 1] * vv.y + go[4 * d4 + 2] * vv.z + go[4 * d4 + 3] * vv.w;
          }
          const float p = __expf(s * scale + bias[j] - ls);
          const float ksc = keep_scale(seed, offset, rowh, j, thr, inv_keep);
          pr[j] = p * ksc;                      // p~_j: what multiplied v_j in the forward
          dr[j] = p;                            // parked: p_j
          Dsum += p * ksc * dpt;                // p_j * dp_j with dp_j = keep_j * dp~_j
        }
        float dqr[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) dqr[d] = 0.f;
        for (int j = 0; j < Lk; ++j) {                                   // pass 2: ds_j = p_j (dp_j - D); dq += ds_j k_j
          const float4* kj = reinterpret_cast<const float4*>(ks + (j * H + h) * HS);
          const float4* vj = reinterpret_cast<const float4*>(vs + (j * H + h) * HS);
          float dpt = 0.f;
#pragma unroll
          for (int d4 = 0; d4 < HD / 4; ++d4) {
            const float4 vv = vj[d4];
            dpt += go[4 * d4] * vv.x + go[4 * d4 + 1] * vv.y + go[4 * d4 + 2] * vv.z + go[4 * d4 + 3] * vv.w;
          }
          const float p = dr[j];
          const float ksc = p > 0.f ? pr[j] / p : 0.f;
          const float ds = p * (ksc * dpt - Dsum) * scale;
          dr[j] = ds;
#pragma unroll
          for (int d4 = 0; d4 < HD / 4; ++d4) {
            const float4 kk = kj[d4];
            dqr[4 * d4] += ds * kk.x; dqr[4 * d4 + 1] += ds * kk.y; dqr[4 * d4 + 2] += ds * kk.z; dqr[4 * d4 + 3] += ds * kk.w;
          }
        }
        T* o = dq + row * lddq + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = Cvt<T>::from(dqr[d]);
      } else {
        for (int j = 0; j < Lk; ++j) { pr[j] = 0.f; dr[j] = 0.f; }
      }
    }
    __syncthreads();
    // phase 2: thread = (head, channel); `nparts` threads share a channel and split the group's tokens
    if (threadIdx.x < nparts * C) {
      for (int t2 = part; t2 < tok_per; t2 += nparts) {
        const int tk = tok0 + t2;
        if (tk >= Lq) break;
        const long long row = (long long)tk * B + b;
        const float g2 = Cvt<T>::to(dout[row * ldo + c2]), q2 = Cvt<T>::to(q[row * ldq + c2]);
        const float* pr = ps + (t2 * H + h2) * LKP;
        const float* dr = dss + (t2 * H + h2) * LKP;
#pragma unroll
        for (int j = 0; j < LKP; ++j) {
          if (j < Lk) {
            accv[j] += pr[j] * g2;
            acck[j] += dr[j] * q2;
          }
        }
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < nparts * C) {
#pragma unroll
    for (int j = 0; j < LKP; ++j) {
      if (j < Lk) {
        atomicAdd(dv + ((long long)j * B + b) * C + c2, accv[j]);
        atomicAdd(dk + ((long long)j * B + b) * C + c2, acck[j]);
      }
    }
  }
}

inline int check_dims(int Lq, int B, int H, int hd, int Lk) {
  if (Lq < 0 || B < 0 || H <= 0 || Lk <= 0) return -1006;
  if (hd != HD || H > 8 || (NT % H) != 0 || Lk > 32 || B > 65535) return -2000;       // shape not served: the caller uses its generic path
  return 0;
}

}  // namespace

extern "C" int ocpg_attn_smallk_fwd(const void* q, long long ldq, const void* k, long long ldk, const void* v, long long ldv,
                                    const unsigned char* key_pad, float scale, int Lq, int B, int H, int hd, int Lk, float pdrop,
                                    unsigned long long seed, unsigned long long offset, const unsigned long long* rng_base, void* out,
                                    long long ldo, float* lse, int dtype, void* stream) {
  if (int e = check_dims(Lq, B, H, hd, Lk)) return e;
  if (Lq == 0 || B == 0) return 0;
  if (!q) return -1001;
  if (!k) return -1003;
  if (!v) return -1005;
  if (!out) return -1017;
  if (!lse) return -1019;
  if (dtype < 0 || dtype > 2) return -1020;
  const size_t lds = ((size_t)2 * Lk * H * HS + 32) * sizeof(float);
  const int tok_per = NT / H;
  const dim3 grid((unsigned)((Lq + tok_per - 1) / tok_per), (unsigned)B);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0)
    attn_smallk_fwd<float><<<grid, NT, lds, st>>>((const float*)q, ldq, (const float*)k, ldk, (const float*)v, ldv, key_pad, scale, Lq, B, H, Lk,
                                                  pdrop, seed, offset, (const uint64_t*)rng_base, (float*)out, ldo, lse);
  else if (dtype == 1)
    attn_smallk_fwd<__hip_bfloat16><<<grid, NT, lds, st>>>((const __hip_bfloat16*)q, ldq, (const __hip_bfloat16*)k, ldk, (const __hip_bfloat16*)v,
                                                           ldv, key_pad, scale, Lq, B, H, Lk, pdrop, seed, offset, (const uint64_t*)rng_base,
                                                           (__hip_bfloat16*)out, ldo, lse);
  else
    attn_smallk_fwd<__half><<<grid, NT, lds, st>>>((const __half*)q, ldq, (const __half*)k, ldk, (const __half*)v, ldv, key_pad, scale, Lq, B, H,
                                                   Lk, pdrop, seed, offset, (const uint64_t*)rng_base, (__half*)out, ldo, lse);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// kernels that may need more than the default 64-KB dynamic-LDS window opt in once per instantiation
template <typename K>
inline void allow_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

#define BWD_LAUNCH(T_, LKP_)                                                                                                         \
  allow_lds(attn_smallk_bwd<T_, LKP_>, lds);                                                                                         \
  attn_smallk_bwd<T_, LKP_><<<grid, NT, lds, st>>>((const T_*)q, ldq, (const T_*)k, ldk, (const T_*)v, ldv, key_pad, (const T_*)dout, ldo, lse, \
                                                   scale, Lq, B, H, Lk, pdrop, seed, offset, (const uint64_t*)rng_base, gpb, (T_*)dq, lddq, dk, dv)
#define BWD_DISPATCH(T_)                           \
  if (Lk <= 8) { BWD_LAUNCH(T_, 8); }              \
  else if (Lk <= 16) { BWD_LAUNCH(T_, 16); }       \
  else { BWD_LAUNCH(T_, 32); }

extern "C" int ocpg_attn_smallk_bwd(const void* q, long long ldq, const void* k, long long ldk, const void* v, long long ldv,
                                    const unsigned char* key_pad, const void* dout, long long ldo, const float* lse, float scale, int Lq,
                                    int B, int H, int hd, int Lk, float pdrop, unsigned long long seed, unsigned long long offset,
                                    const unsigned long long* rng_base, void* dq, long long lddq, float* dk, float* dv, int dtype,
                                    void* stream) {
  if (int e = check_dims(Lq, B, H, hd, Lk)) return e;
  if (Lq == 0 || B == 0) return 0;
  if (!q) return -1001;
  if (!k) return -1003;
  if (!v) return -1005;
  if (!dout) return -1008;
  if (!lse) return -1010;
  if (!dq) return -1020;
  if (!dk) return -1022;
  if (!dv) return -1023;
  if (dtype < 0 || dtype > 2) return -1024;
  const int tok_per = NT / H;
  const int LKP = Lk <= 8 ? 8 : Lk <= 16 ? 16 : 32;
  const size_t lds = ((size_t)2 * Lk * H * HS + 32 + (size_t)2 * tok_per * H * LKP) * sizeof(float);
  if (lds > 150 * 1024) return -2000;
  const int groups = (Lq + tok_per - 1) / tok_per;
  // a workgroup adds `gpb` token groups into its register sums before the flush: few flushes, still >= ~2 workgroups per CU
  int gpb = 1;
  while (gpb < 16 && (long long)(groups / (2 * gpb)) * B >= 512) gpb *= 2;
  const dim3 grid((unsigned)((groups + gpb - 1) / gpb), (unsigned)B);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) { BWD_DISPATCH(float); }
  else if (dtype == 1) { BWD_DISPATCH(__hip_bfloat16); }
  else { BWD_DISPATCH(__half); }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
