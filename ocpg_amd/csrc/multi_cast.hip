// One launch for the dtype cast of MANY tensors (the autocast working copies of all parameters, and their gradients back).
//
// Under autocast the reference casts every Conv2d / Linear weight to half precision with its own kernel per use and casts
// every weight gradient back with another (torch.cuda.amp, engine.py:49-62): ~250 + ~370 tiny launches per step for
// ResNet-101 + the transformer.  torch._foreach_copy_ does not fuse copies whose source and destination dtypes differ
// (it loops), so amp_cache's "one cast per forward" was one autograd node but still hundreds of launches.  This kernel
// takes a device table of (source, destination, element count) and a prefix of 2048-element chunks; each workgroup finds
// its tensor by binary search and moves one chunk, 16 bytes per lane on the wide side.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"

namespace {

constexpr int CHUNK = 2048;     // elements per workgroup: 256 lanes x 8
constexpr int kBiasChunk = 256; // ... and per workgroup of a tensor whose slices are fp32 bias partials (bit 40 of its split count)

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <> __device__ __forceinline__ float to_f<__half>(__half v) { return __half2float(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __hip_bfloat16 from_f<__hip_bfloat16>(float v) { return __float2bfloat16(v); }
template <> __device__ __forceinline__ __half from_f<__half>(float v) { return __float2half_rn(v); }

// The same with a reduction on the way: tensor t is the SUM of splits[t] slices of numels[t] elements, strides[t] elements apart
// (the row-split weight-gradient GEMMs of amp_cache.weight_grad leave [splits, Cout, Cin] partial products: one `sum` launch per
// layer -- ~90 per step in the ResNet body -- and then the cast back to fp32; here both ride in the one launch that casts every
// gradient).  fp32 accumulation, one rounding.
template <typename S, typename D>
__global__ __launch_bounds__(256) void multi_cast_sum(const long long* __restrict__ srcs, const long long* __restrict__ dsts,
                                                      const long long* __restrict__ numels, const long long* __restrict__ chunk_prefix,
                                                      const long long* __restrict__ splits, const long long* __restrict__ strides, int n) {
  const long long blk = blockIdx.x;
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_prefix[mid] <= blk) lo = mid; else hi = mid;
  }
  const S* s = reinterpret_cast<const S*>(srcs[lo]);
  D* d = reinterpret_cast<D*>(dsts[lo]);
  const long long count = numels[lo], stride = strides[lo];
  long long ns = splits[lo];
  const long long base = (blk - chunk_prefix[lo]) * CHUNK + (long long)threadIdx.x * 8;
  if (ns >> 40) {             // bit 40: THIS tensor's slices are fp32 whatever S is (bias-gradient partials of ocpg_colsum_partials)
    // few elements (a bias), many slices: the whole workgroup walks the slices -- thread = (element, group of slices), consecutive
    // threads on consecutive elements (a thread per 8 elements left 32 lanes with 8 x ns dependent scalar loads each: the launch's tail)
    // Such a tensor is cut into chunks of kBiasChunk = 256 elements (the host's chunk prefix agrees: amp_cache._BIAS_CHUNK): a 2 048-wide
    // bias with 256 partial rows on ONE workgroup was a 1 ms serial tail of the launch (round 4, once the encoder FFNs deferred their sums).
    ns &= (1LL << 40) - 1;
    const float* sf = reinterpret_cast<const float*>(srcs[lo]);
    __shared__ float accs[kBiasChunk];
    const long long cb = (blk - chunk_prefix[lo]) * kBiasChunk;
    const int cnt = (int)(count - cb < kBiasChunk ? count - cb : kBiasChunk);
    int groups = kBiasChunk / cnt;            // slices are dealt round-robin to the groups
    if (groups > ns) groups = (int)ns;
    for (int w = threadIdx.x; w < cnt * groups; w += 256) {
      const int e = w % cnt, gk = w / cnt;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      long long k = gk;
      for (; k + 3LL * groups < ns; k += 4LL * groups) {
        a0 += sf[k * stride + cb + e];
        a1 += sf[(k + groups) * stride + cb + e];
        a2 += sf[(k + 2LL * groups) * stride + cb + e];
        a3 += sf[(k + 3LL * groups) * stride + cb + e];
      }
      for (; k < ns; k += groups) a0 += sf[k * stride + cb + e];
      accs[w] = (a0 + a1) + (a2 + a3);        // slot (group, element): w = gk * cnt + e < kBiasChunk
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += 256) {        // the groups are folded in a FIXED order: bit-reproducible run to run
      float t = 0.f;
      for (int g = 0; g < groups; ++g) t += accs[g * cnt + i];
      d[cb + i] = from_f<D>(t);
    }
    return;
  }
  if (base >= count) return;
  const bool wide = base + 8 <= count && ((reinterpret_cast<uintptr_t>(s + base) | reinterpret_cast<uintptr_t>(d + base)) & 15) == 0 &&
                    (ns == 1 || (stride * (long long)sizeof(S)) % 16 == 0);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (wide) {
    // four slices in flight (round 4: with the encoder FFNs' 34-slice partial sums finished here too, the one-load-at-a-time loop was the
    // latency of ns dependent round trips per thread: 1 090 us for ~550 MB = 0.5 TB/s); slices are added in index order, one accumulator
    long long k = 0;
    for (; k + 4 <= ns; k += 4) {
      S in[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const S* sk = s + (k + u) * stride + base;
        if constexpr (sizeof(S) == 4) { *reinterpret_cast<uint4*>(in[u]) = *reinterpret_cast<const uint4*>(sk); *reinterpret_cast<uint4*>(in[u] + 4) = *reinterpret_cast<const uint4*>(sk + 4); }
        else *reinterpret_cast<uint4*>(in[u]) = *reinterpret_cast<const uint4*>(sk);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += to_f<S>(in[u][i]);
    }
    for (; k < ns; ++k) {
      S in[8];
      const S* sk = s + k * stride + base;
      if constexpr (sizeof(S) == 4) { *reinterpret_cast<uint4*>(in) = *reinterpret_cast<const uint4*>(sk); *reinterpret_cast<uint4*>(in + 4) = *reinterpret_cast<const uint4*>(sk + 4); }
      else *reinterpret_cast<uint4*>(in) = *reinterpret_cast<const uint4*>(sk);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += to_f<S>(in[i]);
    }
    D out[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = from_f<D>(acc[i]);
    if constexpr (sizeof(D) == 4) { *reinterpret_cast<uint4*>(d + base) = *reinterpret_cast<uint4*>(out); *reinterpret_cast<uint4*>(d + base + 4) = *reinterpret_cast<uint4*>(out + 4); }
    else *reinterpret_cast<uint4*>(d + base) = *reinterpret_cast<uint4*>(out);
  } else {
    for (long long i = base; i < base + 8 && i < count; ++i) {
      float a = 0.f;
      for (long long k = 0; k < ns; ++k) a += to_f<S>(s[k * stride + i]);
      d[i] = from_f<D>(a);
    }
  }
}

template <typename S, typename D>
__global__ __launch_bounds__(256) void multi_cast(const long long* __restrict__ srcs, const long long* __restrict__ dsts,
                                                  const long long* __restrict__ numels, const long long* __restrict__ chunk_prefix, int n) {
  const long long blk = blockIdx.x;
  int lo = 0, hi = n;                       // largest t with chunk_prefix[t] <= blk
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_prefix[mid] <= blk) lo = mid; else hi = mid;
  }
  const S* s = reinterpret_cast<const S*>(srcs[lo]);
  D* d = reinterpret_cast<D*>(dsts[lo]);
  const long long count = numels[lo];
  const long long base = (blk - chunk_prefix[lo]) * CHUNK + (long long)threadIdx.x * 8;
  if (base + 8 <= count && ((reinterpret_cast<uintptr_t>(s + base) | reinterpret_cast<uintptr_t>(d + base)) & 15) == 0) {
    S in[8];
    D out[8];
    if constexpr (sizeof(S) == 4) { *reinterpret_cast<uint4*>(in) = *reinterpret_cast<const uint4*>(s + base); *reinterpret_cast<uint4*>(in + 4) = *reinterpret_cast<const uint4*>(s + base + 4); }
    else *reinterpret_cast<uint4*>(in) = *reinterpret_cast<const uint4*>(s + base);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = from_f<D>(to_f<S>(in[i]));
    if constexpr (sizeof(D) == 4) { *reinterpret_cast<uint4*>(d + base) = *reinterpret_cast<uint4*>(out); *reinterpret_cast<uint4*>(d + base + 4) = *reinterpret_cast<uint4*>(out + 4); }
    else *reinterpret_cast<uint4*>(d + base) = *reinterpret_cast<uint4*>(out);
  } else {
    for (long long i = base; i < base + 8 && i < count; ++i) d[i] = from_f<D>(to_f<S>(s[i]));
  }
}

}  // namespace

extern "C" int ocpg_multi_cast(const long long* srcs, const long long* dsts, const long long* numels, const long long* chunk_prefix, int n,
                               long long total_chunks, int src_dtype, int dst_dtype, void* stream) {
  if (n < 0 || total_chunks < 0) return -1006;
  if (n == 0 || total_chunks == 0) return 0;
  if (!srcs || !dsts || !numels || !chunk_prefix) return -1001;
  if (total_chunks > 2147483647LL) return -1007;
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = (unsigned)total_chunks;
#define MC(S_, D_) multi_cast<S_, D_><<<g, 256, 0, st>>>(srcs, dsts, numels, chunk_prefix, n)
  if (src_dtype == 0 && dst_dtype == 1) MC(float, __hip_bfloat16);
  else if (src_dtype == 0 && dst_dtype == 2) MC(float, __half);
  else if (src_dtype == 1 && dst_dtype == 0) MC(__hip_bfloat16, float);
  else if (src_dtype == 2 && dst_dtype == 0) MC(__half, float);
  else return -1010;
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// Partial column sums of a row-major [R, C] matrix (bf16 / fp16 / fp32): workgroup s sums its block of rows for every column ->
// part [S, C] fp32, every element written.  The bias gradient of a Linear / 1x1 conv over many rows is the column sum of the output
// gradient; ATen's `sum(0)` is two launches (semaphore fill + reduction) per layer -- here ONE launch leaves S partial rows and the
// fused gradient cast (multi_cast_sum above) finishes the sum while it casts.
template <typename S>
__global__ __launch_bounds__(256) void colsum_partials(const S* __restrict__ x, long long R, int C, long long rows_per_block,
                                                       float* __restrict__ part) {
  const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
  constexpr int VEC = 16 / (int)sizeof(S);
  if (C % VEC == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    // 16-byte loads: thread = (vector column, row phase), TW vector columns side by side (a power of two <= the row's vectors),
    // 256 / TW rows per pass and four passes in flight; the row phases are folded through LDS.  (A thread per column with scalar
    // loads kept 4 two-byte loads in flight per lane: 17-34 us per call where the matrix takes 4-15 us to stream.)
    const int cv = C / VEC;
    int TW = 256;
    while (TW > cv) TW >>= 1;
    const int tcol = threadIdx.x % TW, trow = threadIdx.x / TW, RP = 256 / TW;
    __shared__ float red[256 * VEC];
    for (int v0 = 0; v0 < cv; v0 += TW) {
      const int v = v0 + tcol;
      float acc[VEC];
#pragma unroll
      for (int u = 0; u < VEC; ++u) acc[u] = 0.f;
      if (v < cv) {
        const S* col = x + (long long)v * VEC;
        long long r = r0 + trow;
        for (; r + 3LL * RP < r1; r += 4LL * RP) {
          uint4 q[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) q[k] = *reinterpret_cast<const uint4*>(col + (r + (long long)k * RP) * C);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const S* e = reinterpret_cast<const S*>(&q[k]);
#pragma unroll
            for (int u = 0; u < VEC; ++u) acc[u] += to_f<S>(e[u]);
          }
        }
        for (; r < r1; r += RP) {
          const uint4 q = *reinterpret_cast<const uint4*>(col + r * C);
          const S* e = reinterpret_cast<const S*>(&q);
#pragma unroll
          for (int u = 0; u < VEC; ++u) acc[u] += to_f<S>(e[u]);
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < VEC; ++u) red[(trow * TW + tcol) * VEC + u] = acc[u];
      __syncthreads();
      for (int i = threadIdx.x; i < TW * VEC; i += 256) {
        const int c = v0 * VEC + i;
        if (c < C) {
          float a = 0.f;
          for (int q = 0; q < RP; ++q) a += red[q * TW * VEC + i];
          part[(long long)blockIdx.x * C + c] = a;
        }
      }
    }
    return;
  }
  for (int c = threadIdx.x; c < C; c += 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    long long r = r0;
    for (; r + 4 <= r1; r += 4) {
      a0 += to_f<S>(x[r * C + c]); a1 += to_f<S>(x[(r + 1) * C + c]); a2 += to_f<S>(x[(r + 2) * C + c]); a3 += to_f<S>(x[(r + 3) * C + c]);
    }
    for (; r < r1; ++r) a0 += to_f<S>(x[r * C + c]);
    part[(long long)blockIdx.x * C + c] = (a0 + a1) + (a2 + a3);
  }
}

extern "C" long long ocpg_colsum_blocks(long long R) {
  if (R <= 0) return 0;
  const long long want = (R + 127) / 128;
  return want < 256 ? want : 256;          // the partial rows are folded by ONE workgroup per 2048 columns (multi_cast_sum): keep them few
}

extern "C" int ocpg_colsum_partials(const void* x, long long R, int C, int dtype, float* part, void* stream) {
  if (R < 0 || C <= 0) return -1002;
  if (R == 0) return 0;
  if (!x) return -1001;
  if (!part) return -1005;
  const long long blocks = ocpg_colsum_blocks(R), rpb = (R + blocks - 1) / blocks;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) colsum_partials<float><<<(unsigned)blocks, 256, 0, st>>>((const float*)x, R, C, rpb, part);
  else if (dtype == 1) colsum_partials<__hip_bfloat16><<<(unsigned)blocks, 256, 0, st>>>((const __hip_bfloat16*)x, R, C, rpb, part);
  else if (dtype == 2) colsum_partials<__half><<<(unsigned)blocks, 256, 0, st>>>((const __half*)x, R, C, rpb, part);
  else return -1004;
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ocpg_multi_cast_sum(const long long* srcs, const long long* dsts, const long long* numels, const long long* chunk_prefix,
                                   const long long* splits, const long long* strides, int n, long long total_chunks, int src_dtype,
                                   int dst_dtype, void* stream) {
  if (n < 0 || total_chunks < 0) return -1007;
  if (n == 0 || total_chunks == 0) return 0;
  if (!srcs || !dsts || !numels || !chunk_prefix) return -1001;
  if (!splits || !strides) return -1005;
  if (total_chunks > 2147483647LL) return -1008;
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = (unsigned)total_chunks;
#define MCS(S_, D_) multi_cast_sum<S_, D_><<<g, 256, 0, st>>>(srcs, dsts, numels, chunk_prefix, splits, strides, n)
  if (src_dtype == 1 && dst_dtype == 0) MCS(__hip_bfloat16, float);
  else if (src_dtype == 2 && dst_dtype == 0) MCS(__half, float);
  else if (src_dtype == 0 && dst_dtype == 0) MCS(float, float);
  else return -1011;
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
