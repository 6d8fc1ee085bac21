// MSDeformAttn forward / backward for MI355X (gfx950, wave64) -- hand-written HIP, no CUDA lineage.
//
// What it computes is fixed by the reference (models/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 forward,
// :87-159 + :301-403 backward); how it is mapped onto the machine is not:
//
//   * a "row" is one (batch b, query q, head m); its D channels are contiguous (value is [N,S,M,D]).
//     The fast path gives a row G = D/4 lanes, each lane owning one float4 (16 B) of the head's channels, so
//     a corner fetch of a head is ONE 16-B-per-lane load covering the head's whole 4*D-byte line, and a
//     wave64 carries 64/G rows (D=32: one query x 8 heads = the query's full 1-KiB output line).
//   * the per-sample scalar work (pixel coords, bounds tests, bilinear weights, corner offsets) is done ONCE
//     per row by the row's lanes in parallel (lane j takes samples j, j+G, ...), parked in LDS, and then
//     re-read as wave-broadcast ds_read_b128 -- the reference redoes it in every one of the D channel
//     threads (32x redundant for D=32) together with int64 shape loads.
//   * backward: grad_loc / grad_attn need a sum over the row's D channels; that is a G-lane DPP-free
//     shuffle reduction in registers instead of the reference's shared-memory + thread-0 serial loop
//     (cuh:366-381).  grad_value is a float atomic scatter (one dword per lane per instruction).
//   * no im2col_step chunk loop on the host: one launch per call.
//
// The generic kernels (any D, float or double) keep one wave per row with lanes striding over channels;
// they exist for the reference's test protocol (models/ops/test.py:85: D in {30,71,1025,2048,3096}, fp64).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include <cstdlib>

#include "../../include/ocpg_hip.h"
#include "msda_col.h"
#include "msda_tile.h"
#include "msda_dev.h"

namespace {

constexpr int kMaxLevels = 16;

// One precomputed sample, 32 bytes = two ds_read_b128.
struct __attribute__((aligned(16))) SampleRec {
  int off00;      // element offset (in scalars, relative to value[b, 0, m, 0]) of corner (y0, x0); may be "virtual" (negative) when that corner is outside
  int rowstride;  // W * M * D
  int mask;       // bit k set <=> corner k (0:(y0,x0) 1:(y0,x1) 2:(y1,x0) 3:(y1,x1)) is inside the map; 0 => sample skipped
  float a;        // attention weight
  float ly, lx;   // fractional parts
  float H, W;     // level size as float (grad_loc scaling)
};

template <typename T>
__device__ __forceinline__ void make_sample(T x_n, T y_n, T a, int H, int W, int lstart, int MD, SampleRec& r) {
  const T h_im = y_n * (T)H - (T)0.5;
  const T w_im = x_n * (T)W - (T)0.5;
  r.a = (float)a;
  r.H = (float)H;
  r.W = (float)W;
  r.rowstride = W * MD;
  if (h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W) {
    const int y0 = (int)floor(h_im), x0 = (int)floor(w_im);
    r.ly = (float)(h_im - (T)y0);
    r.lx = (float)(w_im - (T)x0);
    const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= H - 1, x0ok = x0 >= 0, x1ok = x0 + 1 <= W - 1;
    r.mask = (y0ok && x0ok ? 1 : 0) | (y0ok && x1ok ? 2 : 0) | (y1ok && x0ok ? 4 : 0) | (y1ok && x1ok ? 8 : 0);
    r.off00 = (lstart + y0 * W + x0) * MD;
  } else {
    r.ly = r.lx = 0.f;
    r.mask = 0;
    r.off00 = 0;
  }
}

using ocpg_dev::ld4;
using ocpg_dev::group_sum;
using ocpg_dev::reduce_scatter_g8_p4;

// ------------------------------------------------------------------------------------------------------
// Fast forward: D = 4*G, G in {1,2,4,8,16,32,64}.  256 threads = 256/G rows per block.
// LDS: rows_per_block * NS * 32 B (dynamic).
template <int G>
__global__ __launch_bounds__(256) void msda_fwd_fast(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                     const int64_t* __restrict__ level_start, const float* __restrict__ loc,
                                                     const float* __restrict__ attn, int S, int M, int L, int Lq, int P,
                                                     long long rows, float* __restrict__ out) {
  constexpr int D = 4 * G;
  constexpr int ROWS = 256 / G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  SampleRec* recs = reinterpret_cast<SampleRec*>(smem);
  __shared__ int lvlH[kMaxLevels], lvlW[kMaxLevels], lvlS[kMaxLevels];
  const int tid = threadIdx.x;
  if (tid < L) {
    lvlH[tid] = (int)shapes[2 * tid];
    lvlW[tid] = (int)shapes[2 * tid + 1];
    lvlS[tid] = (int)level_start[tid];
  }
  __syncthreads();
  const int NS = L * P;
  const int MD = M * D;
  const int r = tid / G, j = tid % G;
#ifndef MSDA_ROW_MAJOR
  // XCD-aware mapping: blocks are dealt round-robin over the 8 XCDs, so block % M picks the HEAD; with M == 8 every
  // XCD then gathers only its own head's 128-B slices of `value` (1/8 of the map) through its private 4-MiB L2.
  const long long qrow = (long long)(blockIdx.x / M) * ROWS + r;          // flat (b, q)
  const long long row = qrow * M + (blockIdx.x % M);
  const bool live = qrow * M < rows;
#else
  const long long row = (long long)blockIdx.x * ROWS + r;
  const bool live = row < rows;
#endif
  if (live) {
    const float* lrow = loc + row * NS * 2;
    const float* arow = attn + row * NS;
    for (int s = j; s < NS; s += G) {
      const int l = s / P;
      SampleRec rec;
      make_sample<float>(lrow[2 * s], lrow[2 * s + 1], arow[s], lvlH[l], lvlW[l], lvlS[l], MD, rec);
      recs[r * NS + s] = rec;
    }
  }
  __syncthreads();
  if (!live) return;
  const int m = (int)(row % M);
  const long long b = row / ((long long)Lq * M);
  const float* vbase = value + b * (long long)S * MD + m * D + 4 * j;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const SampleRec* rr = recs + r * NS;
  // (a 4-sample batched, branch-free variant of this loop measured SLOWER on MI355X: 84 vs 64 us at the encoder
  //  shape -- the gather is L1/TA-throughput bound, not latency bound, and the batch costs occupancy)
  int s = 0;
  for (; s < NS; ++s) {
    const SampleRec rec = rr[s];
    if (rec.mask == 0) continue;  // uniform across the row's lanes
    const float hy = 1.f - rec.ly, hx = 1.f - rec.lx;
    const float* p00 = vbase + rec.off00;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v1 = (rec.mask & 1) ? ld4(p00) : z;
    const float4 v2 = (rec.mask & 2) ? ld4(p00 + MD) : z;
    const float4 v3 = (rec.mask & 4) ? ld4(p00 + rec.rowstride) : z;
    const float4 v4 = (rec.mask & 8) ? ld4(p00 + rec.rowstride + MD) : z;
    const float w1 = hy * hx, w2 = hy * rec.lx, w3 = rec.ly * hx, w4 = rec.ly * rec.lx;
    acc.x += (w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x) * rec.a;
    acc.y += (w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y) * rec.a;
    acc.z += (w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z) * rec.a;
    acc.w += (w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w) * rec.a;
  }
  *reinterpret_cast<float4*>(out + row * D + 4 * j) = acc;
}

// ------------------------------------------------------------------------------------------------------
// Fused front end (round 4): the module's softmax over the L*P attention logits and `reference point + offset`
// (models/ops/modules/ms_deform_attn.py:96-110, the 2-d reference branch with the offsets already divided by (W_l, H_l) -- the
// caller folds that division into the projection's 256 weight rows) happen in the row's sample-setup phase, where the lanes compute
// per-sample scalars anyway.  Input: the merged query projection qproj [N*Lq, 3*M*NS] = [offsets (M, L, P, 2) | logits (M, L*P)] and
// ref [N*Lq, L, 2].  The sampling locations and attention weights are still WRITTEN (the backward kernels and the module's return
// value read them) but no longer produced by two elementwise passes and read back: -2 launches, -78 MB read per encoder layer at N = 10.
// D = 32 (G = 8), L*P = 16: lane j of a row owns samples j and j + 8.
__global__ __launch_bounds__(256) void msda_fwd_fused8(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ level_start, const float* __restrict__ qproj,
                                                       const float* __restrict__ ref, int S, int M, int L, int Lq, int P, long long rows,
                                                       float* __restrict__ out, float* __restrict__ loc_out, float* __restrict__ attn_out) {
  constexpr int G = 8, D = 32, ROWS = 256 / G, NS = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  SampleRec* recs = reinterpret_cast<SampleRec*>(smem);
  __shared__ int lvlH[kMaxLevels], lvlW[kMaxLevels], lvlS[kMaxLevels];
  const int tid = threadIdx.x;
  if (tid < L) {
    lvlH[tid] = (int)shapes[2 * tid];
    lvlW[tid] = (int)shapes[2 * tid + 1];
    lvlS[tid] = (int)level_start[tid];
  }
  __syncthreads();
  const int MD = M * D;
  const int r = tid / G, j = tid % G;
  const long long qrow = (long long)(blockIdx.x / M) * ROWS + r;          // flat (b, q); head = block % M (one head per XCD L2, see msda_fwd_fast)
  const int m = blockIdx.x % M;
  const long long row = qrow * M + m;
  const bool live = qrow * M < rows;
  if (live) {
    const int QW = 3 * M * NS;
    const float* qo = qproj + qrow * QW + m * NS * 2;
    const float* ql = qproj + qrow * QW + M * NS * 2 + m * NS;
    const float x0 = ql[j], x1 = ql[j + G];
    float mx = fmaxf(x0, x1);
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float e0 = expf(x0 - mx), e1 = expf(x1 - mx);
    const float den = group_sum<G>(e0 + e1);
    const float a[2] = {e0 / den, e1 / den};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int sidx = j + u * G, l = sidx / P;
      const float2 off = *reinterpret_cast<const float2*>(qo + 2 * sidx);
      const float2 rf = *reinterpret_cast<const float2*>(ref + (qrow * L + l) * 2);
      const float lx = rf.x + off.x, ly = rf.y + off.y;
      *reinterpret_cast<float2*>(loc_out + (row * NS + sidx) * 2) = make_float2(lx, ly);
      attn_out[row * NS + sidx] = a[u];
      SampleRec rec;
      make_sample<float>(lx, ly, a[u], lvlH[l], lvlW[l], lvlS[l], MD, rec);
      recs[r * NS + sidx] = rec;
    }
  }
  __syncthreads();
  if (!live) return;
  const long long b = row / ((long long)Lq * M);
  const float* vbase = value + b * (long long)S * MD + m * D + 4 * j;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const SampleRec* rr = recs + r * NS;
  for (int sidx = 0; sidx < NS; ++sidx) {
    const SampleRec rec = rr[sidx];
    if (rec.mask == 0) continue;  // uniform across the row's lanes
    const float hy = 1.f - rec.ly, hx = 1.f - rec.lx;
    const float* p00 = vbase + rec.off00;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v1 = (rec.mask & 1) ? ld4(p00) : z;
    const float4 v2 = (rec.mask & 2) ? ld4(p00 + MD) : z;
    const float4 v3 = (rec.mask & 4) ? ld4(p00 + rec.rowstride) : z;
    const float4 v4 = (rec.mask & 8) ? ld4(p00 + rec.rowstride + MD) : z;
    const float w1 = hy * hx, w2 = hy * rec.lx, w3 = rec.ly * hx, w4 = rec.ly * rec.lx;
    acc.x += (w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x) * rec.a;
    acc.y += (w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y) * rec.a;
    acc.z += (w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z) * rec.a;
    acc.w += (w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w) * rec.a;
  }
  *reinterpret_cast<float4*>(out + row * D + 4 * j) = acc;
}

// ------------------------------------------------------------------------------------------------------
// Generic forward: one wave per row, lanes stride over channels.  Any D, float or double.
template <typename T>
__global__ __launch_bounds__(256) void msda_fwd_generic(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                                        const int64_t* __restrict__ level_start, const T* __restrict__ loc,
                                                        const T* __restrict__ attn, int S, int M, int D, int L, int Lq, int P,
                                                        long long rows, T* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int NS = L * P;
  const long long MD = (long long)M * D;
  const int m = (int)(row % M);
  const long long b = row / ((long long)Lq * M);
  const T* vb = value + b * (long long)S * MD + (long long)m * D;
  for (int c = lane; c < D; c += 64) {
    T acc = 0;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const long long ls = level_start[l];
      for (int p = 0; p < P; ++p) {
        const long long wi = row * NS + (long long)l * P + p;
        const T w_im = loc[2 * wi] * (T)W - (T)0.5, h_im = loc[2 * wi + 1] * (T)H - (T)0.5;
        if (!(h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W)) continue;
        const int y0 = (int)floor(h_im), x0 = (int)floor(w_im);
        const T ly = h_im - (T)y0, lx = w_im - (T)x0, hy = (T)1 - ly, hx = (T)1 - lx;
        const T* p00 = vb + (ls + (long long)y0 * W + x0) * MD + c;
        const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= H - 1, x0ok = x0 >= 0, x1ok = x0 + 1 <= W - 1;
        const T v1 = (y0ok && x0ok) ? p00[0] : (T)0;
        const T v2 = (y0ok && x1ok) ? p00[MD] : (T)0;
        const T v3 = (y1ok && x0ok) ? p00[(long long)W * MD] : (T)0;
        const T v4 = (y1ok && x1ok) ? p00[(long long)W * MD + MD] : (T)0;
        acc += (hy * hx * v1 + hy * lx * v2 + ly * hx * v3 + ly * lx * v4) * attn[wi];
      }
    }
    out[row * D + c] = acc;
  }
}

// ------------------------------------------------------------------------------------------------------
// Gather side of the backward (grad_loc, grad_attn; cuh:87-159 without the col2im scatter): row kernel, used with the
// column-tile scatter kernel (msda_col.hip).  Differences from msda_bwd_fast<G, false>:
//   * validity is folded into the per-axis weights (hy' = y0 >= 0 ? 1-ly : 0 ...) and the corner addresses are clamped
//     into the map, so the inner loop has no masks and no selects: 4 unconditional 16-B loads, packed FMAs;
//   * 4 samples are reduced together by a DPP reduce-scatter (12 moves) instead of 9 LDS-crossbar shuffles per sample.
struct __attribute__((aligned(16))) GatherRec {
  int pk;          // element offset of corner (ya, xa) relative to value[b, 0, m, 0]  |  iy1<<3 | iy0<<2 | ix1<<1 | ix0
  int rowstride;   // W * M * D
  float aW, aH;    // attention weight * level width / height (grad_loc scaling)
  float hy, ly, hx, lx;   // masked by validity
};

// FUSED (round 4; G = 8, L*P = 16): the epilogue applies the backward of the module's softmax and writes the gradient of the merged query
// projection [N*Lq, 3*M*NS] = [d offsets | d logits] (gloc = that matrix, gattn unused) instead of grad_loc / grad_attn -- ATen's
// softmax backward and the `cat` behind the split of the projection (78 MB copied per encoder layer at N = 10) disappear.
template <int G, bool FUSED = false>
__global__ __launch_bounds__(256) void msda_bwd_gather_row(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                           const int64_t* __restrict__ level_start, const float* __restrict__ loc,
                                                           const float* __restrict__ attn, const float* __restrict__ gout, int S, int M,
                                                           int L, int Lq, int P, long long rows, float* __restrict__ gloc,
                                                           float* __restrict__ gattn) {
  constexpr int D = 4 * G;
  constexpr int ROWS = 256 / G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  GatherRec* recs = reinterpret_cast<GatherRec*>(smem);
  __shared__ int lvlH[kMaxLevels], lvlW[kMaxLevels], lvlS[kMaxLevels];
  const int tid = threadIdx.x;
  if (tid < L) {
    lvlH[tid] = (int)shapes[2 * tid];
    lvlW[tid] = (int)shapes[2 * tid + 1];
    lvlS[tid] = (int)level_start[tid];
  }
  __syncthreads();
  const int NS = L * P;
  const int MD = M * D;
  const int r = tid / G, j = tid % G;
  const long long qrow = (long long)(blockIdx.x / M) * ROWS + r;          // head fastest: one head per XCD L2 (see msda_fwd_fast)
  const long long row = qrow * M + (blockIdx.x % M);
  const bool live = qrow * M < rows;
  if (live) {
    const float* lrow = loc + row * NS * 2;
    const float* arow = attn + row * NS;
    for (int s = j; s < NS; s += G) {
      const int l = s / P;
      const int H = lvlH[l], W = lvlW[l];
      GatherRec rec;
      const float h_im = lrow[2 * s + 1] * (float)H - 0.5f, w_im = lrow[2 * s] * (float)W - 0.5f;
      const float a = arow[s];
      rec.rowstride = W * MD;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        const int y0 = (int)floorf(h_im), x0 = (int)floorf(w_im);
        const float ly = h_im - (float)y0, lx = w_im - (float)x0;
        const bool iy0 = y0 >= 0, iy1 = y0 + 1 <= H - 1, ix0 = x0 >= 0, ix1 = x0 + 1 <= W - 1;
        rec.hy = iy0 ? 1.f - ly : 0.f;
        rec.ly = iy1 ? ly : 0.f;
        rec.hx = ix0 ? 1.f - lx : 0.f;
        rec.lx = ix1 ? lx : 0.f;
        rec.aW = a * (float)W;
        rec.aH = a * (float)H;
        rec.pk = ((lvlS[l] + max(y0, 0) * W + max(x0, 0)) * MD) | (iy1 ? 8 : 0) | (iy0 ? 4 : 0) | (ix1 ? 2 : 0) | (ix0 ? 1 : 0);
      } else {
        rec.hy = rec.ly = rec.hx = rec.lx = rec.aW = rec.aH = 0.f;
        rec.pk = 0;
      }
      recs[r * NS + s] = rec;
    }
  }
  __syncthreads();
  if (!live) return;  // whole row groups leave together (G divides 64): the DPP exchanges below stay within live groups
  const int m = (int)(row % M);
  const long long b = row / ((long long)Lq * M);
  const float* vbase = value + b * (long long)S * MD + m * D + 4 * j;
  const float4 go = ld4(gout + row * D + 4 * j);
  const GatherRec* rr = recs + r * NS;
  constexpr int NB = 4;
  float fga[4] = {0.f, 0.f, 0.f, 0.f}, fgx[4] = {0.f, 0.f, 0.f, 0.f}, fgy[4] = {0.f, 0.f, 0.f, 0.f};      // FUSED: this lane pair's sample of each batch
  auto batch = [&](const int s0) {
    float red[NB][3];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const bool have = s0 + i < NS;
      const GatherRec rec = rr[have ? s0 + i : s0];
      const int bits = rec.pk & 15;
      const float* p00 = vbase + (rec.pk & ~15);
      const int dx = (bits & 3) == 3 ? MD : 0, dy = (bits & 12) == 12 ? rec.rowstride : 0;
      const float4 v0 = ld4(p00), v1 = ld4(p00 + dx), v2 = ld4(p00 + dy), v3 = ld4(p00 + dy + dx);
      // per-corner dot products with the output gradient first: the weights then act on 4 scalars, not on 4 x D channels
      const float d0 = go.x * v0.x + go.y * v0.y + go.z * v0.z + go.w * v0.w;
      const float d1 = go.x * v1.x + go.y * v1.y + go.z * v1.z + go.w * v1.w;
      const float d2 = go.x * v2.x + go.y * v2.y + go.z * v2.z + go.w * v2.w;
      const float d3 = go.x * v3.x + go.y * v3.y + go.z * v3.z + go.w * v3.w;
      const float ix0 = (bits & 1) ? 1.f : 0.f, ix1 = (bits & 2) ? 1.f : 0.f, iy0 = (bits & 4) ? 1.f : 0.f, iy1 = (bits & 8) ? 1.f : 0.f;
      const float top = rec.hx * d0 + rec.lx * d1, bot = rec.hx * d2 + rec.lx * d3;          // rows ya / yb, x-interpolated
      const float lef = ix1 * d1 - ix0 * d0, rig = ix1 * d3 - ix0 * d2;                      // d/dx along rows ya / yb
      const float ga = rec.hy * top + rec.ly * bot;
      const float gx = rec.aW * (rec.hy * lef + rec.ly * rig);
      const float gy = rec.aH * (iy1 * bot - iy0 * top);
      red[i][0] = have ? ga : 0.f;
      red[i][1] = have ? gx : 0.f;
      red[i][2] = have ? gy : 0.f;
    }
    if (FUSED) {
      float tot[3];
      (void)reduce_scatter_g8_p4(red, j, tot);       // the lane pair j >> 1 now holds sample s0 + (j >> 1)
#pragma unroll
      for (int k = 0; k < 4; ++k) {       // (selects, not an indexed store: the batch loop stays rolled -- fully unrolled it took 256 registers)
        const bool mine = s0 == NB * k;
        fga[k] = mine ? tot[0] : fga[k]; fgx[k] = mine ? tot[1] : fgx[k]; fgy[k] = mine ? tot[2] : fgy[k];
      }
    } else if (G == 8 && s0 + NB <= NS) {
      float tot[3];
      const int sidx = reduce_scatter_g8_p4(red, j, tot);
      if ((j & 1) == 0) {
        const long long wi = row * NS + s0 + sidx;
        gattn[wi] = tot[0];
        *reinterpret_cast<float2*>(gloc + wi * 2) = make_float2(tot[1], tot[2]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if (s0 + i >= NS) continue;
        const float ga = group_sum<G>(red[i][0]), gx = group_sum<G>(red[i][1]), gy = group_sum<G>(red[i][2]);
        if (j == 0) {
          const long long wi = row * NS + s0 + i;
          gattn[wi] = ga;
          *reinterpret_cast<float2*>(gloc + wi * 2) = make_float2(gx, gy);
        }
      }
    }
  };
#pragma unroll 1
  for (int s0 = 0; s0 < NS; s0 += NB) batch(s0);
  if (FUSED) {
    // softmax backward over the row's 16 weights: d logit_s = a_s (ga_s - sum_t a_t ga_t); the even lane of pair p holds samples 4k + p
    const int p = j >> 1;
    float av[4], dotp = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      av[k] = attn[row * 16 + 4 * k + p];
      dotp += (j & 1) ? 0.f : av[k] * fga[k];
    }
    const float dot = group_sum<G>(dotp);
    if ((j & 1) == 0) {
      const int QW = 3 * M * 16;
      float* go_ = gloc + qrow * QW + (row - qrow * M) * 32;
      float* gl_ = gloc + qrow * QW + M * 32 + (row - qrow * M) * 16;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int sidx = 4 * k + p;
        *reinterpret_cast<float2*>(go_ + 2 * sidx) = make_float2(fgx[k], fgy[k]);
        gl_[sidx] = av[k] * (fga[k] - dot);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Fast backward (plain float-atomic scatter for grad_value).
template <int G, bool SCATTER = true>
__global__ __launch_bounds__(256) void msda_bwd_fast(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                     const int64_t* __restrict__ level_start, const float* __restrict__ loc,
                                                     const float* __restrict__ attn, const float* __restrict__ gout, int S, int M,
                                                     int L, int Lq, int P, long long rows, float* __restrict__ gvalue,
                                                     float* __restrict__ gloc, float* __restrict__ gattn) {
  constexpr int D = 4 * G;
  constexpr int ROWS = 256 / G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  SampleRec* recs = reinterpret_cast<SampleRec*>(smem);
  __shared__ int lvlH[kMaxLevels], lvlW[kMaxLevels], lvlS[kMaxLevels];
  const int tid = threadIdx.x;
  if (tid < L) {
    lvlH[tid] = (int)shapes[2 * tid];
    lvlW[tid] = (int)shapes[2 * tid + 1];
    lvlS[tid] = (int)level_start[tid];
  }
  __syncthreads();
  const int NS = L * P;
  const int MD = M * D;
  const int r = tid / G, j = tid % G;
#ifndef MSDA_ROW_MAJOR
  // XCD-aware mapping: blocks are dealt round-robin over the 8 XCDs, so block % M picks the HEAD; with M == 8 every
  // XCD then gathers only its own head's 128-B slices of `value` (1/8 of the map) through its private 4-MiB L2.
  const long long qrow = (long long)(blockIdx.x / M) * ROWS + r;          // flat (b, q)
  const long long row = qrow * M + (blockIdx.x % M);
  const bool live = qrow * M < rows;
#else
  const long long row = (long long)blockIdx.x * ROWS + r;
  const bool live = row < rows;
#endif
  if (live) {
    const float* lrow = loc + row * NS * 2;
    const float* arow = attn + row * NS;
    for (int s = j; s < NS; s += G) {
      const int l = s / P;
      SampleRec rec;
      make_sample<float>(lrow[2 * s], lrow[2 * s + 1], arow[s], lvlH[l], lvlW[l], lvlS[l], MD, rec);
      recs[r * NS + s] = rec;
    }
  }
  __syncthreads();
  if (!live) return;  // whole row groups leave together (G divides 64): shuffles below stay within live groups
  const int m = (int)(row % M);
  const long long b = row / ((long long)Lq * M);
  const long long boff = b * (long long)S * MD + m * D + 4 * j;
  const float* vbase = value + boff;
  float* gsc = gvalue + b * (long long)S * MD + m * D + j;
  const float4 go = ld4(gout + row * D + 4 * j);
  const float gs[4] = {gout[row * D + j], gout[row * D + j + G], gout[row * D + j + 2 * G], gout[row * D + j + 3 * G]};
  const SampleRec* rr = recs + r * NS;
  for (int s = 0; s < NS; ++s) {
    const SampleRec rec = rr[s];
    float ga = 0.f, gx = 0.f, gy = 0.f;
    if (rec.mask != 0) {
      const float hy = 1.f - rec.ly, hx = 1.f - rec.lx;
      const float w[4] = {hy * hx, hy * rec.lx, rec.ly * hx, rec.ly * rec.lx};
      const float dyc[4] = {-hx, -rec.lx, hx, rec.lx};
      const float dxc[4] = {-hy, hy, -rec.ly, rec.ly};
      const int offs[4] = {0, MD, rec.rowstride, rec.rowstride + MD};
      const float4 tg = make_float4(go.x * rec.a, go.y * rec.a, go.z * rec.a, go.w * rec.a);  // top_grad * attn_weight
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 dxs = val, dys = val;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (rec.mask & (1 << k)) {
          const float4 v = ld4(vbase + rec.off00 + offs[k]);
          if constexpr (SCATTER) {
            float* g = gsc + rec.off00 + offs[k];        // lane j owns channels {j, j+G, j+2G, j+3G}: contiguous 4G-byte segments
#pragma unroll
            for (int c = 0; c < 4; ++c) atomicAdd(g + c * G, w[k] * gs[c] * rec.a);
          }
          val.x += w[k] * v.x; val.y += w[k] * v.y; val.z += w[k] * v.z; val.w += w[k] * v.w;
          dxs.x += dxc[k] * v.x; dxs.y += dxc[k] * v.y; dxs.z += dxc[k] * v.z; dxs.w += dxc[k] * v.w;
          dys.x += dyc[k] * v.x; dys.y += dyc[k] * v.y; dys.z += dyc[k] * v.z; dys.w += dyc[k] * v.w;
        }
      }
      ga = go.x * val.x + go.y * val.y + go.z * val.z + go.w * val.w;
      gx = rec.W * (dxs.x * tg.x + dxs.y * tg.y + dxs.z * tg.z + dxs.w * tg.w);
      gy = rec.H * (dys.x * tg.x + dys.y * tg.y + dys.z * tg.z + dys.w * tg.w);
    }
    ga = group_sum<G>(ga);
    gx = group_sum<G>(gx);
    gy = group_sum<G>(gy);
    if (j == 0) {
      gattn[row * NS + s] = ga;
      *reinterpret_cast<float2*>(gloc + (row * NS + s) * 2) = make_float2(gx, gy);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Tiled backward for self-attention over the value's own pixels (Lq == S, the encoder): LDS-privatised grad_value.
//
// Float atomics execute at the memory side at ~1.3 TB/s chip-wide (MI355X_MICROARCH.md, "Global float atomics"); the
// plain scatter above issues 4*L*P*D*4 = 8 KiB of atomic bytes per (query, head) -- 1.67 GB per encoder layer at
// config #2 -- and is bound by exactly that.  Here one workgroup owns a TILE x TILE patch of query pixels of one
// level for ONE head.  Neighbouring queries sample neighbouring pixels, so for every destination level the
// patch's samples fall into a small window whose origin is found from the data (min corner over the patch) and
// whose size the host derives from the patch extent plus a margin.  Contributions inside the window are summed in
// LDS (ds_add_f32); the window is flushed once with global atomics (one 128-B segment per pixel and head); anything
// outside the window falls back to a direct global atomic, so the result never depends on the locality
// assumption -- only the speed does.
// The LDS accumulators are DOUBLES: measured on gfx950 (tools/ubench/lds_atomic.hip) a ds_add_f32 wave-instruction
// costs ~193 cycles per CU (lane-serial), ds_add_f64 ~9, ds_add_u32 ~5 -- f64 is 20x faster than f32 and also makes
// the in-window sum more accurate than the reference's fp32 atomics; it is rounded to fp32 once at the flush.
constexpr int kTile = 8;
constexpr int kMaxTileLevels = 8;

struct TileGeom {
  int L;
  int ntiles;                               // tiles per (b, m)
  int tile_base[kMaxTileLevels + 1];        // first tile id of each query level
  int tiles_x[kMaxTileLevels];              // tiles per row of each query level
  int win_w[kMaxTileLevels][kMaxTileLevels];    // [query level][dest level] window width  (pixels)
  int win_h[kMaxTileLevels][kMaxTileLevels];
  int win_off[kMaxTileLevels][kMaxTileLevels];  // window start in the LDS accumulator (pixels)
  int win_pixels[kMaxTileLevels];           // total window pixels of a query level
};

struct __attribute__((aligned(16))) TileRec {
  int off00;     // element offset of corner (y0,x0) relative to value[b,0,m,0] (may be virtual)
  int lvl_mask;  // level << 4 | corner mask
  float a, ly, lx;
  int x0, y0;
  int pad;
};

#ifndef MSDA_WPE
#define MSDA_WPE 3      // waves per SIMD the register allocator is held to (152 VGPRs unconstrained = 3)
#endif
template <int G, int NB, bool GATHER = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MSDA_WPE, 8))) void msda_bwd_tiled(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                      const int64_t* __restrict__ level_start, const float* __restrict__ loc,
                                                      const float* __restrict__ attn, const float* __restrict__ gout, int S, int M,
                                                      int P, TileGeom geo, float* __restrict__ gvalue, float* __restrict__ gloc,
                                                      float* __restrict__ gattn) {
  constexpr int D = 4 * G;
  constexpr int ROWS = 256 / G;                      // queries processed per pass
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int lvlH[kMaxTileLevels], lvlW[kMaxTileLevels], lvlS[kMaxTileLevels];
  __shared__ int win_ox[kMaxTileLevels], win_oy[kMaxTileLevels];
  const int L = geo.L;
  const int NS = L * P;
  TileRec* recs = reinterpret_cast<TileRec*>(smem);                                  // [ROWS][P]  (one level at a time)
  double* acc = reinterpret_cast<double*>(smem + (size_t)ROWS * P * sizeof(TileRec));  // [window pixels][D], fp64: see below
  const int tid = threadIdx.x;
  if (tid < L) {
    lvlH[tid] = (int)shapes[2 * tid];
    lvlW[tid] = (int)shapes[2 * tid + 1];
    lvlS[tid] = (int)level_start[tid];
    win_ox[tid] = 0x7fffffff;
    win_oy[tid] = 0x7fffffff;
  }
  // decode (b, m, tile): all uniform
  int bid = blockIdx.x;
#ifndef MSDA_ROW_MAJOR
  const int m = bid % M;          // head fastest: block % 8 == XCD (round-robin dispatch) -> one head per XCD L2
  bid /= M;
  const int tile = bid % geo.ntiles;
  const int b = bid / geo.ntiles;
#else
  const int tile = bid % geo.ntiles;
  bid /= geo.ntiles;
  const int m = bid % M;
  const int b = bid / M;
#endif
  int lq = 0;
  while (lq + 1 < L && tile >= geo.tile_base[lq + 1]) ++lq;
  const int t_in = tile - geo.tile_base[lq];
  const int ty0 = (t_in / geo.tiles_x[lq]) * kTile, tx0 = (t_in % geo.tiles_x[lq]) * kTile;
  __syncthreads();
  const int Hq = lvlH[lq], Wq = lvlW[lq], Sq = lvlS[lq];
  const int th = min(kTile, Hq - ty0), tw = min(kTile, Wq - tx0);
  const int nq = th * tw;                                                 // queries in this tile (<= 64)
  const int MD = M * D;

  // pass A: window origins = min (y0, x0) over the tile's valid samples, per destination level.  With 256 % NS == 0 a lane
  // sees the same sample slot (hence level) in every iteration: running minimum in registers, one LDS atomic pair per lane
  {
    const bool fixed = (256 % NS) == 0;
    int my = 0x7fffffff, mx = 0x7fffffff;
    for (int i = tid; i < nq * NS; i += 256) {
      const int qi = i / NS, s = i % NS, l = s / P;
      const int q = Sq + (ty0 + qi / tw) * Wq + tx0 + qi % tw;
      const long long row = ((long long)b * S + q) * M + m;
      const float2 xy = *reinterpret_cast<const float2*>(loc + (row * NS + s) * 2);
      const int H = lvlH[l], W = lvlW[l];
      const float h_im = xy.y * (float)H - 0.5f, w_im = xy.x * (float)W - 0.5f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        const int y0 = max((int)floorf(h_im), 0), x0 = max((int)floorf(w_im), 0);
        if (fixed) { my = min(my, y0); mx = min(mx, x0); }
        else { atomicMin(&win_oy[l], y0); atomicMin(&win_ox[l], x0); }
      }
    }
    if (fixed && my != 0x7fffffff) {
      const int l = (tid % NS) / P;
      atomicMin(&win_oy[l], my);
      atomicMin(&win_ox[l], mx);
    }
  }
  __syncthreads();

  const int r = tid / G, j = tid % G;
  const long long boff = (long long)b * S * MD + m * D + 4 * j;
  // the output gradient of this lane's query in every pass: loaded ONCE (it does not depend on the level)
  constexpr int PASSES = (kTile * kTile + ROWS - 1) / ROWS;
  float4 go_p[PASSES];
  float gs_p[PASSES][4];
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int qi = ps * ROWS + r;
    go_p[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
    gs_p[ps][0] = gs_p[ps][1] = gs_p[ps][2] = gs_p[ps][3] = 0.f;
    if (qi < nq) {
      const int q = Sq + (ty0 + qi / tw) * Wq + tx0 + qi % tw;
      const long long row = ((long long)b * S + q) * M + m;
      if constexpr (GATHER) go_p[ps] = ld4(gout + row * D + 4 * j);
      gs_p[ps][0] = gout[row * D + j]; gs_p[ps][1] = gout[row * D + j + G]; gs_p[ps][2] = gout[row * D + j + 2 * G]; gs_p[ps][3] = gout[row * D + j + 3 * G];
    }
  }
  // (level, pass) steps; the raw (location, weight) of the NEXT step's records are requested before this step's main loop and
  // consumed after it, so their latency hides behind the scatter/gather work instead of standing alone between two barriers
  const int npass = (nq + ROWS - 1) / ROWS;
  const int nsteps = L * npass;
  const bool pf_ok = ROWS * P <= 256;        // one record per lane per step
  float2 pxy = make_float2(0.f, 0.f);
  float pa = 0.f;
  auto fetch = [&](int st) {
    const int l_ = st / npass, q0_ = (st % npass) * ROWS;
    const int nrow_ = min(ROWS, nq - q0_);
    if (tid < nrow_ * P) {
      const int qi = q0_ + tid / P;
      const int q = Sq + (ty0 + qi / tw) * Wq + tx0 + qi % tw;
      const long long row = ((long long)b * S + q) * M + m;
      const int s = l_ * P + tid % P;
      pxy = *reinterpret_cast<const float2*>(loc + (row * NS + s) * 2);
      pa = attn[row * NS + s];
    }
  };
  if (pf_ok && nsteps > 0) fetch(0);
  int H = 0, W = 0, ww = 0, wh = 0, ox = 0, oy = 0, rowstride = 0;
  const float* vbase = value;
  float* gvalue_l = gvalue;
  for (int st = 0; st < nsteps; ++st) {              // one destination level at a time: one LDS window live
    const int l = st / npass, pass = st % npass, q0 = pass * ROWS;
    if (pass == 0) {
      H = lvlH[l]; W = lvlW[l];
      ww = geo.win_w[lq][l]; wh = geo.win_h[lq][l];
      ox = (win_ox[l] == 0x7fffffff) ? 0 : max(0, min(win_ox[l], W - ww));   // keep the window inside the map
      oy = (win_oy[l] == 0x7fffffff) ? 0 : max(0, min(win_oy[l], H - wh));
      rowstride = W * MD;
      vbase = value + boff + (long long)lvlS[l] * MD;
      gvalue_l = gvalue + (long long)b * S * MD + m * D + (long long)lvlS[l] * MD;
      for (int i = tid; i < ww * wh * D; i += 256) acc[i] = 0.0;
    }
    {
      const int nrow = min(ROWS, nq - q0);
      for (int i = tid; i < nrow * P; i += 256) {     // records of this step: P samples per query
        float2 xy;
        float a_;
        if (pf_ok) { xy = pxy; a_ = pa; }
        else {
          const int qi = q0 + i / P;
          const int q = Sq + (ty0 + qi / tw) * Wq + tx0 + qi % tw;
          const long long row = ((long long)b * S + q) * M + m;
          const int s = l * P + i % P;
          xy = *reinterpret_cast<const float2*>(loc + (row * NS + s) * 2);
          a_ = attn[row * NS + s];
        }
        const float h_im = xy.y * (float)H - 0.5f, w_im = xy.x * (float)W - 0.5f;
        TileRec rec;
        rec.a = a_;
        rec.pad = 0;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
          const int y0 = (int)floorf(h_im), x0 = (int)floorf(w_im);
          rec.ly = h_im - (float)y0;
          rec.lx = w_im - (float)x0;
          const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= H - 1, x0ok = x0 >= 0, x1ok = x0 + 1 <= W - 1;
          rec.lvl_mask = (y0ok && x0ok ? 1 : 0) | (y0ok && x1ok ? 2 : 0) | (y1ok && x0ok ? 4 : 0) | (y1ok && x1ok ? 8 : 0);
          rec.off00 = (y0 * W + x0) * MD;
          rec.x0 = x0;
          rec.y0 = y0;
        } else {
          rec.ly = rec.lx = 0.f;
          rec.lvl_mask = 0;
          rec.off00 = 0;
          rec.x0 = rec.y0 = 0;
        }
        recs[i] = rec;
      }
      if (pf_ok && st + 1 < nsteps) fetch(st + 1);
      __syncthreads();
      if (r < nrow) {     // whole G-lane groups take the branch together
        const int qi = q0 + r;
        const int q = Sq + (ty0 + qi / tw) * Wq + tx0 + qi % tw;
        const long long row = ((long long)b * S + q) * M + m;
        float4 go = go_p[0];
        float gs[4] = {gs_p[0][0], gs_p[0][1], gs_p[0][2], gs_p[0][3]};
#pragma unroll
        for (int ps = 1; ps < PASSES; ++ps)
          if (pass == ps) { go = go_p[ps]; gs[0] = gs_p[ps][0]; gs[1] = gs_p[ps][1]; gs[2] = gs_p[ps][2]; gs[3] = gs_p[ps][3]; }
        const TileRec* rr = recs + r * P;
        const int offs[4] = {0, MD, rowstride, rowstride + MD};
        for (int p0 = 0; p0 < P; p0 += NB) {
          TileRec rec[NB];
          float4 v[NB][4];
          float red[NB][3];
#pragma unroll
          for (int i = 0; i < NB; ++i) rec[i] = rr[p0 + i];
          if constexpr (GATHER) {
#pragma unroll
            for (int i = 0; i < NB; ++i)    // all corner loads of the batch in flight before the first use
#pragma unroll
              for (int k = 0; k < 4; ++k) v[i][k] = ld4((rec[i].lvl_mask & (1 << k)) ? vbase + rec[i].off00 + offs[k] : vbase);
          }
#pragma unroll
          for (int i = 0; i < NB; ++i) {
            const int mask = rec[i].lvl_mask;
            const float hy = 1.f - rec[i].ly, hx = 1.f - rec[i].lx;
            const float w[4] = {hy * hx, hy * rec[i].lx, rec[i].ly * hx, rec[i].ly * rec[i].lx};
            const float dyc[4] = {-hx, -rec[i].lx, hx, rec[i].lx};
            const float dxc[4] = {-hy, hy, -rec[i].ly, rec[i].ly};
            const float a = mask ? rec[i].a : 0.f;
            const float4 tg = make_float4(go.x * a, go.y * a, go.z * a, go.w * a);
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 dxs = val, dys = val;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (mask & (1 << k)) {
                const int px = rec[i].x0 + (k & 1) - ox, py = rec[i].y0 + (k >> 1) - oy;
                // scatter: lane j owns channels {j, j+G, j+2G, j+3G} here, so one atomic instruction covers G
                // CONSECUTIVE floats per row (contiguous 4G-byte segments in memory, G consecutive LDS banks)
                if (px >= 0 && px < ww && py >= 0 && py < wh) {      // inside the privatised window: ds_add_f32
                  const int pp = py * ww + px;
                  double* gl_ = acc + pp * D + j;                     // bank-group swizzle: channel group i sits in slot (i+pp)&3
#ifndef EXP_NO_LDS_ATOMIC
#pragma unroll
                  for (int c = 0; c < 4; ++c) atomicAdd(gl_ + ((c + pp) & 3) * G, (double)(w[k] * gs[c] * a));   // ds_add_f64
#else
                  asm volatile("" ::"v"(gl_), "v"(w[k] * gs[0] * a));
#endif
                } else {                                              // outside: straight to memory
                  float* g = gvalue_l + rec[i].off00 + offs[k] + j;
#pragma unroll
                  for (int c = 0; c < 4; ++c) atomicAdd(g + c * G, w[k] * gs[c] * a);
                }
                if constexpr (GATHER) {
                  const float4 vv = v[i][k];
                  val.x += w[k] * vv.x; val.y += w[k] * vv.y; val.z += w[k] * vv.z; val.w += w[k] * vv.w;
                  dxs.x += dxc[k] * vv.x; dxs.y += dxc[k] * vv.y; dxs.z += dxc[k] * vv.z; dxs.w += dxc[k] * vv.w;
                  dys.x += dyc[k] * vv.x; dys.y += dyc[k] * vv.y; dys.z += dyc[k] * vv.z; dys.w += dyc[k] * vv.w;
                }
              }
            }
            red[i][0] = mask ? go.x * val.x + go.y * val.y + go.z * val.z + go.w * val.w : 0.f;
            red[i][1] = (float)W * (dxs.x * tg.x + dxs.y * tg.y + dxs.z * tg.z + dxs.w * tg.w);
            red[i][2] = (float)H * (dys.x * tg.x + dys.y * tg.y + dys.z * tg.z + dys.w * tg.w);
          }
          if constexpr (!GATHER) {
            // scatter-only instantiation: grad_loc / grad_attn come from the gather-only kernel
          } else if constexpr (G == 8 && NB == 4) {
            float tot[3];
            const int sidx = reduce_scatter_g8_p4(red, j, tot);
            if ((j & 1) == 0) {
              const long long wi = row * NS + l * P + p0 + sidx;
              gattn[wi] = tot[0];
              *reinterpret_cast<float2*>(gloc + wi * 2) = make_float2(tot[1], tot[2]);
            }
          } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
              const float ga = group_sum<G>(red[i][0]), gx = group_sum<G>(red[i][1]), gy = group_sum<G>(red[i][2]);
              if (j == 0) {
                const long long wi = row * NS + l * P + p0 + i;
                gattn[wi] = ga;
                *reinterpret_cast<float2*>(gloc + wi * 2) = make_float2(gx, gy);
              }
            }
          }
        }
      }
      __syncthreads();
    }
    if (pass == npass - 1) {
      // flush this level's window: one global atomic per touched (pixel, channel)
      float* gl = gvalue + (long long)b * S * MD + (long long)lvlS[l] * MD + m * D;
      for (int i = tid; i < ww * wh * D; i += 256) {
        const float v = (float)acc[i];
        if (v != 0.f) {
          const int pp = i / D, slot = (i % D) / G, jj = i % G;
          const int c = jj + G * ((slot - pp) & 3);                  // undo the bank-group swizzle
#ifndef EXP_NO_FLUSH
          atomicAdd(gl + ((long long)(oy + pp / ww) * W + ox + pp % ww) * MD + c, v);
#endif
        }
      }
      __syncthreads();
    }
  }
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Generic backward: one wave per row; any D; float or double.
template <typename T>
__global__ __launch_bounds__(256) void msda_bwd_generic(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                                        const int64_t* __restrict__ level_start, const T* __restrict__ loc,
                                                        const T* __restrict__ attn, const T* __restrict__ gout, int S, int M, int D,
                                                        int L, int Lq, int P, long long rows, T* __restrict__ gvalue,
                                                        T* __restrict__ gloc, T* __restrict__ gattn) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;  // wave-uniform
  const int NS = L * P;
  const long long MD = (long long)M * D;
  const int m = (int)(row % M);
  const long long b = row / ((long long)Lq * M);
  const long long boff = b * (long long)S * MD + (long long)m * D;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const long long ls = level_start[l];
    for (int p = 0; p < P; ++p) {
      const long long wi = row * NS + (long long)l * P + p;
      const T w_im = loc[2 * wi] * (T)W - (T)0.5, h_im = loc[2 * wi + 1] * (T)H - (T)0.5;
      T ga = 0, gx = 0, gy = 0;
      if (h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W) {
        const int y0 = (int)floor(h_im), x0 = (int)floor(w_im);
        const T ly = h_im - (T)y0, lx = w_im - (T)x0, hy = (T)1 - ly, hx = (T)1 - lx;
        const T a = attn[wi];
        const bool ok[4] = {y0 >= 0 && x0 >= 0, y0 >= 0 && x0 + 1 <= W - 1, y0 + 1 <= H - 1 && x0 >= 0,
                            y0 + 1 <= H - 1 && x0 + 1 <= W - 1};
        const T w[4] = {hy * hx, hy * lx, ly * hx, ly * lx};
        const T dyc[4] = {-hx, -lx, hx, lx};
        const T dxc[4] = {-hy, hy, -ly, ly};
        const long long o00 = boff + (ls + (long long)y0 * W + x0) * MD;
        const long long offs[4] = {0, MD, (long long)W * MD, (long long)W * MD + MD};
        for (int c = lane; c < D; c += 64) {
          const T tg = gout[row * D + c];
          const T tga = tg * a;
          T val = 0, dx = 0, dy = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (ok[k]) {
              const T v = value[o00 + offs[k] + c];
              atomicAdd(gvalue + o00 + offs[k] + c, w[k] * tga);
              val += w[k] * v;
              dx += dxc[k] * v;
              dy += dyc[k] * v;
            }
          }
          ga += tg * val;
          gx += (T)W * dx * tga;
          gy += (T)H * dy * tga;
        }
      }
      ga = wave_sum<T>(ga);
      gx = wave_sum<T>(gx);
      gy = wave_sum<T>(gy);
      if (lane == 0) {
        gattn[wi] = ga;
        gloc[2 * wi] = gx;
        gloc[2 * wi + 1] = gy;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Host side of the tiled backward: window sizes per (query level, destination level) from the level shapes.
// margin: how far (in destination pixels) beyond the patch's own footprint the window extends; the origin is
// data-driven, so the margin only has to cover the SPREAD of the offsets, not their common shift.
inline bool make_tile_geom(const int64_t* sh, int L, int D, int rows_per_pass, int P, TileGeom& g, size_t& lds_bytes) {
  const int margin = 3;
#ifndef MSDA_LDS_CAP_KB
#define MSDA_LDS_CAP_KB 52
#endif
  const size_t cap = MSDA_LDS_CAP_KB * 1024;            // LDS budget per workgroup: 52 KB -> 3 workgroups (12 waves) per CU
  const size_t rec_bytes = (size_t)rows_per_pass * P * sizeof(TileRec);
  g.L = L;
  int base = 0;
  int max_pix = 0;
  for (int lq = 0; lq < L; ++lq) {
    const int Hq = (int)sh[2 * lq], Wq = (int)sh[2 * lq + 1];
    if (Hq <= 0 || Wq <= 0) return false;
    g.tile_base[lq] = base;
    g.tiles_x[lq] = (Wq + kTile - 1) / kTile;
    base += g.tiles_x[lq] * ((Hq + kTile - 1) / kTile);
    int off = 0;
    for (int l = 0; l < L; ++l) {
      const int H = (int)sh[2 * l], W = (int)sh[2 * l + 1];
      const int fw = (std::min(kTile, Wq) * W + Wq - 1) / Wq, fh = (std::min(kTile, Hq) * H + Hq - 1) / Hq;   // patch footprint
      int ww = std::min(W, fw + 2 * margin + 2), wh = std::min(H, fh + 2 * margin + 2);
      while ((size_t)ww * wh * D * sizeof(double) + rec_bytes > cap) {     // shrink to the budget (more fallback atomics, same result)
        if (ww >= wh && ww > 1) --ww; else if (wh > 1) --wh; else return false;
      }
      g.win_w[lq][l] = ww;
      g.win_h[lq][l] = wh;
      g.win_off[lq][l] = 0;
      off += ww * wh;
      max_pix = std::max(max_pix, ww * wh);
    }
    g.win_pixels[lq] = off;
  }
  g.tile_base[L] = base;
  g.ntiles = base;
  lds_bytes = rec_bytes + (size_t)max_pix * D * sizeof(double);
  return lds_bytes <= cap;
}

inline int fast_group(int D) {
  if (D % 4) return 0;
  const int g = D / 4;
  if (g < 1 || g > 64 || (g & (g - 1))) return 0;
  return g;
}

inline int check_common(const void* a, const void* b, const void* c, const void* d, const void* e, int N, int S, int M, int D,
                        int L, int Lq, int P) {
  if (N < 0) return -1006;
  if (S <= 0) return -1007;
  if (M <= 0) return -1008;
  if (D <= 0) return -1009;
  if (L <= 0) return -1010;
  if (Lq < 0) return -1011;
  if (P <= 0) return -1012;
  if ((long long)N * Lq == 0) return 0;  // empty problem: pointers may legitimately be null
  if (!a) return -1001;
  if (!b) return -1002;
  if (!c) return -1003;
  if (!d) return -1004;
  if (!e) return -1005;
  return 0;
}

// OCPG_MSDA_COL=0 keeps the row / tiled kernels (A/B timing and parity of the older paths); default: column kernels
inline bool col_enabled() {
  const char* e = std::getenv("OCPG_MSDA_COL");      // read per call: tests toggle it in-process
  return !(e && e[0] == '0');
}

// The column-tile FORWARD (LDS-staged value windows) is correct but, at config #2, slower than the row kernel
// (185 vs 100 us: DESIGN.md section 4.1): opt-in with OCPG_MSDA_FWD=col.
inline bool col_fwd_enabled() {
  const char* e = std::getenv("OCPG_MSDA_FWD");
  return col_enabled() && e && e[0] == 'c';
}

inline int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ---- the two halves of the self-attention backward (also exported on their own: include/ocpg_hip.h) -----------------
// 1 = launched, 0 = shape not served by these kernels
inline int launch_bwd_value_col(const float* loc, const float* attn, const float* grad_out, const int64_t* shapes_host, int N, int S,
                                int M, int D, int L, int Lq, int P, float* grad_value, hipStream_t st) {
  const int G = fast_group(D);
  if (!(shapes_host && Lq == S && col_enabled() && (G == 4 || G == 8) && (long long)S * M * D < (1LL << 31))) return 0;
  // round 3: output-tiled kernels (plain stores for the fine levels, no halo re-flush: csrc/msda_tile.hip).  Measured at config #2,
  // N = 10 (tools/bench_msda_gv.py, profiles/r03_msda_gv_paths.json): 261-280 us against the column scatter's 213 us on the model's
  // INITIAL ring offsets, 334-364 against 502 us on perturbed ("trained") offsets -- opt-in (OCPG_MSDA_TILE=1) until it wins both.
  const char* te = std::getenv("OCPG_MSDA_TILE");
  if (te && te[0] == '1' && ocpg_tile::bwd_value_tile(loc, attn, grad_out, shapes_host, N, S, M, D, L, P, grad_value, st)) return 1;
  ocpg_col::ColGeom cg;
  const char* tw = std::getenv("OCPG_MSDA_TILEW");      // experiment switch: scatter tile width on the finest level
  static const int mlo = [] { const char* e = std::getenv("OCPG_MSDA_MARGIN_LO"); return e ? std::atoi(e) : ocpg_col::kScatterMarginLo; }();    // A/B
  if (!ocpg_col::make_col_geom(shapes_host, L, S, M, P, 8, tw ? std::atoi(tw) : 16, cg, mlo, ocpg_col::kMarginHi)) return 0;
  return ocpg_col::bwd_scatter_col(loc, attn, grad_out, N, S, M, D, P, cg, grad_value, st);
}

inline int launch_bwd_locattn_row(const float* value, const int64_t* shapes, const int64_t* level_start, const float* loc,
                                  const float* attn, const float* grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                                  float* grad_loc, float* grad_attn, hipStream_t st) {
  const int G = fast_group(D);
  if (!((G == 4 || G == 8) && L <= kMaxLevels && (long long)S * M * D < (1LL << 31))) return 0;
  const int rpb = 256 / G;
  const size_t glds = rpb * (size_t)L * P * sizeof(GatherRec);
  if (glds > 48 * 1024) return 0;
  const long long rows = (long long)N * Lq * M;
  const unsigned ggrid = (unsigned)((((long long)N * Lq + rpb - 1) / rpb) * M);
  if (G == 4) msda_bwd_gather_row<4><<<ggrid, 256, glds, st>>>(value, shapes, level_start, loc, attn, grad_out, S, M, L, Lq, P, rows, grad_loc, grad_attn);
  else msda_bwd_gather_row<8><<<ggrid, 256, glds, st>>>(value, shapes, level_start, loc, attn, grad_out, S, M, L, Lq, P, rows, grad_loc, grad_attn);
  return 1;
}

}  // namespace

#define FAST_DISPATCH(G_, KERNEL, ...)                                                          \
  switch (G_) {                                                                                 \
    case 1: KERNEL<1><<<grid, 256, lds, st>>>(__VA_ARGS__); break;                               \
    case 2: KERNEL<2><<<grid, 256, lds, st>>>(__VA_ARGS__); break;                               \
    case 4: KERNEL<4><<<grid, 256, lds, st>>>(__VA_ARGS__); break;                               \
    case 8: KERNEL<8><<<grid, 256, lds, st>>>(__VA_ARGS__); break;                               \
    case 16: KERNEL<16><<<grid, 256, lds, st>>>(__VA_ARGS__); break;                             \
    case 32: KERNEL<32><<<grid, 256, lds, st>>>(__VA_ARGS__); break;                             \
    default: KERNEL<64><<<grid, 256, lds, st>>>(__VA_ARGS__); break;                             \
  }

#define FAST_DISPATCH_GATHER(G_, GRID_, LDS_)                                                                                     \
  switch (G_) {                                                                                                                 \
    case 4: msda_bwd_fast<4, false><<<GRID_, 256, LDS_, st>>>(value, shapes, level_start, loc, attn, grad_out, S, M, L, Lq, P, rows, grad_value, grad_loc, grad_attn); break; \
    default: msda_bwd_fast<8, false><<<GRID_, 256, LDS_, st>>>(value, shapes, level_start, loc, attn, grad_out, S, M, L, Lq, P, rows, grad_value, grad_loc, grad_attn); break; \
  }

extern "C" {

const char* ocpg_hip_version(void) { return "ocpg_hip gfx950 r4"; }

int ocpg_msda_fwd_f32(const float* value, const int64_t* shapes, const int64_t* level_start, const float* loc, const float* attn,
                      int N, int S, int M, int D, int L, int Lq, int P, float* out, const int64_t* shapes_host, void* stream) {
  if (int e = check_common(value, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P)) return e;
  const long long rows = (long long)N * Lq * M;
  if (rows == 0) return 0;
  if (!out) return -1013;
  hipStream_t st = (hipStream_t)stream;
  const int G = fast_group(D);
  if (shapes_host && Lq == S && col_fwd_enabled() && (long long)S * M * D < (1LL << 31)) {
    ocpg_col::ColGeom cg;
    if (ocpg_col::make_col_geom(shapes_host, L, S, M, P, 8, 8, cg) && ocpg_col::fwd_col(value, loc, attn, N, S, M, D, P, cg, out, st))
      return launch_status();
  }
  const size_t rec_bytes = (size_t)L * P * sizeof(SampleRec);
  if (G && L <= kMaxLevels && (256 / G) * rec_bytes <= 48 * 1024 && (long long)S * M * D < (1LL << 31)) {
    const int rpb = 256 / G;
#ifndef MSDA_ROW_MAJOR
    const unsigned grid = (unsigned)((((long long)N * Lq + rpb - 1) / rpb) * M);
#else
    const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
#endif
    const size_t lds = rpb * rec_bytes;
    // (a mask-free variant with validity folded into the weights, as in msda_bwd_gather_row, measured 109 vs 100 us: the
    //  forward is bound by the 16-B-per-lane gather path, not by its selects)
    FAST_DISPATCH(G, msda_fwd_fast, value, shapes, level_start, loc, attn, S, M, L, Lq, P, rows, out)
  } else {
    const unsigned grid = (unsigned)((rows + 3) / 4);
    msda_fwd_generic<float><<<grid, 256, 0, st>>>(value, shapes, level_start, loc, attn, S, M, D, L, Lq, P, rows, out);
  }
  return launch_status();
}

int ocpg_msda_fwd_f64(const double* value, const int64_t* shapes, const int64_t* level_start, const double* loc,
                      const double* attn, int N, int S, int M, int D, int L, int Lq, int P, double* out, void* stream) {
  if (int e = check_common(value, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P)) return e;
  const long long rows = (long long)N * Lq * M;
  if (rows == 0) return 0;
  if (!out) return -1013;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  msda_fwd_generic<double><<<grid, 256, 0, (hipStream_t)stream>>>(value, shapes, level_start, loc, attn, S, M, D, L, Lq, P, rows, out);
  return launch_status();
}

int ocpg_msda_bwd_f32(const float* value, const int64_t* shapes, const int64_t* level_start, const float* loc, const float* attn,
                      const float* grad_out, int N, int S, int M, int D, int L, int Lq, int P, float* grad_value, float* grad_loc,
                      float* grad_attn, const int64_t* shapes_host, void* stream) {
  if (int e = check_common(value, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P)) return e;
  const long long rows = (long long)N * Lq * M;
  if (rows == 0) return 0;
  if (!grad_out) return -1013;
  if (!grad_value) return -1014;
  if (!grad_loc) return -1015;
  if (!grad_attn) return -1016;
  hipStream_t st = (hipStream_t)stream;
  const int G = fast_group(D);
  const size_t rec_bytes = (size_t)L * P * sizeof(SampleRec);
  if (G && L <= kMaxLevels && (256 / G) * rec_bytes <= 48 * 1024 && (long long)S * M * D < (1LL << 31)) {
    const int rpb = 256 / G;
    // Self-attention backward = two independent kernels: the column-tile scatter (grad_value; reads loc / attn / grad_out)
    // and the row gather (grad_loc, grad_attn; reads value too).  (Running them on two streams was measured: 405 vs
    // 415 us -- each fills the chip on its own -- so both stay on the caller's stream.)
    if (launch_bwd_value_col(loc, attn, grad_out, shapes_host, N, S, M, D, L, Lq, P, grad_value, st)) {
      if (launch_bwd_locattn_row(value, shapes, level_start, loc, attn, grad_out, N, S, M, D, L, Lq, P, grad_loc, grad_attn, st))
        return launch_status();
      // (not reachable: the scatter accepts a subset of the gather's shapes) finish with the round-1 gather-only kernel
      const unsigned ggrid = (unsigned)((((long long)N * Lq + rpb - 1) / rpb) * M);
      FAST_DISPATCH_GATHER(G, ggrid, rpb * rec_bytes)
      return launch_status();
    }
    TileGeom geo;
    size_t tiled_lds = 0;
    if (shapes_host && Lq == S && L <= kMaxTileLevels && G >= 4 && G <= 16 && make_tile_geom(shapes_host, L, D, rpb, P, geo, tiled_lds)) {
      const unsigned grid = (unsigned)((long long)N * M * geo.ntiles);
      const size_t lds = tiled_lds;
#ifndef MSDA_SPLIT
#define MSDA_SPLIT 0      // 1: scatter-only tiled kernel (grad_value) + gather-only row kernel (grad_loc / grad_attn): measured 746 us vs 701 us for the single kernel (0)
#endif
#ifndef MSDA_NB
#define MSDA_NB 2
#endif
#if MSDA_SPLIT
#define TILED_LAUNCH(G_, NB_) msda_bwd_tiled<G_, NB_, false><<<grid, 256, lds, st>>>(value, shapes, level_start, loc, attn, grad_out, S, M, P, geo, grad_value, grad_loc, grad_attn)
#else
#define TILED_LAUNCH(G_, NB_) msda_bwd_tiled<G_, NB_, true><<<grid, 256, lds, st>>>(value, shapes, level_start, loc, attn, grad_out, S, M, P, geo, grad_value, grad_loc, grad_attn)
#endif
      const bool b4 = (P % MSDA_NB) == 0;
      switch (G) {
        case 4: if (b4) TILED_LAUNCH(4, MSDA_NB); else TILED_LAUNCH(4, 1); break;
        case 8: if (b4) TILED_LAUNCH(8, MSDA_NB); else TILED_LAUNCH(8, 1); break;
        default: if (b4) TILED_LAUNCH(16, MSDA_NB); else TILED_LAUNCH(16, 1); break;
      }
#if MSDA_SPLIT
      {   // the gather side at full occupancy (no LDS window, ~half the registers): same row kernel as the cross-attention path
        const unsigned ggrid = (unsigned)((((long long)N * Lq + rpb - 1) / rpb) * M);
        const size_t glds = rpb * rec_bytes;
        switch (G) {
          case 4: msda_bwd_fast<4, false><<<ggrid, 256, glds, st>>>(value, shapes, level_start, loc, attn, grad_out, S, M, L, Lq, P, rows, grad_value, grad_loc, grad_attn); break;
          case 8: msda_bwd_fast<8, false><<<ggrid, 256, glds, st>>>(value, shapes, level_start, loc, attn, grad_out, S, M, L, Lq, P, rows, grad_value, grad_loc, grad_attn); break;
          default: msda_bwd_fast<16, false><<<ggrid, 256, glds, st>>>(value, shapes, level_start, loc, attn, grad_out, S, M, L, Lq, P, rows, grad_value, grad_loc, grad_attn); break;
        }
      }
#endif
#undef TILED_LAUNCH
      return launch_status();
    }
#ifndef MSDA_ROW_MAJOR
    const unsigned grid = (unsigned)((((long long)N * Lq + rpb - 1) / rpb) * M);
#else
    const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
#endif
    const size_t lds = rpb * rec_bytes;
    FAST_DISPATCH(G, msda_bwd_fast, value, shapes, level_start, loc, attn, grad_out, S, M, L, Lq, P, rows, grad_value, grad_loc,
                  grad_attn)
  } else {
    const unsigned grid = (unsigned)((rows + 3) / 4);
    msda_bwd_generic<float><<<grid, 256, 0, st>>>(value, shapes, level_start, loc, attn, grad_out, S, M, D, L, Lq, P, rows,
                                                  grad_value, grad_loc, grad_attn);
  }
  return launch_status();
}

int ocpg_msda_bwd_value_f32(const float* loc, const float* attn, const float* grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                            float* grad_value, const int64_t* shapes_host, void* stream) {
  if (N < 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0) return -1006;
  if ((long long)N * Lq == 0) return 0;
  if (!loc) return -1001;
  if (!attn) return -1002;
  if (!grad_out) return -1003;
  if (!grad_value) return -1011;
  if (!launch_bwd_value_col(loc, attn, grad_out, shapes_host, N, S, M, D, L, Lq, P, grad_value, (hipStream_t)stream)) return -2000;
  return launch_status();
}

// grad_value with per-call path selection (include/ocpg_hip.h).  Both paths' kernels are launched; the call site's state decides on the
// device which of them runs (csrc/msda_col.h).  Shapes, or forced paths (OCPG_MSDA_TILE / OCPG_MSDA_COL / OCPG_MSDA_COL_LP), that do not
// allow the choice take the plain entry point's route and leave the state untouched.
int ocpg_msda_bwd_value_sel_f32(const float* loc, const float* attn, const float* grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                                float* grad_value, const int64_t* shapes_host, int* sel_state, void* stream) {
  if (N < 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0) return -1006;
  if ((long long)N * Lq == 0) return 0;
  if (!loc) return -1001;
  if (!attn) return -1002;
  if (!grad_out) return -1003;
  if (!grad_value) return -1011;
  hipStream_t st = (hipStream_t)stream;
  const bool forced = std::getenv("OCPG_MSDA_TILE") != nullptr || !col_enabled();
  if (sel_state && !forced && shapes_host && Lq == S && D == 32 && (long long)S * M * D < (1LL << 31)) {
    // thresholds (percent of far samples as the ACTIVE family counts them), measured at config #2, N = 10 (tools/bench_msda_gv.py, GV_SELECT=1):
    //   offsets                      column: far share, us      tiled: far share, us
    //   initial ring                      0.0 %   169                0.0 %   279
    //   ring + N(0, 1.5 px) + 2 % far     3.0 %   241                3.6 %   305
    //   ring + N(0, 3 px) + 5 % far       9.3 %   432               12.5 %   361
    static const int to_tile = [] { const char* e = std::getenv("OCPG_MSDA_SEL_TO_TILE"); return e ? std::atoi(e) : 6; }();
    static const int to_col = [] { const char* e = std::getenv("OCPG_MSDA_SEL_TO_COL"); return e ? std::atoi(e) : 6; }();
    static const int mlo = [] { const char* e = std::getenv("OCPG_MSDA_MARGIN_LO"); return e ? std::atoi(e) : ocpg_col::kScatterMarginLo; }();
    ocpg_col::ColGeom cg;
    if (ocpg_col::make_col_geom(shapes_host, L, S, M, P, 8, 16, cg, mlo, ocpg_col::kMarginHi) && ocpg_col::select_supported(cg, D, P) &&
        ocpg_tile::tile_supported(shapes_host, N, L, S, M, P, D)) {
      if (ocpg_col::bwd_scatter_col(loc, attn, grad_out, N, S, M, D, P, cg, grad_value, st, sel_state, to_tile) != 2) return -2001;
      if (!ocpg_tile::bwd_value_tile(loc, attn, grad_out, shapes_host, N, S, M, D, L, P, grad_value, st, sel_state, to_col)) return -2002;
      ocpg_col::select_commit(sel_state, to_tile, to_col, st);
      return launch_status();
    }
  }
  if (!launch_bwd_value_col(loc, attn, grad_out, shapes_host, N, S, M, D, L, Lq, P, grad_value, st)) return -2000;
  return launch_status();
}

// ---- fused front end (include/ocpg_hip.h): D = 32, L * P = 16; anything else returns -2000 (the caller keeps the unfused path) ----------
int ocpg_msda_fused_fwd_f32(const float* value, const int64_t* shapes, const int64_t* level_start, const float* qproj, const float* ref,
                            int N, int S, int M, int D, int L, int Lq, int P, float* out, float* loc_out, float* attn_out, void* stream) {
  if (int e = check_common(value, shapes, level_start, qproj, ref, N, S, M, D, L, Lq, P)) return e;
  const long long rows = (long long)N * Lq * M;
  if (rows == 0) return 0;
  if (!out) return -1013;
  if (!loc_out) return -1014;
  if (!attn_out) return -1015;
  if (D != 32 || L * P != 16 || L > kMaxLevels || (long long)S * M * D >= (1LL << 31)) return -2000;
  const int rpb = 256 / 8;
  const unsigned grid = (unsigned)((((long long)N * Lq + rpb - 1) / rpb) * M);
  msda_fwd_fused8<<<grid, 256, rpb * 16 * sizeof(SampleRec), (hipStream_t)stream>>>(value, shapes, level_start, qproj, ref, S, M, L, Lq, P, rows, out,
                                                                                   loc_out, attn_out);
  return launch_status();
}

int ocpg_msda_fused_bwd_qproj_f32(const float* value, const int64_t* shapes, const int64_t* level_start, const float* loc, const float* attn,
                                  const float* grad_out, int N, int S, int M, int D, int L, int Lq, int P, float* grad_qproj, void* stream) {
  if (int e = check_common(value, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P)) return e;
  const long long rows = (long long)N * Lq * M;
  if (rows == 0) return 0;
  if (!grad_out) return -1013;
  if (!grad_qproj) return -1014;
  if (D != 32 || L * P != 16 || L > kMaxLevels || (long long)S * M * D >= (1LL << 31)) return -2000;
  const int rpb = 256 / 8;
  const unsigned ggrid = (unsigned)((((long long)N * Lq + rpb - 1) / rpb) * M);
  msda_bwd_gather_row<8, true><<<ggrid, 256, rpb * (size_t)16 * sizeof(GatherRec), (hipStream_t)stream>>>(value, shapes, level_start, loc, attn, grad_out, S, M, L,
                                                                                                      Lq, P, rows, grad_qproj, nullptr);
  return launch_status();
}

int ocpg_msda_bwd_locattn_f32(const float* value, const int64_t* shapes, const int64_t* level_start, const float* loc,
                              const float* attn, const float* grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                              float* grad_loc, float* grad_attn, void* stream) {
  if (int e = check_common(value, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P)) return e;
  if ((long long)N * Lq * M == 0) return 0;
  if (!grad_out) return -1013;
  if (!grad_loc) return -1014;
  if (!grad_attn) return -1015;
  if (!launch_bwd_locattn_row(value, shapes, level_start, loc, attn, grad_out, N, S, M, D, L, Lq, P, grad_loc, grad_attn, (hipStream_t)stream))
    return -2000;
  return launch_status();
}

int ocpg_msda_bwd_f64(const double* value, const int64_t* shapes, const int64_t* level_start, const double* loc,
                      const double* attn, const double* grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                      double* grad_value, double* grad_loc, double* grad_attn, void* stream) {
  if (int e = check_common(value, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P)) return e;
  const long long rows = (long long)N * Lq * M;
  if (rows == 0) return 0;
  if (!grad_out) return -1013;
  if (!grad_value) return -1014;
  if (!grad_loc) return -1015;
  if (!grad_attn) return -1016;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  msda_bwd_generic<double><<<grid, 256, 0, (hipStream_t)stream>>>(value, shapes, level_start, loc, attn, grad_out, S, M, D, L,
                                                                   Lq, P, rows, grad_value, grad_loc, grad_attn);
  return launch_status();
}

}  // extern "C"
