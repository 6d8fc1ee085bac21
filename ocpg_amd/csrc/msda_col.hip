// MSDeformAttn for self-attention over the value's own pixels (Lq == S, the deformable ENCODER) on MI355X (gfx950):
// "pyramid column" kernels.  Semantics = models/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 (forward), :87-159 +
// :301-403 (backward); the mapping onto the machine is new.
//
// Why columns.  One (frame, head) problem has S queries (5 100 at config #2), each sampling L*P points; a query at level
// lq, pixel (y, x) samples every level l around ITS OWN normalised position.  Cut the finest level into <= 8x8 blocks;
// a column = one block plus the pixels of every coarser level that scale onto it (8x8 + 4x4 + 2x2 + 1 = 85 queries).
// All samples a column takes at destination level l fall (for the offsets a deformable encoder learns: a few pixels)
// into the column's footprint at l plus a margin: ONE small window per (column, level).
//   * forward / backward-gather: the window of `value` is staged once in LDS with coalesced 16-B loads; the 4 bilinear
//     corners of every sample are then ds_read_b128 (256 B/clk/CU) instead of 16-B-per-lane global gathers (the row
//     kernels are bound by the texture-address path: 16 cycles per 1-KiB wave load -> ~85 us per call at N=10).
//   * backward-scatter (grad_value): LDS float atomics are the wrong tool on gfx950 -- measured
//     (tools/ubench/lds_atomic2.hip) ds_add_f32 193 cycles per wave-instruction (lane-serial), ds_add_f64 8.8 and 92 with
//     8 lanes on one address, which is the COMMON case for coarse destination levels.  Instead the column's corner
//     contributions (1 360 per level) are BINNED by window pixel with integer LDS atomics (counting sort), and every
//     window pixel is then summed in REGISTERS by the 8 lanes that own it (gather form: one broadcast ds_read_b64 of the
//     item + one ds_read_b128 of the staged grad_out row per contribution) and leaves the kernel as ONE global atomic
//     per touched (pixel, channel).  Coarse queries share their column with the fine ones, so their sparse contributions
//     merge with the dense ones before the flush (round 1 flushed 389 MB of atomics for a 52 MB tensor).
//   * samples outside the window take a direct global path (loads / atomics): results never depend on locality.
//   * every global load of a workgroup (windows of all levels, all (location, weight) pairs) is issued in the prologue:
//     one memory round trip per workgroup, the per-level phases are LDS-only.  Index arithmetic uses multiply-high
//     "magic" reciprocals (gfx950 has no integer divide: ~35 instructions each; the first version spent more
//     instructions on divisions than on the bilinear arithmetic).
#include "msda_col.h"

#include <algorithm>
#include <cstdlib>

#include "msda_dev.h"

namespace ocpg_col {
namespace {

using ocpg_dev::ld4;

constexpr int kNT = 384;     // threads per workgroup: 48 row groups of 8 lanes -> a column's 85 queries in 2 passes
constexpr int kLM = 4;       // levels the column kernels are compiled for

__host__ __device__ __forceinline__ unsigned magic_of(unsigned d) { return d <= 1 ? 0xffffffffu : (unsigned)(0xffffffffu / d); }
// n / d for n < 2^31 with m = magic_of(d): the estimate is at most one low
__device__ __forceinline__ unsigned udiv(unsigned n, unsigned d, unsigned m) {
  unsigned q = __umulhi(n, m);
  if (n - q * d >= d) ++q;
  return q;
}

// Barrier for LDS hand-offs only: waits for this wave's LDS operations, NOT for its global stores / atomics
// (__syncthreads() also drains vmcnt: every level's flush atomics would be waited for at the next barrier --
// measured 80 us of the scatter kernel -- although nothing in the workgroup ever reads them back).
#ifndef EXP_SYNC
#define EXP_SYNC 1
#endif
__device__ __forceinline__ void lds_barrier() {
#if EXP_SYNC
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
  __syncthreads();
#endif
}

#ifdef EXP_STAMPS
// Diagnostic build only (never shipped): per-phase cycle sums of wave 0 of every scatter workgroup.
__device__ unsigned long long g_stamps[16];
#define STAMP(k) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_stamps[k], t_ - tprev); tprev = t_; } } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

struct TileCtx {
  int ry0[kLM], cx0[kLM], cw[kLM], nq[kLM];
  unsigned m_cw[kLM], m_ww[kLM];
  int qbase[kLM + 1];
  int wy0[kLM], wx0[kLM], wh[kLM], ww[kLM];
  int woff[kLM + 1];       // flat offset of each level's window in the column's window-pixel list
  int H[kLM], W[kLM], S0[kLM];
};

// threads 0..L-1 fill their level, then (after a barrier) the prefix sums; ends with a barrier
__device__ __forceinline__ void tile_setup(const ColGeom& geo, int tile, TileCtx& t, int tid) {
  if (tid < geo.L) {
    const int l = tid;
    const int ty = (int)udiv(tile, geo.ntx, geo.m_ntx), tx = tile - ty * geo.ntx;
    const int H = geo.H[l], W = geo.W[l];
    const int ry0 = (int)udiv(ty * H, geo.nty, geo.m_nty), ry1 = (int)udiv((ty + 1) * H, geo.nty, geo.m_nty);
    const int cx0 = (int)udiv(tx * W, geo.ntx, geo.m_ntx), cx1 = (int)udiv((tx + 1) * W, geo.ntx, geo.m_ntx);
    t.H[l] = H; t.W[l] = W; t.S0[l] = geo.S0[l];
    t.ry0[l] = ry0; t.cx0[l] = cx0; t.cw[l] = cx1 - cx0;
    t.m_cw[l] = magic_of(cx1 - cx0);
    t.nq[l] = (ry1 - ry0) * (cx1 - cx0);
    const int fy1 = max(ry1, ry0 + 1), fx1 = max(cx1, cx0 + 1);     // footprint even when the column has no query here
    const int wy0 = max(0, ry0 - geo.mlo), wx0 = max(0, cx0 - geo.mlo);
    t.wy0[l] = wy0;
    t.wx0[l] = wx0;
    t.wh[l] = min(H, fy1 + geo.mhi) - wy0;
    t.ww[l] = min(W, fx1 + geo.mhi) - wx0;
    t.m_ww[l] = magic_of(t.ww[l]);
  }
  __syncthreads();
  if (tid <= geo.L) {
    int q = 0, w = 0;
    for (int k = 0; k < tid; ++k) { q += t.nq[k]; w += t.wh[k] * t.ww[k]; }
    t.qbase[tid] = q;
    t.woff[tid] = w;
  }
  __syncthreads();
}

__device__ __forceinline__ int local_to_query(const TileCtx& t, int L, int i) {
  int l = 0;
  while (l + 1 < L && i >= t.qbase[l + 1]) ++l;
  const int r = i - t.qbase[l];
  const int dy = (int)udiv(r, t.cw[l], t.m_cw[l]);
  return t.S0[l] + (t.ry0[l] + dy) * t.W[l] + t.cx0[l] + r - dy * t.cw[l];
}

// ---- sample records ----------------------------------------------------------------------------------------------------
// Gather kernels: validity is FOLDED INTO THE WEIGHTS.  The two corner columns are xa = max(x0, 0), xb = min(x0+1, W-1)
// (rows alike): always inside the map, so every corner address is loadable; a corner outside the map gets weight 0
// (hx' = x0 >= 0 ? 1-lx : 0, lx' = x0+1 <= W-1 ? lx : 0, ...).  No masks, no selects in the inner loop.
//   pk = (pix << 5) | (global << 4) | iy1 << 3 | iy0 << 2 | ix1 << 1 | ix0;   pix = pixel (ya, xa) in the window, or in the
//   level map when global.  An invalid sample: pk = 0 and all weights 0 (it reads pixel 0 of the window and ignores it).
struct GRec {
  int pk;
  float hy, ly, hx, lx;   // masked by validity
};

__device__ __forceinline__ GRec make_grec(float x_n, float y_n, int H, int W, int wy0, int wx0, int wh, int ww) {
  GRec r;
  const float h_im = y_n * (float)H - 0.5f, w_im = x_n * (float)W - 0.5f;
  if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {          // cuh:268 / cuh:332
    const int y0 = (int)floorf(h_im), x0 = (int)floorf(w_im);
    const float ly = h_im - (float)y0, lx = w_im - (float)x0;
    const bool iy0 = y0 >= 0, iy1 = y0 + 1 <= H - 1, ix0 = x0 >= 0, ix1 = x0 + 1 <= W - 1;
    r.hy = iy0 ? 1.f - ly : 0.f;
    r.ly = iy1 ? ly : 0.f;
    r.hx = ix0 ? 1.f - lx : 0.f;
    r.lx = ix1 ? lx : 0.f;
    const int ya = max(y0, 0), yb = min(y0 + 1, H - 1), xa = max(x0, 0), xb = min(x0 + 1, W - 1);
    const bool in = ya >= wy0 && yb < wy0 + wh && xa >= wx0 && xb < wx0 + ww;
    const int pix = in ? (ya - wy0) * ww + (xa - wx0) : ya * W + xa;
    r.pk = (pix << 5) | (in ? 0 : 16) | (iy1 ? 8 : 0) | (iy0 ? 4 : 0) | (ix1 ? 2 : 0) | (ix0 ? 1 : 0);
  } else {
    r.hy = r.ly = r.hx = r.lx = 0.f;
    r.pk = 0;
  }
  return r;
}

// ----------------------------------------------------------------------------------------------------------------------
// Forward with LDS-staged value windows.  NT threads = NT/G row groups of G lanes; a row group owns one query of the
// column per pass and one float4 (4 channels) per lane.  (Opt-in, OCPG_MSDA_FWD=col: at config #2 it measures 185 us
// against the row kernel's 100 us -- same instruction count, a third of the resident waves)
template <int G>
__global__ __launch_bounds__(kNT) void k_fwd_col(const float* __restrict__ value, const float* __restrict__ loc,
                                                 const float* __restrict__ attn, int S, int M, int P, ColGeom geo,
                                                 float* __restrict__ out) {
  constexpr int D = 4 * G, ROWS = kNT / G, PASSES = (96 + ROWS - 1) / ROWS;
  constexpr int WU0 = (320 + ROWS - 1) / ROWS;      // level 0's window goes registers -> LDS inside the prologue (<= 320 pixels)
  constexpr int WU = (384 + ROWS - 1) / ROWS;       // window pixels of the OTHER levels stay in registers (<= 384 per column)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* win = reinterpret_cast<float*>(smem);                                          // [wmax][D]   (one level at a time)
  float4* rec_w = reinterpret_cast<float4*>(smem + (size_t)geo.wmax * D * sizeof(float));  // [tmax * P]  4 corner weights * attention weight
  int* rec_pk = reinterpret_cast<int*>(rec_w + (size_t)geo.tmax * P);                    // [tmax * P]
  int* qg = reinterpret_cast<int*>(rec_pk + (size_t)geo.tmax * P);                       // [tmax]
  __shared__ TileCtx tc;
  const int tid = threadIdx.x;
  const int bid = blockIdx.x;
  const int bt = (int)udiv(bid, M, geo.m_M);
  const int m = bid - bt * M;      // head fastest: blocks are dealt round-robin over the 8 XCDs -> one head per XCD L2
  const int b = (int)udiv(bt, geo.ntiles, geo.m_ntiles), tile = bt - b * geo.ntiles;
  tile_setup(geo, tile, tc, tid);
  const int L = geo.L, NS = L * P, MD = M * D;
  const int T = tc.qbase[L], TP = T * P;
  const int r = tid / G, j = tid % G;
  const float* vb = value + (long long)b * S * MD + m * D + 4 * j;
  // ---- prologue: all global loads ---------------------------------------------------------------------------------
  // (loads are UNCONDITIONAL on clamped indices: a load inside a branch makes hipcc wait for it right there -- the
  //  first build had one s_waitcnt vmcnt per load, i.e. a dozen dependent memory round trips per workgroup)
  float4 wreg0[WU0], wreg[WU];
  const int wtot = tc.woff[L], w1st = tc.woff[1];
  {
    const int ww = tc.ww[0];
    const unsigned mw = tc.m_ww[0];
    const float* v0 = vb + ((long long)tc.S0[0] + (long long)tc.wy0[0] * tc.W[0] + tc.wx0[0]) * MD;
#pragma unroll
    for (int u = 0; u < WU0; ++u) {
      const int px = min(r + u * ROWS, w1st - 1);
      const int dy = (int)udiv(px, ww, mw);
      wreg0[u] = ld4(v0 + ((long long)dy * tc.W[0] + px - dy * ww) * MD);
    }
  }
  if (wtot > w1st) {       // uniform
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int idx = min(w1st + r + u * ROWS, wtot - 1);      // flat window pixel over levels 1..L-1; this lane's float4 of it
      int l = 1;
      while (l + 1 < L && idx >= tc.woff[l + 1]) ++l;
      const int px = idx - tc.woff[l], ww = tc.ww[l];
      const int dy = (int)udiv(px, ww, tc.m_ww[l]);
      wreg[u] = ld4(vb + ((long long)tc.S0[l] + (long long)(tc.wy0[l] + dy) * tc.W[l] + tc.wx0[l] + px - dy * ww) * MD);
    }
  } else {
#pragma unroll
    for (int u = 0; u < WU; ++u) wreg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int i = tid; i < T; i += kNT) qg[i] = local_to_query(tc, L, i);
  __syncthreads();
  // this thread's sample of every level: (query tid / P, point tid % P)
  float2 sxy[kLM];
  float sa[kLM];
  {
    const int ts = min(tid, TP - 1);
    const int ql = (int)udiv(ts, P, geo.m_P), p = ts - ql * P;
    const long long wi0 = (((long long)b * S + qg[ql]) * M + m) * NS + p;
#pragma unroll
    for (int l = 0; l < kLM; ++l) {
      const long long wi = wi0 + min(l, L - 1) * P;
      sxy[l] = *reinterpret_cast<const float2*>(loc + wi * 2);
      sa[l] = attn[wi];
    }
  }
  float4 acc[PASSES];
  long long rowp[PASSES];
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int ql = ps * ROWS + r;
    rowp[ps] = ((long long)b * S + qg[min(ql, T - 1)]) * M + m;
    acc[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < WU0; ++u) {
    const int px = r + u * ROWS;
    if (px < w1st) *reinterpret_cast<float4*>(win + px * D + 4 * j) = wreg0[u];
  }
#pragma unroll
  for (int l = 0; l < kLM; ++l) {
    if (l < L) {
      const int H = tc.H[l], W = tc.W[l], ww = tc.ww[l];
      const float* vl = vb + (long long)tc.S0[l] * MD;
      // (1) registers -> LDS: this level's window and records
      if (l > 0) {
        const int w0 = tc.woff[l], w1 = tc.woff[l + 1];
#pragma unroll
        for (int u = 0; u < WU; ++u) {
          const int idx = w1st + r + u * ROWS;
          if (idx >= w0 && idx < w1) *reinterpret_cast<float4*>(win + (idx - w0) * D + 4 * j) = wreg[u];
        }
      }
      if (tid < TP) {
        const GRec g = make_grec(sxy[l].x, sxy[l].y, H, W, tc.wy0[l], tc.wx0[l], tc.wh[l], ww);
        rec_pk[tid] = g.pk;
        const float a = sa[l];
        rec_w[tid] = make_float4(g.hy * g.hx * a, g.hy * g.lx * a, g.ly * g.hx * a, g.ly * g.lx * a);
      }
      __syncthreads();
      // (2) gather
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int ql = ps * ROWS + r;
        if (ql < T) {                                       // whole G-lane groups take the branch together
          const int rb = ql * P;
          {
            float4 o = acc[ps];
            for (int p0 = 0; p0 < P; p0 += 2) {
              const bool two = p0 + 1 < P;
              const int i0 = rb + p0, i1 = two ? i0 + 1 : i0;
              const int pk[2] = {rec_pk[i0], rec_pk[i1]};
              float4 w[2] = {rec_w[i0], rec_w[i1]};
              if (!two) w[1] = make_float4(0.f, 0.f, 0.f, 0.f);
              float4 v[2][4];
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                const int pix = pk[i] >> 5;
                const bool sx = (pk[i] & 3) == 3, sy = (pk[i] & 12) == 12;       // both columns / rows inside the map
                if (!(pk[i] & 16)) {
                  const float* p00 = win + pix * D + 4 * j;
                  const int dx = sx ? D : 0, dy = sy ? ww * D : 0;
                  v[i][0] = ld4(p00); v[i][1] = ld4(p00 + dx); v[i][2] = ld4(p00 + dy); v[i][3] = ld4(p00 + dy + dx);
                } else {
                  const float* p00 = vl + (long long)pix * MD;
                  const long long dx = sx ? MD : 0, dy = sy ? (long long)W * MD : 0;
                  v[i][0] = ld4(p00); v[i][1] = ld4(p00 + dx); v[i][2] = ld4(p00 + dy); v[i][3] = ld4(p00 + dy + dx);
                }
              }
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                o.x += w[i].x * v[i][0].x + w[i].y * v[i][1].x + w[i].z * v[i][2].x + w[i].w * v[i][3].x;
                o.y += w[i].x * v[i][0].y + w[i].y * v[i][1].y + w[i].z * v[i][2].y + w[i].w * v[i][3].y;
                o.z += w[i].x * v[i][0].z + w[i].y * v[i][1].z + w[i].z * v[i][2].z + w[i].w * v[i][3].z;
                o.w += w[i].x * v[i][0].w + w[i].y * v[i][1].w + w[i].z * v[i][2].w + w[i].w * v[i][3].w;
              }
            }
            acc[ps] = o;
          }
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps)
    if (ps * ROWS + r < T) *reinterpret_cast<float4*>(out + rowp[ps] * D + 4 * j) = acc[ps];
}

// ----------------------------------------------------------------------------------------------------------------------
// Backward-scatter: grad_value of one column, one destination level at a time (same latency plan: every (location,
// weight) pair and the column's grad_out rows are loaded in the prologue).
struct __attribute__((aligned(8))) Item {
  float w;   // bilinear weight * attention weight
  int q;     // float offset of the query's staged grad_out row (query index in the column * D)
};

template <int G, int NT>
__global__ __launch_bounds__(NT) void k_scatter_col(const float* __restrict__ loc, const float* __restrict__ attn,
                                                     const float* __restrict__ gout, int S, int M, int P, ColGeom geo,
                                                     float* __restrict__ gvalue) {
  constexpr int D = 4 * G, GROUPS = NT / G, NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* gs = reinterpret_cast<float*>(smem);                                           // [tmax][D], channel j + G*c at 4*j + c
  Item* items = reinterpret_cast<Item*>(smem + (size_t)geo.tmax * D * sizeof(float));    // [tmax * P * 4]
  int* qg = reinterpret_cast<int*>(items + (size_t)geo.tmax * P * 4);                    // [tmax]
  __shared__ int cnt[NT], start[NT], nz[NT], wtot[NW];
  __shared__ TileCtx tc;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bid = blockIdx.x;
  const int bt = (int)udiv(bid, M, geo.m_M);
  const int m = bid - bt * M;
  const int b = (int)udiv(bt, geo.ntiles, geo.m_ntiles), tile = bt - b * geo.ntiles;
#ifdef EXP_STAMPS
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
  cnt[tid] = 0;
  tile_setup(geo, tile, tc, tid);
  STAMP(0);
  const int L = geo.L, NS = L * P, MD = M * D;
  const int T = tc.qbase[L], TP = T * P;
  for (int i = tid; i < T; i += NT) qg[i] = local_to_query(tc, L, i);
  __syncthreads();
  const int j = tid % G;
  const int ts = min(tid, TP - 1);
  const int qloc = (int)udiv(ts, P, geo.m_P);           // this thread's sample of every level: (query tid / P, point tid % P)
  float2 sxy[kLM];
  float sa[kLM];
  {                                                     // unconditional loads on clamped indices (see the gather kernel)
    const int p = ts - qloc * P;
    const long long wi0 = (((long long)b * S + qg[qloc]) * M + m) * NS + p;
#pragma unroll
    for (int l = 0; l < kLM; ++l) {
      const long long wi = wi0 + min(l, L - 1) * P;
      sxy[l] = *reinterpret_cast<const float2*>(loc + wi * 2);
      sa[l] = attn[wi];
    }
  }
  {       // stage grad_out of the column's queries, channel-interleaved per lane (tmax * G <= 2 * NT)
    float4 gq[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ql = min(tid / G + u * GROUPS, T - 1);
      const float* g = gout + (((long long)b * S + qg[ql]) * M + m) * D + j;
      gq[u] = make_float4(g[0], g[G], g[2 * G], g[3 * G]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ql = tid / G + u * GROUPS;
      if (ql < T) *reinterpret_cast<float4*>(gs + ql * D + 4 * j) = gq[u];
    }
  }
  __syncthreads();
  STAMP(1);
  float* gvb = gvalue + (long long)b * S * MD + m * D;
#pragma unroll
  for (int l = 0; l < kLM; ++l) {
    if (l < L) {
      const int H = tc.H[l], W = tc.W[l], wy0 = tc.wy0[l], wx0 = tc.wx0[l], wh = tc.wh[l], ww = tc.ww[l];
      const int wpx = wh * ww;
      float* gvl = gvb + (long long)tc.S0[l] * MD;
      // (1) bin: one sample of this level per lane (TP <= NT); a window corner takes a slot in its pixel's list
      int pix0 = 0, inm = 0, ovm = 0, slotk[4] = {0, 0, 0, 0};
      float wk[4] = {0.f, 0.f, 0.f, 0.f};
      if (tid < TP) {
        const float h_im = sxy[l].y * (float)H - 0.5f, w_im = sxy[l].x * (float)W - 0.5f;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
          const int y0 = (int)floorf(h_im), x0 = (int)floorf(w_im);
          const float ly = h_im - (float)y0, lx = w_im - (float)x0, hy = 1.f - ly, hx = 1.f - lx, a = sa[l];
          const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= H - 1, x0ok = x0 >= 0, x1ok = x0 + 1 <= W - 1;
          const int mask = (y0ok && x0ok ? 1 : 0) | (y0ok && x1ok ? 2 : 0) | (y1ok && x0ok ? 4 : 0) | (y1ok && x1ok ? 8 : 0);
          wk[0] = hy * hx * a; wk[1] = hy * lx * a; wk[2] = ly * hx * a; wk[3] = ly * lx * a;
          const bool in = max(y0, 0) >= wy0 && min(y0 + 1, H - 1) < wy0 + wh && max(x0, 0) >= wx0 && min(x0 + 1, W - 1) < wx0 + ww;
          if (in) {
            inm = mask;
            pix0 = (y0 - wy0) * ww + (x0 - wx0);          // may be "virtual" (row / column -1): only corners in the mask are used
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (mask & (1 << k)) slotk[k] = atomicAdd(&cnt[pix0 + (k & 1) + (k >> 1) * ww], 1);
          } else {
            ovm = mask;
            pix0 = y0 * W + x0;
          }
        }
      }
      {
        // corners outside the window: straight to memory, one contribution per wave step, D lanes x 4 B contiguous
        unsigned long long bal = __ballot(ovm != 0);
        while (bal) {
          const int src = __ffsll((long long)bal) - 1;
          bal &= bal - 1;
          const int om = __builtin_amdgcn_readlane(ovm, src), gp = __builtin_amdgcn_readlane(pix0, src),
                    qs = __builtin_amdgcn_readlane(qloc, src);
          float wsrc[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) wsrc[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wk[k]), src));
          if (lane < D) {
            const float g = gs[qs * D + 4 * (lane % G) + lane / G];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (om & (1 << k)) atomicAdd(gvl + (long long)(gp + (k & 1) + (k >> 1) * W) * MD + lane, wsrc[k] * g);
          }
        }
      }
      lds_barrier();
      STAMP(2);
      // (2) one scan for both the list starts and the compaction of the non-empty pixels (packed: count | flag << 16)
      int nnz;
      {
        const int c = tid < wpx ? cnt[tid] : 0;
        const int v = c | (c ? 1 << 16 : 0);
        int s = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(s, o, 64);
          if (lane >= o) s += t;
        }
        if (lane == 63) wtot[wave] = s;
        lds_barrier();
        int base = 0, all = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const int t = wtot[w];
          if (w < wave) base += t;
          all += t;
        }
        nnz = all >> 16;
        const int excl = base + s - v;
        start[tid] = excl & 0xffff;
        if (c) nz[excl >> 16] = tid;
      }
      lds_barrier();
      STAMP(3);
      // (3) drop the items into their lists
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (inm & (1 << k)) {
          Item it;
          it.w = wk[k];
          it.q = qloc * D;
          items[start[pix0 + (k & 1) + (k >> 1) * ww] + slotk[k]] = it;
        }
      lds_barrier();
      STAMP(4);
      // (4) every touched window pixel is summed in registers by the G lanes that own it, then flushed once
      const unsigned mw = tc.m_ww[l];
#ifdef EXP_NO_ACC
      nnz = 0;
#endif
      // Two neighbouring lane groups (16 lanes) flush together so that every atomic request carries a full 64-byte line of
      // ONE pixel (G = 8): the memory-side atomic path is request-bound, and 32-byte requests were half empty.
      for (int kk0 = (tid / (2 * G)) * 2; kk0 < nnz; kk0 += GROUPS) {
        const int odd = (tid / G) & 1;
        const int kk = kk0 + odd;
        const bool valid = kk < nnz;
        const int pp = valid ? nz[kk] : 0;
        const int n = valid ? cnt[pp] : 0;
        const Item* lst = items + start[pp];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int t = 0;
        for (; t + 4 <= n; t += 4) {
          const Item i0 = lst[t], i1 = lst[t + 1], i2 = lst[t + 2], i3 = lst[t + 3];
          const float4 g0 = ld4(gs + i0.q + 4 * j), g1 = ld4(gs + i1.q + 4 * j), g2 = ld4(gs + i2.q + 4 * j),
                       g3 = ld4(gs + i3.q + 4 * j);
          acc.x += i0.w * g0.x + i1.w * g1.x + i2.w * g2.x + i3.w * g3.x;
          acc.y += i0.w * g0.y + i1.w * g1.y + i2.w * g2.y + i3.w * g3.y;
          acc.z += i0.w * g0.z + i1.w * g1.z + i2.w * g2.z + i3.w * g3.z;
          acc.w += i0.w * g0.w + i1.w * g1.w + i2.w * g2.w + i3.w * g3.w;
        }
        for (; t < n; ++t) {
          const Item i0 = lst[t];
          const float4 g0 = ld4(gs + i0.q + 4 * j);
          acc.x += i0.w * g0.x; acc.y += i0.w * g0.y; acc.z += i0.w * g0.z; acc.w += i0.w * g0.w;
        }
        const int dy = (int)udiv(pp, ww, mw);
        const int gpix = (wy0 + dy) * W + wx0 + pp - dy * ww;                             // pixel of this group in the level
#ifndef EXP_NO_FLUSH
        if (G == 8) {
          // lane j of the even group holds channels j, j+8, j+16, j+24 of pixel A, the odd group the same of pixel B; after the
          // exchange the 16 lanes write A[0..15], A[16..31], B[0..15], B[16..31]: one 64-byte request each
          const float send1 = odd ? acc.x : acc.y, send2 = odd ? acc.z : acc.w;
          const float got1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send1), 0x128, 0xF, 0xF, true));   // row_ror:8 = lane ^ 8
          const float got2 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send2), 0x128, 0xF, 0xF, true));
          const int opix = __builtin_amdgcn_mov_dpp(valid ? gpix : -1, 0x128, 0xF, 0xF, true);
          const int pixA = odd ? opix : (valid ? gpix : -1), pixB = odd ? (valid ? gpix : -1) : opix;
          const int jj = j + (odd ? G : 0);
          if (pixA >= 0) {
            float* g = gvl + (long long)pixA * MD + jj;
            atomicAdd(g, odd ? got1 : acc.x);
            atomicAdd(g + 2 * G, odd ? got2 : acc.z);
          }
          if (pixB >= 0) {
            float* g = gvl + (long long)pixB * MD + jj;
            atomicAdd(g, odd ? acc.y : got1);
            atomicAdd(g + 2 * G, odd ? acc.w : got2);
          }
        } else if (valid) {
          float* g = gvl + (long long)gpix * MD + j;      // lane j owns channels j, j+G, j+2G, j+3G: 4*G contiguous bytes per instruction
          atomicAdd(g, acc.x);
          atomicAdd(g + G, acc.y);
          atomicAdd(g + 2 * G, acc.z);
          atomicAdd(g + 3 * G, acc.w);
        }
#endif
        if (valid && j == 0) cnt[pp] = 0;        // ready for the next level
      }
      STAMP(5);
      lds_barrier();
      STAMP(6);
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// Backward-scatter, TWO destination levels per pass (round 3).  The single-level kernel above spends 5 workgroup barriers per
// level on one sample per thread (stamps: binning 31 %, end-of-level barrier 14 %); here a pass bins the samples of a level PAIR
// into the concatenation of their two windows: half the barriers, twice the independent LDS work between them; the list starts
// and the compaction of the non-empty pixels come from ONE wave (12 bins per lane) instead of a two-barrier cross-wave scan, and
// a thread's eight counter atomics / eight list writes are issued back to back (their results are first used after the last one
// is in flight).  Same results as k_scatter_col up to the order of the fp32 sums.
template <int G, int NT>
__global__ __launch_bounds__(NT) void k_scatter_col2(const float* __restrict__ loc, const float* __restrict__ attn,
                                                      const float* __restrict__ gout, int S, int M, int P, ColGeom geo,
                                                      float* __restrict__ gvalue) {
  constexpr int D = 4 * G, GROUPS = NT / G, BPL = NT / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* gs = reinterpret_cast<float*>(smem);                                           // [tmax][D], channel j + G*c at 4*j + c
  Item* items = reinterpret_cast<Item*>(smem + (size_t)geo.tmax * D * sizeof(float));    // [tmax * P * 8]
  int* qg = reinterpret_cast<int*>(items + (size_t)geo.tmax * P * 8);                    // [tmax]
  __shared__ int cnt[NT], start[NT], nz[NT], nnz_s;
  __shared__ TileCtx tc;
  const int tid = threadIdx.x, lane = tid & 63;
  const int bid = blockIdx.x;
  const int bt = (int)udiv(bid, M, geo.m_M);
  const int m = bid - bt * M;
  const int b = (int)udiv(bt, geo.ntiles, geo.m_ntiles), tile = bt - b * geo.ntiles;
  cnt[tid] = 0;
  tile_setup(geo, tile, tc, tid);
  const int L = geo.L, NS = L * P, MD = M * D;
  const int T = tc.qbase[L], TP = T * P;
  for (int i = tid; i < T; i += NT) qg[i] = local_to_query(tc, L, i);
  __syncthreads();
  const int j = tid % G;
  const int ts = min(tid, TP - 1);
  const int qloc = (int)udiv(ts, P, geo.m_P);
  float2 sxy[kLM];
  float sa[kLM];
  {
    const int p = ts - qloc * P;
    const long long wi0 = (((long long)b * S + qg[qloc]) * M + m) * NS + p;
#pragma unroll
    for (int l = 0; l < kLM; ++l) {
      const long long wi = wi0 + min(l, L - 1) * P;
      sxy[l] = *reinterpret_cast<const float2*>(loc + wi * 2);
      sa[l] = attn[wi];
    }
  }
  {
    float4 gq[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ql = min(tid / G + u * GROUPS, T - 1);
      const float* g = gout + (((long long)b * S + qg[ql]) * M + m) * D + j;
      gq[u] = make_float4(g[0], g[G], g[2 * G], g[3 * G]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ql = tid / G + u * GROUPS;
      if (ql < T) *reinterpret_cast<float4*>(gs + ql * D + 4 * j) = gq[u];
    }
  }
  __syncthreads();
  float* gvb = gvalue + (long long)b * S * MD + m * D;
#pragma unroll
  for (int l0 = 0; l0 < kLM; l0 += 2) {
    if (l0 < L) {
      const int nl = min(2, L - l0);
      const int wpx0 = tc.wh[l0] * tc.ww[l0];
      const int l1c = l0 + 1 < kLM ? l0 + 1 : l0;          // known after unrolling (register arrays stay in registers); guarded by nl below
      const int wtot2 = wpx0 + (nl > 1 ? tc.wh[l1c] * tc.ww[l1c] : 0);
      // (1) bin both levels' samples: bins [0, wpx0) = window of level l0, [wpx0, wtot2) = window of level l0 + 1
      int pid[8], ovm[2] = {0, 0}, opix[2] = {0, 0};
      float wk[8];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int l = u ? l1c : l0;
        const int H = tc.H[l], W = tc.W[l], wy0 = tc.wy0[l], wx0 = tc.wx0[l], wh = tc.wh[l], ww = tc.ww[l];
        const int boff = u ? wpx0 : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { pid[u * 4 + k] = -1; wk[u * 4 + k] = 0.f; }
        if (tid < TP && u < nl) {
          const float h_im = sxy[l].y * (float)H - 0.5f, w_im = sxy[l].x * (float)W - 0.5f;
          if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
            const int y0 = (int)floorf(h_im), x0 = (int)floorf(w_im);
            const float ly = h_im - (float)y0, lx = w_im - (float)x0, hy = 1.f - ly, hx = 1.f - lx, a = sa[l];
            const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= H - 1, x0ok = x0 >= 0, x1ok = x0 + 1 <= W - 1;
            const int mask = (y0ok && x0ok ? 1 : 0) | (y0ok && x1ok ? 2 : 0) | (y1ok && x0ok ? 4 : 0) | (y1ok && x1ok ? 8 : 0);
            wk[u * 4 + 0] = hy * hx * a; wk[u * 4 + 1] = hy * lx * a; wk[u * 4 + 2] = ly * hx * a; wk[u * 4 + 3] = ly * lx * a;
            const bool in = max(y0, 0) >= wy0 && min(y0 + 1, H - 1) < wy0 + wh && max(x0, 0) >= wx0 && min(x0 + 1, W - 1) < wx0 + ww;
            if (in) {
              const int p0 = boff + (y0 - wy0) * ww + (x0 - wx0);        // may be "virtual" (row / column -1): only corners in the mask are used
#pragma unroll
              for (int k = 0; k < 4; ++k)
                if (mask & (1 << k)) pid[u * 4 + k] = p0 + (k & 1) + (k >> 1) * ww;
            } else {
              ovm[u] = mask;
              opix[u] = y0 * W + x0;
            }
          }
        }
      }
      int slot[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) slot[i] = pid[i] >= 0 ? atomicAdd(&cnt[pid[i]], 1) : 0;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        // corners outside the window: straight to memory, one contribution per wave step, D lanes x 4 B contiguous
        const int l = u ? l1c : l0;
        const int W = tc.W[l];
        float* gvl = gvb + (long long)tc.S0[l] * MD;
        unsigned long long bal = __ballot(ovm[u] != 0);
        while (bal) {
          const int src = __ffsll((long long)bal) - 1;
          bal &= bal - 1;
          const int om = __builtin_amdgcn_readlane(ovm[u], src), gp = __builtin_amdgcn_readlane(opix[u], src),
                    qs = __builtin_amdgcn_readlane(qloc, src);
          float wsrc[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) wsrc[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wk[u * 4 + k]), src));
          if (lane < D) {
            const float g = gs[qs * D + 4 * (lane % G) + lane / G];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (om & (1 << k)) atomicAdd(gvl + (long long)(gp + (k & 1) + (k >> 1) * W) * MD + lane, wsrc[k] * g);
          }
        }
      }
      lds_barrier();
      // (2) list starts + compaction of the non-empty pixels, by one wave (BPL consecutive bins per lane, packed count | flag << 16)
      if (tid < 64) {
        int c[BPL], s = 0;
#pragma unroll
        for (int i = 0; i < BPL; ++i) {
          const int bin = lane * BPL + i;
          c[i] = bin < wtot2 ? cnt[bin] : 0;
          s += c[i] | (c[i] ? 1 << 16 : 0);
        }
        int inc = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(inc, o, 64);
          if (lane >= o) inc += t;
        }
        int run = inc - s;
#pragma unroll
        for (int i = 0; i < BPL; ++i) {
          const int bin = lane * BPL + i;
          start[bin] = run & 0xffff;
          if (c[i]) nz[run >> 16] = bin;
          run += c[i] | (c[i] ? 1 << 16 : 0);
        }
        if (lane == 63) nnz_s = inc >> 16;
      }
      lds_barrier();
      // (3) drop the items into their lists: all starts first, then the writes
      {
        int st[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) st[i] = start[max(pid[i], 0)];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (pid[i] >= 0) {
            Item it;
            it.w = wk[i];
            it.q = qloc * D;
            items[st[i] + slot[i]] = it;
          }
      }
      lds_barrier();
      // (4) every touched pixel is summed in registers by the G lanes that own it, then flushed once (pairs of groups: full 64-B lines)
      const int nnz = nnz_s;
      for (int kk0 = (tid / (2 * G)) * 2; kk0 < nnz; kk0 += GROUPS) {
        const int odd = (tid / G) & 1;
        const int kk = kk0 + odd;
        const bool valid = kk < nnz;
        const int pp = valid ? nz[kk] : 0;
        const int n = valid ? cnt[pp] : 0;
        const Item* lst = items + start[pp];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int t = 0;
        for (; t + 4 <= n; t += 4) {
          const Item i0 = lst[t], i1 = lst[t + 1], i2 = lst[t + 2], i3 = lst[t + 3];
          const float4 g0 = ld4(gs + i0.q + 4 * j), g1 = ld4(gs + i1.q + 4 * j), g2 = ld4(gs + i2.q + 4 * j),
                       g3 = ld4(gs + i3.q + 4 * j);
          acc.x += i0.w * g0.x + i1.w * g1.x + i2.w * g2.x + i3.w * g3.x;
          acc.y += i0.w * g0.y + i1.w * g1.y + i2.w * g2.y + i3.w * g3.y;
          acc.z += i0.w * g0.z + i1.w * g1.z + i2.w * g2.z + i3.w * g3.z;
          acc.w += i0.w * g0.w + i1.w * g1.w + i2.w * g2.w + i3.w * g3.w;
        }
        for (; t < n; ++t) {
          const Item i0 = lst[t];
          const float4 g0 = ld4(gs + i0.q + 4 * j);
          acc.x += i0.w * g0.x; acc.y += i0.w * g0.y; acc.z += i0.w * g0.z; acc.w += i0.w * g0.w;
        }
        const int u = pp >= wpx0 ? 1 : 0;
        const int l = u ? l1c : l0;
        const int pl = pp - (u ? wpx0 : 0), ww = tc.ww[l];
        const int dy = (int)udiv(pl, ww, tc.m_ww[l]);
        const int gpix = tc.S0[l] + (tc.wy0[l] + dy) * tc.W[l] + tc.wx0[l] + pl - dy * ww;      // pixel of this group in the whole map
#ifndef EXP_NO_FLUSH
        {
          const float send1 = odd ? acc.x : acc.y, send2 = odd ? acc.z : acc.w;
          const float got1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send1), 0x128, 0xF, 0xF, true));   // row_ror:8 = lane ^ 8
          const float got2 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send2), 0x128, 0xF, 0xF, true));
          const int opx = __builtin_amdgcn_mov_dpp(valid ? gpix : -1, 0x128, 0xF, 0xF, true);
          const int pixA = odd ? opx : (valid ? gpix : -1), pixB = odd ? (valid ? gpix : -1) : opx;
          const int jj = j + (odd ? G : 0);
          if (pixA >= 0) {
            float* g = gvb + (long long)pixA * MD + jj;
            atomicAdd(g, odd ? got1 : acc.x);
            atomicAdd(g + 2 * G, odd ? got2 : acc.z);
          }
          if (pixB >= 0) {
            float* g = gvb + (long long)pixB * MD + jj;
            atomicAdd(g, odd ? acc.y : got1);
            atomicAdd(g + 2 * G, odd ? acc.w : got2);
          }
        }
#else
        if (valid && acc.x + acc.y + acc.z + acc.w == 1.2345f) gvb[gpix] = acc.x;     // timing-only build: keeps the sums alive, never stores
#endif
        if (valid && j == 0) cnt[pp] = 0;        // ready for the next pass
      }
      lds_barrier();
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// Backward-scatter, ALL destination levels in ONE pass (round 4).  Measured on the level-pair kernel above (a build without the
// flush atomics, tools/r4_exp1.sh): 179 us with or without them -- the kernel is bound by its own phase structure, not by the
// memory-side atomic rate: two passes x five barriers, one LDS atomic + one list entry per CORNER, and a final phase in which the few
// coarse-level pixels (136 / 340 contributions each at levels 2 / 3 against 11 at level 0) keep a quarter of the lane groups busy.
// Here:
//   * the unit that is sorted is the SAMPLE, not the corner: one counter atomic and one 16-byte item {lx, ly, a, row} per sample,
//     binned by its top-left pixel on a (wh + 1) x (ww + 1) grid per level (row / column -1 of the window included); a pixel then
//     walks the lists of the four bins that can hold a sample touching it and forms the corner weight on the fly (3 multiplies);
//   * all levels are binned together (965 bins at config #2): one bin phase, one scan, one item phase, one sum phase -- 5 barriers;
//   * the sum phase is a TASK list built from the counts: a pixel with few contributions is one lane group's task, a heavy pixel
//     (coarse levels) is a whole wave's -- its 8 lane groups stride over the lists and add their partial sums through three
//     cross-lane exchanges -- so every lane group carries about the same number of contributions.
// Same results as the kernels above up to the order of the fp32 sums.
#ifdef EXP_STAMPS
#define S3_INIT unsigned tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tprev3 = __builtin_amdgcn_s_memtime()
#define S3(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc_[(k) & 7] += (unsigned)(t_ - tprev3); tprev3 = t_; } while (0)
#define S3_FLUSH do { if (tid == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_stamps[8 + i_], (unsigned long long)tacc_[i_]); } } while (0)
#else
#define S3_INIT do { } while (0)
#define S3(k) do { } while (0)
#define S3_FLUSH do { } while (0)
#endif

struct __attribute__((aligned(16))) SItem {
  float lx, ly, a;
  int q;        // float offset of the query's staged grad_out row
};

constexpr int kBins3 = 1280;       // sum over levels of (wh + 1) * (ww + 1) at most (1 070 at config #2 with 5 + 5 margins)
constexpr int kHeavy = 48;         // contributions from which a pixel becomes a whole wave's task

template <int NT>
__global__ __launch_bounds__(NT, 6) void k_scatter_col3(const float* __restrict__ loc, const float* __restrict__ attn,
                                                         const float* __restrict__ gout, int S, int M, int P, ColGeom geo,
                                                         float* __restrict__ gvalue) {
  constexpr int G = 8, D = 32, GROUPS = NT / G, NW = NT / 64, BPL = kBins3 / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* gs = reinterpret_cast<float*>(smem);                                             // [tmax][D], channel j + G*c at 4*j + c
  SItem* items = reinterpret_cast<SItem*>(smem + (size_t)geo.tmax * D * sizeof(float));    // [L * tmax * P]
  int* qg = reinterpret_cast<int*>(items + (size_t)geo.L * geo.tmax * P);                  // [tmax]
  __shared__ int cnt[kBins3], start[kBins3 + 1];
  __shared__ unsigned short task_n[kBins3], task_w[256];
  __shared__ int boff[kLM + 1], n_narrow, n_wide;
  __shared__ TileCtx tc;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bid = blockIdx.x;
  const int bt = (int)udiv(bid, M, geo.m_M);
  const int m = bid - bt * M;
  const int b = (int)udiv(bt, geo.ntiles, geo.m_ntiles), tile = bt - b * geo.ntiles;
  S3_INIT;
  for (int i = tid; i < kBins3; i += NT) cnt[i] = 0;
  if (tid == 0) { n_narrow = 0; n_wide = 0; }
  tile_setup(geo, tile, tc, tid);
  const int L = geo.L, NS = L * P, MD = M * D;
  const int T = tc.qbase[L], TP = T * P;
  if (tid <= L) {
    int o = 0;
    for (int k = 0; k < tid; ++k) o += (tc.wh[k] + 1) * (tc.ww[k] + 1);
    boff[tid] = o;
  }
  for (int i = tid; i < T; i += NT) qg[i] = local_to_query(tc, L, i);
  __syncthreads();
  S3(0);
  const int j = tid % G;
  const int ts = min(tid, TP - 1);
  const int qloc = (int)udiv(ts, P, geo.m_P);
  float2 sxy[kLM];
  float sa[kLM];
  {
    const int p = ts - qloc * P;
    const long long wi0 = (((long long)b * S + qg[qloc]) * M + m) * NS + p;
#pragma unroll
    for (int l = 0; l < kLM; ++l) {
      const long long wi = wi0 + min(l, L - 1) * P;
      sxy[l] = *reinterpret_cast<const float2*>(loc + wi * 2);
      sa[l] = attn[wi];
    }
  }
  {
    float4 gq[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ql = min(tid / G + u * GROUPS, T - 1);
      const float* g = gout + (((long long)b * S + qg[ql]) * M + m) * D + j;
      gq[u] = make_float4(g[0], g[G], g[2 * G], g[3 * G]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ql = tid / G + u * GROUPS;
      if (ql < T) *reinterpret_cast<float4*>(gs + ql * D + 4 * j) = gq[u];
    }
  }
  float* gvb = gvalue + (long long)b * S * MD + m * D;
  // (1) bin this thread's sample of every level by its top-left pixel.  bin >= 0: inside the window; bin = -1 - mask: outside (mask = its
  // corners inside the map, 0 = no sample), and `slot` then holds the corner-0 pixel of the direct path instead of a list slot
  int bin[kLM], slot[kLM];
  float slx[kLM], sly[kLM];
#pragma unroll
  for (int l = 0; l < kLM; ++l) {
    bin[l] = -1; slot[l] = 0; slx[l] = 0.f; sly[l] = 0.f;
    if (l < L && tid < TP) {
      const int H = tc.H[l], W = tc.W[l], wy0 = tc.wy0[l], wx0 = tc.wx0[l], wh = tc.wh[l], ww = tc.ww[l];
      const float h_im = sxy[l].y * (float)H - 0.5f, w_im = sxy[l].x * (float)W - 0.5f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        const int y0 = (int)floorf(h_im), x0 = (int)floorf(w_im);
        sly[l] = h_im - (float)y0; slx[l] = w_im - (float)x0;
        const bool in = max(y0, 0) >= wy0 && min(y0 + 1, H - 1) < wy0 + wh && max(x0, 0) >= wx0 && min(x0 + 1, W - 1) < wx0 + ww;
        if (in) {
          bin[l] = boff[l] + (y0 - wy0 + 1) * (ww + 1) + (x0 - wx0 + 1);
        } else {
          const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= H - 1, x0ok = x0 >= 0, x1ok = x0 + 1 <= W - 1;
          bin[l] = -1 - ((y0ok && x0ok ? 1 : 0) | (y0ok && x1ok ? 2 : 0) | (y1ok && x0ok ? 4 : 0) | (y1ok && x1ok ? 8 : 0));
          slot[l] = y0 * W + x0;
        }
      }
    }
  }
#pragma unroll
  for (int l = 0; l < kLM; ++l)
    if (bin[l] >= 0) slot[l] = atomicAdd(&cnt[bin[l]], 1);
  __syncthreads();           // gs staged (the direct path below reads it), counts complete
  S3(1);
#pragma unroll
  for (int l = 0; l < kLM; ++l) {
    if (l < L) {
      // samples outside the window: straight to memory, one sample per wave step, D lanes x 4 B contiguous per corner
      const int W = tc.W[l];
      float* gvl = gvb + (long long)tc.S0[l] * MD;
      unsigned long long bal = __ballot(bin[l] < -1);
      while (bal) {
        const int src = __ffsll((long long)bal) - 1;
        bal &= bal - 1;
        const int om = -1 - __builtin_amdgcn_readlane(bin[l], src), gp = __builtin_amdgcn_readlane(slot[l], src),
                  qs = __builtin_amdgcn_readlane(qloc, src);
        const float lx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(slx[l]), src));
        const float ly = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sly[l]), src));
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sa[l]), src));
        const float hy = 1.f - ly, hx = 1.f - lx;
        if (lane < D) {
          const float g = gs[qs * D + 4 * (lane % G) + lane / G];
          if (om & 1) atomicAdd(gvl + (long long)gp * MD + lane, hy * hx * a * g);
          if (om & 2) atomicAdd(gvl + (long long)(gp + 1) * MD + lane, hy * lx * a * g);
          if (om & 4) atomicAdd(gvl + (long long)(gp + W) * MD + lane, ly * hx * a * g);
          if (om & 8) atomicAdd(gvl + (long long)(gp + W + 1) * MD + lane, ly * lx * a * g);
        }
      }
    }
  }
  // (2) wave 0: list starts of all bins; the other waves: the task lists (a pixel's contributions = the counts of its four bins)
  S3(2);
  const int npix = tc.woff[L];
  if (wave == 0) {
    int c[BPL], s = 0;
#pragma unroll
    for (int i = 0; i < BPL; ++i) { c[i] = cnt[lane * BPL + i]; s += c[i]; }
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    int run = inc - s;
#pragma unroll
    for (int i = 0; i < BPL; ++i) { start[lane * BPL + i] = run; run += c[i]; }
    if (lane == 63) start[kBins3] = run;
  } else {
    for (int p0 = 0; p0 < npix; p0 += NT - 64) {
      const int pix = p0 + tid - 64;
      int work = 0;
      if (pix < npix) {
        int l = 0;
#pragma unroll
        for (int k = 1; k < kLM; ++k) if (k < L && pix >= tc.woff[k]) l = k;
        const int r = pix - tc.woff[l], ww = tc.ww[l];
        const int dy = (int)udiv(r, ww, tc.m_ww[l]), dx = r - dy * ww;
        const int b3 = boff[l] + dy * (ww + 1) + dx;
        work = cnt[b3] + cnt[b3 + 1] + cnt[b3 + ww + 1] + cnt[b3 + ww + 2];
      }
      const bool nar = work > 0 && work <= kHeavy, wid = work > kHeavy;
      const unsigned long long bn = __ballot(nar), bw = __ballot(wid);
      int basen = 0, basew = 0;
      if (lane == 0) {
        if (bn) basen = atomicAdd(&n_narrow, __popcll(bn));
        if (bw) basew = atomicAdd(&n_wide, __popcll(bw));
      }
      basen = __builtin_amdgcn_readfirstlane(basen);
      basew = __builtin_amdgcn_readfirstlane(basew);
      const unsigned long long below = (1ull << lane) - 1ull;
      if (nar) task_n[basen + __popcll(bn & below)] = (unsigned short)pix;
      if (wid) task_w[basew + __popcll(bw & below)] = (unsigned short)pix;
    }
  }
  lds_barrier();
  S3(3);
  // (3) the items into their lists
  {
    int st[kLM];
#pragma unroll
    for (int l = 0; l < kLM; ++l) st[l] = start[max(bin[l], 0)];
#pragma unroll
    for (int l = 0; l < kLM; ++l)
      if (bin[l] >= 0) {
        SItem it;
        it.lx = slx[l]; it.ly = sly[l]; it.a = sa[l]; it.q = qloc * D;
        items[st[l] + slot[l]] = it;
      }
  }
  lds_barrier();
  S3(4);
  // (4) sums.  Corner k of a sample is pixel (y0 + (k >> 1), x0 + (k & 1)): pixel (dy, dx) of the window takes corner 3 from bin
  // (dy, dx), corner 2 from (dy, dx + 1), corner 1 from (dy + 1, dx), corner 0 from (dy + 1, dx + 1) of the bin grid.
  const int nw = n_wide, nn = n_narrow;
  const int grp8 = lane >> 3;
  const float* gsj = gs + 4 * j;
  for (int t = wave; t < nw; t += NW) {         // heavy pixels: one wave each, its 8 lane groups stride over the lists
    const int pix = task_w[t];
    int l = 0;
#pragma unroll
    for (int k = 1; k < kLM; ++k) if (k < L && pix >= tc.woff[k]) l = k;
    const int r = pix - tc.woff[l], ww = tc.ww[l];
    const int dy = (int)udiv(r, ww, tc.m_ww[l]), dx = r - dy * ww;
    const int b3 = boff[l] + dy * (ww + 1) + dx;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      const int bk = b3 + ((k & 2) ? 0 : ww + 1) + ((k & 1) ? 0 : 1);
      const SItem* lst = items + start[bk];
      const int n = cnt[bk];
      const float sy = (k & 2) ? 1.f : -1.f, oy = (k & 2) ? 0.f : 1.f, sx = (k & 1) ? 1.f : -1.f, ox = (k & 1) ? 0.f : 1.f;   // corner weight = (oy + sy ly)(ox + sx lx) a
      for (int i = grp8; i < n; i += 16) {
        const bool two = i + 8 < n;
        const SItem i0 = lst[i], i1 = lst[two ? i + 8 : i];
        const float4 g0 = ld4(gsj + i0.q), g1 = ld4(gsj + i1.q);
        const float w0 = (oy + sy * i0.ly) * (ox + sx * i0.lx) * i0.a;
        const float w1 = two ? (oy + sy * i1.ly) * (ox + sx * i1.lx) * i1.a : 0.f;
        acc.x += w0 * g0.x + w1 * g1.x; acc.y += w0 * g0.y + w1 * g1.y; acc.z += w0 * g0.z + w1 * g1.z; acc.w += w0 * g0.w + w1 * g1.w;
      }
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {          // every lane group ends with the pixel's total
      acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
      acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
    }
    if (lane < D) {                             // lane i: channel i = component i / 8 of lane group i / 8 (whose j is i % 8)
      const int gpix = tc.S0[l] + (tc.wy0[l] + dy) * tc.W[l] + tc.wx0[l] + dx;
      const float v = grp8 == 0 ? acc.x : grp8 == 1 ? acc.y : grp8 == 2 ? acc.z : acc.w;
      atomicAdd(gvb + (long long)gpix * MD + lane, v);
    }
  }
  S3(5);
  const int odd = (tid / G) & 1;
  for (int kk = (tid / (2 * G)) * 2 + odd; kk - odd < nn; kk += GROUPS) {       // light pixels: one lane group each; pairs flush full 64-B lines
    const bool valid = kk < nn;
    const int pix = valid ? task_n[kk] : 0;
    int l = 0;
#pragma unroll
    for (int k = 1; k < kLM; ++k) if (k < L && pix >= tc.woff[k]) l = k;
    const int r = pix - tc.woff[l], ww = tc.ww[l];
    const int dy = (int)udiv(r, ww, tc.m_ww[l]), dx = r - dy * ww;
    const int b3 = boff[l] + dy * (ww + 1) + dx;
    const int gpix = valid ? tc.S0[l] + (tc.wy0[l] + dy) * tc.W[l] + tc.wx0[l] + dx : -1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      const int bk = b3 + ((k & 2) ? 0 : ww + 1) + ((k & 1) ? 0 : 1);
      const SItem* lst = items + start[bk];
      const int n = valid ? cnt[bk] : 0;
      const float sy = (k & 2) ? 1.f : -1.f, oy = (k & 2) ? 0.f : 1.f, sx = (k & 1) ? 1.f : -1.f, ox = (k & 1) ? 0.f : 1.f;
      for (int i = 0; i < n; i += 2) {
        const bool two = i + 1 < n;
        const SItem i0 = lst[i], i1 = lst[two ? i + 1 : i];
        const float4 g0 = ld4(gsj + i0.q), g1 = ld4(gsj + i1.q);
        const float w0 = (oy + sy * i0.ly) * (ox + sx * i0.lx) * i0.a;
        const float w1 = two ? (oy + sy * i1.ly) * (ox + sx * i1.lx) * i1.a : 0.f;
        acc.x += w0 * g0.x + w1 * g1.x; acc.y += w0 * g0.y + w1 * g1.y; acc.z += w0 * g0.z + w1 * g1.z; acc.w += w0 * g0.w + w1 * g1.w;
      }
    }
#ifndef EXP_NO_FLUSH
    {
      const float send1 = odd ? acc.x : acc.y, send2 = odd ? acc.z : acc.w;
      const float got1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send1), 0x128, 0xF, 0xF, true));   // row_ror:8 = lane ^ 8
      const float got2 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send2), 0x128, 0xF, 0xF, true));
      const int opx = __builtin_amdgcn_mov_dpp(gpix, 0x128, 0xF, 0xF, true);
      const int pixA = odd ? opx : gpix, pixB = odd ? gpix : opx;
      const int jj = j + (odd ? G : 0);
      if (pixA >= 0) {
        float* g = gvb + (long long)pixA * MD + jj;
        atomicAdd(g, odd ? got1 : acc.x);
        atomicAdd(g + 2 * G, odd ? got2 : acc.z);
      }
      if (pixB >= 0) {
        float* g = gvb + (long long)pixB * MD + jj;
        atomicAdd(g, odd ? acc.y : got1);
        atomicAdd(g + 2 * G, odd ? acc.w : got2);
      }
    }
#else
    if (valid && acc.x + acc.y + acc.z + acc.w == 1.2345f) gvb[gpix] = acc.x;     // timing-only build
#endif
  }
  S3(6);
  S3_FLUSH;
}

// ----------------------------------------------------------------------------------------------------------------------
// Backward-scatter, one pass, PATCH-owned sums (round 4, second step).  Phase cuts of the kernel above (tools/r4_exp5.sh, N = 10, ring
// offsets): loads + binning 73 us, light-pixel sums 81 us, heavy-pixel sums 63 us of 204 -- (i) the loads were 4- and 8-byte
// pieces (a thread = one point of every level: 16 wave-level loads of 16 partial lines each) and (ii) the sums are bound by LDS
// BANDWIDTH: every corner contribution re-reads its query's 128-byte grad_out row, 26 M rows = 3.3 GB = 42 us at 128 B/clk/CU before
// any inefficiency.  Here
//   * a thread owns the four POINTS of one (query, level): two 16-byte loads of locations, one of weights; rows come in as 16-byte
//     pieces and are transposed to the flush layout on their way into LDS;
//   * the sums are owned by 2 x 2-pixel PATCHES: a lane group keeps the patch's 4 pixels in registers (16 floats per lane), walks
//     the 3 x 3 bins whose samples can touch the patch and reads each sample's row ONCE for all its corners inside the patch
//     (2.25 row reads per sample instead of 4), four items in flight; the bin position relative to the patch is a template
//     parameter, so which accumulators a sample feeds is known at compile time (no predicated FMAs).  (2 x 4 patches = 1.9 reads
//     per sample were built first: 32 accumulators left no registers for a second item in flight at 6 waves per SIMD.)
//   * heavy patches (coarse levels) are a wave's task: its 8 lane groups stride over the lists, then a 12-exchange reduce-scatter
//     leaves pixel g / 2 of the patch with lane groups g, g ^ 1, and the even ones flush.
struct __attribute__((aligned(16))) PItem {
  float wy0, wy1;   // (1 - ly) * a, ly * a
  float lx;
  int q;            // float offset of the query's staged grad_out row
};

#ifndef EXP_HEAVYP
#define EXP_HEAVYP 64
#endif
constexpr int kHeavyP = EXP_HEAVYP;        // sample reads from which a patch becomes a whole wave's task
constexpr int kPatchMax = 512;
// Light patches in order of their work (round 4, third step).  A wave's eight lane groups walk eight patches in lockstep, so a trip
// costs the wave what its LONGEST list costs: patches are ranked by work (a 16-bucket counting sort in LDS, heaviest first) and a wave
// draws eight NEIGHBOURS of that order at a time from a workgroup counter -- groups of a wave see lists of like length and the waves
// balance themselves (longest tasks first).  0 = the arrival order and the static stride of the first col4.
#ifndef EXP_TASK_SORT
#define EXP_TASK_SORT 1
#endif
#ifndef EXP_INFLIGHT
#define EXP_INFLIGHT 4
#endif
#ifndef EXP_INFLIGHT_W
#define EXP_INFLIGHT_W 1
#endif
constexpr int kWorkBuckets = 16;
constexpr int kInFlight = EXP_INFLIGHT;    // items a lane group of a LIGHT patch keeps in flight per trip

// One bin of a patch.  The bin's position (BY, BX) relative to the patch is static, so which accumulators a sample feeds is known at
// compile time.  Four items in flight per trip.  (Also built and measured, tools/r4_exp9.sh: the three bins of a bin row as ONE contiguous
// run of items -- they are consecutive in the scan order -- with the bin column carried in the item: 7 instead of 27 dependent LDS round
// trips per patch, but four selects per item for the column weights: 192 against 177 us.  The sums are bound by instruction issue at
// low lane efficiency -- lane groups of a wave walk lists of different lengths -- not by LDS latency or bandwidth.)
#ifndef EXP_LEAN_SUM
#define EXP_LEAN_SUM 1
#endif
constexpr int kItemTail = 32;              // zeroed items behind the last list: a trip may read up to 3 * 8 + 7 items past its list's end
template <int BY, int BX, int step, int NF = 4>
__device__ __forceinline__ void patch_bin_sum(const PItem* __restrict__ lst, int n, int first, const float* __restrict__ gsj, int zq,
                                              const PItem* __restrict__ zit, float4 (&acc)[4]) {
#if EXP_LEAN_SUM
  // Issue-bound loop (section 4.2c): no per-item index clamp and no per-item weight select.  A trip reads its NF items at constant
  // offsets from one address -- past the end of a list lie the next lists' items (finite weights, valid rows) and behind the last list
  // kItemTail zeroed items -- and an item past the end takes the ZERO ROW (zq) instead of its query's row: finite x 0 adds nothing.
  // (Item weights are attention weight x bilinear weight: finite whenever the attention weights are.  A non-finite attention weight
  // makes its own pixels non-finite in any implementation; here it can also reach the pixels of the list read past -- a step whose
  // attention weights overflowed has no usable gradient either way.  grad_out rows are never read past a list: that is the zero row.)
  // (no unrolling across trips: hipcc's own 2 x unroll of this loop -- nine instances per patch kind -- cost 28-36 bytes of scratch at the 80-register cap, a
  // remainder loop per instance and 11-14 us of the launch: tools/ab_gv_variants.sh, 161 -> 147 us on cold operands)
#if !defined(EXP_UNROLL) || !EXP_UNROLL
#pragma unroll 1
#endif
  for (int i = first; i < n; i += NF * step) {
    const PItem* p = lst + i;
    PItem it[NF];
    float4 g[NF];
#if defined(EXP_ZERO_ITEM) && EXP_ZERO_ITEM
    // variant (unmeasured: DESIGN section 9): an item past the end is replaced by a ZEROED item whose row is the zero row, so nothing
    // of a neighbouring list is ever multiplied -- exact for non-finite attention weights too; + 1 instruction per item
#pragma unroll
    for (int u = 0; u < NF; ++u) it[u] = *(i + u * step < n ? p + u * step : zit);
#pragma unroll
    for (int u = 0; u < NF; ++u) g[u] = ld4(gsj + it[u].q);
#else
#pragma unroll
    for (int u = 0; u < NF; ++u) it[u] = p[u * step];
#pragma unroll
    for (int u = 0; u < NF; ++u) g[u] = ld4(gsj + (i + u * step < n ? it[u].q : zq));
#endif
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const float wy[2] = {it[u].wy0, it[u].wy1};
      const float wx[2] = {1.f - it[u].lx, it[u].lx};
#else
  for (int i = first; i < n; i += NF * step) {
    PItem it[NF];
#pragma unroll
    for (int u = 0; u < NF; ++u) it[u] = lst[min(i + u * step, n - 1)];
    float4 g[NF];
#pragma unroll
    for (int u = 0; u < NF; ++u) g[u] = ld4(gsj + it[u].q);
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const bool on = i + u * step < n;
      const float wy[2] = {on ? it[u].wy0 : 0.f, on ? it[u].wy1 : 0.f};
      const float wx[2] = {1.f - it[u].lx, it[u].lx};
#endif
#pragma unroll
      for (int cy = 0; cy < 2; ++cy) {
        const int py = BY - 1 + cy;
        if (py < 0 || py > 1) continue;
#pragma unroll
        for (int cx = 0; cx < 2; ++cx) {
          const int px = BX - 1 + cx;
          if (px < 0 || px > 1) continue;
          const float w = wy[cy] * wx[cx];
          float4& a = acc[py * 2 + px];
          a.x += w * g[u].x; a.y += w * g[u].y; a.z += w * g[u].z; a.w += w * g[u].w;
        }
      }
    }
  }
}

// two neighbouring lane groups (16 lanes) flush their pixels together: every atomic request carries a full 64-byte line of ONE pixel
__device__ __forceinline__ void pair_flush(const float4& acc, int gpix, int odd, int j, float* __restrict__ gvb, int MD) {
#ifdef EXP_NO_FLUSH
  if (gpix >= 0 && acc.x + acc.y + acc.z + acc.w == 1.2345f) gvb[gpix] = acc.x;     // timing-only build: keeps the sums alive, never stores
  return;
#endif
  const float send1 = odd ? acc.x : acc.y, send2 = odd ? acc.z : acc.w;
  const float got1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send1), 0x128, 0xF, 0xF, true));   // row_ror:8 = lane ^ 8
  const float got2 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send2), 0x128, 0xF, 0xF, true));
  const int opx = __builtin_amdgcn_mov_dpp(gpix, 0x128, 0xF, 0xF, true);
  const int pixA = odd ? opx : gpix, pixB = odd ? gpix : opx;
  const int jj = j + (odd ? 8 : 0);
  if (pixA >= 0) {
    float* g = gvb + (long long)pixA * MD + jj;
    atomicAdd(g, odd ? got1 : acc.x);
    atomicAdd(g + 16, odd ? got2 : acc.z);
  }
  if (pixB >= 0) {
    float* g = gvb + (long long)pixB * MD + jj;
    atomicAdd(g, odd ? acc.y : got1);
    atomicAdd(g + 16, odd ? acc.w : got2);
  }
}

template <int NT>
__global__ __launch_bounds__(NT, 6) void k_scatter_col4(const float* __restrict__ loc, const float* __restrict__ attn,
                                                         const float* __restrict__ gout, int S, int M, ColGeom geo,
                                                         float* __restrict__ gvalue, int* __restrict__ sel, int to_tile_pct) {
  // path selection (msda_col.h): this kernel is path 0; when the call site's state says the output-tiled kernels serve this call, every
  // workgroup leaves at once
#ifndef EXP_SEL_CHECK
#define EXP_SEL_CHECK 1
#endif
#ifndef EXP_SEL_STAT
#define EXP_SEL_STAT 1
#endif
#if EXP_SEL_CHECK
  if (sel != nullptr && sel[kSelCur] != 0) return;
#endif
  constexpr int G = 8, D = 32, P = 4, GROUPS = NT / G, NW = NT / 64, BPL = kBins3 / 64;
  static_assert(kPatchMax <= NT - 64, "a thread of waves 1.. holds at most one patch");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* gs = reinterpret_cast<float*>(smem);                                             // [tmax + 1][D], channel j + G*c at 4*j + c; row tmax = zeros
  PItem* items = reinterpret_cast<PItem*>(smem + (size_t)(geo.tmax + 1) * D * sizeof(float));    // [L * tmax * P + kItemTail]
  int* qg = reinterpret_cast<int*>(items + (size_t)geo.L * geo.tmax * P + kItemTail);      // [tmax]
  __shared__ int cnt[kBins3], start[kBins3 + 1];
  __shared__ unsigned short task_n[kPatchMax], task_w[kPatchMax];
  __shared__ int boff[kLM + 1], poff[kLM + 1], pcw[kLM], n_narrow, n_wide, n_far;
  __shared__ int whist[kWorkBuckets], next_task;
  __shared__ unsigned m_pcw[kLM];
  __shared__ TileCtx tc;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bid = blockIdx.x;
  const int bt = (int)udiv(bid, M, geo.m_M);
  const int m = bid - bt * M;
  const int b = (int)udiv(bt, geo.ntiles, geo.m_ntiles), tile = bt - b * geo.ntiles;
  for (int i = tid; i < kBins3; i += NT) cnt[i] = 0;
  if (tid < kWorkBuckets) whist[tid] = 0;
  if (tid >= 64 && tid < 64 + D) gs[geo.tmax * D + tid - 64] = 0.f;      // the zero row (patch_bin_sum)
  if (tid == 0) { n_narrow = 0; n_wide = 0; n_far = 0; next_task = 0; }
  tile_setup(geo, tile, tc, tid);
  const int L = geo.L, NS = L * P, MD = M * D;
  const int T = tc.qbase[L], TL = T * L;
  if (tid <= L) {
    int o = 0, po = 0;
    for (int k = 0; k < tid; ++k) {
      o += (tc.wh[k] + 1) * (tc.ww[k] + 1);
      po += ((tc.wh[k] + 1) >> 1) * ((tc.ww[k] + 1) >> 1);
    }
    boff[tid] = o;
    poff[tid] = po;
    if (tid < L) {
      pcw[tid] = (tc.ww[tid] + 1) >> 1;
      m_pcw[tid] = magic_of((unsigned)((tc.ww[tid] + 1) >> 1));
    }
  }
  for (int i = tid; i < T; i += NT) qg[i] = local_to_query(tc, L, i);
  __syncthreads();
  const int j = tid % G;
  // this thread: the four points of (query ql, level lv)
  const int ts = min(tid, TL - 1);
  const int ql = L == 4 ? ts >> 2 : L == 2 ? ts >> 1 : L == 3 ? ts / 3 : ts;
  const int lv = ts - ql * L;
  float4 la, lb, wa;
  {
    const long long wi = (((long long)b * S + qg[ql]) * M + m) * NS + lv * P;
    la = ld4(loc + wi * 2); lb = ld4(loc + wi * 2 + 4); wa = ld4(attn + wi);
  }
  {
    float4 gq[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int qr = min(tid / G + u * GROUPS, T - 1);
      gq[u] = ld4(gout + (((long long)b * S + qg[qr]) * M + m) * D + 4 * j);           // channels 4j .. 4j+3
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int qr = tid / G + u * GROUPS;
      if (qr < T) {                 // channel c sits at 4 * (c % 8) + c / 8: position 16 * (j & 1) + 4 * i + (j >> 1) for c = 4j + i
        float* d = gs + qr * D + 16 * (j & 1) + (j >> 1);
        d[0] = gq[u].x; d[4] = gq[u].y; d[8] = gq[u].z; d[12] = gq[u].w;
      }
    }
  }
  float* gvb = gvalue + (long long)b * S * MD + m * D;
  // (1) bin the four samples by their top-left pixel.  bin >= 0: inside the window; bin = -1 - mask: outside (mask = corners inside the map,
  // 0 = no sample) and `slot` = the corner-0 pixel for the direct path
  const int lH = tc.H[lv], lW = tc.W[lv];
  int bin[4], slot[4];
  float wy0[4], wy1[4], slx[4];
  {
    const int wy0_ = tc.wy0[lv], wx0_ = tc.wx0[lv], wh = tc.wh[lv], ww = tc.ww[lv], bo = boff[lv];
    const float px_[4] = {la.x, la.z, lb.x, lb.z}, py_[4] = {la.y, la.w, lb.y, lb.w}, pa_[4] = {wa.x, wa.y, wa.z, wa.w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      bin[p] = -1; slot[p] = 0; wy0[p] = 0.f; wy1[p] = 0.f; slx[p] = 0.f;
      const float h_im = py_[p] * (float)lH - 0.5f, w_im = px_[p] * (float)lW - 0.5f;
      if (tid < TL && h_im > -1.f && w_im > -1.f && h_im < (float)lH && w_im < (float)lW) {
        const int y0 = (int)floorf(h_im), x0 = (int)floorf(w_im);
        const float ly = h_im - (float)y0;
        slx[p] = w_im - (float)x0;
        wy0[p] = (1.f - ly) * pa_[p];
        wy1[p] = ly * pa_[p];
        const bool in = max(y0, 0) >= wy0_ && min(y0 + 1, lH - 1) < wy0_ + wh && max(x0, 0) >= wx0_ && min(x0 + 1, lW - 1) < wx0_ + ww;
        if (in) {
          bin[p] = bo + (y0 - wy0_ + 1) * (ww + 1) + (x0 - wx0_ + 1);
        } else {
          const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= lH - 1, x0ok = x0 >= 0, x1ok = x0 + 1 <= lW - 1;
          bin[p] = -1 - ((y0ok && x0ok ? 1 : 0) | (y0ok && x1ok ? 2 : 0) | (y1ok && x0ok ? 4 : 0) | (y1ok && x1ok ? 8 : 0));
          slot[p] = y0 * lW + x0;
        }
      }
    }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p)
    if (bin[p] >= 0) slot[p] = atomicAdd(&cnt[bin[p]], 1);
  if (EXP_SEL_STAT == 1 ? sel != nullptr : EXP_SEL_STAT == 2 ? (sel != nullptr && (blockIdx.x & 15) == 0) : false) {      // this call's share of samples outside the windows: what the NEXT call at this site is dispatched on
    int nf = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) nf += __popcll(__ballot(bin[p] < -1));
    if (lane == 0 && nf) atomicAdd(&n_far, nf);
  }
  __syncthreads();           // gs staged (the direct path below reads it), counts complete
  // the call site's statistics: every 16th workgroup reports (400 k samples estimate a share well enough) with one fire-and-forget atomic
  if (EXP_SEL_STAT != 0 && sel != nullptr && tid == NT - 1 && (bid & 15) == 0) sel_report(sel, n_far, TL * 4);
#if defined(EXP4_CUT) && EXP4_CUT == 1
  return;                    // timing-only build: loads + binning
#endif
  {
    // samples outside the window: straight to memory, one sample per wave step, D lanes x 4 B contiguous per corner
    const int lS0 = tc.S0[lv];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      unsigned long long bal = __ballot(bin[p] < -1);
#ifdef EXP_NO_FAR
      bal = 0;               // timing-only build: samples outside the windows are dropped
#endif
      while (bal) {
        const int src = __ffsll((long long)bal) - 1;
        bal &= bal - 1;
        const int om = -1 - __builtin_amdgcn_readlane(bin[p], src), gp = __builtin_amdgcn_readlane(slot[p], src),
                  qs = __builtin_amdgcn_readlane(ql, src), W = __builtin_amdgcn_readlane(lW, src), s0 = __builtin_amdgcn_readlane(lS0, src);
        const float lx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(slx[p]), src));
        const float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wy0[p]), src));
        const float a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wy1[p]), src));
        if (lane < D) {
          const float g = gs[qs * D + 4 * (lane % G) + lane / G];
          float* gvl = gvb + (long long)s0 * MD + lane;
          if (om & 1) atomicAdd(gvl + (long long)gp * MD, a0 * (1.f - lx) * g);
          if (om & 2) atomicAdd(gvl + (long long)(gp + 1) * MD, a0 * lx * g);
          if (om & 4) atomicAdd(gvl + (long long)(gp + W) * MD, a1 * (1.f - lx) * g);
          if (om & 8) atomicAdd(gvl + (long long)(gp + W + 1) * MD, a1 * lx * g);
        }
      }
    }
  }
  // (2) wave 0: list starts of all bins; the other waves: the patch task lists (a patch's work = the counts of its 3 x 3 bins)
  const int npatch = poff[L];
#if EXP_TASK_SORT
  int my_pt = -1, my_bucket = 0, my_rank = 0;     // this thread's light patch, its work bucket and its arrival rank inside the bucket
#endif
  if (wave == 0) {
    int c[BPL], s = 0;
#pragma unroll
    for (int i = 0; i < BPL; ++i) { c[i] = cnt[lane * BPL + i]; s += c[i]; }
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    int run = inc - s;
#pragma unroll
    for (int i = 0; i < BPL; ++i) { start[lane * BPL + i] = run; run += c[i]; }
    if (lane == 63) start[kBins3] = run;
  } else {
    for (int p0 = 0; p0 < npatch; p0 += NT - 64) {
      const int pt = p0 + tid - 64;
      int work = 0;
      if (pt < npatch) {
        int l = 0;
#pragma unroll
        for (int k = 1; k < kLM; ++k) if (k < L && pt >= poff[k]) l = k;
        const int r = pt - poff[l], ww = tc.ww[l], wh = tc.wh[l];
        const int pr = (int)udiv(r, pcw[l], m_pcw[l]), pc = r - pr * pcw[l];
        const int b0 = boff[l] + 2 * pr * (ww + 1) + 2 * pc;
#pragma unroll
        for (int by = 0; by < 3; ++by)
#pragma unroll
          for (int bx = 0; bx < 3; ++bx)
            if (2 * pr + by <= wh && 2 * pc + bx <= ww) work += cnt[b0 + by * (ww + 1) + bx];
      }
      const bool nar = work > 0 && work <= kHeavyP, wid = work > kHeavyP;
      const unsigned long long bn = __ballot(nar), bw = __ballot(wid);
      (void)bn;
      int basen = 0, basew = 0;
      if (lane == 0) {
#if !EXP_TASK_SORT
        if (bn) basen = atomicAdd(&n_narrow, __popcll(bn));
#endif
        if (bw) basew = atomicAdd(&n_wide, __popcll(bw));
      }
      basen = __builtin_amdgcn_readfirstlane(basen);
      basew = __builtin_amdgcn_readfirstlane(basew);
      const unsigned long long below = (1ull << lane) - 1ull;
#if EXP_TASK_SORT
      if (nar) {                  // npatch <= kPatchMax <= NT - 64: one trip of this loop, at most one patch per thread
        my_pt = pt;
        my_bucket = (kHeavyP - work) * kWorkBuckets / kHeavyP;
        my_rank = atomicAdd(&whist[my_bucket], 1);
      }
#else
      if (nar) task_n[basen + __popcll(bn & below)] = (unsigned short)pt;
#endif
      if (wid) task_w[basew + __popcll(bw & below)] = (unsigned short)pt;
    }
  }
  lds_barrier();
  // (3) the items into their lists
  {
    int st[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) st[p] = start[max(bin[p], 0)];
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (bin[p] >= 0) {
        PItem it;
        it.wy0 = wy0[p]; it.wy1 = wy1[p]; it.lx = slx[p]; it.q = ql * D;
        items[st[p] + slot[p]] = it;
      }
  }
  if (tid >= NT - kItemTail) {        // zeroed items behind the last list (patch_bin_sum reads past the end of a list)
    PItem z;
    z.wy0 = 0.f; z.wy1 = 0.f; z.lx = 0.f; z.q = geo.tmax * D;       // (its row: the zero row)
    items[start[kBins3] + tid - (NT - kItemTail)] = z;
  }
#if EXP_TASK_SORT
  {
    // the light patches into their ranked order: bucket starts from the histogram the barrier above completed
    int base = 0, all = 0;
#pragma unroll
    for (int k = 0; k < kWorkBuckets; ++k) {
      const int h = whist[k];
      base += k < my_bucket ? h : 0;
      all += h;
    }
    if (my_pt >= 0) task_n[base + my_rank] = (unsigned short)my_pt;
    if (tid == 64) n_narrow = all;
  }
#endif
  lds_barrier();
#if defined(EXP4_CUT) && EXP4_CUT == 2
  return;                    // timing-only build: everything but the sums and the flush
#endif
  // (4) sums.  Bin (br, bc) of a level's grid holds the samples whose top-left pixel is window pixel (br - 1, bc - 1); patch (pr, pc) owns window
  // pixels rows 2pr .. 2pr+1, columns 2pc .. 2pc+1 and is touched by bins rows 2pr .. 2pr+2, columns 2pc .. 2pc+2.
  const int nw = n_wide, nn = n_narrow;
  const int grp8 = lane >> 3;
  const int odd = grp8 & 1;
  const float* gsj = gs + 4 * j;
  const int zq = geo.tmax * D;
  const PItem* zit = items + start[kBins3];       // the first of the zeroed items behind the last list
#define OCPG_PATCH_BIN(BY, BX)                                                                                     \
  {                                                                                                                \
    const bool ok_ = live_ && 2 * pr + BY <= wh && 2 * pc + BX <= ww;                                              \
    const int bk_ = ok_ ? b0 + BY * (ww + 1) + BX : b0;                                                            \
    patch_bin_sum<BY, BX, STEP_, NF_>(items + start[bk_], ok_ ? cnt[bk_] : 0, first, gsj, zq, zit, acc);          \
  }
#define OCPG_PATCH_ALL                                                                                             \
  OCPG_PATCH_BIN(0, 0) OCPG_PATCH_BIN(0, 1) OCPG_PATCH_BIN(0, 2) OCPG_PATCH_BIN(1, 0) OCPG_PATCH_BIN(1, 1) OCPG_PATCH_BIN(1, 2)  \
  OCPG_PATCH_BIN(2, 0) OCPG_PATCH_BIN(2, 1) OCPG_PATCH_BIN(2, 2)
  for (int t = wave; t < nw; t += NW) {         // heavy patches: one wave each
    const int pt = task_w[t];
    int l = 0;
#pragma unroll
    for (int k = 1; k < kLM; ++k) if (k < L && pt >= poff[k]) l = k;
    const int r = pt - poff[l], ww = tc.ww[l], wh = tc.wh[l];
    const int pr = (int)udiv(r, pcw[l], m_pcw[l]), pc = r - pr * pcw[l];
    const int b0 = boff[l] + 2 * pr * (ww + 1) + 2 * pc;
    float4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int first = grp8;
    const bool live_ = true;
#define STEP_ 8
#define NF_ EXP_INFLIGHT_W
    OCPG_PATCH_ALL
#undef NF_
#undef STEP_
    // reduce-scatter over the 8 lane groups: lane groups g, g ^ 1 end with the total of pixel g / 2 of the patch
    float4 kb[2], kc;
    {
      const bool hi = (grp8 & 4) != 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 keep = hi ? acc[2 + i] : acc[i], send = hi ? acc[i] : acc[2 + i];
        kb[i] = make_float4(keep.x + __shfl_xor(send.x, 32, 64), keep.y + __shfl_xor(send.y, 32, 64), keep.z + __shfl_xor(send.z, 32, 64),
                            keep.w + __shfl_xor(send.w, 32, 64));
      }
      const bool mid = (grp8 & 2) != 0;
      const float4 keep = mid ? kb[1] : kb[0], send = mid ? kb[0] : kb[1];
      kc = make_float4(keep.x + __shfl_xor(send.x, 16, 64), keep.y + __shfl_xor(send.y, 16, 64), keep.z + __shfl_xor(send.z, 16, 64),
                       keep.w + __shfl_xor(send.w, 16, 64));
      kc = make_float4(kc.x + __shfl_xor(kc.x, 8, 64), kc.y + __shfl_xor(kc.y, 8, 64), kc.z + __shfl_xor(kc.z, 8, 64), kc.w + __shfl_xor(kc.w, 8, 64));
    }
    const int prow = 2 * pr + (grp8 >> 2), pcol = 2 * pc + ((grp8 >> 1) & 1);
    const bool live = !odd && prow < wh && pcol < ww && (kc.x != 0.f || kc.y != 0.f || kc.z != 0.f || kc.w != 0.f);
    pair_flush(kc, live ? tc.S0[l] + (tc.wy0[l] + prow) * tc.W[l] + tc.wx0[l] + pcol : -1, odd, j, gvb, MD);
  }
#if EXP_TASK_SORT
  for (;;) {                                                  // light patches: a wave draws eight neighbours of the ranked order
    int kbase = 0;
    if (lane == 0) kbase = atomicAdd(&next_task, 8);
    kbase = __builtin_amdgcn_readfirstlane(kbase);
    if (kbase >= nn) break;
    const int kk = kbase + grp8;
    const bool valid = kk < nn;
#else
  for (int kk = tid / G; kk - odd < nn; kk += GROUPS) {       // light patches: one lane group each (kk - odd: pairs of groups leave the loop together)
    const bool valid = kk < nn;
#endif
    const int pt = valid ? task_n[kk] : 0;
    int l = 0;
#pragma unroll
    for (int k = 1; k < kLM; ++k) if (k < L && pt >= poff[k]) l = k;
    const int r = pt - poff[l], ww = tc.ww[l], wh = tc.wh[l];
    const int pr = (int)udiv(r, pcw[l], m_pcw[l]), pc = r - pr * pcw[l];
    const int b0 = boff[l] + 2 * pr * (ww + 1) + 2 * pc;
    float4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int first = 0;
    const bool live_ = valid;
#define STEP_ 1
#define NF_ kInFlight
    OCPG_PATCH_ALL
#undef NF_
#undef STEP_
    const int pbase = tc.S0[l] + (tc.wy0[l] + 2 * pr) * tc.W[l] + tc.wx0[l] + 2 * pc, lw = tc.W[l];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool live = valid && 2 * pr + (k >> 1) < wh && 2 * pc + (k & 1) < ww && (acc[k].x != 0.f || acc[k].y != 0.f || acc[k].z != 0.f || acc[k].w != 0.f);
      pair_flush(acc[k], live ? pbase + (k >> 1) * lw + (k & 1) : -1, odd, j, gvb, MD);
    }
  }
#undef OCPG_PATCH_ALL
#undef OCPG_PATCH_BIN

}

inline size_t scatter4_lds(const ColGeom& g) {
  return (size_t)(g.tmax + 1) * 32 * sizeof(float) + ((size_t)g.L * g.tmax * 4 + kItemTail) * sizeof(PItem) + (size_t)g.tmax * sizeof(int);
}

// the patch kernel: D = 32, P = 4, 768 threads (a thread per (query, level)), bin grids and patches within the static tables
inline bool scatter4_ok(const ColGeom& g, int D, int P) {
  if (D != 32 || P != 4 || g.L > kLM || g.tmax * g.L > 768 || g.tmax * 8 > 2 * 768) return false;
  if (scatter4_lds(g) > 68 * 1024) return false;
  int bins = 0, patches = 0;
  for (int l = 0; l < g.L; ++l) {
    const int rmax = (g.H[l] + g.nty - 1) / g.nty, cmax = (g.W[l] + g.ntx - 1) / g.ntx;
    const int wh = std::min(g.H[l], std::max(rmax, 1) + g.mlo + g.mhi), ww = std::min(g.W[l], std::max(cmax, 1) + g.mlo + g.mhi);
    bins += (wh + 1) * (ww + 1);
    patches += ((wh + 1) / 2) * ((ww + 1) / 2);
  }
  return bins <= kBins3 && patches <= kPatchMax;
}

inline size_t scatter3_lds(const ColGeom& g, int D, int P) {
  return (size_t)g.tmax * D * sizeof(float) + (size_t)g.L * g.tmax * P * sizeof(SItem) + (size_t)g.tmax * sizeof(int);
}

// the one-pass kernel: G = 8, 768 threads, every level's bin grid and window pixels within the static tables, LDS for two workgroups per CU
inline bool scatter3_ok(const ColGeom& g, int D, int P) {
  if (D != 32 || g.L > kLM || P < 1 || g.tmax * P > 768 || g.tmax * 8 > 2 * 768) return false;
  if (scatter3_lds(g, D, P) > 68 * 1024) return false;
  int bins = 0, pix = 0;
  for (int l = 0; l < g.L; ++l) {
    const int rmax = (g.H[l] + g.nty - 1) / g.nty, cmax = (g.W[l] + g.ntx - 1) / g.ntx;
    const int wh = std::min(g.H[l], std::max(rmax, 1) + g.mlo + g.mhi), ww = std::min(g.W[l], std::max(cmax, 1) + g.mlo + g.mhi);
    bins += (wh + 1) * (ww + 1);
    pix += wh * ww;
  }
  return bins <= kBins3 && pix <= kBins3 && pix <= 0xffff;
}

inline size_t gather_lds(const ColGeom& g, int D, int P) {
  return (size_t)g.wmax * D * sizeof(float) + (size_t)g.tmax * P * (sizeof(float4) + sizeof(int)) + (size_t)g.tmax * sizeof(int);
}

inline size_t scatter_lds(const ColGeom& g, int D, int P) {
  return (size_t)g.tmax * D * sizeof(float) + (size_t)g.tmax * P * 4 * sizeof(Item) + (size_t)g.tmax * sizeof(int);
}

inline size_t scatter2_lds(const ColGeom& g, int D, int P) {
  return (size_t)g.tmax * D * sizeof(float) + (size_t)g.tmax * P * 8 * sizeof(Item) + (size_t)g.tmax * sizeof(int);
}

// the level-pair kernel: G = 8 only, 768 threads, every pair of consecutive windows fits the 768 bins, LDS for two workgroups per CU
inline bool scatter2_ok(const ColGeom& g, int D, int P) {
  if (D != 32 || g.L > kLM || P < 1 || g.tmax * P > 768 || g.tmax * 8 > 2 * 768) return false;
  if (scatter2_lds(g, D, P) > 70 * 1024) return false;
  for (int l = 0; l < g.L; l += 2) {
    int px = 0;
    for (int u = l; u < std::min(l + 2, g.L); ++u) {
      const int rmax = (g.H[u] + g.nty - 1) / g.nty, cmax = (g.W[u] + g.ntx - 1) / g.ntx;
      px += std::min(g.H[u], std::max(rmax, 1) + g.mlo + g.mhi) * std::min(g.W[u], std::max(cmax, 1) + g.mlo + g.mhi);
    }
    if (px > 768) return false;
  }
  return true;
}

inline bool gather_ok(const ColGeom& g, int D, int P) {
  const int G = D / 4;
  if (D % 4 || (G != 4 && G != 8)) return false;
  const int rows = kNT / G, passes = (96 + rows - 1) / rows;
  return g.L <= kLM && g.tmax <= rows * passes && g.tmax * P <= kNT && g.wmax <= 320 && g.wrest <= 384 && P >= 1 &&
         gather_lds(g, D, P) <= 64 * 1024;
}

// threads of the scatter kernel for this geometry (0: not supported): one sample of a level per thread
inline int scatter_threads(const ColGeom& g, int D, int P) {
  const int G = D / 4;
  if (D % 4 || (G != 4 && G != 8) || g.L > kLM || P < 1 || scatter_lds(g, D, P) > 56 * 1024) return 0;
  for (int nt : {384, 768})
    if (g.tmax * P <= nt && g.wmax <= nt && g.tmax * G <= 2 * nt) return nt;
  return 0;
}

}  // namespace

bool make_col_geom(const int64_t* sh, int L, int S, int M, int P, int tile_h, int tile_w, ColGeom& g, int margin_lo, int margin_hi) {
  if (!sh || L < 1 || L > kLM || M < 1 || P < 1) return false;
  long long tot = 0, best = -1;
  int base = 0;
  for (int l = 0; l < L; ++l) {
    const long long H = sh[2 * l], W = sh[2 * l + 1];
    if (H <= 0 || W <= 0 || H > (1 << 14) || W > (1 << 14)) return false;
    g.H[l] = (int)H;
    g.W[l] = (int)W;
    g.S0[l] = (int)tot;
    if (H * W > best) { best = H * W; base = l; }
    tot += H * W;
  }
  if (tot != S) return false;
  g.L = L;
  g.mlo = margin_lo;
  g.mhi = margin_hi;
  g.nty = (g.H[base] + tile_h - 1) / tile_h;
  g.ntx = (g.W[base] + tile_w - 1) / tile_w;
  g.ntiles = g.nty * g.ntx;
  g.m_ntx = magic_of(g.ntx);
  g.m_nty = magic_of(g.nty);
  g.m_ntiles = magic_of(g.ntiles);
  g.m_M = magic_of(M);
  g.m_P = magic_of(P);
  g.tmax = 0;
  g.wmax = 0;
  g.wrest = 0;
  for (int l = 0; l < L; ++l) {
    const int rmax = (g.H[l] + g.nty - 1) / g.nty, cmax = (g.W[l] + g.ntx - 1) / g.ntx;
    g.tmax += rmax * cmax;
    const int wh = std::min(g.H[l], std::max(rmax, 1) + g.mlo + g.mhi), ww = std::min(g.W[l], std::max(cmax, 1) + g.mlo + g.mhi);
    g.wmax = std::max(g.wmax, wh * ww);
    if (l > 0) g.wrest += wh * ww;
  }
  return true;
}

bool scatter_supported(const ColGeom& g, int D, int P) { return scatter_threads(g, D, P) != 0; }

int fwd_col(const float* value, const float* loc, const float* attn, int N, int S, int M, int D, int P, const ColGeom& g, float* out,
            hipStream_t st) {
  if (!gather_ok(g, D, P)) return 0;
  const size_t lds = gather_lds(g, D, P);
  const unsigned grid = (unsigned)((long long)N * g.ntiles * M);
  if (D == 16) k_fwd_col<4><<<grid, kNT, lds, st>>>(value, loc, attn, S, M, P, g, out);
  else k_fwd_col<8><<<grid, kNT, lds, st>>>(value, loc, attn, S, M, P, g, out);
  return 1;
}

__global__ void k_sel_commit(int* sel, int to_tile_pct, int to_col_pct) {
  if (threadIdx.x != 0) return;
  unsigned long long* word = reinterpret_cast<unsigned long long*>(sel + kSelFar);
  const unsigned long long w = *word;
  const long long f = (long long)(w >> 32), t = (long long)(w & 0xffffffffull);
  if (t > 0) {        // (a call whose active family reported nothing keeps the path)
    const int cur = sel[kSelCur];
    sel[kSelNext] = cur == 0 ? (f * 100 > (long long)to_tile_pct * t ? 1 : 0) : (f * 100 > (long long)to_col_pct * t ? 1 : 0);
    sel[kSelLastFar] = (int)f;
    sel[kSelLastTotal] = (int)t;
    *word = 0ull;
  }
  sel[kSelCur] = sel[kSelNext];
}

void select_commit(int* sel, int to_tile_pct, int to_col_pct, hipStream_t st) { k_sel_commit<<<1, 64, 0, st>>>(sel, to_tile_pct, to_col_pct); }

bool select_supported(const ColGeom& g, int D, int P) {
  const char* e = std::getenv("OCPG_MSDA_COL_LP");
  return (!e || std::atoi(e) >= 4) && scatter4_ok(g, D, P);
}

int bwd_scatter_col(const float* loc, const float* attn, const float* gout, int N, int S, int M, int D, int P, const ColGeom& g,
                    float* gvalue, hipStream_t st, int* sel, int to_tile_pct) {
  {
    const char* e = std::getenv("OCPG_MSDA_COL_LP");      // A/B (read per call: tests toggle it): 4 = one pass, patch-owned sums (default), 3 = one pass, pixel-owned sums, 2 = level pairs, 1 = one level per pass
    const int lp = e ? std::atoi(e) : 4;
    if (lp >= 4 && scatter4_ok(g, D, P)) {
      const size_t lds4 = scatter4_lds(g);
      if (lds4 > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter_col4<768>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
      k_scatter_col4<768><<<(unsigned)((long long)N * g.ntiles * M), 768, lds4, st>>>(loc, attn, gout, S, M, g, gvalue, sel, to_tile_pct);
      return 2;
    }
    if (lp >= 3 && scatter3_ok(g, D, P)) {
      const size_t lds3 = scatter3_lds(g, D, P);
      if (lds3 > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter_col3<768>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      k_scatter_col3<768><<<(unsigned)((long long)N * g.ntiles * M), 768, lds3, st>>>(loc, attn, gout, S, M, P, g, gvalue);
      return 1;
    }
    if (lp >= 2 && scatter2_ok(g, D, P)) {
      const size_t lds2 = scatter2_lds(g, D, P);
      if (lds2 > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter_col2<8, 768>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      k_scatter_col2<8, 768><<<(unsigned)((long long)N * g.ntiles * M), 768, lds2, st>>>(loc, attn, gout, S, M, P, g, gvalue);
      return 1;
    }
  }
  const int nt = scatter_threads(g, D, P);
  if (!nt) return 0;
  const size_t lds = scatter_lds(g, D, P);
  const unsigned grid = (unsigned)((long long)N * g.ntiles * M);
  if (D == 16) {
    if (nt == 384) k_scatter_col<4, 384><<<grid, 384, lds, st>>>(loc, attn, gout, S, M, P, g, gvalue);
    else k_scatter_col<4, 768><<<grid, 768, lds, st>>>(loc, attn, gout, S, M, P, g, gvalue);
  } else {
    if (nt == 384) k_scatter_col<8, 384><<<grid, 384, lds, st>>>(loc, attn, gout, S, M, P, g, gvalue);
    else k_scatter_col<8, 768><<<grid, 768, lds, st>>>(loc, attn, gout, S, M, P, g, gvalue);
  }
  return 1;
}

}  // namespace ocpg_col

#ifdef EXP_STAMPS
extern "C" int ocpg_debug_stamps(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ocpg_col::g_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(ocpg_col::g_stamps), z, sizeof(z)) != hipSuccess) return -2;
  }
  return 0;
}
#endif
