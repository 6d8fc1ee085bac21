// grad_value of the MSDeformAttn backward for self-attention over the value's own pixels (the deformable ENCODER, Lq == S) on
// MI355X (gfx950), OUTPUT-TILED.  Semantics = models/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159 (ms_deform_attn_col2im_bilinear:
// every sample adds attn * bilinear weight * grad_out[query, head, :] to its four corner pixels of grad_value) as driven by
// :301-403; the reference does it with one global atomicAdd per (corner, channel).
//
// Why a new shape (round 3).  The round-2 column kernel (msda_col.hip) owns the QUERIES of a pyramid column and flushes a
// window around it with global float atomics: neighbouring columns' windows overlap, 125 MB of atomics for a 52 MB tensor,
// and gfx950 executes float atomics at the memory side at ~1.3 TB/s (MI355X_MICROARCH.md).  Here the DESTINATION is owned:
//   * k_gv_tile   -- levels 0..LA-1 (the fine ones, 94 % of the bytes): one workgroup per (frame, tile, head) owns the
//     tile's pixels; it walks the CANDIDATE queries (those within kMargin pixels of the tile at that level, every query level),
//     bins their corner contributions by pixel with integer LDS atomics (counting sort, as round 2 found: ds_add_f32 is
//     lane-serial on gfx950), sums every pixel in REGISTERS over all rounds and leaves with ONE plain 16-byte store per lane:
//     zero global atomics, every byte of those levels written exactly once.
//   * k_gv_coarse -- levels LA..L-1 (a few hundred pixels): one workgroup per (frame, query chunk, head); the chunk's queries
//     are interleaved over the map so every pixel-owning lane group gets work; all pixels of those levels live in register
//     accumulators across the rounds and are flushed once with global atomics (K chunks x 38 KB per (frame, head)).  The same
//     kernel sends the FAR corners of the fine levels (a sample further than kMargin from its query's own position: none at
//     the model's initial offsets, a few after training) straight to memory with atomics -- it runs AFTER k_gv_tile's stores.
// Both kernels decide "candidate / far" from the same host-built integer tables (msda_tile.h), so the split is exact for any
// sampling locations.  A round = 256 candidate queries x the 4 points of one level: 4 barriers per round with 16 corner
// contributions per thread between them (round 2: 5 barriers per level with one sample per thread).
#include "msda_tile.h"
#include "msda_col.h"       // path-selection state (SelState slots, sel_report / sel_decide)

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "msda_dev.h"

namespace ocpg_tile {
namespace {

using ocpg_dev::ld4;

constexpr int NT = 256, G = 8, D = 32, NGRP = NT / G;     // 32 lane groups of 8 lanes: a group owns pixels, a lane 4 channels
constexpr int NPA = 5;                                    // pixels per lane group in the tile kernel (160 footprint pixels)

__host__ __device__ __forceinline__ unsigned magic_of(unsigned d) { return d <= 1 ? 0xffffffffu : (unsigned)(0xffffffffu / d); }
__device__ __forceinline__ unsigned udiv(unsigned n, unsigned d, unsigned m) {     // n / d for n < 2^31, m = magic_of(d)
  unsigned q = __umulhi(n, m);
  if (n - q * d >= d) ++q;
  return q;
}

// barrier for LDS hand-offs only (does not drain this wave's global stores / atomics)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef EXP_STAMPS
// Diagnostic build only (never shipped): per-phase cycle sums of wave 0 of every workgroup ([0..7] tile kernel, [8..15] coarse),
// kept in registers and added to memory ONCE per workgroup (a global atomic per stamp perturbs what it measures).
__device__ unsigned long long g_tstamps[16];
#define STAMP_INIT unsigned tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tprev = __builtin_amdgcn_s_memtime()
#define STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc_[(k) & 7] += (unsigned)(t_ - tprev); tprev = t_; } while (0)
#define STAMP_FLUSH(base) do { if (tid == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_tstamps[(base) + i_], (unsigned long long)tacc_[i_]); } } while (0)
#else
#define STAMP(k) do { } while (0)
#define STAMP_INIT do { } while (0)
#define STAMP_FLUSH(base) do { } while (0)
#endif

struct __attribute__((aligned(8))) Item {
  float w;   // bilinear weight * attention weight
  int q;     // float offset of the query's staged grad_out row
};

// The four points of one (query, head, level): locations and attention weights, as they sit in memory (32 + 16 contiguous bytes)
struct Pts {
  float x[4], y[4], a[4];
};
__device__ __forceinline__ Pts load_pts(const float* __restrict__ loc, const float* __restrict__ attn, long long row, int L, int l) {
  const long long wi = (row * L + l) * 4;
  const float4 A = ld4(loc + wi * 2), B = ld4(loc + wi * 2 + 4), W = ld4(attn + wi);
  Pts p;
  p.x[0] = A.x; p.y[0] = A.y; p.x[1] = A.z; p.y[1] = A.w; p.x[2] = B.x; p.y[2] = B.y; p.x[3] = B.z; p.y[3] = B.w;
  p.a[0] = W.x; p.a[1] = W.y; p.a[2] = W.z; p.a[3] = W.w;
  return p;
}

// exclusive scan of cnt[0 .. 64 * BPL) by ONE wave (BPL consecutive bins per lane) -> start[]
template <int BPL>
__device__ __forceinline__ void wave_scan_bins(const int* cnt, int* start, int lane) {
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
}

// sum of one pixel's list into the lane's 4 channels
__device__ __forceinline__ void sum_list(const Item* __restrict__ lst, int n, const float* __restrict__ gs, int j, float4& acc) {
  int t = 0;
  for (; t + 4 <= n; t += 4) {
    const Item i0 = lst[t], i1 = lst[t + 1], i2 = lst[t + 2], i3 = lst[t + 3];
    const float4 g0 = ld4(gs + i0.q + 4 * j), g1 = ld4(gs + i1.q + 4 * j), g2 = ld4(gs + i2.q + 4 * j), g3 = ld4(gs + i3.q + 4 * j);
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
}

// ======================================================================================================================
// A round of either kernel = NB entries (an entry = one query at one destination level: its 4 points), TWO threads per entry
// (2 points = 8 corner contributions each), so a round bins at most 16 * NB items and stages NB grad_out rows: 32 KB of LDS per
// workgroup for NB = 128 -> 4 workgroups (16 waves) per CU; every phase is bound by LDS / memory LATENCY, not throughput (stamps,
// tools/stamps_tile.py), so occupancy and batched independent LDS operations are what pay.
constexpr int NB = NT / 2;                                // entries per round
constexpr int kItems = NB * 16;                           // corner contributions of one round at most
constexpr int kListMax = 2048;                            // compacted (query, level) entries of a tile at most (= its candidates)

// this thread's two points of one (query, head, level): 16 + 8 contiguous bytes
struct Pts2 {
  float x[2], y[2], a[2];
};
__device__ __forceinline__ Pts2 load_pts2(const float* __restrict__ loc, const float* __restrict__ attn, long long row, int L, int l, int half) {
  const long long wi = (row * L + l) * 4 + 2 * half;
  const float4 A = ld4(loc + wi * 2);
  const float2 W = *reinterpret_cast<const float2*>(attn + wi);
  Pts2 p;
  p.x[0] = A.x; p.y[0] = A.y; p.x[1] = A.z; p.y[1] = A.w;
  p.a[0] = W.x; p.a[1] = W.y;
  return p;
}

// Corner contributions of two points inside the pixel box [ry0, ry1) x [cx0, cx1) of an H x W map (the box lies inside the map):
// bin = pixbase + (y - ry0) * cw + (x - cx0), or -1.  All eight LDS atomics are issued back to back (their results are first
// used after the last one is in flight: one LDS latency per thread, not eight).
__device__ __forceinline__ void bin_points(const Pts2& pt, bool active, int H, int W, int ry0, int ry1, int cx0, int cx1, int cw, int pixbase,
                                           int* cnt, int (&key)[8], float (&wv)[8]) {
  int pid[8];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float h_im = pt.y[p] * (float)H - 0.5f, w_im = pt.x[p] * (float)W - 0.5f;
    const bool ok = active && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;       // cuh:268 / cuh:332
    const int y0 = (int)floorf(h_im), x0 = (int)floorf(w_im);
    const float ly = h_im - (float)y0, lx = w_im - (float)x0, hy = 1.f - ly, hx = 1.f - lx, a = pt.a[p];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
      const bool in = ok && yy >= ry0 && yy < ry1 && xx >= cx0 && xx < cx1;
      pid[p * 4 + k] = in ? pixbase + (yy - ry0) * cw + (xx - cx0) : -1;
      wv[p * 4 + k] = ((k >> 1) ? ly : hy) * ((k & 1) ? lx : hx) * a;
    }
  }
  int slot[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) slot[i] = pid[i] >= 0 ? atomicAdd(&cnt[pid[i]], 1) : 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) key[i] = pid[i] >= 0 ? (pid[i] << 16) | slot[i] : -1;
}

// items of this thread into their lists: all list starts are read first (independent LDS reads), then written
__device__ __forceinline__ void drop_items(const int (&key)[8], const float (&wv)[8], int goff, const int* start, Item* items) {
  int st[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) st[i] = start[max(key[i], 0) >> 16];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (key[i] >= 0) {
      Item it;
      it.w = wv[i];
      it.q = goff;
      items[st[i] + (key[i] & 0xffff)] = it;
    }
}

// the NP pixels of a lane group: counts and starts of all lists first, then the sums
template <int NP>
__device__ __forceinline__ void sum_pixels(int grp, int j, int nb, int* cnt, const int* start, const Item* items, const float* gs,
                                           float4 (&acc)[NP]) {
  int n[NP], st[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int pid = min(grp + NGRP * k, nb - 1);
    n[k] = grp + NGRP * k < nb ? cnt[pid] : 0;
    st[k] = start[pid];
  }
#pragma unroll
  for (int k = 0; k < NP; ++k)
    if (n[k]) {
      sum_list(items + st[k], n[k], gs, j, acc[k]);
      if (j == 0) cnt[grp + NGRP * k] = 0;          // ready for the next round
    }
}

// ======================================================================================================================
// Tile kernel: levels l < LA, plain stores.
struct TileSh {
  int cnt[192], start[192], nlist, pad_[3];
  int qya[kLA][kLM], qxa[kLA][kLM], nc[kLA][kLM], cbase[kLA][kLM + 1];
  unsigned m_nc[kLA][kLM];
  int ry0[kLA], cx0[kLA], rh[kLA], cw[kLA], pixbase[kLA + 1];
  unsigned m_cw[kLA];
  unsigned short list[kListMax];          // (query | level << 15) of every candidate with at least one corner inside the footprint
};

__global__ __launch_bounds__(NT, 4) void k_gv_tile(const float* __restrict__ loc, const float* __restrict__ attn,
                                                const float* __restrict__ gout, int S, int M, TileTab tab, float* __restrict__ gvalue,
                                                const int* __restrict__ sel) {
  if (sel != nullptr && sel[ocpg_col::kSelCur] != 1) return;      // path selection: the column scatter serves this call
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Item* items = reinterpret_cast<Item*>(smem);
  float* gs = reinterpret_cast<float*>(smem + sizeof(Item) * kItems);
  TileSh& sh = *reinterpret_cast<TileSh*>(smem + sizeof(Item) * kItems + sizeof(float) * NB * D);
  const int tid = threadIdx.x, lane = tid & 63, grp = tid >> 3, j = tid & 7;
  const int bid = blockIdx.x;
  const int bt = (int)udiv(bid, M, tab.m_M);
  const int m = bid - bt * M;            // head fastest: blocks are dealt round-robin over the 8 XCDs -> one head per XCD L2
  const int b = (int)udiv(bt, tab.ntiles, tab.m_ntiles), tile = bt - b * tab.ntiles;
  const int ty = (int)udiv(tile, tab.ntx, tab.m_ntx), tx = tile - ty * tab.ntx;
  const int L = tab.L, LA = tab.LA;
  STAMP_INIT;
  if (tid < 192) sh.cnt[tid] = 0;
  if (tid == 192) sh.nlist = 0;
  if (tid < LA * kLM) {
    const int l = tid / kLM, lq = tid % kLM;
    int a0 = 0, n_r = 0, c0 = 0, n_c = 0;
    if (lq < L) {
      a0 = tab.candY[l][lq][ty][0]; n_r = tab.candY[l][lq][ty][1] - a0;
      c0 = tab.candX[l][lq][tx][0]; n_c = tab.candX[l][lq][tx][1] - c0;
    }
    sh.qya[l][lq] = a0; sh.qxa[l][lq] = c0;
    sh.nc[l][lq] = n_c;
    sh.m_nc[l][lq] = magic_of((unsigned)n_c);
    sh.cbase[l][lq + 1] = max(n_r, 0) * max(n_c, 0);      // counts; prefix below
  }
  if (tid >= 64 && tid < 64 + LA) {
    const int l = tid - 64;
    sh.ry0[l] = tab.ry0[l][ty]; sh.rh[l] = tab.ry0[l][ty + 1] - tab.ry0[l][ty];
    sh.cx0[l] = tab.cx0[l][tx]; sh.cw[l] = tab.cx0[l][tx + 1] - tab.cx0[l][tx];
    sh.m_cw[l] = magic_of((unsigned)sh.cw[l]);
  }
  __syncthreads();
  if (tid < LA) {
    int s = 0;
    sh.cbase[tid][0] = 0;
    for (int lq = 0; lq < kLM; ++lq) { s += sh.cbase[tid][lq + 1]; sh.cbase[tid][lq + 1] = s; }
  }
  if (tid == 64) {
    int s = 0;
    for (int l = 0; l < LA; ++l) { sh.pixbase[l] = s; s += sh.rh[l] * sh.cw[l]; }
    sh.pixbase[LA] = s;
  }
  __syncthreads();
  const int nb = sh.pixbase[LA];
  const long long rowb = (long long)b * S;
  STAMP(0);
  // ---- prefilter: which candidates have a corner inside the footprint?  (no atomics, no barrier: position arithmetic only) ----
  for (int l = 0; l < LA; ++l) {
    const int H = tab.H[l], W = tab.W[l];
    const int ry0 = sh.ry0[l], ry1 = ry0 + sh.rh[l], cx0 = sh.cx0[l], cx1 = cx0 + sh.cw[l];
    const int ncand = sh.cbase[l][kLM];                      // (the host checked: a tile's candidates of all levels fit the list)
    if (sh.rh[l] * sh.cw[l] == 0) continue;                  // (uniform) the tile owns no pixel of this level
    for (int c0 = 0; c0 < ncand; c0 += 4 * NT) {
      int qv[4];
      float4 xy[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {                           // four candidates per pass: all eight loads in flight together
        const int c = min(c0 + u * NT + tid, ncand - 1);
        int lq = 0;
        while (lq + 1 < kLM && c >= sh.cbase[l][lq + 1]) ++lq;
        const int r = c - sh.cbase[l][lq];
        const int dy = (int)udiv(r, sh.nc[l][lq], sh.m_nc[l][lq]);
        qv[u] = tab.S0[lq] + (sh.qya[l][lq] + dy) * tab.W[lq] + sh.qxa[l][lq] + r - dy * sh.nc[l][lq];
        const float* lp = loc + ((((rowb + qv[u]) * M + m) * L + l) * 4) * 2;
        xy[u][0] = ld4(lp);
        xy[u][1] = ld4(lp + 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float px[4] = {xy[u][0].x, xy[u][0].z, xy[u][1].x, xy[u][1].z}, py[4] = {xy[u][0].y, xy[u][0].w, xy[u][1].y, xy[u][1].w};
        bool hit = false;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const float h_im = py[p] * (float)H - 0.5f, w_im = px[p] * (float)W - 0.5f;
          const bool ok = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
          const int y0 = (int)floorf(h_im), x0 = (int)floorf(w_im);
          hit |= ok && y0 + 1 >= ry0 && y0 < ry1 && x0 + 1 >= cx0 && x0 < cx1;       // one of the 2 rows and one of the 2 columns inside
        }
        hit = hit && c0 + u * NT + tid < ncand;
        const unsigned long long hb = __ballot(hit);
        int base = 0;
        if (lane == 0 && hb) base = atomicAdd(&sh.nlist, __popcll(hb));
        base = __builtin_amdgcn_readfirstlane(base);
        if (hit) sh.list[base + __popcll(hb & ((1ull << lane) - 1ull))] = (unsigned short)(qv[u] | (l << 15));
      }
    }
  }
  __syncthreads();
  const int nent = sh.nlist;
  STAMP(1);
  float4 acc[NPA];
#pragma unroll
  for (int k = 0; k < NPA; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int half = tid & 1;
  for (int e0 = 0; e0 < nent; e0 += NB) {
    // ---- (1) this thread's entry (2 of its 4 points); the grad_out rows of the round's entries ---------------------------------
    const bool active = e0 + (tid >> 1) < nent;
    const int e = sh.list[min(e0 + (tid >> 1), nent - 1)];
    const int q = e & 0x7fff, l = e >> 15;
    const Pts2 pt = load_pts2(loc, attn, (rowb + q) * M + m, L, l, half);
    float4 gq[NB / NGRP];
#pragma unroll
    for (int u = 0; u < NB / NGRP; ++u) {
      const int qr = sh.list[min(e0 + grp + u * NGRP, nent - 1)] & 0x7fff;
      gq[u] = ld4(gout + ((rowb + qr) * M + m) * D + 4 * j);
    }
    int key[8];
    float wv[8];
    bin_points(pt, active, tab.H[l], tab.W[l], sh.ry0[l], sh.ry0[l] + sh.rh[l], sh.cx0[l], sh.cx0[l] + sh.cw[l], sh.cw[l], sh.pixbase[l],
               sh.cnt, key, wv);
#pragma unroll
    for (int u = 0; u < NB / NGRP; ++u) *reinterpret_cast<float4*>(gs + (grp + u * NGRP) * D + 4 * j) = gq[u];
    lds_barrier();
    STAMP(2);
    if (tid < 64) wave_scan_bins<3>(sh.cnt, sh.start, lane);
    lds_barrier();
    STAMP(3);
    drop_items(key, wv, (tid >> 1) * D, sh.start, items);
    lds_barrier();
    STAMP(4);
    sum_pixels<NPA>(grp, j, nb, sh.cnt, sh.start, items, gs, acc);
    STAMP(5);
    lds_barrier();
    STAMP(6);
  }
  // ---- every pixel of the footprint is written exactly once ---------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < NPA; ++k) {
    const int pid = grp + NGRP * k;
    if (pid < nb) {
      int l = 0;
      while (l + 1 < LA && pid >= sh.pixbase[l + 1]) ++l;
      const int r = pid - sh.pixbase[l];
      const int dy = (int)udiv(r, sh.cw[l], sh.m_cw[l]);
      const int gp = tab.S0[l] + (sh.ry0[l] + dy) * tab.W[l] + sh.cx0[l] + r - dy * sh.cw[l];
      *reinterpret_cast<float4*>(gvalue + ((rowb + gp) * M + m) * D + 4 * j) = acc[k];
    }
  }
  STAMP(7);
  STAMP_FLUSH(0);
}

// ======================================================================================================================
// Coarse kernel: levels l >= LA in register accumulators (global atomics at the end) + the far corners of the fine levels.
// Lane j of a group holds channels j, j+8, j+16, j+24 (the flush then writes 32 contiguous bytes per group and instruction).
template <int NPX>
struct CoarseSh {
  int cnt[NGRP * NPX], start[NGRP * NPX];
};

template <int NPX>
__global__ __launch_bounds__(NT, NPX <= 10 ? 3 : 1) void k_gv_coarse(const float* __restrict__ loc, const float* __restrict__ attn,
                                                  const float* __restrict__ gout, int S, int M, TileTab tab, float* __restrict__ gvalue,
                                                  int* __restrict__ sel, int to_col_pct) {
  // path selection: runs only when the call site's state says so; it then also proposes the next call's path (stay while the share of
  // far samples stays above to_col_pct)
  if (sel != nullptr && sel[ocpg_col::kSelCur] != 1) return;
  int far_seen = 0, fine_seen = 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Item* items = reinterpret_cast<Item*>(smem);
  float* gs = reinterpret_cast<float*>(smem + sizeof(Item) * kItems);
  CoarseSh<NPX>& sh = *reinterpret_cast<CoarseSh<NPX>*>(smem + sizeof(Item) * kItems + sizeof(float) * NB * D);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = tid >> 3, j = tid & 7;
  const int bid = blockIdx.x;
  const int bt = (int)udiv(bid, M, tab.m_M);
  const int m = bid - bt * M;
  const int K = tab.K;
  const int b = (int)udiv(bt, K, tab.m_K), kq = bt - b * K;
  const int L = tab.L, LA = tab.LA, MD = M * D;
  const int nq = (S - kq + K - 1) / K;                 // this chunk's queries: kq, kq + K, kq + 2K, ...
  const int nb = tab.nbB;
  for (int i = tid; i < NGRP * NPX; i += NT) sh.cnt[i] = 0;
  float4 acc[NPX];
#pragma unroll
  for (int k = 0; k < NPX; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long rowb = (long long)b * S;
  float* gvb = gvalue + rowb * MD + m * D;
  const int half = tid & 1, ql = tid >> 1;             // two threads per query: points {0,1} and {2,3}
  STAMP_INIT;
  __syncthreads();

  const int nbatch = (nq + NB - 1) / NB;
  for (int bi = 0; bi < nbatch; ++bi) {
    // batch bi takes the chunk's queries bi, bi + nbatch, bi + 2 nbatch, ...: every batch spans the whole map, so the lanes of a
    // wave hit different coarse pixels (the LDS counters of a 6x10 level serialise when neighbours share a pixel)
    const bool active = bi + nbatch * ql < nq;
    const int q = kq + K * min(bi + nbatch * ql, nq - 1);
    const long long row = (rowb + q) * M + m;
    // ---- the batch's global loads: the points of the fine levels (far test), the grad_out rows (channel-interleaved per lane) ----
    Pts2 pts[kLM];
#pragma unroll
    for (int l = 0; l < kLA; ++l) pts[l] = load_pts2(loc, attn, row, L, min(l, L - 1), half);
    {
      float4 gq[NB / NGRP];
#pragma unroll
      for (int u = 0; u < NB / NGRP; ++u) {
        const int qi = min(bi + nbatch * (grp + u * NGRP), nq - 1);
        const float* g = gout + ((rowb + kq + (long long)K * qi) * M + m) * D + j;
        gq[u] = make_float4(g[0], g[G], g[2 * G], g[3 * G]);
      }
#pragma unroll
      for (int u = 0; u < NB / NGRP; ++u) *reinterpret_cast<float4*>(gs + (grp + u * NGRP) * D + 4 * j) = gq[u];
    }
    lds_barrier();
    STAMP(8);
    // ---- far corners of the fine levels: straight to memory (after k_gv_tile's stores: stream order) -------------------------
    if (LA > 0) {
      int lq = 0;
      while (lq + 1 < L && q >= tab.S0[lq + 1]) ++lq;
      const int rq = q - tab.S0[lq];
      const int qy = (int)udiv(rq, tab.W[lq], tab.m_W[lq]), qx = rq - qy * tab.W[lq];
#pragma unroll
      for (int l = 0; l < kLA; ++l) {
        if (l < LA) {
          const int H = tab.H[l], W = tab.W[l];
          const int ya = tab.nearY[l][tab.ybase[lq] + qy][0], yb = tab.nearY[l][tab.ybase[lq] + qy][1];
          const int xa = tab.nearX[l][tab.xbase[lq] + qx][0], xb = tab.nearX[l][tab.xbase[lq] + qx][1];
          float* gvl = gvb + (long long)tab.S0[l] * MD;
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const float h_im = pts[l].y[p] * (float)H - 0.5f, w_im = pts[l].x[p] * (float)W - 0.5f;
            const bool ok = active && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
            const int y0 = (int)floorf(h_im), x0 = (int)floorf(w_im);
            fine_seen += ok ? 1 : 0;
            int ovm = 0;
            if (ok) {
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
                const bool inmap = yy >= 0 && yy <= H - 1 && xx >= 0 && xx <= W - 1;
                const bool near = yy >= ya && yy < yb && xx >= xa && xx < xb;
                if (inmap && !near) ovm |= 1 << k;
              }
            }
            far_seen += ovm != 0 ? 1 : 0;
            unsigned long long bal = __ballot(ovm != 0);
            if (bal) {             // (wave-uniform) rare: one far sample per wave step, lanes 0..31 = its 32 channels
              const float ly = h_im - (float)y0, lx = w_im - (float)x0, hy = 1.f - ly, hx = 1.f - lx, a = pts[l].a[p];
              const int pix0 = y0 * W + x0;
              while (bal) {
                const int src = __ffsll((long long)bal) - 1;
                bal &= bal - 1;
                const int om = __builtin_amdgcn_readlane(ovm, src), gp = __builtin_amdgcn_readlane(pix0, src);
                const float w00 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hy * hx * a), src));
                const float w01 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hy * lx * a), src));
                const float w10 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ly * hx * a), src));
                const float w11 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ly * lx * a), src));
                if (lane < D) {
                  const float g = gs[(wave * 32 + (src >> 1)) * D + 4 * (lane % G) + lane / G];
                  if (om & 1) atomicAdd(gvl + (long long)gp * MD + lane, w00 * g);
                  if (om & 2) atomicAdd(gvl + (long long)(gp + 1) * MD + lane, w01 * g);
                  if (om & 4) atomicAdd(gvl + (long long)(gp + W) * MD + lane, w10 * g);
                  if (om & 8) atomicAdd(gvl + (long long)(gp + W + 1) * MD + lane, w11 * g);
                }
              }
            }
          }
        }
      }
    }
    STAMP(9);
    // ---- the coarse levels: bin, sum in registers (their points are loaded together, after the fine ones are dead) ----------------
#pragma unroll
    for (int l = kLA; l < kLM; ++l) pts[l] = load_pts2(loc, attn, row, L, min(l, L - 1), half);
    if (LA < kLA) {
#pragma unroll
      for (int l = 0; l < kLA; ++l) pts[l] = load_pts2(loc, attn, row, L, min(l, L - 1), half);     // (L < 2: a coarse level sits below kLA)
    }
#pragma unroll
    for (int l = 0; l < kLM; ++l) {
      if (l >= LA && l < L) {
        const int H = tab.H[l], W = tab.W[l];
        int key[8];
        float wv[8];
        bin_points(pts[l], active, H, W, 0, H, 0, W, W, tab.offB[l], sh.cnt, key, wv);
        lds_barrier();
        STAMP(10);
        if (tid < 64) wave_scan_bins<NPX / 2>(sh.cnt, sh.start, lane);
        lds_barrier();
        STAMP(11);
        drop_items(key, wv, ql * D, sh.start, items);
        lds_barrier();
        STAMP(12);
        sum_pixels<NPX>(grp, j, nb, sh.cnt, sh.start, items, gs, acc);
        STAMP(13);
        lds_barrier();
        STAMP(14);
      }
    }
    if (LA == L) lds_barrier();      // (uniform) no coarse round followed: the far pass's reads of gs must finish before the next batch stages
  }
  // ---- flush: one global atomic per (pixel, channel) of the coarse levels per workgroup.  The sums go through LDS so that a
  // wave-instruction adds two FULL 128-byte rows (float atomics run at the memory side in 64-byte requests: half-filled
  // requests halve the rate, MI355X_MICROARCH.md "Global float atomics")
  {
    float* T = reinterpret_cast<float*>(smem);                 // [NGRP * HP][D] floats (20 KB) <= items + gs
    constexpr int HP = 5;
    static_assert(NPX % HP == 0, "the flush walks the pixel slots HP at a time");
#pragma unroll
    for (int h = 0; h < NPX / HP; ++h) {
      lds_barrier();
#pragma unroll
      for (int kk = 0; kk < HP; ++kk) {
        const float4 v = acc[h * HP + kk];
        float* t = T + (kk * NGRP + grp) * D + j;
        t[0] = v.x; t[G] = v.y; t[2 * G] = v.z; t[3 * G] = v.w;
      }
      lds_barrier();
      for (int r = wave * 2 + (lane >> 5); r < HP * NGRP; r += 2 * (NT / 64)) {
        const int pid = h * HP * NGRP + r;
        if (pid < nb) {
          int l = LA;
          while (l + 1 < L && pid >= tab.offB[l + 1]) ++l;
          const float v = T[r * D + (lane & 31)];
          if (v != 0.f) atomicAdd(gvb + (long long)(tab.S0[l] + pid - tab.offB[l]) * MD + (lane & 31), v);
        }
      }
    }
  }
  if (sel != nullptr && (blockIdx.x & 7) == 0) {       // every 8th workgroup's (far, seen) fine-level samples -> the call site's state; the last of them proposes
    __shared__ int sel_far, sel_seen;
    if (tid == 0) { sel_far = 0; sel_seen = 0; }
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) { far_seen += __shfl_xor(far_seen, o, 64); fine_seen += __shfl_xor(fine_seen, o, 64); }
    if (lane == 0) { atomicAdd(&sel_far, far_seen); atomicAdd(&sel_seen, fine_seen); }
    __syncthreads();
    if (tid == 0) ocpg_col::sel_report(sel, sel_far, sel_seen);
  }
  STAMP(15);
  STAMP_FLUSH(8);
}

template <typename Kn>
inline void allow_lds(Kn kernel, size_t bytes) {
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

bool make_tile_tab(const int64_t* sh, int N, int L, int S, int M, int P, int Dm, TileTab& t) {
  if (!sh || L < 1 || L > kLM || M < 1 || P != 4 || Dm != D || N < 1) return false;
  std::memset(&t, 0, sizeof(t));
  long long tot = 0;
  int sumH = 0, sumW = 0;
  for (int l = 0; l < L; ++l) {
    const long long H = sh[2 * l], W = sh[2 * l + 1];
    if (H <= 0 || W <= 0 || H > kMaxH0 || W > kMaxW0) return false;
    if (l > 0 && (H > t.H[0] || W > t.W[0])) return false;           // level 0 is the finest
    t.H[l] = (int)H; t.W[l] = (int)W; t.S0[l] = (int)tot;
    t.ybase[l] = sumH; t.xbase[l] = sumW;
    t.m_W[l] = magic_of((unsigned)W);
    sumH += (int)H; sumW += (int)W;
    tot += H * W;
  }
  if (tot != S || sumH > kSumH || sumW > kSumW || (long long)N * S * M * D >= (1LL << 31)) return false;
  t.L = L;
  t.LA = std::min(L, kLA);
  t.nty = (t.H[0] + 7) / 8;
  t.ntx = (t.W[0] + 15) / 16;
  if (t.nty > kMaxT || t.ntx > kMaxT) return false;
  t.ntiles = t.nty * t.ntx;
  t.m_M = magic_of((unsigned)M);
  t.m_ntiles = magic_of((unsigned)t.ntiles);
  t.m_ntx = magic_of((unsigned)t.ntx);
  // footprints (bands) and the band of every pixel row / column
  uint8_t rowtile[kLA][kMaxH0], coltile[kLA][kMaxW0];
  t.nbA = 0;
  for (int l = 0; l < t.LA; ++l) {
    int rmax = 0, cmax = 0;
    for (int ty = 0; ty <= t.nty; ++ty) t.ry0[l][ty] = (uint8_t)((long long)ty * t.H[l] / t.nty);
    for (int tx = 0; tx <= t.ntx; ++tx) t.cx0[l][tx] = (uint8_t)((long long)tx * t.W[l] / t.ntx);
    for (int ty = 0; ty < t.nty; ++ty) {
      rmax = std::max(rmax, t.ry0[l][ty + 1] - t.ry0[l][ty]);
      for (int y = t.ry0[l][ty]; y < t.ry0[l][ty + 1]; ++y) rowtile[l][y] = (uint8_t)ty;
    }
    for (int tx = 0; tx < t.ntx; ++tx) {
      cmax = std::max(cmax, t.cx0[l][tx + 1] - t.cx0[l][tx]);
      for (int x = t.cx0[l][tx]; x < t.cx0[l][tx + 1]; ++x) coltile[l][x] = (uint8_t)tx;
    }
    t.nbA += rmax * cmax;
  }
  if (t.nbA > NGRP * NPA) return false;
  // candidate / near tables: along each axis separately
  for (int l = 0; l < t.LA; ++l)
    for (int lq = 0; lq < L; ++lq)
      for (int axis = 0; axis < 2; ++axis) {
        const int nl = axis ? t.W[l] : t.H[l], nqd = axis ? t.W[lq] : t.H[lq], nt = axis ? t.ntx : t.nty;
        const uint8_t* band0 = axis ? t.cx0[l] : t.ry0[l];
        const uint8_t* pix2band = axis ? coltile[l] : rowtile[l];
        uint8_t(*cand)[2] = axis ? t.candX[l][lq] : t.candY[l][lq];
        uint8_t(*near)[2] = axis ? &t.nearX[l][t.xbase[lq]] : &t.nearY[l][t.ybase[lq]];
        int lo[kMaxT], hi[kMaxT];
        for (int b = 0; b < nt; ++b) { lo[b] = 1 << 30; hi[b] = 0; }
        for (int qc = 0; qc < nqd; ++qc) {
          const int c = (int)(((long long)(2 * qc + 1) * nl) / (2 * nqd));       // the query's own position, as a pixel of level l
          const int p0 = std::max(0, c - kMargin), p1 = std::min(nl - 1, c + kMargin);
          const int b0 = pix2band[p0], b1 = pix2band[p1];
          near[qc][0] = band0[b0];
          near[qc][1] = band0[b1 + 1];
          for (int b = b0; b <= b1; ++b) { lo[b] = std::min(lo[b], qc); hi[b] = std::max(hi[b], qc + 1); }
        }
        for (int b = 0; b < nt; ++b) {
          cand[b][0] = (uint8_t)(hi[b] > 0 ? lo[b] : 0);
          cand[b][1] = (uint8_t)(hi[b] > 0 ? hi[b] : 0);
        }
      }
  // coarse levels: bins, query chunks
  int off = 0;
  for (int l = 0; l <= L; ++l) {
    t.offB[l] = off;
    if (l >= t.LA && l < L) off += t.H[l] * t.W[l];
  }
  t.nbB = off;
  if (t.nbB > NGRP * 20) return false;
  // every tile's candidate list (all levels) must fit the tile kernel's LDS list; queries are stored in 15 bits
  if (S > 0x7fff) return false;
  for (int ty = 0; ty < t.nty; ++ty)
    for (int tx = 0; tx < t.ntx; ++tx) {
      int tot_c = 0;
      for (int l = 0; l < t.LA; ++l)
        for (int lq = 0; lq < L; ++lq)
          tot_c += (t.candY[l][lq][ty][1] - t.candY[l][lq][ty][0]) * (t.candX[l][lq][tx][1] - t.candX[l][lq][tx][0]);
      if (tot_c > kListMax) return false;
    }
  // query chunks of the coarse kernel: ~1024 workgroups, each a whole number of NB-query batches (r of them)
  int r = std::max(1, (int)(((long long)S * N * M + (long long)NB * 512) / ((long long)NB * 1024)));
  if (const char* e = std::getenv("OCPG_MSDA_TILE_BATCHES")) r = std::max(1, std::atoi(e));       // experiment switch
  const int K = std::max(1, (S + NB * r - 1) / (NB * r));
  t.K = K;
  t.m_K = magic_of((unsigned)K);
  return true;
}

bool tile_supported(const int64_t* shapes_host, int N, int L, int S, int M, int P, int Dm) {
  TileTab tab;
  return make_tile_tab(shapes_host, N, L, S, M, P, Dm, tab);
}

int bwd_value_tile(const float* loc, const float* attn, const float* gout, const int64_t* shapes_host, int N, int S, int M, int Dm, int L,
                   int P, float* gvalue, hipStream_t st, int* sel, int to_col_pct) {
  static_assert(sizeof(TileTab) <= 3600, "TileTab travels as a kernel argument (4 KB limit with the other arguments)");
  TileTab tab;
  if (!make_tile_tab(shapes_host, N, L, S, M, P, Dm, tab)) return 0;
  if ((reinterpret_cast<uintptr_t>(loc) | reinterpret_cast<uintptr_t>(attn) | reinterpret_cast<uintptr_t>(gout) |
       reinterpret_cast<uintptr_t>(gvalue)) & 15) return 0;
  const size_t ldsA = sizeof(Item) * kItems + sizeof(float) * NB * D + sizeof(TileSh);
  allow_lds(k_gv_tile, ldsA);
  k_gv_tile<<<(unsigned)((long long)N * tab.ntiles * M), NT, ldsA, st>>>(loc, attn, gout, S, M, tab, gvalue, sel);
  const unsigned gridB = (unsigned)((long long)N * tab.K * M);
  if (tab.nbB <= NGRP * 10) {
    const size_t ldsB = sizeof(Item) * kItems + sizeof(float) * NB * D + sizeof(CoarseSh<10>);
    allow_lds(k_gv_coarse<10>, ldsB);
    k_gv_coarse<10><<<gridB, NT, ldsB, st>>>(loc, attn, gout, S, M, tab, gvalue, sel, to_col_pct);
  } else {
    const size_t ldsB = sizeof(Item) * kItems + sizeof(float) * NB * D + sizeof(CoarseSh<20>);
    allow_lds(k_gv_coarse<20>, ldsB);
    k_gv_coarse<20><<<gridB, NT, ldsB, st>>>(loc, attn, gout, S, M, tab, gvalue, sel, to_col_pct);
  }
  return 1;
}

}  // namespace ocpg_tile

#ifdef EXP_STAMPS
extern "C" int ocpg_debug_stamps_tile(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ocpg_tile::g_tstamps), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(ocpg_tile::g_tstamps), z, sizeof(z)) != hipSuccess) return -2;
  }
  return 0;
}
#endif
