// 3x3 / stride-1 / padding-1 convolution of channels-last bf16 maps on the matrix cores, HALO-STAGED: a workgroup owns a TH x TW pixel
// tile (<= 128 pixels) x 64 output channels; per 64-channel chunk the (TH + 2) x (TW + 2) input halo is staged in LDS ONCE and serves
// all nine taps (conv3x3_mfma.hip gathers the shifted tile from global memory for every tap: nine times the activation traffic and the
// address arithmetic of a gather in every K step); only the 64 x 64 weight tile of the next tap streams through registers while the
// current tap multiplies.  Per tap and wave: 12 fragment reads feed 8 MFMAs (32 pixels x 64 channels = two 32x32x16 tiles).
// Same operands and epilogue as conv3x3_mfma<DGRAD, BN>; `flip` turns it into the input gradient (tap t multiplies w[.][8 - t][.]).
#include "conv3x3_halo.h"

#include <stdint.h>

namespace ocpg_halo {
namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BN = 64, CK = 64, NT = 256;
constexpr int ROW = CK + 8;                      // LDS row: 64 channels + 16 B pad (144 B: conflict-free 16-byte fragment reads)
constexpr int MAXHALO = 180;                     // (TH + 2) * (TW + 2) at most
constexpr int XI = (MAXHALO * 8 + NT - 1) / NT;  // 16-byte halo segments per thread and chunk (6)

#ifdef EXP_STAMPS
// phase cycle sums per wave (s_memtime): [block][wave][phase]; phases: 0 fetch issue, 1 fragment reads + MFMA issue, 2 wait for the weight
// loads + park, 3 barrier, 4 whole kernel
__device__ unsigned long long g_conv_stamps[1024 * 4 * 8];
#define STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define STAMP(v)
#endif

struct Geo {
  int N, H, W, C, Cout, TH, TW, tiles_x, tiles_y, relu, flip;
};

__global__ __launch_bounds__(NT) void k_conv3x3_halo(const __hip_bfloat16* __restrict__ x, const __hip_bfloat16* __restrict__ w,
                                                     const float* __restrict__ scale, const float* __restrict__ bias, const Geo g,
                                                     __hip_bfloat16* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) short xl[MAXHALO * ROW];
  __shared__ __attribute__((aligned(16))) short wl[2][BN * ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % g.tiles_x;
  t /= g.tiles_x;
  const int ty = t % g.tiles_y, n = t / g.tiles_y;
  const int ty0 = ty * g.TH, tx0 = tx * g.TW, n0 = blockIdx.y * BN;
  const int HW2 = g.TW + 2, nhalo = (g.TH + 2) * HW2, npix = g.TH * g.TW;
  const int C = g.C;

  // ---- halo staging identity: segment it = tid + i * NT  ->  (halo pixel it / 8, 16-byte segment it % 8); element offset or -1
  long long xoff[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int it = tid + i * NT, hp = it >> 3;
    const int hy = hp / HW2, hx = hp - hy * HW2;
    const int yy = ty0 + hy - 1, xx = tx0 + hx - 1;
    xoff[i] = (hp < nhalo && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) ? (((long long)n * g.H + yy) * g.W + xx) * C + (it & 7) * 8 : -1;
  }
  // ---- weight staging identity: rows tid / 8 and tid / 8 + 32 of the 64-row tile, segment tid % 8
  const __hip_bfloat16* wp[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) wp[j] = w + (long long)(n0 + (tid >> 3) + 32 * j) * 9 * C + (tid & 7) * 8;

  // ---- fragment identity: A row = tile pixel wave * 32 + fr (pixels past the tile read pixel 0: never stored)
  int p = wave * 32 + fr;
  if (p >= npix) p = 0;
  const int prow = p / g.TW, pcol = p - prow * g.TW;
  const int abase = (prow * HW2 + pcol) * ROW + fh * 8;
  const int bbase = fr * ROW + fh * 8;

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  uint4 xr[XI], wr[2];
  const int nchunks = C / CK;
  auto fetch_x = [&](int c0) {
#pragma unroll
    for (int i = 0; i < XI; ++i)
      xr[i] = xoff[i] >= 0 ? *reinterpret_cast<const uint4*>(x + xoff[i] + c0) : make_uint4(0u, 0u, 0u, 0u);
  };
  auto park_x = [&]() {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int it = tid + i * NT;
      if (it < nhalo * 8) *reinterpret_cast<uint4*>(&xl[(it >> 3) * ROW + (it & 7) * 8]) = xr[i];
    }
  };
  auto fetch_w = [&](int c0, int tap) {
    const int tw = g.flip ? 8 - tap : tap;
    const long long off = (long long)tw * C + c0;
#pragma unroll
    for (int j = 0; j < 2; ++j) wr[j] = *reinterpret_cast<const uint4*>(wp[j] + off);
  };
  auto park_w = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) *reinterpret_cast<uint4*>(&wl[buf][((tid >> 3) + 32 * j) * ROW + (tid & 7) * 8]) = wr[j];
  };

#ifdef EXP_STAMPS
  unsigned long long ph[5] = {0, 0, 0, 0, 0};
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
  fetch_x(0);
  fetch_w(0, 0);
  park_x();
  park_w(0);
  __syncthreads();
  int buf = 0;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int c0 = ch * CK;
    const bool more = ch + 1 < nchunks;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      STAMP(s0);
      // the next step's weight tile (same chunk's next tap, or tap 0 of the next chunk; past the end: a repeated, unused load)
      if (tap < 8) fetch_w(c0, tap + 1);
      else fetch_w(more ? c0 + CK : c0, more ? 0 : 8);
      if (tap == 0 && more) fetch_x(c0 + CK);             // the next chunk's halo: eight taps to arrive
      STAMP(s1);
      const int dy = tap / 3, dx = tap - dy * 3;
      const short* ap = xl + abase + (dy * HW2 + dx) * ROW;
      const short* bp = wl[buf] + bbase;
#pragma unroll
      for (int kk = 0; kk < CK / 16; ++kk) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + kk * 16);
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(bp + kk * 16);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(bp + 32 * ROW + kk * 16);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc[1], 0, 0, 0);
      }
#ifdef EXP_STAMPS
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 0" ::: "memory");
#endif
      STAMP(s2);
      if (tap == 8 && more) {                              // the halo buffer turns over: everybody is done with this chunk's tile
        __syncthreads();
        park_x();
      }
      park_w(buf ^ 1);                                     // that buffer was last read in the previous step (barrier since)
#ifdef EXP_STAMPS
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
      STAMP(s3);
      __syncthreads();
      STAMP(s4);
#ifdef EXP_STAMPS
      ph[0] += s1 - s0, ph[1] += s2 - s1, ph[2] += s3 - s2, ph[3] += s4 - s3;
#endif
      buf ^= 1;
    }
  }

#ifdef EXP_STAMPS
  ph[4] = __builtin_amdgcn_s_memtime() - t_begin;
  if (lane == 0 && blockIdx.y == 0 && blockIdx.x < 1024)
    for (int q = 0; q < 5; ++q) g_conv_stamps[(blockIdx.x * 4 + wave) * 8 + q] = ph[q];
#endif
  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + j * 32 + (lane & 31);
    if (col >= g.Cout) continue;
    const float sv = scale ? scale[col] : 1.f, bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pp = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (pp >= npix) continue;
      const int pr = pp / g.TW, pc = pp - pr * g.TW;
      const int yy = ty0 + pr, xx = tx0 + pc;
      if (yy >= g.H || xx >= g.W) continue;
      float v = acc[j][r] * sv + bv;
      if (g.relu) v = fmaxf(v, 0.f);
      y[(((long long)n * g.H + yy) * g.W + xx) * g.Cout + col] = __float2bfloat16(v);
    }
  }
}

// tile shape: TH x TW <= 128 pixels with (TH + 2)(TW + 2) <= MAXHALO, least padded area over the map
void pick_tile(int H, int W, int& TH, int& TW) {
  static const int cand[][2] = {{8, 16}, {6, 20}, {4, 24}, {7, 18}, {5, 24}, {6, 16}, {8, 12}, {4, 16}, {8, 8}};
  long long best = -1;
  for (const auto& c : cand) {
    const int th = c[0], tw = c[1];
    if (th * tw > 128 || (th + 2) * (tw + 2) > MAXHALO) continue;
    const long long tiles = (long long)((H + th - 1) / th) * ((W + tw - 1) / tw);
    if (best < 0 || tiles * 128 < best) { best = tiles * 128; TH = th; TW = tw; }   // cost = workgroups (each pays a full 128-row MFMA tile)
  }
}

}  // namespace

bool conv3x3_halo(const __hip_bfloat16* x, const __hip_bfloat16* w, const float* scale, const float* bias, int relu, int flip, int N, int H, int W,
                  int C, int Cout, __hip_bfloat16* y, hipStream_t st) {
  if (C % CK != 0 || Cout % BN != 0 || N <= 0) return false;
  Geo g;
  g.N = N, g.H = H, g.W = W, g.C = C, g.Cout = Cout, g.relu = relu, g.flip = flip;
  g.TH = 8, g.TW = 16;
  pick_tile(H, W, g.TH, g.TW);
  g.tiles_x = (W + g.TW - 1) / g.TW, g.tiles_y = (H + g.TH - 1) / g.TH;
  const long long blocks = (long long)N * g.tiles_x * g.tiles_y;
  if (blocks > 0x7fffffffLL) return false;
  k_conv3x3_halo<<<dim3((unsigned)blocks, (unsigned)(Cout / BN)), NT, 0, st>>>(x, w, scale, bias, g, y);
  return true;
}

}  // namespace ocpg_halo

#ifdef EXP_STAMPS
extern "C" int ocpg_debug_conv_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ocpg_halo::g_conv_stamps), sizeof(unsigned long long) * n);
}
#endif
