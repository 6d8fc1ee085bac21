// The LFM block's two Fourier transforms on CHANNELS-LAST maps, hand-written (round 4) -- reference models/modules.py:44-56:
//     spec = fft2(x.float()) * (1 - coef * high);  z = cat([spec.real, spec.imag], 1)   ... two 1x1 convs ...
//     y = ifft2(complex(y[:, :C], y[:, C:]), s=(h, w)).real;  return x + y
// Rounds 1-3 ran rocFFT on [N, C, h, w] planes between two transposing kernels (c2p / p2c, csrc/spectral.hip): per LFM call and
// direction 2 rocFFT launches + its 2 transposes + conjugate-fill / .real / residual copies, 8 calls per step each way -- 3.9 ms of
// the 40 ms step.  Here a 2-D transform is two passes over the channels-last map, one along w and one along h; a lane owns ONE
// CHANNEL of a line, so every global access is 64 consecutive channels (256 B / 512 B) and the 1x1 convs' GEMM operand [N, h, w, 2C]
// is written / read in place -- no transposes, no complex staging, gate, normalisation, Hermitian mirror, .real and the residual in
// the passes' prologues / epilogues:
//   rows_fwd   x real [N,H,W,C] fp32            -> T [N,H,W/2+1,C] complex  (half spectrum along w: the input is real)
//   cols_fwd   T                                -> pair [N,H,W,2C] (Re || Im) * norm * (1 - coef[n] high[u,v]); column W-v is the
//                                                  conjugate mirror of column v
//   cols_inv   pair (* gate)                    -> T' [N,H,W/2+1,C]: Hermitian part of the spectrum (Re ifft2(Y) = ifft2 of it),
//                                                  inverse transform along h; + the partial sums of the gate's coefficient gradient
//   rows_inv   T'                               -> real [N,H,W,C] = norm * Re(inverse along w) (+ residual)
// Forward LFM: rows_fwd, cols_fwd(gate) ... cols_inv, rows_inv(1/hw, + x).  Backward: the gradient of `ifft2(.).real` is
// fft2(g) / hw = rows_fwd, cols_fwd(norm 1/hw); the gradient of fft2 on a real input is Re(unnormalised inverse of gate * gz)
// = cols_inv(gate), rows_inv(1).
//
// A line transform of length L = L1 * L2 (both <= 16; the host picks the pair with the smallest sum; a prime L <= 16 is 1 x L) is
// one Cooley-Tukey split with direct small DFTs on the VALU: step 1 = L2 column DFTs of length L1 (+ twiddle W_L^(j2 k1)), step 2 =
// L1 row DFTs of length L2, in place in an LDS tile [L][64 channels] of float2; a thread keeps the <= 16 inputs of its small DFT in
// registers, the twiddles are powers of W_L read from a table with wave-uniform indices (scalar loads).  fp32 throughout, as the
// reference's FFT; L^(1/2)-ish rounding growth like any FFT.  Cost at the finest level (48 x 80, 10 frames, 256 channels): 5 760 FMAs
// per line and channel along w, 2 688 along h = 1.0 GFMA per 2-D transform (fp32 matrix cores have the VALU's rate on CDNA4, so
// there is nothing to gain from MFMA here); the passes are HBM-bound: 39 + 40 MB, 40 + 39 MB.
// Lengths with a prime factor > 16, or > 128, are not served: ocpg_lfm_dft_supported() says so and the caller keeps rocFFT.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"

namespace {

constexpr int MAXR = 16, NT = 256, NW = NT / 64, CH = 64;
constexpr int U = 8;                    // global loads of a line in flight per thread

struct Split { int L, L1, L2; };

inline Split split_of(int L) {
  Split s{L, 0, 0};
  if (L > 128) return s;                 // the LDS tile [L][64] float2 stays within 64 KB
  int best = 1 << 30;
  for (int a = 1; a <= MAXR; ++a)
    if (L % a == 0 && L / a <= MAXR && a + L / a < best) { best = a + L / a; s.L1 = a; s.L2 = L / a; }
  return s;
}

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return (float)*p; }
template <> __device__ __forceinline__ float ldf<__hip_bfloat16>(const __hip_bfloat16* p) { return __bfloat162float(*p); }
template <> __device__ __forceinline__ float ldf<__half>(const __half* p) { return __half2float(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v) { *p = (T)v; }
template <> __device__ __forceinline__ void stf<__hip_bfloat16>(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }
template <> __device__ __forceinline__ void stf<__half>(__half* p, float v) { *p = __float2half(v); }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// One Cooley-Tukey step: nb small DFTs of length R, in place.  Batch b reads slots b * bstride + j * jstride (j < R) and leaves output
// k in slot b * bstride + k * jstride.  twR = the R-point DFT matrix W_R^(jk) at [k * R + j]: the twiddles of one output are ONE row of
// it, R consecutive float2 with compile-time offsets = a couple of wide scalar loads (wave-uniform address), then R complex
// multiply-adds.  (Indexing one table of W_L powers with (L / R) j k mod L instead cost ~14 scalar instructions per twiddle -- 64-bit
// address arithmetic, compare / select -- against its 4 FMAs: 117 us per launch at 48 x 80 where the arithmetic is 30.)
// STEP1: output k of batch b is also multiplied by W_L^(b k) (twL) and the inputs may be real (REAL_IN: imaginary parts are zero, half
// the products); step 2 (STEP1 false): batch b = k1, output k = k2 is spectrum element k1 + L1 k2, skipped past kmax; REAL_OUT: only
// the real part is wanted.
template <int R, bool INV, bool STEP1, bool REAL_IN, bool REAL_OUT>
__device__ __forceinline__ void dft_batch(float2* __restrict__ tile, const float2* __restrict__ twR, const float2* __restrict__ twL, int nb, int bstride,
                                          int jstride, int L1, int kmax, int lane, int wave) {
  for (int b = wave; b < nb; b += NW) {
    float2 r[R];
#pragma unroll
    for (int j = 0; j < R; ++j) r[j] = tile[(b * bstride + j * jstride) * CH + lane];
    // (outputs go straight back to the batch's own slots: its inputs are all in registers by now)
#pragma unroll 2
    for (int k = 0; k < R; ++k) {
      if (!STEP1 && b + L1 * k > kmax) break;
      const float2* __restrict__ row = twR + k * R;
      float2 t[R];
#pragma unroll
      for (int j = 1; j < R; ++j) t[j] = row[j];
      float2 acc = REAL_IN ? make_float2(r[0].x, 0.f) : r[0];
#pragma unroll
      for (int j = 1; j < R; ++j) {
        const float ty = INV ? -t[j].y : t[j].y;
        acc.x += r[j].x * t[j].x;
        if (!REAL_OUT) acc.y += r[j].x * ty;
        if (!REAL_IN) {
          acc.x -= r[j].y * ty;
          if (!REAL_OUT) acc.y += r[j].y * t[j].x;
        }
      }
      if (STEP1 && k > 0) {
        float2 w = twL[b * k];                  // b < L2, k < L1: b k < L
        if (INV) w.y = -w.y;
        acc = cmul(acc, w);
      }
      tile[(b * bstride + k * jstride) * CH + lane] = acc;
    }
  }
}

#define OCPG_DFT_CASE(R_) case R_: dft_batch<R_, INV, STEP1, REAL_IN, REAL_OUT>(tile, twR, twL, nb, bstride, jstride, L1, kmax, lane, wave); break;
template <bool INV, bool STEP1, bool REAL_IN, bool REAL_OUT>
__device__ __forceinline__ void dft_step(int R, float2* __restrict__ tile, const float2* __restrict__ twR, const float2* __restrict__ twL, int nb, int bstride,
                                         int jstride, int L1, int kmax, int lane, int wave) {
  switch (R) {
    OCPG_DFT_CASE(2) OCPG_DFT_CASE(3) OCPG_DFT_CASE(4) OCPG_DFT_CASE(5) OCPG_DFT_CASE(6) OCPG_DFT_CASE(7) OCPG_DFT_CASE(8) OCPG_DFT_CASE(9)
    OCPG_DFT_CASE(10) OCPG_DFT_CASE(11) OCPG_DFT_CASE(12) OCPG_DFT_CASE(13) OCPG_DFT_CASE(14) OCPG_DFT_CASE(15) OCPG_DFT_CASE(16)
    default: break;                      // R == 1: the identity
  }
}
#undef OCPG_DFT_CASE

// tw: the tables of one length L (float2; forward sign, INV conjugates): [0, L) W_L^m = exp(-2 pi i m / L); [L, L + L1^2) the L1-point
// DFT matrix W_L1^(jk) at k * L1 + j; then the L2-point one (ocpg_lfm_dft_split gives L1, L2; ops/functions/spectral_func.py builds
// them in fp64).  Element j of the line sits at tile[j * CH + lane] on entry; output element k sits at slot (k % L1) * L2 + k / L1 on
// exit (slot_of).  kmax: only outputs k <= kmax are needed (half spectrum of a real line).
template <bool INV, bool REAL_IN = false, bool REAL_OUT = false>
__device__ __forceinline__ void line_dft(float2* __restrict__ tile, const Split s, const float2* __restrict__ tw, int lane, int wave, int kmax) {
  const int L = s.L, L1 = s.L1, L2 = s.L2;
#ifdef DFT_CUT_COMPUTE
  __syncthreads();
  return;
#endif
  const float2* __restrict__ tw1 = tw + L;
  const float2* __restrict__ tw2 = tw1 + L1 * L1;
  if (L1 > 1) {
    // step 1: L2 column DFTs of length L1 over slots j1 * L2 + j2, then the twiddle W_L^(j2 k1)
    dft_step<INV, true, REAL_IN, false>(L1, tile, tw1, tw, L2, 1, L2, L1, kmax, lane, wave);
    __syncthreads();
    // step 2: L1 row DFTs of length L2 over slots k1 * L2 + j2
    dft_step<INV, false, false, REAL_OUT>(L2, tile, tw2, tw, L1, L2, 1, L1, kmax, lane, wave);
  } else {
    dft_step<INV, false, REAL_IN, REAL_OUT>(L2, tile, tw2, tw, 1, L2, 1, 1, kmax, lane, wave);
  }
  __syncthreads();
}

__device__ __forceinline__ int slot_of(const Split& s, int k) { return (k % s.L1) * s.L2 + k / s.L1; }

// ---- along w, real -> half spectrum.  grid (N*H, ceil(C / 64))
__global__ __launch_bounds__(NT) void rows_fwd(const float* __restrict__ x, int W, int C, Split s, const float2* __restrict__ tw, float2* __restrict__ T) {
  extern __shared__ float2 tile[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long line = blockIdx.x;
  const int c = blockIdx.y * CH + lane;
  const bool ok = c < C;
  const int Wh = W / 2 + 1;
  // (loads in batches of U: a load + LDS store per iteration serialises one HBM round trip per element of the line)
  for (int j0 = wave; j0 < W; j0 += NW * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * NW;
      v[u] = (ok && j < W) ? x[(line * W + j) * C + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * NW;
      if (j < W) tile[j * CH + lane] = make_float2(v[u], 0.f);
    }
  }
  __syncthreads();
  line_dft<false, true, false>(tile, s, tw, lane, wave, W / 2);
  if (ok)
    for (int v = wave; v < Wh; v += NW) T[(line * Wh + v) * C + c] = tile[slot_of(s, v) * CH + lane];
}

// ---- along h, half spectrum -> gated [Re || Im] pair, both mirror halves.  grid (N * (W/2+1), ceil(C / 64))
template <typename P>
__global__ __launch_bounds__(NT) void cols_fwd(const float2* __restrict__ T, const float* __restrict__ coef, const float* __restrict__ high, int H, int W,
                                               int C, Split s, const float2* __restrict__ tw, float norm, P* __restrict__ pair) {
  extern __shared__ float2 tile[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Wh = W / 2 + 1;
  const int n = blockIdx.x / Wh, v = blockIdx.x - n * Wh;
  const int c = blockIdx.y * CH + lane;
  const bool ok = c < C;
  for (int y0 = wave; y0 < H; y0 += NW * U) {
    float2 q[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const int y = y0 + i * NW;
      q[i] = (ok && y < H) ? T[(((long long)n * H + y) * Wh + v) * C + c] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const int y = y0 + i * NW;
      if (y < H) tile[y * CH + lane] = q[i];
    }
  }
  __syncthreads();
  line_dft<false>(tile, s, tw, lane, wave, H);
  if (!ok) return;
  const float cf = coef ? coef[n] : 0.f;
  const int vm = W - v;                                   // the mirror column (v = 0 and v = W / 2 mirror onto themselves)
  const bool mirror = v > 0 && vm > v;
  for (int u = wave; u < H; u += NW) {
    const float2 sv = tile[slot_of(s, u) * CH + lane];
    const float g = norm * (coef ? 1.f - cf * high[u * W + v] : 1.f);
    P* d = pair + (((long long)n * H + u) * W + v) * 2 * C + c;
    stf(d, sv.x * g);
    stf(d + C, sv.y * g);
    if (mirror) {                                         // S[(H - u) % H][W - v] = conj(S[u][v])
      const int um = u ? H - u : 0;
      const float gm = norm * (coef ? 1.f - cf * high[um * W + vm] : 1.f);
      P* dm = pair + (((long long)n * H + um) * W + vm) * 2 * C + c;
      stf(dm, sv.x * gm);
      stf(dm + C, -sv.y * gm);
    }
  }
}

// ---- pair (* gate) -> Hermitian part -> inverse along h.  grid (N * (W/2+1), ceil(C / 64)); part [N][(W/2+1) * gridDim.y]
template <typename P>
__global__ __launch_bounds__(NT) void cols_inv(const P* __restrict__ pair, const float* __restrict__ coef, const float* __restrict__ high,
                                               const P* __restrict__ zs, float* __restrict__ part, int H, int W, int C, Split s,
                                               const float2* __restrict__ tw, float2* __restrict__ T) {
  extern __shared__ float2 tile[];
  __shared__ float red[NW];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Wh = W / 2 + 1;
  const int n = blockIdx.x / Wh, v = blockIdx.x - n * Wh;
  const int c = blockIdx.y * CH + lane;
  const bool ok = c < C;
  const float cf = coef ? coef[n] : 0.f;
  const int vm = v ? W - v : 0;
  float dsum = 0.f;
  constexpr int UI = 4;
  for (int u0 = wave; u0 < H; u0 += NW * UI) {
    float ar[UI], ai[UI], br[UI], bi[UI], za[UI], zb[UI], zc[UI], zd[UI], ha[UI], hb[UI];
#pragma unroll
    for (int i = 0; i < UI; ++i) {
      const int u = u0 + i * NW;
      ar[i] = ai[i] = br[i] = bi[i] = za[i] = zb[i] = zc[i] = zd[i] = ha[i] = hb[i] = 0.f;
      if (ok && u < H) {
        const int um = u ? H - u : 0;
        const long long ia = (((long long)n * H + u) * W + v) * 2 * C + c, ib = (((long long)n * H + um) * W + vm) * 2 * C + c;
        ar[i] = ldf(pair + ia), ai[i] = ldf(pair + ia + C), br[i] = ldf(pair + ib), bi[i] = ldf(pair + ib + C);
        if (coef) ha[i] = high[u * W + v], hb[i] = high[um * W + vm];
        if (part) {
          za[i] = ldf(zs + ia), zb[i] = ldf(zs + ia + C);
          if (vm != v) zc[i] = ldf(zs + ib), zd[i] = ldf(zs + ib + C);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < UI; ++i) {
      const int u = u0 + i * NW;
      if (u >= H) continue;
      const float ga = 1.f - cf * ha[i], gb = 1.f - cf * hb[i];
      // d/dcoef of S * (1 - coef * high): -high * (g_re S_re + g_im S_im), with S = z / gate from the saved gated spectrum (a gate
      // of exactly 0 -- coef == 1.0f at the one frequency where high == 1 -- has lost S: that single term is dropped)
      if (part && ok) {
        if (ga != 0.f) dsum -= ha[i] * (ar[i] * za[i] + ai[i] * zb[i]) / ga;
        if (vm != v && gb != 0.f) dsum -= hb[i] * (br[i] * zc[i] + bi[i] * zd[i]) / gb;
      }
      tile[u * CH + lane] = make_float2(0.5f * (ar[i] * ga + br[i] * gb), 0.5f * (ai[i] * ga - bi[i] * gb));
    }
  }
  __syncthreads();
  line_dft<true>(tile, s, tw, lane, wave, H);
  if (ok)
    for (int y = wave; y < H; y += NW) T[(((long long)n * H + y) * Wh + v) * C + c] = tile[slot_of(s, y) * CH + lane];
  if (part) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dsum += __shfl_down(dsum, o, 64);
    if (lane == 0) red[wave] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int i = 0; i < NW; ++i) t += red[i];
      part[((long long)n * Wh + v) * gridDim.y + blockIdx.y] = t;
    }
  }
}

// ---- half spectrum -> real line (+ residual).  grid (N*H, ceil(C / 64))
__global__ __launch_bounds__(NT) void rows_inv(const float2* __restrict__ T, int W, int C, Split s, const float2* __restrict__ tw, float norm,
                                               const float* __restrict__ residual, float* __restrict__ out) {
  extern __shared__ float2 tile[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long line = blockIdx.x;
  const int c = blockIdx.y * CH + lane;
  const bool ok = c < C;
  const int Wh = W / 2 + 1;
  for (int v0 = wave; v0 < W; v0 += NW * U) {
    float2 q[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const int v = v0 + i * NW;
      q[i] = make_float2(0.f, 0.f);
      if (ok && v < W) q[i] = T[(line * Wh + (v < Wh ? v : W - v)) * C + c];
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const int v = v0 + i * NW;
      if (v < W) tile[v * CH + lane] = make_float2(q[i].x, v < Wh ? q[i].y : -q[i].y);
    }
  }
  __syncthreads();
  line_dft<true, false, true>(tile, s, tw, lane, wave, W);
  if (ok)
    for (int j0 = wave; j0 < W; j0 += NW * U) {
      float rs[U];
#pragma unroll
      for (int i = 0; i < U; ++i) {
        const int j = j0 + i * NW;
        rs[i] = (residual && j < W) ? residual[(line * W + j) * C + c] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < U; ++i) {
        const int j = j0 + i * NW;
        if (j < W) out[(line * W + j) * C + c] = norm * tile[slot_of(s, j) * CH + lane].x + rs[i];
      }
    }
}

inline int status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace

extern "C" {

int ocpg_lfm_dft_split(int L) {                  /* L1 * 256 + L2 of the line transform of length L; 0: not served */
  if (L < 1) return 0;
  const Split s = split_of(L);
  return s.L1 ? s.L1 * 256 + s.L2 : 0;
}

int ocpg_lfm_dft_supported(int H, int W) {
  if (H < 1 || W < 1) return 0;
  return split_of(H).L1 > 0 && split_of(W).L1 > 0;
}

int ocpg_lfm_spectrum_fwd(const float* x, const float* coef, const float* high, int N, int H, int W, int C, const void* tw_h, const void* tw_w,
                          float norm, void* tmp, void* pair, int pair_dt, void* stream) {
  if (N < 0 || H < 1 || W < 1 || C < 1) return -1003;
  if (N == 0) return 0;
  const Split sh = split_of(H), sw = split_of(W);
  if (!sh.L1 || !sw.L1) return -2000;
  if (!x) return -1001;
  if (!tw_h || !tw_w) return -1005;
  if (!tmp) return -1011;
  if (!pair) return -1012;
  if ((coef == nullptr) != (high == nullptr)) return -1002;
  if (pair_dt < 0 || pair_dt > 2) return -1013;
  hipStream_t st = (hipStream_t)stream;
  const int Wh = W / 2 + 1, slabs = (C + CH - 1) / CH;
  if ((long long)N * H > 2147483647LL || (long long)N * Wh > 2147483647LL) return -1003;
  rows_fwd<<<dim3((unsigned)(N * H), slabs), NT, (size_t)W * CH * sizeof(float2), st>>>(x, W, C, sw, (const float2*)tw_w, (float2*)tmp);
  const dim3 g((unsigned)(N * Wh), slabs);
  const size_t lds = (size_t)H * CH * sizeof(float2);
  if (pair_dt == 0) cols_fwd<float><<<g, NT, lds, st>>>((const float2*)tmp, coef, high, H, W, C, sh, (const float2*)tw_h, norm, (float*)pair);
  else if (pair_dt == 1) cols_fwd<__hip_bfloat16><<<g, NT, lds, st>>>((const float2*)tmp, coef, high, H, W, C, sh, (const float2*)tw_h, norm, (__hip_bfloat16*)pair);
  else cols_fwd<__half><<<g, NT, lds, st>>>((const float2*)tmp, coef, high, H, W, C, sh, (const float2*)tw_h, norm, (__half*)pair);
  return status();
}

int ocpg_lfm_spectrum_inv(const void* pair, int pair_dt, const float* coef, const float* high, const void* z_saved, float* coef_part, int N, int H,
                          int W, int C, const void* tw_h, const void* tw_w, float norm, void* tmp, const float* residual, float* out, void* stream) {
  if (N < 0 || H < 1 || W < 1 || C < 1) return -1007;
  if (N == 0) return 0;
  const Split sh = split_of(H), sw = split_of(W);
  if (!sh.L1 || !sw.L1) return -2000;
  if (!pair) return -1001;
  if (!tw_h || !tw_w) return -1011;
  if (!tmp) return -1014;
  if (!out) return -1016;
  if ((coef == nullptr) != (high == nullptr)) return -1003;
  if (coef_part && (!coef || !z_saved)) return -1005;
  if (pair_dt < 0 || pair_dt > 2) return -1002;
  hipStream_t st = (hipStream_t)stream;
  const int Wh = W / 2 + 1, slabs = (C + CH - 1) / CH;
  if ((long long)N * H > 2147483647LL || (long long)N * Wh > 2147483647LL) return -1007;
  const dim3 g((unsigned)(N * Wh), slabs);
  const size_t lds = (size_t)H * CH * sizeof(float2);
  if (pair_dt == 0) cols_inv<float><<<g, NT, lds, st>>>((const float*)pair, coef, high, (const float*)z_saved, coef_part, H, W, C, sh, (const float2*)tw_h, (float2*)tmp);
  else if (pair_dt == 1) cols_inv<__hip_bfloat16><<<g, NT, lds, st>>>((const __hip_bfloat16*)pair, coef, high, (const __hip_bfloat16*)z_saved, coef_part, H, W, C, sh, (const float2*)tw_h, (float2*)tmp);
  else cols_inv<__half><<<g, NT, lds, st>>>((const __half*)pair, coef, high, (const __half*)z_saved, coef_part, H, W, C, sh, (const float2*)tw_h, (float2*)tmp);
  rows_inv<<<dim3((unsigned)(N * H), slabs), NT, (size_t)W * CH * sizeof(float2), st>>>((const float2*)tmp, W, C, sw, (const float2*)tw_w, norm, residual, out);
  return status();
}

}  // extern "C"
