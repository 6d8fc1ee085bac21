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

// tw[k] = exp(-2 pi i k / L) (forward); INV conjugates.  Element j of the line sits at tile[j * CH + lane] on entry; output element
// k sits at slot (k % L1) * L2 + k / L1 on exit (slot_of).  kmax: only outputs k <= kmax are needed (half spectrum of a real line).
template <bool INV>
__device__ __forceinline__ void line_dft(float2* __restrict__ tile, const Split s, const float2* __restrict__ tw, int lane, int wave, int kmax) {
  const int L = s.L, L1 = s.L1, L2 = s.L2;
  if (L1 > 1) {
    for (int j2 = wave; j2 < L2; j2 += NW) {
      float2 r[MAXR];
#pragma unroll
      for (int j1 = 0; j1 < MAXR; ++j1)
        if (j1 < L1) r[j1] = tile[(j1 * L2 + j2) * CH + lane];
      for (int k1 = 0; k1 < L1; ++k1) {
        float2 acc = r[0];
        const int st = (L2 * k1) % L;          // W_L1^(j1 k1) = W_L^(L2 j1 k1)
        int ti = 0;
#pragma unroll
        for (int j1 = 1; j1 < MAXR; ++j1)
          if (j1 < L1) {
            ti += st;
            if (ti >= L) ti -= L;
            float2 t = tw[ti];
            if (INV) t.y = -t.y;
            acc.x += r[j1].x * t.x - r[j1].y * t.y;
            acc.y += r[j1].x * t.y + r[j1].y * t.x;
          }
        float2 t = tw[(j2 * k1) % L];
        if (INV) t.y = -t.y;
        tile[(k1 * L2 + j2) * CH + lane] = cmul(acc, t);
      }
    }
    __syncthreads();
  }
  for (int k1 = wave; k1 < L1; k1 += NW) {
    float2 r[MAXR];
#pragma unroll
    for (int j2 = 0; j2 < MAXR; ++j2)
      if (j2 < L2) r[j2] = tile[(k1 * L2 + j2) * CH + lane];
    for (int k2 = 0; k2 < L2; ++k2) {
      if (k1 + L1 * k2 > kmax) break;
      float2 acc = r[0];
      const int st = (L1 * k2) % L;            // W_L2^(j2 k2) = W_L^(L1 j2 k2)
      int ti = 0;
#pragma unroll
      for (int j2 = 1; j2 < MAXR; ++j2)
        if (j2 < L2) {
          ti += st;
          if (ti >= L) ti -= L;
          float2 t = tw[ti];
          if (INV) t.y = -t.y;
          acc.x += r[j2].x * t.x - r[j2].y * t.y;
          acc.y += r[j2].x * t.y + r[j2].y * t.x;
        }
      tile[(k1 * L2 + k2) * CH + lane] = acc;
    }
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
  for (int j = wave; j < W; j += NW) tile[j * CH + lane] = make_float2(ok ? x[(line * W + j) * C + c] : 0.f, 0.f);
  __syncthreads();
  line_dft<false>(tile, s, tw, lane, wave, W / 2);
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
  for (int y = wave; y < H; y += NW) tile[y * CH + lane] = ok ? T[(((long long)n * H + y) * Wh + v) * C + c] : make_float2(0.f, 0.f);
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
  for (int u = wave; u < H; u += NW) {
    float2 h = make_float2(0.f, 0.f);
    if (ok) {
      const int um = u ? H - u : 0;
      const long long ia = (((long long)n * H + u) * W + v) * 2 * C + c, ib = (((long long)n * H + um) * W + vm) * 2 * C + c;
      const float ha = coef ? high[u * W + v] : 0.f, hb = coef ? high[um * W + vm] : 0.f;
      const float ga = 1.f - cf * ha, gb = 1.f - cf * hb;
      const float ar = ldf(pair + ia), ai = ldf(pair + ia + C), br = ldf(pair + ib), bi = ldf(pair + ib + C);
      h = make_float2(0.5f * (ar * ga + br * gb), 0.5f * (ai * ga - bi * gb));
      if (part) {
        // d/dcoef of S * (1 - coef * high): -high * (g_re S_re + g_im S_im), with S = z / gate from the saved gated spectrum (a gate
        // of exactly 0 -- coef == 1.0f at the one frequency where high == 1 -- has lost S: that single term is dropped)
        if (ga != 0.f) dsum -= ha * (ar * ldf(zs + ia) + ai * ldf(zs + ia + C)) / ga;
        if (vm != v && gb != 0.f) dsum -= hb * (br * ldf(zs + ib) + bi * ldf(zs + ib + C)) / gb;
      }
    }
    tile[u * CH + lane] = h;
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
  for (int v = wave; v < W; v += NW) {
    float2 t = make_float2(0.f, 0.f);
    if (ok) {
      if (v < Wh) t = T[(line * Wh + v) * C + c];
      else { t = T[(line * Wh + (W - v)) * C + c]; t.y = -t.y; }
    }
    tile[v * CH + lane] = t;
  }
  __syncthreads();
  line_dft<true>(tile, s, tw, lane, wave, W);
  if (ok)
    for (int j = wave; j < W; j += NW) {
      const long long i = (line * W + j) * C + c;
      out[i] = norm * tile[slot_of(s, j) * CH + lane].x + (residual ? residual[i] : 0.f);
    }
}

inline int status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace

extern "C" {

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
