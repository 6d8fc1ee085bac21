// Level-set loss of the weakly-supervised mask criterion, forward and backward, for all decoder layers at once.
//
// Reference: levelset_loss + region_levelset + length_regularization (models/segmentation.py:279-315), called from
// SetCriterion.loss_masks (models/criterion.py:160-178) once per decoder layer and resolution: ~60 elementwise /
// reduction kernels forward and ~120 backward per call, 8 calls per step.  Here: 2 launches forward, 1 backward.
//
//   x     [Lr, N, h, w]   mask logits (Lr decoder layers, N = clips x frames)
//   feats [N, CF, h, w]   level-set features, the first C (<= CF) channels are used (the criterion drops the last one)
//   box   [N, h, w]       {0,1} box region
//   p = sigmoid(x); fg = p box; bg = (1 - p) box; t_c = feats_c box; t2 = sum_c t_c^2; pixels_n = max(sum box, 1)
//   for w in {fg, bg}:  W = sum w;  D = max(W, 1e-5);  S_c = sum w t_c;  c_c = S_c / D;
//                       E = sum w t2 - 2 sum_c c_c S_c + sum_c c_c^2 W          ( = sum_c sum_p w (t_c - c_c)^2 when W >= 1e-5 )
//                       len = sum |w[y+1,x] - w[y,x]| + sum |w[y,x+1] - w[y,x]|
//   loss_l = mean_n [ (E_fg + E_bg) / C / pixels_n + 1e-5 (len_fg + len_bg) / pixels_n ]
//
// Backward (r = W / D, delta = [W > 1e-5]; r = delta = 1 in the regular case):
//   dE/dw_p  = t2_p - (4 - 2 r) c.t_p + (1 + 2 delta - 2 r delta) |c|^2                 ( = sum_c (t_pc - c_c)^2 regular )
//   dE/dt_pc = 2 w_p (t_pc - (2 - r) c_c)
//   dlen/dw_p = sgn(w_p - w_up) - sgn(w_down - w_p) + sgn(w_p - w_left) - sgn(w_right - w_p)   (terms that exist)
//   dx = (dfg - dbg) p (1 - p) box;   dfeats_c = box sum_l dt_c
//
// levelset_sums: one lane per pixel, 2C+7 partial sums per (layer, frame) reduced wave -> block -> one plain store per sum into
// the workgroup's partial row; levelset_final reduces the rows (240 workgroups adding into the same 29 addresses serialise).
// HBM-bound streaming passes: (Lr + C + 1) floats in per pixel forward; backward the same in, (Lr + CF) out.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"
#include "fill.h"

namespace {

constexpr int CMAX = 16;                 // level-set feature channels used (the reference uses 11)
constexpr float EPS_W = 0.00001f;        // clamp of the region mass (segmentation.py:286-287)
constexpr int PPT = 4;                   // pixels per lane in levelset_sums

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float sgnf_(float v) { return (v > 0.f) - (v < 0.f); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// sums layout per (l, n): [0] W_fg [1] W_bg [2] A_fg [3] A_bg [4] len_fg [5] len_bg [6] sum box [7..7+C) S_fg [7+C..7+2C) S_bg
__global__ __launch_bounds__(256) void levelset_sums(const float* __restrict__ x, const float* __restrict__ feats,
                                                     const float* __restrict__ box, int N, int C, int CF, int h, int w,
                                                     float* __restrict__ sums) {
  const int hw = h * w;
  const int n = blockIdx.y, l = blockIdx.z;
  const int NS = 7 + 2 * C;
  float acc[7 + 2 * CMAX];
#pragma unroll
  for (int i = 0; i < 7 + 2 * CMAX; ++i) acc[i] = 0.f;
  const float* xl = x + ((long long)l * N + n) * hw;
  const float* bx = box + (long long)n * hw;
#pragma unroll 1
  for (int it = 0; it < PPT; ++it) {               // PPT pixels per lane: the 29-value workgroup reduction is paid once per 1024 pixels
    const int px = blockIdx.x * (256 * PPT) + it * 256 + threadIdx.x;
    if (px >= hw) break;
    const float b = bx[px];
    const float p = sigmoidf_(xl[px]);
    const float fg = p * b, bg = (1.f - p) * b;
    const int yy = px / w, xx = px % w;
    if (yy + 1 < h) {
      const float pd = sigmoidf_(xl[px + w]), bd = bx[px + w];
      acc[4] += fabsf(pd * bd - fg);
      acc[5] += fabsf((1.f - pd) * bd - bg);
    }
    if (xx + 1 < w) {
      const float pr = sigmoidf_(xl[px + 1]), br = bx[px + 1];
      acc[4] += fabsf(pr * br - fg);
      acc[5] += fabsf((1.f - pr) * br - bg);
    }
    acc[6] += b;
    if (b != 0.f) {
      const float* f = feats + (long long)n * CF * hw + px;
      float t2 = 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        if (c < C) {
          const float t = f[(long long)c * hw] * b;
          t2 += t * t;
          acc[7 + c] += fg * t;
          acc[7 + CMAX + c] += bg * t;
        }
      }
      acc[0] += fg; acc[1] += bg; acc[2] += fg * t2; acc[3] += bg * t2;
    }
  }
  __shared__ float red[4][7 + 2 * CMAX];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 7 + 2 * CMAX; ++i) {
    const bool used = i < 7 || (i < 7 + CMAX ? i - 7 < C : i - 7 - CMAX < C);
    if (used) {
      const float v = wave_sum(acc[i]);
      if (lane == 0) red[wave][i] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < 7 + 2 * CMAX) {
    const int i = threadIdx.x;
    const bool used = i < 7 || (i < 7 + CMAX ? i - 7 < C : i - 7 - CMAX < C);
    if (used) {
      const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
      const int slot = i < 7 ? i : (i < 7 + CMAX ? 7 + (i - 7) : 7 + C + (i - 7 - CMAX));
      // per-workgroup partial (plain store): part [Lr, N, gridDim.x, NS]; levelset_final reduces them
      sums[(((long long)l * N + n) * gridDim.x + blockIdx.x) * NS + slot] = v;
    }
  }
}

// coef layout per (l, n): [0] k1_fg [1] k2_fg [2] k1_bg [3] k2_bg [4] (2-r)_fg [5] (2-r)_bg [6] 1/(C pixels) [7] 1e-5/pixels
//                         [8..8+C) c_fg [8+C..8+2C) c_bg
__global__ __launch_bounds__(64) void levelset_final(const float* __restrict__ part, int nblk, int N, int C, float* __restrict__ coef,
                                                     float* __restrict__ loss) {
  __shared__ float s[7 + 2 * CMAX];
  const int n = blockIdx.x, l = blockIdx.y;
  const int NS = 7 + 2 * C, NC = 8 + 2 * C;
  const float* p = part + ((long long)l * N + n) * nblk * NS;
  for (int slot = 0; slot < NS; ++slot) {            // reduce the workgroup partials of this (layer, frame)
    float v = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 64) v += p[(long long)b * NS + slot];
    v = wave_sum(v);
    if (threadIdx.x == 0) s[slot] = v;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float* k = coef + ((long long)l * N + n) * NC;
  const float pixels = fmaxf(s[6], 1.f);
  float e_tot = 0.f;
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const float W = s[side], A = s[2 + side];
    const float D = fmaxf(W, EPS_W);
    const float r = W / D, delta = W > EPS_W ? 1.f : 0.f;
    float cs = 0.f, cc = 0.f;
    for (int c = 0; c < C; ++c) {
      const float S = s[7 + side * C + c];
      const float cv = S / D;
      k[8 + side * C + c] = cv;
      cs += cv * S;
      cc += cv * cv;
    }
    e_tot += A - 2.f * cs + cc * W;
    k[2 * side] = 4.f - 2.f * r;
    k[2 * side + 1] = (1.f + 2.f * delta - 2.f * r * delta) * cc;
    k[4 + side] = 2.f - r;
  }
  k[6] = 1.f / ((float)C * pixels);
  k[7] = 0.00001f / pixels;
  atomicAdd(loss + l, (e_tot / (float)C / pixels + 0.00001f * (s[4] + s[5]) / pixels) / (float)N);      // N adds per address
}

__global__ __launch_bounds__(256) void levelset_bwd(const float* __restrict__ x, const float* __restrict__ feats,
                                                    const float* __restrict__ box, const float* __restrict__ coef,
                                                    const float* __restrict__ gloss, int Lr, int N, int C, int CF, int h, int w,
                                                    float* __restrict__ gx, float* __restrict__ gfeat) {
  const int hw = h * w;
  const int n = blockIdx.y;
  const int px = blockIdx.x * 256 + threadIdx.x;
  if (px >= hw) return;
  const int NC = 8 + 2 * C;
  const float* bx = box + (long long)n * hw;
  const float b = bx[px];
  const int yy = px / w, xx = px % w;
  float t[CMAX], gt[CMAX];
  float t2 = 0.f;
  const float* f = feats + (long long)n * CF * hw + px;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    t[c] = (c < C && b != 0.f) ? f[(long long)c * hw] * b : 0.f;
    t2 += t[c] * t[c];
    gt[c] = 0.f;
  }
  const bool up = yy > 0, down = yy + 1 < h, left = xx > 0, right = xx + 1 < w;
  const float b_u = up ? bx[px - w] : 0.f, b_d = down ? bx[px + w] : 0.f, b_l = left ? bx[px - 1] : 0.f, b_r = right ? bx[px + 1] : 0.f;
  for (int l = 0; l < Lr; ++l) {
    const float* xl = x + ((long long)l * N + n) * hw;
    float g = 0.f;
    if (b != 0.f) {
      const float* k = coef + ((long long)l * N + n) * NC;
      const float G = gloss[l] / (float)N;
      const float p = sigmoidf_(xl[px]);
      const float fg = p * b, bg = (1.f - p) * b;
      float dot_f = 0.f, dot_b = 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        if (c < C) {
          dot_f += k[8 + c] * t[c];
          dot_b += k[8 + C + c] * t[c];
        }
      }
      const float de_f = t2 - k[0] * dot_f + k[1], de_b = t2 - k[2] * dot_b + k[3];
      float dl_f = 0.f, dl_b = 0.f;
      if (up)    { const float q = sigmoidf_(xl[px - w]); dl_f += sgnf_(fg - q * b_u); dl_b += sgnf_(bg - (1.f - q) * b_u); }
      if (down)  { const float q = sigmoidf_(xl[px + w]); dl_f -= sgnf_(q * b_d - fg); dl_b -= sgnf_((1.f - q) * b_d - bg); }
      if (left)  { const float q = sigmoidf_(xl[px - 1]); dl_f += sgnf_(fg - q * b_l); dl_b += sgnf_(bg - (1.f - q) * b_l); }
      if (right) { const float q = sigmoidf_(xl[px + 1]); dl_f -= sgnf_(q * b_r - fg); dl_b -= sgnf_((1.f - q) * b_r - bg); }
      const float gfg = G * (k[6] * de_f + k[7] * dl_f), gbg = G * (k[6] * de_b + k[7] * dl_b);
      g = (gfg - gbg) * p * (1.f - p) * b;
      const float sf = 2.f * G * k[6] * fg, sb = 2.f * G * k[6] * bg;
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < C) gt[c] += sf * (t[c] - k[4] * k[8 + c]) + sb * (t[c] - k[5] * k[8 + C + c]);
    }
    gx[((long long)l * N + n) * hw + px] = g;
  }
  if (gfeat) {
    float* gf = gfeat + (long long)n * CF * hw + px;
    for (int c = 0; c < CF; ++c) gf[(long long)c * hw] = c < C ? gt[c < CMAX ? c : 0] * b : 0.f;
  }
}

}  // namespace

extern "C" {

int ocpg_levelset_fwd_f32(const float* x, const float* feats, const float* box, int Lr, int N, int C, int CF, int h, int w, float* sums,
                          float* coef, float* loss, void* stream) {
  if (Lr <= 0 || N <= 0 || C <= 0 || C > CMAX || CF < C || h <= 0 || w <= 0) return -1006;
  if (N > 65535 || Lr > 65535) return -1007;
  if (!x) return -1001;
  if (!feats) return -1002;
  if (!box) return -1003;
  if (!sums || !coef || !loss) return -1010;
  hipStream_t st = (hipStream_t)stream;
  const int hw = h * w;
  const int nblk = (hw + 256 * PPT - 1) / (256 * PPT);
  hipError_t e = ocpg_fill::zero_async(loss, sizeof(float) * (size_t)Lr, st);
  if (e != hipSuccess) return -(int)e;
  levelset_sums<<<dim3(nblk, N, Lr), 256, 0, st>>>(x, feats, box, N, C, CF, h, w, sums);
  levelset_final<<<dim3(N, Lr), 64, 0, st>>>(sums, nblk, N, C, coef, loss);
  e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

int ocpg_levelset_bwd_f32(const float* x, const float* feats, const float* box, const float* coef, const float* gloss, int Lr, int N,
                          int C, int CF, int h, int w, float* gx, float* gfeat, void* stream) {
  if (Lr <= 0 || N <= 0 || C <= 0 || C > CMAX || CF < C || h <= 0 || w <= 0) return -1006;
  if (N > 65535) return -1007;
  if (!x) return -1001;
  if (!feats) return -1002;
  if (!box) return -1003;
  if (!coef) return -1004;
  if (!gloss) return -1005;
  if (!gx) return -1010;
  const int hw = h * w;
  levelset_bwd<<<dim3((hw + 255) / 256, N), 256, 0, (hipStream_t)stream>>>(x, feats, box, coef, gloss, Lr, N, C, CF, h, w, gx, gfeat);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // extern "C"
