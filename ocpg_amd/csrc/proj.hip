// Box-projection loss of the weakly-supervised mask criterion (with its mean term), forward and backward, all decoder
// layers at once.
//
// Reference: proj_loss (models/segmentation.py:253-277) + dice_coefficient (:203-211), called from
// SetCriterion.loss_masks (models/criterion.py:141-158) per decoder layer for the full- and the low-resolution masks:
// ~30 small kernels forward and ~60 backward per call, 8 calls per step.  Here: 3 launches forward, 2 backward.
//
//   x [Lr, B, T, H, W] logits;  p = sigmoid(x)
//   column stats (reduce H): cmax / cmean [Lr,B,T,W];  row stats (reduce W): rmax / rmean [Lr,B,T,H]
//   targets (no gradient, computed by the caller): tcmax = region.amax(H), tcmean = weak.mean(H)  [B,T,W];
//                                                  trmax = region.amax(W), trmean = weak.mean(W)  [B,T,H]
//   dice(a, t) = 1 - 2 sum(a t) / (sum a^2 + sum t^2 + 1e-5)   over the T*W (or T*H) entries of one (layer, clip)
//   loss_l = mean_b [ dice(cmax, tcmax) + dice(rmax, trmax) ] + 0.1 mean_b [ dice(cmean, tcmean) + dice(rmean, trmean) ]
// Backward: d dice / d a_i = -2 (t_i U - 2 I a_i) / U^2 (I = sum a t, U = the denominator); amax routes it to the
// elements equal to the maximum (shared evenly among ties, as torch.amax's backward does), the mean spreads it by
// 1/H (1/W); times p (1 - p).
// Layout of the work: column stats = one lane per (frame, column) walking down the rows (coalesced across the wave);
// row stats = one wave per row; both single passes with a running (max, tie count, sum).  HBM-bound: x is read twice
// forward and once backward, everything else is O(H + W) per frame.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"
#include "fill.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max_all(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// colstat [F, 3, W]: max, tie count, mean.  A workgroup owns 64 columns of a frame; its 4 waves walk interleaved rows
// (coalesced 256-B row segments) and merge their (max, ties, sum) through LDS: 4x the lanes in flight of one lane per column.
__global__ __launch_bounds__(256) void proj_col_stats(const float* __restrict__ x, int H, int W, float* __restrict__ colstat) {
  __shared__ float sm[3][4][64];
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  float mx = -1.f, cnt = 0.f, sum = 0.f;
  if (col < W) {
    const float* xf = x + (long long)f * H * W + col;
    for (int y = wave; y < H; y += 4) {
      const float p = sigmoidf_(xf[(long long)y * W]);
      sum += p;
      if (p > mx) { mx = p; cnt = 1.f; }
      else if (p == mx) cnt += 1.f;
    }
  }
  sm[0][wave][lane] = mx; sm[1][wave][lane] = cnt; sm[2][wave][lane] = sum;
  __syncthreads();
  if (wave == 0 && col < W) {
    float m = sm[0][0][lane], c = sm[1][0][lane], s_ = sm[2][0][lane];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float mk = sm[0][k][lane], ck = sm[1][k][lane];
      if (mk > m) { m = mk; c = ck; }
      else if (mk == m) c += ck;
      s_ += sm[2][k][lane];
    }
    float* o = colstat + (long long)f * 3 * W + col;
    o[0] = m; o[W] = c; o[2 * W] = s_ / (float)H;
  }
}

// rowstat [F, 3, H]: one wave per row
__global__ __launch_bounds__(256) void proj_row_stats(const float* __restrict__ x, int F, int H, int W, float* __restrict__ rowstat) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)F * H) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + row * W;
  float mx = -1.f, sum = 0.f;
  for (int i = lane; i < W; i += 64) {
    const float p = sigmoidf_(xr[i]);
    sum += p;
    mx = fmaxf(mx, p);
  }
  mx = wave_max_all(mx);
  float cnt = 0.f;
  for (int i = lane; i < W; i += 64) cnt += (sigmoidf_(xr[i]) == mx) ? 1.f : 0.f;     // second read: L1/L2 hit
  sum = wave_sum(sum);
  cnt = wave_sum(cnt);
  if (lane == 0) {
    const long long f = row / H;
    const int y = (int)(row % H);
    float* o = rowstat + f * 3 * H + y;
    o[0] = mx; o[H] = cnt; o[2 * H] = sum / (float)W;
  }
}

__device__ __forceinline__ void block_sum3(float (&v)[3], float (*red)[3]) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float s = wave_sum(v[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
  __syncthreads();
}

// pair k: 0 column max, 1 row max, 2 column mean, 3 row mean
__device__ __forceinline__ void pair_ptrs(int k, const float* colstat, const float* rowstat, const float* tcmax, const float* trmax,
                                          const float* tcmean, const float* trmean, int f0, int bt0, int H, int W, const float*& a,
                                          const float*& t, int& len, int& astride) {
  const bool col = (k & 1) == 0;
  len = col ? W : H;
  astride = 3 * len;
  const float* stat = col ? colstat : rowstat;
  a = stat + (long long)f0 * 3 * len + (k >= 2 ? 2 * len : 0);
  const float* tt = k == 0 ? tcmax : k == 1 ? trmax : k == 2 ? tcmean : trmean;
  t = tt + (long long)bt0 * len;
}

// one block per (l, b): the four dice terms; IU [Lr, B, 4, 2]
__global__ __launch_bounds__(256) void proj_dice(const float* __restrict__ colstat, const float* __restrict__ rowstat,
                                                 const float* __restrict__ tcmax, const float* __restrict__ trmax,
                                                 const float* __restrict__ tcmean, const float* __restrict__ trmean, int B, int T, int H,
                                                 int W, float* __restrict__ IU, float* __restrict__ loss) {
  __shared__ float red[4][3];
  const int b = blockIdx.x, l = blockIdx.y;
  const int f0 = (l * B + b) * T, bt0 = b * T;
  float total = 0.f;
  for (int k = 0; k < 4; ++k) {
    const float *a, *t;
    int len, astride;
    pair_ptrs(k, colstat, rowstat, tcmax, trmax, tcmean, trmean, f0, bt0, H, W, a, t, len, astride);
    float v[3] = {0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < T * len; i += 256) {
      const int tt = i / len, j = i % len;
      const float av = a[(long long)tt * astride + j], tv = t[(long long)tt * len + j];
      v[0] += av * tv; v[1] += av * av; v[2] += tv * tv;
    }
    block_sum3(v, red);
    const float I = v[0], U = v[1] + v[2] + 0.00001f;
    if (threadIdx.x == 0) {
      IU[(((long long)l * B + b) * 4 + k) * 2] = I;
      IU[(((long long)l * B + b) * 4 + k) * 2 + 1] = U;
    }
    total += (k < 2 ? 1.f : 0.1f) * (1.f - 2.f * I / U);
  }
  if (threadIdx.x == 0) atomicAdd(loss + l, total / (float)B);
}

// gradient coefficients: Gc [F, 2, W] (max term / mean term), Gr [F, 2, H]
__global__ __launch_bounds__(256) void proj_coef(const float* __restrict__ colstat, const float* __restrict__ rowstat,
                                                 const float* __restrict__ tcmax, const float* __restrict__ trmax,
                                                 const float* __restrict__ tcmean, const float* __restrict__ trmean,
                                                 const float* __restrict__ IU, const float* __restrict__ gloss, int B, int T, int H,
                                                 int W, float* __restrict__ Gc, float* __restrict__ Gr) {
  const int b = blockIdx.x, l = blockIdx.y;
  const int f0 = (l * B + b) * T, bt0 = b * T;
  const float gl = gloss[l] / (float)B;
  for (int k = 0; k < 4; ++k) {
    const float *a, *t;
    int len, astride;
    pair_ptrs(k, colstat, rowstat, tcmax, trmax, tcmean, trmean, f0, bt0, H, W, a, t, len, astride);
    const float I = IU[(((long long)l * B + b) * 4 + k) * 2], U = IU[(((long long)l * B + b) * 4 + k) * 2 + 1];
    const float scale = gl * (k < 2 ? 1.f : 0.1f) * -2.f / (U * U);
    const bool col = (k & 1) == 0;
    float* G = (col ? Gc : Gr) + (long long)f0 * 2 * len + (k >= 2 ? len : 0);
    const float* cnt = a + len;                         // tie counts sit right after the maxima (max pairs only)
    const float inv_n = 1.f / (float)(col ? H : W);     // the mean spreads over the reduced axis
    for (int i = threadIdx.x; i < T * len; i += 256) {
      const int tt = i / len, j = i % len;
      const float av = a[(long long)tt * astride + j], tv = t[(long long)tt * len + j];
      const float g = scale * (tv * U - 2.f * I * av);
      G[(long long)tt * 2 * len + j] = k < 2 ? g / cnt[(long long)tt * astride + j] : g * inv_n;
    }
  }
}

__global__ __launch_bounds__(256) void proj_bwd(const float* __restrict__ x, const float* __restrict__ colstat,
                                                const float* __restrict__ rowstat, const float* __restrict__ Gc,
                                                const float* __restrict__ Gr, int H, int W, float* __restrict__ gx) {
  const int f = blockIdx.z, y = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= W) return;
  const long long idx = ((long long)f * H + y) * W + col;
  const float p = sigmoidf_(x[idx]);
  const float* cs = colstat + (long long)f * 3 * W;
  const float* rs = rowstat + (long long)f * 3 * H;
  const float* gc = Gc + (long long)f * 2 * W;
  const float* gr = Gr + (long long)f * 2 * H;
  float g = gc[W + col] + gr[H + y];
  if (p == cs[col]) g += gc[col];
  if (p == rs[y]) g += gr[y];
  gx[idx] = g * p * (1.f - p);
}

}  // namespace

extern "C" {

int ocpg_proj_fwd_f32(const float* x, const float* tcmax, const float* trmax, const float* tcmean, const float* trmean, int Lr, int B,
                      int T, int H, int W, float* colstat, float* rowstat, float* IU, float* loss, void* stream) {
  if (Lr <= 0 || B <= 0 || T <= 0 || H <= 0 || W <= 0) return -1006;
  const long long F = (long long)Lr * B * T;
  if (F > 65535 || B > 65535 || Lr > 65535) return -1007;
  if (!x) return -1001;
  if (!tcmax || !trmax || !tcmean || !trmean) return -1002;
  if (!colstat || !rowstat || !IU || !loss) return -1010;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = ocpg_fill::zero_async(loss, sizeof(float) * Lr, st);
  if (e != hipSuccess) return -(int)e;
  proj_col_stats<<<dim3((W + 63) / 64, (unsigned)F), 256, 0, st>>>(x, H, W, colstat);
  proj_row_stats<<<(unsigned)((F * H + 3) / 4), 256, 0, st>>>(x, (int)F, H, W, rowstat);
  proj_dice<<<dim3(B, Lr), 256, 0, st>>>(colstat, rowstat, tcmax, trmax, tcmean, trmean, B, T, H, W, IU, loss);
  e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

int ocpg_proj_bwd_f32(const float* x, const float* tcmax, const float* trmax, const float* tcmean, const float* trmean,
                      const float* colstat, const float* rowstat, const float* IU, const float* gloss, int Lr, int B, int T, int H, int W,
                      float* Gc, float* Gr, float* gx, void* stream) {
  if (Lr <= 0 || B <= 0 || T <= 0 || H <= 0 || W <= 0) return -1006;
  const long long F = (long long)Lr * B * T;
  if (F > 65535 || H > 65535) return -1007;
  if (!x) return -1001;
  if (!tcmax || !trmax || !tcmean || !trmean) return -1002;
  if (!colstat || !rowstat || !IU) return -1003;
  if (!gloss) return -1004;
  if (!Gc || !Gr || !gx) return -1010;
  hipStream_t st = (hipStream_t)stream;
  proj_coef<<<dim3(B, Lr), 256, 0, st>>>(colstat, rowstat, tcmax, trmax, tcmean, trmean, IU, gloss, B, T, H, W, Gc, Gr);
  proj_bwd<<<dim3((W + 255) / 256, H, (unsigned)F), 256, 0, st>>>(x, colstat, rowstat, Gc, Gr, H, W, gx);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // extern "C"
