/* CPU oracle for MSDeformAttn forward/backward -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's CUDA op (models/ops/src/cuda/ms_deform_im2col_cuda.cuh):
 *   forward   ms_deformable_im2col_gpu_kernel            cuh:237-299  (bilinear fetch cuh:33-84)
 *   backward  ms_deformable_col2im_gpu_kernel_*          cuh:301-403  (per-sample math cuh:87-159)
 * Semantics restated (not the code): for every (b, q, m) and every (level l, point p)
 *   h_im = loc_y * H_l - 0.5,  w_im = loc_x * W_l - 0.5              (align_corners=False)
 *   the sample contributes only if -1 < h_im < H_l and -1 < w_im < W_l (cuh:287);
 *   each of the 4 bilinear corners is used only if it lies inside the map (zero padding);
 *   out[b,q,m,:] = sum_{l,p} attn[b,q,m,l,p] * bilinear(value_l[b,:,m,:], h_im, w_im).
 * Backward, per sample and channel c (cuh:111-158):
 *   grad_value[corner] += w_corner * attn * go[c]
 *   grad_attn          += go[c] * bilinear
 *   grad_loc_x         += W_l * attn * go[c] * d(bilinear)/d(w_im),   grad_loc_y likewise with H_l.
 * Layouts: value [N,S,M,D], shapes [L,2] int64 (H,W), level_start [L] int64, loc [N,Lq,M,L,P,2] (x,y),
 * attn [N,Lq,M,L,P], out/grad_out [N,Lq,M*D].  Caller zeroes nothing: outputs are fully overwritten
 * (grad_value is zeroed here, matching at::zeros_like in ms_deform_attn_cuda.cu:121-123).
 *
 * Pinned by tests/golden/msda_testpy.npz and msda_cases.npz (generated from the reference's own
 * ms_deform_attn_core_pytorch, the ground truth of the reference's test.py).
 * Build: gcc -O2 -fopenmp -shared -fPIC (see oracle/build.py). Threads: OpenMP over the batch index.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define DEFINE_MSDA(T, SUF)                                                                        \
  void msda_oracle_fwd_##SUF(const T* value, const int64_t* shapes, const int64_t* lstart,          \
                             const T* loc, const T* attn, int N, int S, int M, int D, int L,        \
                             int Lq, int P, T* out) {                                               \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                        \
    for (int b = 0; b < N; ++b)                                                                     \
      for (int q = 0; q < Lq; ++q)                                                                  \
        for (int m = 0; m < M; ++m) {                                                               \
          T* o = out + (((int64_t)b * Lq + q) * M + m) * D;                                         \
          for (int c = 0; c < D; ++c) o[c] = 0;                                                     \
          const int64_t wbase = (((int64_t)b * Lq + q) * M + m) * L * P;                            \
          for (int l = 0; l < L; ++l) {                                                             \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                           \
            const T* v = value + ((int64_t)b * S + lstart[l]) * M * D + (int64_t)m * D;             \
            for (int p = 0; p < P; ++p) {                                                           \
              const int64_t wi = wbase + (int64_t)l * P + p;                                        \
              const T x = loc[2 * wi] * W - (T)0.5, y = loc[2 * wi + 1] * H - (T)0.5;               \
              if (!(y > -1 && x > -1 && y < H && x < W)) continue;                                  \
              const int y0 = (int)floor((double)y), x0 = (int)floor((double)x);                     \
              const T ly = y - y0, lx = x - x0, hy = 1 - ly, hx = 1 - lx;                           \
              const T a = attn[wi];                                                                 \
              const T cw[4] = {hy * hx, hy * lx, ly * hx, ly * lx};                                 \
              const int cy[4] = {y0, y0, y0 + 1, y0 + 1}, cx[4] = {x0, x0 + 1, x0, x0 + 1};         \
              for (int k = 0; k < 4; ++k) {                                                         \
                if (cy[k] < 0 || cy[k] > H - 1 || cx[k] < 0 || cx[k] > W - 1) continue;             \
                const T* px = v + ((int64_t)cy[k] * W + cx[k]) * M * D;                             \
                const T wk = cw[k] * a;                                                             \
                for (int c = 0; c < D; ++c) o[c] += wk * px[c];                                     \
              }                                                                                     \
            }                                                                                       \
          }                                                                                         \
        }                                                                                           \
  }                                                                                                 \
  void msda_oracle_bwd_##SUF(const T* value, const int64_t* shapes, const int64_t* lstart,          \
                             const T* loc, const T* attn, const T* go, int N, int S, int M, int D,  \
                             int L, int Lq, int P, T* gvalue, T* gloc, T* gattn) {                  \
    memset(gvalue, 0, sizeof(T) * (size_t)N * S * M * D);                                           \
    _Pragma("omp parallel for schedule(static)")                                                    \
    for (int b = 0; b < N; ++b)                                                                     \
      for (int q = 0; q < Lq; ++q)                                                                  \
        for (int m = 0; m < M; ++m) {                                                               \
          const T* g = go + (((int64_t)b * Lq + q) * M + m) * D;                                    \
          const int64_t wbase = (((int64_t)b * Lq + q) * M + m) * L * P;                            \
          for (int l = 0; l < L; ++l) {                                                             \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                           \
            const int64_t voff = ((int64_t)b * S + lstart[l]) * M * D + (int64_t)m * D;             \
            for (int p = 0; p < P; ++p) {                                                           \
              const int64_t wi = wbase + (int64_t)l * P + p;                                        \
              gattn[wi] = 0; gloc[2 * wi] = 0; gloc[2 * wi + 1] = 0;                                \
              const T x = loc[2 * wi] * W - (T)0.5, y = loc[2 * wi + 1] * H - (T)0.5;               \
              if (!(y > -1 && x > -1 && y < H && x < W)) continue;                                  \
              const int y0 = (int)floor((double)y), x0 = (int)floor((double)x);                     \
              const T ly = y - y0, lx = x - x0, hy = 1 - ly, hx = 1 - lx;                           \
              const T a = attn[wi];                                                                 \
              const T cw[4] = {hy * hx, hy * lx, ly * hx, ly * lx};                                 \
              /* d(bilinear)/dy and /dx coefficients of each corner (cuh:115-146) */                \
              const T dy[4] = {-hx, -lx, hx, lx}, dx[4] = {-hy, hy, -ly, ly};                       \
              const int cy[4] = {y0, y0, y0 + 1, y0 + 1}, cx[4] = {x0, x0 + 1, x0, x0 + 1};         \
              T ga = 0, gx = 0, gy = 0;                                                             \
              for (int k = 0; k < 4; ++k) {                                                         \
                if (cy[k] < 0 || cy[k] > H - 1 || cx[k] < 0 || cx[k] > W - 1) continue;             \
                const int64_t off = voff + ((int64_t)cy[k] * W + cx[k]) * M * D;                    \
                for (int c = 0; c < D; ++c) {                                                       \
                  const T vv = value[off + c], tg = g[c];                                           \
                  gvalue[off + c] += cw[k] * a * tg;                                                \
                  ga += tg * cw[k] * vv;                                                            \
                  gx += dx[k] * vv * tg * a;                                                        \
                  gy += dy[k] * vv * tg * a;                                                        \
                }                                                                                   \
              }                                                                                     \
              gattn[wi] = ga; gloc[2 * wi] = W * gx; gloc[2 * wi + 1] = H * gy;                     \
            }                                                                                       \
          }                                                                                         \
        }                                                                                           \
  }

DEFINE_MSDA(float, f32)
DEFINE_MSDA(double, f64)
