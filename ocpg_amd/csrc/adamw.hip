// Gradient clipping + AdamW for MANY parameter tensors in three launches ("fused multi-tensor AdamW + grad-clip
// kernel").  Reference: engine.py:100-106 (`clip_grad_norm_(model.parameters(), max_norm)` then `optimizer.step()`) with main.py:76-99's
// four-group torch.optim.AdamW.  torch runs this as ~10 foreach launches for the norm + one multiply of every gradient by the clip
// coefficient (a full read + write of all gradients) + 12 multi-tensor AdamW launches; here:
//   (1) sqnorm_partials : one partial sum of squares per 2048-element chunk of every gradient (device table of pointers, as multi_cast)
//   (2) norm_finish     : one workgroup folds the partials -> out[0] = total 2-norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))
//   (3) adamw_apply     : p, m, v updated in one pass; the gradient is scaled by the coefficient ON THE FLY (never rewritten);
//                         per-tensor learning rate and weight decay (the optimizer's groups), bias corrections passed by value.
// Arithmetic = torch's _single_tensor_adamw (decoupled weight decay first, then the moment updates, then
// p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps)).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocpg_hip.h"

namespace {

constexpr int CHUNK = 2048;     // elements per workgroup: 256 lanes x 8

__device__ __forceinline__ int find_tensor(const long long* __restrict__ chunk_prefix, int n, long long blk) {
  int lo = 0, hi = n;                       // largest t with chunk_prefix[t] <= blk
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_prefix[mid] <= blk) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void sqnorm_partials(const long long* __restrict__ grads, const long long* __restrict__ numels,
                                                       const long long* __restrict__ chunk_prefix, int n, float* __restrict__ part) {
  const long long blk = blockIdx.x;
  const int t = find_tensor(chunk_prefix, n, blk);
  const float* g = reinterpret_cast<const float*>(grads[t]);
  const long long count = numels[t];
  const long long base = (blk - chunk_prefix[t]) * CHUNK + (long long)threadIdx.x * 8;
  float s = 0.f;
  if (base + 8 <= count && (reinterpret_cast<uintptr_t>(g + base) & 15) == 0) {
    const float4 a = *reinterpret_cast<const float4*>(g + base), b = *reinterpret_cast<const float4*>(g + base + 4);
    s = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
  } else {
    for (long long i = base; i < base + 8 && i < count; ++i) s += g[i] * g[i];
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blk] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

__global__ __launch_bounds__(1024) void norm_finish(const float* __restrict__ part, long long n, float max_norm, float* __restrict__ out) {
  double s = 0.0;
  for (long long i = threadIdx.x; i < n; i += 1024) s += (double)part[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ double ws[16];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += ws[i];
    const float norm = (float)sqrt(t);
    out[0] = norm;
    // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1 (a NaN norm stays NaN)
    const float coef = max_norm / (norm + 1e-6f);
    out[1] = max_norm > 0.f ? (coef < 1.f ? coef : (coef != coef ? coef : 1.f)) : 1.f;
  }
}

// The same under torch.amp.GradScaler (engine.py:98-106 with --amp on fp16): the gradients are still multiplied by `scale`.  One thread
// folds 1 / scale into the clip coefficient, decides whether the step is skipped (a non-finite norm -- some gradient overflowed -- or the
// scaler's own found_inf flag), advances the DEVICE step counter only for steps that are taken (GradScaler skips optimizer.step() as a
// whole, so the bias corrections must not see skipped steps) and leaves the bias corrections of that count for adamw_apply.  Nothing
// here needs the host to know whether the step was taken: no .item() sync.
//   st[0] = norm of the UNSCALED gradients before clipping, st[1] = coefficient for the scaled gradients, st[2] = 1 when skipped,
//   st[3] = steps taken (in/out), st[4] = 1 - beta1^step, st[5] = 1 / sqrt(1 - beta2^step)
__global__ __launch_bounds__(1024) void norm_finish_amp(const float* __restrict__ part, long long n, float max_norm, const float* __restrict__ grad_scale,
                                                        const float* __restrict__ found_inf, double beta1, double beta2, float* __restrict__ st) {
  double s = 0.0;
  for (long long i = threadIdx.x; i < n; i += 1024) s += (double)part[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ double ws[16];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += ws[i];
    const float scaled = (float)sqrt(t);
    const float inv = grad_scale ? 1.f / grad_scale[0] : 1.f;
    const float norm = scaled * inv;
    const bool skip = !(fabsf(scaled) <= 3.402823466e38f) || (found_inf && found_inf[0] != 0.f);
    const float coef = max_norm / (norm + 1e-6f);
    st[0] = norm;
    st[1] = (max_norm > 0.f ? (coef < 1.f ? coef : 1.f) : 1.f) * inv;
    st[2] = skip ? 1.f : 0.f;
    const float step = st[3] + (skip ? 0.f : 1.f);
    st[3] = step;
    st[4] = (float)(1.0 - pow(beta1, (double)step));
    st[5] = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)step)));
  }
}

constexpr int GROUP = 4;        // table chunks per workgroup: 4 x (4 reads + 3 writes) of 32 B per lane in flight

__global__ __launch_bounds__(256) void adamw_apply(const long long* __restrict__ params, const long long* __restrict__ grads,
                                                   const long long* __restrict__ exp_avg, const long long* __restrict__ exp_avg_sq,
                                                   const long long* __restrict__ numels, const long long* __restrict__ chunk_prefix,
                                                   const float* __restrict__ lr, const float* __restrict__ wd, int n, long long total_chunks,
                                                   const float* __restrict__ clip, float om_beta1, float beta2, float om_beta2, float eps, float bc1,
                                                   float rsqrt_bc2, int amp) {
  const long long blk0 = (long long)blockIdx.x * GROUP;
  const float coef = clip ? clip[1] : 1.f;
  if (amp) {                                // (norm_finish_amp's verdict and bias corrections; uniform over the grid)
    if (clip[2] != 0.f) return;
    bc1 = clip[4], rsqrt_bc2 = clip[5];
  }
  int t = find_tensor(chunk_prefix, n, blk0);
  float pv[GROUP][8], gv[GROUP][8], mv[GROUP][8], vv[GROUP][8];
  float* pp[GROUP];
  float* mp[GROUP];
  float* vp[GROUP];
  float decay[GROUP], step_size[GROUP];
  int cnt[GROUP];
  bool wide[GROUP];
  // ---- all loads of the group first
#pragma unroll
  for (int q = 0; q < GROUP; ++q) {
    const long long blk = blk0 + q;
    cnt[q] = 0;
    wide[q] = false;
    if (blk >= total_chunks) continue;
    while (t + 1 < n && chunk_prefix[t + 1] <= blk) ++t;        // chunks of one tensor are consecutive: at most a step or two
    float* p = reinterpret_cast<float*>(params[t]);
    const float* g = reinterpret_cast<const float*>(grads[t]);
    float* m = reinterpret_cast<float*>(exp_avg[t]);
    float* v = reinterpret_cast<float*>(exp_avg_sq[t]);
    const long long count = numels[t];
    const float lr_t = lr[t];
    decay[q] = 1.f - lr_t * wd[t];
    step_size[q] = lr_t / bc1;
    const long long base = (blk - chunk_prefix[t]) * CHUNK + (long long)threadIdx.x * 8;
    if (base >= count) continue;
    pp[q] = p + base, mp[q] = m + base, vp[q] = v + base;
    wide[q] = base + 8 <= count && ((reinterpret_cast<uintptr_t>(p + base) | reinterpret_cast<uintptr_t>(g + base) |
                                     reinterpret_cast<uintptr_t>(m + base) | reinterpret_cast<uintptr_t>(v + base)) & 15) == 0;
    cnt[q] = wide[q] ? 8 : (int)((count - base) < 8 ? (count - base) : 8);
    if (wide[q]) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        *reinterpret_cast<float4*>(pv[q] + 4 * h) = *reinterpret_cast<const float4*>(p + base + 4 * h);
        *reinterpret_cast<float4*>(gv[q] + 4 * h) = *reinterpret_cast<const float4*>(g + base + 4 * h);
        *reinterpret_cast<float4*>(mv[q] + 4 * h) = *reinterpret_cast<const float4*>(m + base + 4 * h);
        *reinterpret_cast<float4*>(vv[q] + 4 * h) = *reinterpret_cast<const float4*>(v + base + 4 * h);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < cnt[q]) { pv[q][i] = p[base + i]; gv[q][i] = g[base + i]; mv[q][i] = m[base + i]; vv[q][i] = v[base + i]; }
    }
  }
#pragma unroll
  for (int q = 0; q < GROUP; ++q) {
    if (cnt[q] == 0) continue;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < cnt[q]) {
        const float gi = gv[q][i] * coef;
        float pi = pv[q][i] * decay[q];
        // (1 - beta) comes from the host in double precision, as torch's Python scalars do: 1.f - 0.999f is 4.7e-5 off 0.001
        const float mi = mv[q][i] + (gi - mv[q][i]) * om_beta1;            // lerp form, as torch: exp_avg.lerp_(grad, 1 - beta1)
        const float vi = vv[q][i] * beta2 + gi * gi * om_beta2;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
        pi -= step_size[q] * (mi / denom);
        pv[q][i] = pi; mv[q][i] = mi; vv[q][i] = vi;
      }
    }
    if (wide[q]) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        *reinterpret_cast<float4*>(pp[q] + 4 * h) = *reinterpret_cast<float4*>(pv[q] + 4 * h);
        *reinterpret_cast<float4*>(mp[q] + 4 * h) = *reinterpret_cast<float4*>(mv[q] + 4 * h);
        *reinterpret_cast<float4*>(vp[q] + 4 * h) = *reinterpret_cast<float4*>(vv[q] + 4 * h);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < cnt[q]) { pp[q][i] = pv[q][i]; mp[q][i] = mv[q][i]; vp[q][i] = vv[q][i]; }
    }
  }
}

inline int status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace

extern "C" {

int ocpg_grad_norm_clip(const long long* grads, const long long* numels, const long long* chunk_prefix, int n, long long total_chunks,
                        float max_norm, float* partials, float* norm_and_coef, void* stream) {
  if (n < 0 || total_chunks < 0) return -1004;
  if (!norm_and_coef) return -1008;
  hipStream_t st = (hipStream_t)stream;
  if (n > 0 && total_chunks > 0) {
    if (!grads || !numels || !chunk_prefix) return -1001;
    if (!partials) return -1007;
    if (total_chunks > 2147483647LL) return -1005;
    sqnorm_partials<<<(unsigned)total_chunks, 256, 0, st>>>(grads, numels, chunk_prefix, n, partials);
  }
  norm_finish<<<1, 1024, 0, st>>>(partials, n > 0 ? total_chunks : 0, max_norm, norm_and_coef);
  return status();
}

int ocpg_adamw_step(const long long* params, const long long* grads, const long long* exp_avg, const long long* exp_avg_sq,
                    const long long* numels, const long long* chunk_prefix, const float* lr, const float* weight_decay, int n,
                    long long total_chunks, const float* norm_and_coef, double beta1, double beta2, double eps, long long step, void* stream) {
  if (n < 0 || total_chunks < 0 || step < 1) return -1009;
  if (n == 0 || total_chunks == 0) return 0;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numels || !chunk_prefix || !lr || !weight_decay) return -1001;
  if (total_chunks > 2147483647LL) return -1010;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  adamw_apply<<<(unsigned)((total_chunks + GROUP - 1) / GROUP), 256, 0, (hipStream_t)stream>>>(params, grads, exp_avg, exp_avg_sq, numels, chunk_prefix, lr,
                                                                                              weight_decay, n, total_chunks, norm_and_coef, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)bc1,
                                                                       (float)(1.0 / sqrt(bc2)), 0);
  return status();
}

int ocpg_grad_norm_clip_amp(const long long* grads, const long long* numels, const long long* chunk_prefix, int n, long long total_chunks,
                            float max_norm, float* partials, const float* grad_scale, const float* found_inf, double beta1, double beta2,
                            float* amp_state, void* stream) {
  if (n < 0 || total_chunks < 0) return -1004;
  if (!amp_state) return -1008;
  hipStream_t st = (hipStream_t)stream;
  if (n > 0 && total_chunks > 0) {
    if (!grads || !numels || !chunk_prefix || !partials) return -1001;
    if (total_chunks > 2147483647LL) return -1005;
    sqnorm_partials<<<(unsigned)total_chunks, 256, 0, st>>>(grads, numels, chunk_prefix, n, partials);
  }
  norm_finish_amp<<<1, 1024, 0, st>>>(partials, n > 0 ? total_chunks : 0, max_norm, grad_scale, found_inf, beta1, beta2, amp_state);
  return status();
}

int ocpg_adamw_step_amp(const long long* params, const long long* grads, const long long* exp_avg, const long long* exp_avg_sq,
                        const long long* numels, const long long* chunk_prefix, const float* lr, const float* weight_decay, int n,
                        long long total_chunks, const float* amp_state, double beta1, double beta2, double eps, void* stream) {
  if (n < 0 || total_chunks < 0) return -1009;
  if (n == 0 || total_chunks == 0) return 0;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numels || !chunk_prefix || !lr || !weight_decay || !amp_state) return -1001;
  if (total_chunks > 2147483647LL) return -1010;
  adamw_apply<<<(unsigned)((total_chunks + GROUP - 1) / GROUP), 256, 0, (hipStream_t)stream>>>(params, grads, exp_avg, exp_avg_sq, numels, chunk_prefix, lr,
                                                                                              weight_decay, n, total_chunks, amp_state, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, 1.f,
                                                                                              1.f, 1);
  return status();
}

}  // extern "C"
