// Microbenchmark (round 2): what limits LDS float accumulation on gfx950?
//   f32 add in the default float mode (denormals kept) vs with MODE.fp_denorm(single) = flush, set in-kernel;
//   f64 / u32 for comparison; a non-atomic read-add-write of 16 B per lane; same-address contention (k lanes per address).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

enum Mode { F32 = 0, F32_FLUSH = 1, F64 = 2, U32 = 3, RMW128 = 4, F32_RTN = 5 };

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int share) {
  __shared__ __attribute__((aligned(16))) unsigned char raw[65536];
  float* bf = reinterpret_cast<float*>(raw);
  double* bd = reinterpret_cast<double*>(raw);
  unsigned* bu = reinterpret_cast<unsigned*>(raw);
  float4* b4 = reinterpret_cast<float4*>(raw);
  for (int i = threadIdx.x; i < 16384; i += 256) bf[i] = 0.f;
  __syncthreads();
  if (MODE == F32_FLUSH) __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 0);   // MODE[5:4] = 0: flush f32 denormals
  const int lane = (threadIdx.x / share);            // `share` consecutive lanes hit the same address
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = (lane + u * 256 + it * 7) & 4095;
      if (MODE == F32 || MODE == F32_FLUSH) atomicAdd(&bf[idx], 1.5f);
      else if (MODE == F32_RTN) acc += atomicAdd(&bf[idx], 1.5f);
      else if (MODE == F64) atomicAdd(&bd[idx], 1.5);
      else if (MODE == U32) atomicAdd(&bu[idx], 3u);
      else { float4 v = b4[idx]; v.x += 1.5f; v.y += 1.5f; v.z += 1.5f; v.w += 1.5f; b4[idx] = v; }
    }
  }
  __syncthreads();
  if (MODE == F32_FLUSH) __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 3);
  if (threadIdx.x == 1) out[blockIdx.x] = bf[threadIdx.x] + acc;
}

template <int MODE>
void run(const char* name, int share) {
  float* d;
  hipMalloc(&d, sizeof(float) * 4096);
  const int iters = 2000, blocks = 1024;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 10, share);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, iters, share);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_cu = (double)blocks * 4 * iters * 8 / 256.0;
  float h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  printf("%-10s share %2d: %.3f ms, %6.1f clk (2.4GHz) per wave-instr per CU  (check %g)\n", name, share, ms, ms * 1e-3 * 2.4e9 / per_cu, h);
  hipFree(d);
}

int main() {
  for (int share : {1, 2, 8, 64}) {
    run<F32>("f32", share);
    run<F32_FLUSH>("f32_flush", share);
    run<F32_RTN>("f32_rtn", share);
    run<F64>("f64", share);
    run<U32>("u32", share);
  }
  run<RMW128>("rmw128", 1);
  return 0;
}
