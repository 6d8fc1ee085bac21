// Microbenchmark: LDS atomic add throughput on gfx950 for f32 / u32 / u64 / f64, distinct addresses per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T>
__global__ __launch_bounds__(256) void k(T* out, int iters, int stride) {
  __shared__ T buf[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) buf[i] = 0;
  __syncthreads();
  const int base = (threadIdx.x * stride) & 4095;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) atomicAdd(&buf[(base + u * 256 + it) & 8191], (T)1);
  }
  __syncthreads();
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (T)(t1 - t0);
  if (threadIdx.x == 1) out[1 + blockIdx.x] = buf[threadIdx.x];
}

template <typename T>
void run(const char* name, int stride) {
  T* d;
  hipMalloc(&d, sizeof(T) * 4096);
  int iters = 2000;
  for (int blocks : {256, 1024}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<T><<<blocks, 256>>>(d, 10, stride);
    hipEventRecord(e0);
    k<T><<<blocks, 256>>>(d, iters, stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * 4 * iters * 8;          // wave-instructions
    double per_cu = instr / 256.0;
    printf("%s stride %d blocks %d: %.3f ms, %.1f clk (2.4GHz) per wave-instr per CU\n", name, stride, blocks, ms, ms * 1e-3 * 2.4e9 / per_cu);
  }
  hipFree(d);
}

int main() {
  run<float>("f32", 1);
  run<unsigned int>("u32", 1);
  run<unsigned long long>("u64", 1);
  run<double>("f64", 1);
  run<float>("f32", 33);
  return 0;
}
