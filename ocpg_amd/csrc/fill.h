// Zero / byte-pattern fill as an ordinary kernel.  The library never uses hipMemsetAsync: on ROCm 7.x (the libamdhip64 that
// ships with torch 2.10+rocm7.0) a memset node captured into a HIP graph writes a corrupted 16-byte pattern from its second
// launch on (tools/dbg_graph_memset.py), so anything that must start from zero has to be zeroed by a kernel node.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ocpg_fill {

// dst rows of `row_bytes` bytes, `rows` of them `pitch` bytes apart, every byte group of elem (1, 2, 4) bytes = low bytes of value
static __global__ void k_fill(unsigned char* __restrict__ dst, unsigned value, int elem, size_t row_bytes, size_t rows, size_t pitch) {
  const unsigned pat = elem == 1 ? (value & 0xffu) * 0x01010101u : elem == 2 ? (value & 0xffffu) * 0x00010001u : value;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t r = 0; r < rows; ++r) {
    unsigned char* p = dst + r * pitch;
    // head up to 16-byte alignment, 16-byte body, tail: the pattern phase is kept by shifting with the byte offset
    const size_t head = min(row_bytes, (size_t)((16 - ((uintptr_t)p & 15)) & 15));
    const size_t body = (row_bytes - head) / 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < head; i += stride) p[i] = (unsigned char)(pat >> (8 * (i % elem)));
    // after `head` bytes the element phase is head % elem (elem divides 4, rows are elem-aligned in every caller's use)
    const unsigned ph = (unsigned)(head % 4);
    const unsigned rot = ph ? (pat >> (8 * ph)) | (pat << (32 - 8 * ph)) : pat;
    uint4* b = reinterpret_cast<uint4*>(p + head);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < body; i += stride) b[i] = make_uint4(rot, rot, rot, rot);
    const size_t done = head + body * 16;
    for (size_t i = done + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_bytes; i += stride)
      p[i] = (unsigned char)(pat >> (8 * (i % 4)));
  }
}

inline unsigned fill_blocks(size_t row_bytes) {
  const size_t b = (row_bytes / 16 + 255) / 256;
  return (unsigned)(b < 1 ? 1 : b > 2048 ? 2048 : b);
}

inline hipError_t zero_async(void* dst, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  k_fill<<<fill_blocks(bytes), 256, 0, st>>>((unsigned char*)dst, 0u, 1, bytes, 1, 0);
  return hipGetLastError();
}

}  // namespace ocpg_fill
