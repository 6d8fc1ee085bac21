// Device helpers shared by the MSDeformAttn kernels (row kernels in msda.hip, column-tile kernels in msda_col.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace ocpg_dev {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- cross-lane sums inside an 8-lane row group without touching LDS (DPP) ---------------------------------
__device__ __forceinline__ float dpp_xor1(float v) {   // quad_perm [1,0,3,2]
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {   // quad_perm [2,3,0,1]
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_mirror8(float v) {   // row_half_mirror: lane i <-> 7-i inside each 8-lane group
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));
}

// Sum 4 samples x {ga, gx, gy} over the 8 lanes of a row as a reduce-scatter: after it, the lane pair
// (j>>1) == s holds sample s's three totals (12 DPP moves instead of 36 LDS-crossbar shuffles).
// v[s][c] in; returns this lane's sample index, totals in out[0..2].
__device__ __forceinline__ int reduce_scatter_g8_p4(const float (&v)[4][3], int j, float (&out)[3]) {
  const bool hi = (j & 4) != 0;          // step 1: partner 7-j (opposite bit 2); keep samples {0,1} (low half) or {2,3}
  float a[2][3];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float keep = hi ? v[2 + s][c] : v[s][c];
      const float send = hi ? v[s][c] : v[2 + s][c];
      a[s][c] = keep + dpp_mirror8(send);
    }
  const bool mid = (j & 2) != 0;         // step 2: partner j^2; keep sample 0 or 1 of the pair
  float b[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float keep = mid ? a[1][c] : a[0][c];
    const float send = mid ? a[0][c] : a[1][c];
    b[c] = keep + dpp_xor2(send);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) out[c] = b[c] + dpp_xor1(b[c]);   // step 3: all-reduce over the last pair
  return (hi ? 2 : 0) + (mid ? 1 : 0);
}

}  // namespace ocpg_dev
