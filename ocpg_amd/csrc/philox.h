// Counter-based generator shared by the kernels that apply dropout without storing a mask (fused_ln.hip, attn_smallk.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ocpg_dev {

// ---- Philox-4x32-10 ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox(uint2 key, uint4 c) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t lo0 = 0xD2511F53u * c.x, hi0 = __umulhi(0xD2511F53u, c.x);
    const uint32_t lo1 = 0xCD9E8D57u * c.z, hi1 = __umulhi(0xCD9E8D57u, c.z);
    c = make_uint4(hi1 ^ c.y ^ key.x, lo1, hi0 ^ c.w ^ key.y, lo0);
    key.x += 0x9E3779B9u;
    key.y += 0xBB67AE85u;
  }
  return c;
}

}  // namespace ocpg_dev
