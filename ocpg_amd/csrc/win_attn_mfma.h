// Matrix-core variants of the window-attention kernels (csrc/win_attn_mfma.hip); dispatched from csrc/win_attn.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace ocpg_win_mfma {

bool supported(int N, int head_dim, int dtype);      // bf16 / fp16 storage, head_dim 32, the tiles fit the CU's LDS
int fwd(const void* qkv, const float* biasT, const int* region, float scale, int BW, int NW, int N, int H, void* out, float* lse, int dtype,
        hipStream_t st);

// dS: [BW, H, N, N] in the storage dtype, (key, query) order, fully written -- the caller sums it over BW for the bias gradient
// (NULL: the bias needs no gradient)
int bwd(const void* qkv, const float* bias, const float* biasT, const int* region, float scale, int BW, int NW, int N, int H, const void* out,
        const void* dout, const float* lse, void* dqkv, float* Dbuf, void* dS, int dtype, hipStream_t st);

}  // namespace ocpg_win_mfma
