// Halo-staged variant of the 3x3 / stride-1 implicit-GEMM convolution (csrc/conv3x3_halo.hip); declared for conv3x3_mfma.hip's entry points.
#pragma once
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

namespace ocpg_halo {
// y [N,H,W,Cout] = act(conv3x3(x [N,H,W,C], w [Cout][9][C], pad 1, stride 1) * scale + bias); flip: tap t reads w[.][8 - t][.] (the input
// gradient with the channel-swapped weight).  false: shape not served (C, Cout multiples of 64 only).
bool conv3x3_halo(const __hip_bfloat16* x, const __hip_bfloat16* w, const float* scale, const float* bias, int relu, int flip, int N, int H, int W,
                  int C, int Cout, __hip_bfloat16* y, hipStream_t st);
}  // namespace ocpg_halo
