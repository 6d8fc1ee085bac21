// 3x3 convolution of channels-last bf16 maps as an implicit GEMM on the matrix cores (gfx950 MFMA 32x32x16 bf16, fp32
// accumulation) -- forward and input-gradient of the 3x3 convs on the path: torchvision Bottleneck.conv2 inside the ResNet
// body (models/backbone.py:86-117), the stride-2 neck conv (models/ocpg.py:118-126), MSO's refinement convs
// (models/decoder.py:22-46).  Round 1 ran them on MIOpen: at the ResNet-101 layer3 shape (10 frames x 24 x 40, 256 -> 256
// channels, 11.3 GFLOP) its forward measured 90 us and its backward 187 us inside the step (~125 TFLOP/s, 5 % of the bf16
// MFMA peak, plus layout shuffles around its NCHW-minded solvers).
//
// GEMM view: M = N*Ho*Wo output pixels, Ngemm = Cout, K = 9*Cin ordered (ky, kx, ci) = the physical order of a
// channels-last weight [Cout, 3, 3, Cin].  No im2col buffer: the A tile of a K step (one tap, 32 input channels) is gathered
// straight from the shifted input pixels (64 contiguous bytes per row; taps outside the map read as zeros).
//   * workgroup tile 64 (pixels) x 128 (output channels), 4 waves as 2 x 2, each wave 32 x 64 = two 32x32 accumulators;
//   * K step 32: one 16-B global load per thread for A, two for B, register-prefetched one step ahead and parked in a
//     double-buffered LDS image (rows padded to 80 B: the 16 lanes of a ds_read_b128 group then hit 16 disjoint bank
//     quartets), one barrier per K step;
//   * DGRAD = true turns the same kernel into the input gradient: rows are INPUT pixels, the tap's source is the output
//     pixel (yi + 1 - ky) / stride (when divisible and inside), and the weight operand is the [Cin, 3, 3, Cout] transpose.
// The weight gradient stays a dense GEMM over the im2col matrix (csrc/im2col.hip + hipBLASLt): its K dimension is the pixel
// index, along which neither operand is contiguous.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "../../include/ocpg_hip.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BK = 64, NT = 256;        // BN (64 or 128) is a template parameter: 64 doubles the workgroups of the under-filled shapes
constexpr int LDS_ROW = BK + 8;        // bf16 elements per LDS row (144 B: 16-B aligned, rows 4 banks apart)
constexpr int SEGS = BK / 8;           // 16-B segments per staged row
constexpr int ROWS_PER_PASS = NT / SEGS;
constexpr int A_L = BM / ROWS_PER_PASS;   // 16-B loads per thread per K step (B: BN / ROWS_PER_PASS)
constexpr int NSETS = 2;               // register sets in flight

struct ConvGeom {
  int N, H, W, C;          // rows' map: output map (forward) or input map (dgrad); C = channels of the GATHERED operand
  int Hs, Ws;              // the gathered map (forward: input; dgrad: output-gradient map)
  int Cout;                // GEMM N
  int stride;
  long long M;             // N * H * W
};

// source pixel of GEMM row (n, y, x) for tap (ky, kx); false = zero padding
template <bool DGRAD>
__device__ __forceinline__ bool tap_source(const ConvGeom& g, int y, int x, int ky, int kx, int& ys, int& xs) {
  if (!DGRAD) {
    ys = y * g.stride + ky - 1;
    xs = x * g.stride + kx - 1;
    return ys >= 0 && ys < g.Hs && xs >= 0 && xs < g.Ws;
  } else {
    const int ty = y + 1 - ky, tx = x + 1 - kx;
    if (ty < 0 || tx < 0 || (g.stride == 2 && ((ty | tx) & 1))) return false;
    ys = g.stride == 2 ? ty >> 1 : ty;
    xs = g.stride == 2 ? tx >> 1 : tx;
    return ys < g.Hs && xs < g.Ws;
  }
}

template <bool DGRAD, int BN>
__global__ __launch_bounds__(NT) void conv3x3_mfma(const __hip_bfloat16* __restrict__ x, const __hip_bfloat16* __restrict__ w,
                                                   const float* __restrict__ scale, const float* __restrict__ bias, int relu, ConvGeom g,
                                                   __hip_bfloat16* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) short As[2][BM * LDS_ROW];
  constexpr int B_L = BN / ROWS_PER_PASS, NJ = BN / 64, WN = BN / 2;      // a wave's tile: 32 rows x WN columns = NJ MFMA tiles
  __shared__ __attribute__((aligned(16))) short Bs[2][BN * LDS_ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;                 // wave tile: rows wm*32.., cols wn*64..
  const long long m0 = (long long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  // ---- staging identity: 16-B segment sseg of rows srow + i * ROWS_PER_PASS (A: i < A_L, B: i < B_L)
  const int srow = tid / SEGS, sseg = tid % SEGS;
  int pn[A_L], py[A_L], px[A_L];
  bool row_ok[A_L];
  {
    const int hw = g.H * g.W;
#pragma unroll
    for (int i = 0; i < A_L; ++i) {
      const long long mrow = m0 + srow + i * ROWS_PER_PASS;
      row_ok[i] = mrow < g.M;
      const long long mr = row_ok[i] ? mrow : 0;
      pn[i] = (int)(mr / hw);
      const int r = (int)(mr - (long long)pn[i] * hw);
      py[i] = r / g.W;
      px[i] = r - py[i] * g.W;
    }
  }
  const int C = g.C, ksteps_per_tap = C / BK, ksteps = 9 * ksteps_per_tap;
  const long long wrow_stride = 9LL * C;                   // elements between consecutive GEMM-N rows of the weight operand
  const __hip_bfloat16* wp[B_L];                           // rows past Cout are clamped: their columns are never stored
#pragma unroll
  for (int i = 0; i < B_L; ++i) wp[i] = w + (long long)min(n0 + srow + i * ROWS_PER_PASS, g.Cout - 1) * wrow_stride + sseg * 8;

  // NSETS register sets: the loads of K step s + NSETS + 1 are issued while step s computes, so a load has NSETS MFMA phases
  // (not a fraction of one) to come back -- at ~1 workgroup per CU (300 workgroups for ResNet-101's layer3 shape) nothing
  // else hides it.  Loads are unconditional (clamped address, zeroed at park time): a load inside a branch gets its own
  // s_waitcnt and serialises the batch.
  struct Regs { uint4 a[A_L], b[B_L]; unsigned z; };     // z bit i: A row i of this set is zero padding
  Regs S0, S1;
  int f_tap = 0, f_c = 0;                                  // K position of the NEXT fetch (fetches are issued in K order)
  auto fetch = [&](Regs& R) {
    uint4 (&a)[A_L] = R.a; uint4 (&bq)[B_L] = R.b; unsigned& z = R.z;
    const int ky = (f_tap * 11) >> 5, kx = f_tap - ky * 3;  // tap / 3 for tap < 9
    const int c0 = f_c * BK + sseg * 8;
    z = 0;
#pragma unroll
    for (int i = 0; i < A_L; ++i) {
      int ys = 0, xs = 0;
      const bool ok = row_ok[i] && tap_source<DGRAD>(g, py[i], px[i], ky, kx, ys, xs);
      if (!ok) { ys = 0; xs = 0; z |= 1u << i; }
      a[i] = *reinterpret_cast<const uint4*>(x + (((long long)pn[i] * g.Hs + ys) * g.Ws + xs) * C + c0);
    }
    const long long woff = (long long)f_tap * C + f_c * BK;
#pragma unroll
    for (int i = 0; i < B_L; ++i) bq[i] = *reinterpret_cast<const uint4*>(wp[i] + woff);
    if (++f_c == ksteps_per_tap) { f_c = 0; ++f_tap; }
  };
  auto park = [&](int buf, const Regs& R) {
    const uint4 (&a)[A_L] = R.a; const uint4 (&bq)[B_L] = R.b; const unsigned z = R.z;
#pragma unroll
    for (int i = 0; i < A_L; ++i)
      *reinterpret_cast<uint4*>(&As[buf][(srow + i * ROWS_PER_PASS) * LDS_ROW + sseg * 8]) = ((z >> i) & 1u) ? make_uint4(0u, 0u, 0u, 0u) : a[i];
#pragma unroll
    for (int i = 0; i < B_L; ++i) *reinterpret_cast<uint4*>(&Bs[buf][(srow + i * ROWS_PER_PASS) * LDS_ROW + sseg * 8]) = bq[i];
  };

  f32x16 acc[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;                // fragment row / k-half of this lane
  auto compute = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(&As[buf][(wm * 32 + fr) * LDS_ROW + kk * 16 + fh * 8]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Bs[buf][(wn * WN + j * 32 + fr) * LDS_ROW + kk * 16 + fh * 8]);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
      }
    }
  };
  // step s waits in register set s % 2 and is parked into LDS buffer s % 2; ksteps = 9 * C / BK >= 9
  static_assert(NSETS == 2, "the k loop below is written for two register sets");
  fetch(S0);
  fetch(S1);
  park(0, S0);
  fetch(S0);                                               // step 2
  __syncthreads();
  auto step = [&](int s_, Regs& nxt) {                     // nxt holds step s_ + 1
    if (s_ < ksteps) {
      if (s_ + 1 < ksteps) park((s_ & 1) ^ 1, nxt);        // that buffer was last read in step s_ - 1 (barrier since)
      if (s_ + 3 < ksteps) fetch(nxt);                     // step s_ + 3
      compute(s_ & 1);
    }
    __syncthreads();
  };
  for (int ks = 0; ks < ksteps; ks += 2) {
    step(ks, S1);
    step(ks + 1, S0);
  }
  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = n0 + wn * WN + j * 32 + (lane & 31);
    if (col >= g.Cout) continue;
    const float sv = scale ? scale[col] : 1.f, bv = bias ? bias[col] : 0.f;      // frozen-BN affine / conv bias in the epilogue
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      float v = acc[j][r] * sv + bv;
      if (relu) v = fmaxf(v, 0.f);
      if (row < g.M) y[row * g.Cout + col] = __float2bfloat16(v);
    }
  }
}

// 64-column tiles when the 128-column grid would leave the chip under-filled or badly quantised (< 3 workgroups per CU):
// layer3 of ResNet-101 at 10 frames of 384x640 is 150 x 2 = 300 workgroups on 256 CUs, layer4 38 x 4 = 152
inline bool narrow_tiles(unsigned mtiles, int ncols) {
  static const int mode = [] { const char* e = std::getenv("OCPG_CONV3X3_BN"); return e ? std::atoi(e) : 0; }();
  if (mode == 64) return true;
  if (mode == 128) return false;
  return (long long)mtiles * ((ncols + 127) / 128) < 3 * 256;
}

}  // namespace

// x [N,H,W,Cin] bf16 channels-last, w [Cout,3,3,Cin] bf16 -> y [N,Ho,Wo,Cout] bf16 = act(conv(x, w) * scale + bias); pad 1, stride 1 / 2;
// scale / bias fp32 [Cout] or NULL (1 / 0): the frozen-BN affine of the ResNet body, or a plain conv bias
extern "C" int ocpg_conv3x3_mfma_fwd(const void* x, const void* w, const float* scale, const float* bias, int relu, int N, int H, int W,
                                     int Cin, int Cout, int stride, void* y, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1006;
  if ((stride != 1 && stride != 2) || Cin % BK != 0) return -2000;
  if (N == 0) return 0;
  if (!x) return -1001;
  if (!w) return -1002;
  if (!y) return -1010;
  ConvGeom g;
  g.N = N; g.H = (H - 1) / stride + 1; g.W = (W - 1) / stride + 1; g.C = Cin; g.Hs = H; g.Ws = W; g.Cout = Cout; g.stride = stride;
  g.M = (long long)N * g.H * g.W;
  const unsigned mt = (unsigned)((g.M + BM - 1) / BM);
  if (narrow_tiles(mt, Cout))
    conv3x3_mfma<false, 64><<<dim3(mt, (unsigned)((Cout + 63) / 64)), NT, 0, (hipStream_t)stream>>>(
        (const __hip_bfloat16*)x, (const __hip_bfloat16*)w, scale, bias, relu, g, (__hip_bfloat16*)y);
  else
    conv3x3_mfma<false, 128><<<dim3(mt, (unsigned)((Cout + 127) / 128)), NT, 0, (hipStream_t)stream>>>(
        (const __hip_bfloat16*)x, (const __hip_bfloat16*)w, scale, bias, relu, g, (__hip_bfloat16*)y);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// dy [N,Ho,Wo,Cout] bf16, wT [Cin,3,3,Cout] bf16 (the weight with its channel axes swapped) -> dx [N,H,W,Cin] bf16 fully written
extern "C" int ocpg_conv3x3_mfma_dgrad(const void* dy, const void* wT, int N, int H, int W, int Cin, int Cout, int stride, void* dx,
                                       void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1006;
  if ((stride != 1 && stride != 2) || Cout % BK != 0) return -2000;
  if (N == 0) return 0;
  if (!dy) return -1001;
  if (!wT) return -1002;
  if (!dx) return -1009;
  ConvGeom g;
  g.N = N; g.H = H; g.W = W; g.C = Cout; g.Hs = (H - 1) / stride + 1; g.Ws = (W - 1) / stride + 1; g.Cout = Cin; g.stride = stride;
  g.M = (long long)N * H * W;
  const unsigned mt = (unsigned)((g.M + BM - 1) / BM);
  if (narrow_tiles(mt, Cin))
    conv3x3_mfma<true, 64><<<dim3(mt, (unsigned)((Cin + 63) / 64)), NT, 0, (hipStream_t)stream>>>(
        (const __hip_bfloat16*)dy, (const __hip_bfloat16*)wT, nullptr, nullptr, 0, g, (__hip_bfloat16*)dx);
  else
    conv3x3_mfma<true, 128><<<dim3(mt, (unsigned)((Cin + 127) / 128)), NT, 0, (hipStream_t)stream>>>(
        (const __hip_bfloat16*)dy, (const __hip_bfloat16*)wT, nullptr, nullptr, 0, g, (__hip_bfloat16*)dx);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
