// Weight gradient of the 3x3 convolutions of the ResNet body (torchvision Bottleneck.conv2 inside models/backbone.py:86-117) on the matrix
// cores, straight from the two channels-last maps (round 4) -- rounds 1-3 wrote the patch (im2col) matrix [pixels, 9 Cin] (csrc/im2col.hip:
// 55 MB per layer3 convolution, 30 launches per step) and ran a row-split hipBLASLt GEMM over it.
//   gw[co][ky][kx][ci] = sum_{n, yo, xo} gz[n, yo, xo, co] * x[n, yo s + ky - 1, xo s + kx - 1, ci]          (padding 1, stride s = 1 | 2)
// GEMM view per tap: M = co, N = ci, K = output pixel.  The reduction axis is the PIXEL, along which neither operand is contiguous
// (both maps are [pixel][channel]) -- exactly the shape gfx950's transposing LDS read serves: a K chunk (a segment of <= 64 / 32 output
// pixels of one output row) is staged as it lies, Gs[pixel][64 co] and Xs[3 input rows][segment + halo][64 ci] (out-of-image rows /
// columns as zeros: that IS the padding), and both MFMA operands -- 8 consecutive pixels of one channel -- come out of
// ds_read_b64_tr_b16 (each lane supplies its own row address, so the stride-2 gather "pixel k -> column k s + kx" costs nothing).
// A workgroup owns a 64 (co) x 64 (ci) tile of ALL NINE taps (the gz fragment of a k step feeds nine MFMAs; 4 waves as 2 x 2, 9 x 16
// accumulator registers each) over a range of output rows (split K: blockIdx.z), and writes its partial tile in bf16 as
// part[z][co][9][ci] -- the layout and dtype of the row-split GEMM it replaces, so the summing stays where it was (the fused gradient
// cast, csrc/multi_cast.hip, or one ATen sum).
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "../../include/ocpg_hip.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256, TM = 64, TN = 64, LROW = 72;       // tile edges (channels), LDS row in bf16 elements (144 B: 16-byte aligned)

// fragment: 8 consecutive k (rows r0 + 8 fh + 0..7 at row pitch `pitch` LDS rows) of column c0 + (lane & 31) of a [row][LROW] tile
__device__ __forceinline__ bf16x8 tr_frag(const short* tile, int r0, int pitch, int c0, int lane) {
  const int li = lane & 15, grp = lane >> 4, fh = lane >> 5;
  const short* p = tile + (r0 + (8 * fh + (li >> 2)) * pitch) * LROW + c0 + 16 * (grp & 1) + 4 * (li & 3);
  const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
  const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p + 4 * pitch * LROW));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// v or zeros, component by component (a select between two uint4 OBJECTS is lowered through their addresses and drags both into scratch)
__device__ __forceinline__ uint4 keep(uint4 v, bool yes) { return make_uint4(yes ? v.x : 0u, yes ? v.y : 0u, yes ? v.z : 0u, yes ? v.w : 0u); }

struct WG {
  int N, H, W, Ho, Wo, Cin, Cout, stride, seg, xw, rows_per_split;
};

// S = stride, SEG = output pixels per chunk (64 / 32), XW = input columns per chunk incl. the halo
template <int S, int SEG>
#ifdef WG_ONE_SET
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_wgrad(
#else
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_wgrad(
#endif
    const __hip_bfloat16* __restrict__ gz, const __hip_bfloat16* __restrict__ x, WG g,
                                                    __hip_bfloat16* __restrict__ part) {
  constexpr int XW = (SEG - 1) * S + 3;
  constexpr int GL = SEG * 8 / NT;                       // 16-byte loads per thread and chunk: gz segment
  constexpr int XL = (3 * XW * 8 + NT - 1) / NT;         // ... and the three input rows
  __shared__ __attribute__((aligned(16))) short Gs[SEG * LROW];
  __shared__ __attribute__((aligned(16))) short Xs[3 * XW * LROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
  const int co0 = blockIdx.x * TM, ci0 = blockIdx.y * TN;
  const long long rows = (long long)g.N * g.Ho;
  const long long r_lo = (long long)blockIdx.z * g.rows_per_split, r_hi = min(rows, r_lo + g.rows_per_split);
  const int nseg = (g.Wo + SEG - 1) / SEG;
  const int nchunks = (int)(r_hi > r_lo ? r_hi - r_lo : 0) * nseg;
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // Global loads of chunk c + 1 are issued before chunk c is multiplied (registers), parked after it: one workgroup per SIMD set (the nine
  // accumulator tiles take 144 registers) leaves nothing else to hide a load behind.  Loads are unconditional on clamped addresses and
  // zeroed at park time (a load inside a branch waits for its own data before the next one is issued).
  struct Regs { uint4 rg[GL], rx[XL]; unsigned zg, zx; };   // z bit i: element i is padding / outside the map -> parked as zeros
  Regs RA, RB;                                           // two chunks in flight: one chunk's MFMAs (~1 us) do not cover an HBM round trip
  // the fetch cursor (image, output row, segment) advances without divisions; past the last chunk it stays on valid memory
  const long long r0c = min(r_lo, rows - 1);
  int fn = (int)(r0c / g.Ho), fy = (int)(r0c - (long long)fn * g.Ho), fs = 0;
  auto fetch = [&](Regs& R) __attribute__((always_inline)) {
    uint4 (&rg)[GL] = R.rg; uint4 (&rx)[XL] = R.rx;
    unsigned zg = 0, zx = 0;
    const int xo0 = fs * SEG, npx = min(SEG, g.Wo - xo0);
    const long long grow = ((long long)fn * g.Ho + fy) * g.Wo;
#pragma unroll
    for (int i = 0; i < GL; ++i) {
      const int e = tid + i * NT, k = e >> 3, sg = e & 7;
      const bool ok = k < npx && co0 + sg * 8 + 8 <= g.Cout;
      zg |= ok ? 0u : 1u << i;
      rg[i] = *reinterpret_cast<const uint4*>(gz + (grow + min(xo0 + k, g.Wo - 1)) * g.Cout + min(co0 + sg * 8, g.Cout - 8));
    }
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int e = min(tid + i * NT, 3 * XW * 8 - 1), sg = e & 7, j = (e >> 3) % XW, ky = (e >> 3) / XW;
      const int yi = fy * S + ky - 1, xi = xo0 * S - 1 + j;
      const bool ok = yi >= 0 && yi < g.H && xi >= 0 && xi < g.W && ci0 + sg * 8 + 8 <= g.Cin;
      zx |= ok ? 0u : 1u << i;
      rx[i] = *reinterpret_cast<const uint4*>(x + (((long long)fn * g.H + min(max(yi, 0), g.H - 1)) * g.W + min(max(xi, 0), g.W - 1)) * g.Cin +
                                              min(ci0 + sg * 8, g.Cin - 8));
    }
    R.zg = zg, R.zx = zx;
    // advance (never past the workgroup's last row: the extra fetches after the last chunk re-read it)
    if (fs + 1 < nseg) ++fs;
    else if ((long long)fn * g.Ho + fy + 1 < r_hi) { fs = 0; if (++fy == g.Ho) { fy = 0; ++fn; } }
  };
  auto park = [&](const Regs& R) __attribute__((always_inline)) {
    const uint4 (&rg)[GL] = R.rg; const uint4 (&rx)[XL] = R.rx;
    const unsigned zg = R.zg, zx = R.zx;
#pragma unroll
    for (int i = 0; i < GL; ++i) {
      const int e = tid + i * NT;
      *reinterpret_cast<uint4*>(Gs + (e >> 3) * LROW + (e & 7) * 8) = keep(rg[i], !((zg >> i) & 1u));
    }
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int e = tid + i * NT;
      if (e < 3 * XW * 8) *reinterpret_cast<uint4*>(Xs + (e >> 3) * LROW + (e & 7) * 8) = keep(rx[i], !((zx >> i) & 1u));
    }
  };
  fetch(RA);
#ifndef WG_ONE_SET
  fetch(RB);
#endif
  int ps = 0;                                            // segment of the chunk being parked
  auto chunk = [&](Regs& R) __attribute__((always_inline)) {
    const int npx = min(SEG, g.Wo - ps * SEG), kp = (npx + 15) & ~15;
    if (++ps == nseg) ps = 0;
    __syncthreads();                                     // the previous chunk's fragment reads are done
#ifndef WG_CUT_LOAD
    park(R);
    __syncthreads();
    fetch(R);                                            // two chunks ahead (unconditional)
#endif
#ifndef WG_CUT_COMPUTE
    for (int k0 = 0; k0 < kp; k0 += 16) {
      const bf16x8 a = tr_frag(Gs, k0, 1, wm * 32, lane);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t - ky * 3;
        const bf16x8 b = tr_frag(Xs + ky * XW * LROW, k0 * S + kx, S, wn * 32, lane);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
      }
    }
#endif
  };
#ifdef WG_ONE_SET
  for (int c = 0; c < nchunks; ++c) chunk(RA);
#else
  for (int c = 0; c < nchunks; c += 2) {
    chunk(RA);
    if (c + 1 < nchunks) chunk(RB);
  }
#endif
  // ---- partial tile -> part[z][co][tap][ci] (bf16).  C/D layout: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int ci = ci0 + wn * 32 + (lane & 31);
  if (ci < g.Cin) {
    __hip_bfloat16* out = part + (long long)blockIdx.z * g.Cout * 9 * g.Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < g.Cout) out[((long long)co * 9 + t) * g.Cin + ci] = __float2bfloat16(acc[t][r]);
      }
  }
}

inline int seg_of(int stride) { return stride == 1 ? 64 : 32; }

}  // namespace

extern "C" {

/* number of row ranges (the leading dimension of `part`) ocpg_conv3x3_mfma_wgrad uses for this shape: one workgroup per CU (every extra
 * range is another [Cout, 9 Cin] partial for the summing pass to read) */
int ocpg_conv3x3_mfma_wgrad_splits(int N, int H, int W, int Cin, int Cout, int stride) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (stride != 1 && stride != 2)) return 0;
  const long long rows = (long long)N * ((H - 1) / stride + 1);
  const long long tiles = (long long)((Cout + TM - 1) / TM) * ((Cin + TN - 1) / TN);
  static const int target = [] { const char* e = std::getenv("OCPG_WGRAD_WGS"); return e && *e ? std::atoi(e) : 256; }();
  long long s = (target + tiles - 1) / tiles;
  if (s > rows) s = rows;
  if (s < 1) s = 1;
  const long long rps = (rows + s - 1) / s;
  return (int)((rows + rps - 1) / rps);
}

int ocpg_conv3x3_mfma_wgrad(const void* gz, const void* x, int N, int H, int W, int Cin, int Cout, int stride, void* part, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1003;
  if ((stride != 1 && stride != 2) || Cin % 8 != 0 || Cout % 8 != 0) return -2000;
  if (N == 0) return 0;
  if (!gz) return -1001;
  if (!x) return -1002;
  if (!part) return -1009;
  WG g;
  g.N = N; g.H = H; g.W = W; g.Ho = (H - 1) / stride + 1; g.Wo = (W - 1) / stride + 1; g.Cin = Cin; g.Cout = Cout; g.stride = stride;
  g.seg = seg_of(stride);
  g.xw = (g.seg - 1) * stride + 3;
  const int splits = ocpg_conv3x3_mfma_wgrad_splits(N, H, W, Cin, Cout, stride);
  const long long rows = (long long)N * g.Ho;
  g.rows_per_split = (int)((rows + splits - 1) / splits);
  const dim3 grid((unsigned)((Cout + TM - 1) / TM), (unsigned)((Cin + TN - 1) / TN), (unsigned)splits);
  if (stride == 1)
    conv3x3_wgrad<1, 64><<<grid, NT, 0, (hipStream_t)stream>>>((const __hip_bfloat16*)gz, (const __hip_bfloat16*)x, g, (__hip_bfloat16*)part);
  else
    conv3x3_wgrad<2, 32><<<grid, NT, 0, (hipStream_t)stream>>>((const __hip_bfloat16*)gz, (const __hip_bfloat16*)x, g, (__hip_bfloat16*)part);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // extern "C"
