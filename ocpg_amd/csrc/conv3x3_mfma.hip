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
#include "conv3x3_halo.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BK = 64, NT = 256;        // BN (64 or 128) is a template parameter: 64 doubles the workgroups of the under-filled shapes
constexpr int LDS_ROW = BK + 8;        // bf16 elements per LDS row (144 B: 16-B aligned, rows 4 banks apart)
constexpr int SEGS = BK / 8;           // 16-B segments per staged row
constexpr int ROWS_PER_PASS = NT / SEGS;
constexpr int A_L = BM / ROWS_PER_PASS;   // 16-B loads per thread per K step (B: BN / ROWS_PER_PASS)
constexpr int NSETS = 3;               // register sets in flight

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

// SPLITK (64-column tiles only; round 4): blockIdx.z takes a contiguous range of the 64-channel chunks of K and the tile leaves as fp32
// partial sums part[z][row][col] (y = that buffer; no affine / ReLU: ocpg_splitk_reduce finishes) -- for convolutions whose GEMM has few
// rows and a long K: the neck's stride-2 level (models/ocpg.py:119-123: 600 output pixels x 256 channels, K = 18 432 = 288 steps on 40
// workgroups without the split).
// BTR (input gradient only; round 4): the weight operand is the convolution's OWN weight [Cout_conv, 3, 3, Cin_conv] (no per-step
// transposed copy: 30 ATen transposes, 0.26 ms per step).  For the input gradient the GEMM's K axis is (tap, output channel) and its N axis
// the input channel, so a K step's B tile is 64 weight ROWS (k) x BN contiguous input channels (n): it is staged as it lies, [k][n], and
// the MFMA fragments -- 8 consecutive k of one n -- are read TRANSPOSED with gfx950's ds_read_b64_tr_b16 (per 16-lane group a 4-row x
// 16-column block, delivered column-major: lane 4q + p supplies row q, columns 4p..4p+3, lane i receives column i of the 4 rows).
template <bool DGRAD, int BN, bool SPLITK = false, bool BTR = false>
__global__ __launch_bounds__(NT) void conv3x3_mfma(const __hip_bfloat16* __restrict__ x, const __hip_bfloat16* __restrict__ w,
                                                   const float* __restrict__ scale, const float* __restrict__ bias, int relu, ConvGeom g,
                                                   __hip_bfloat16* __restrict__ y, __hip_bfloat16* __restrict__ cols,
                                                   const __hip_bfloat16* __restrict__ mask = nullptr) {
  // mask (round 4; same [row][column] layout as y, or null): an element is kept only where mask > 0 -- the input-gradient launch then ALSO
  // does the frozen-BN + ReLU backward of the layer in front (gz = gx * scale[c] * [y_prev > 0], csrc/bn_act.hip's job until round 3)
  static_assert(!SPLITK || BN == 64, "the split-K epilogue is the 64-column one");
  static_assert(!BTR || DGRAD, "the untransposed weight operand is the input gradient's");
  constexpr int BROW = BN + 8;           // BTR: LDS row of the [k][n] B tile (bf16 elements; 16-byte aligned, rows 4 banks apart)
  constexpr int BSEG = BN / 8;           // BTR: 16-byte segments per staged k row
  static_assert(BK * BROW <= BN * LDS_ROW, "the [k][n] image fits the [n][k] one's buffer");
  constexpr int B_L = BN / ROWS_PER_PASS, NJ = BN / 64, WN = BN / 2;      // a wave's tile: 32 rows x WN columns = NJ MFMA tiles
  // 64-column tiles: the four waves split the K STEP instead of the tile (wave w takes the 16-wide slice w of every 64-wide step and
  // accumulates the whole 64 x 64 tile = 2 x 2 MFMA tiles): two A and two B fragments feed four MFMAs, where a 32 x 32 wave tile reads
  // two fragments per MFMA -- the kernel was bound by LDS read bandwidth (32 KB of fragment reads per workgroup and K step against
  // 16 KB of global data), not by the matrix cores or HBM.  The four partial tiles are summed through LDS once, after the K loop.
  constexpr bool KSPLIT = BN == 64;
  constexpr int NACC = KSPLIT ? 4 : NJ;
  __shared__ __attribute__((aligned(16))) short smem[2 * BM * LDS_ROW + 2 * BN * LDS_ROW];
  short (*As)[BM * LDS_ROW] = reinterpret_cast<short (*)[BM * LDS_ROW]>(smem);
  short (*Bs)[BN * LDS_ROW] = reinterpret_cast<short (*)[BN * LDS_ROW]>(smem + 2 * BM * LDS_ROW);
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
  const int C = g.C, ksteps_per_tap = C / BK;
  const int cps = SPLITK ? ksteps_per_tap / (int)gridDim.z : ksteps_per_tap;      // 64-channel chunks of this workgroup (the host made it divide)
  const int c_lo = SPLITK ? (int)blockIdx.z * cps : 0, c_hi = c_lo + cps;
  const int ksteps = 9 * cps;
  const long long wrow_stride = 9LL * C;                   // elements between consecutive GEMM-N rows of the weight operand
  const __hip_bfloat16* wp[B_L];                           // rows past Cout are clamped: their columns are never stored
#pragma unroll
  for (int i = 0; i < B_L; ++i) {
    if constexpr (BTR) {                                     // segment e of the [64 k][BN n] tile: k row e / BSEG, columns 8 (e % BSEG)..
      const int e = tid + i * NT, kr = e / BSEG, ns = e % BSEG;
      wp[i] = w + (long long)kr * (9LL * g.Cout) + min(n0 + ns * 8, g.Cout - 8);      // (a weight row is 9 * Cin_conv long; Cin_conv = g.Cout here)
    } else {
      wp[i] = w + (long long)min(n0 + srow + i * ROWS_PER_PASS, g.Cout - 1) * wrow_stride + sseg * 8;
    }
  }

  // NSETS register sets: the loads of K step s + NSETS + 1 are issued while step s computes, so a load has NSETS MFMA phases
  // (not a fraction of one) to come back -- at ~1 workgroup per CU (300 workgroups for ResNet-101's layer3 shape) nothing
  // else hides it.  Loads are unconditional (clamped address, zeroed at park time): a load inside a branch gets its own
  // s_waitcnt and serialises the batch.
  struct Regs { uint4 a[A_L], b[B_L]; unsigned z; int coff; };     // z bit i: A row i of this set is zero padding; coff: its column in the patch matrix (-1: past the end)
  Regs S0, S1;
  int f_tap = 0, f_c = c_lo;                               // K position of the NEXT fetch (fetches are issued in K order)
  // Fetches are UNCONDITIONAL, also past the last K step (clamped to the last tap: valid memory, never parked into a buffer that is
  // read): with `if (s + 3 < ksteps) fetch(...)` the compiler has to assume the path on which the fetch did not happen, on which the
  // register set about to be parked holds the MOST RECENT loads -- it then waits with vmcnt(3..0), i.e. also for the four loads issued
  // one step ago, and the two-step prefetch distance silently became one (seen in the ISA; 1 830 cycles per K step and wave).
  auto fetch = [&](Regs& R) {
    uint4 (&a)[A_L] = R.a; uint4 (&bq)[B_L] = R.b; unsigned& z = R.z;
    // K order: the nine taps of one 64-channel chunk, then the next chunk -- consecutive steps then read the SAME 128-byte lines of
    // neighbouring pixels (a tap shifts the tile by one pixel or one row), which the L1 still holds; tap-major order re-read every
    // line from L2 nine times, and the launch is bound by the CU's L1-miss bandwidth (10 B/cycle with one workgroup per CU, 19 with three)
    const int tap = f_tap, fc = min(f_c, c_hi - 1);
    const int ky = (tap * 11) >> 5, kx = tap - ky * 3;      // tap / 3 for tap < 9
    const int c0 = fc * BK + sseg * 8;
    R.coff = f_c < c_hi ? tap * C + c0 : -1;
    z = 0;
#pragma unroll
    for (int i = 0; i < A_L; ++i) {
      int ys = 0, xs = 0;
      const bool ok = row_ok[i] && tap_source<DGRAD>(g, py[i], px[i], ky, kx, ys, xs);
      if (!ok) { ys = 0; xs = 0; z |= 1u << i; }
      a[i] = *reinterpret_cast<const uint4*>(x + (((long long)pn[i] * g.Hs + ys) * g.Ws + xs) * C + c0);
    }
    const long long woff = BTR ? (long long)fc * BK * (9LL * g.Cout) + (long long)tap * g.Cout : (long long)tap * C + fc * BK;
#pragma unroll
    for (int i = 0; i < B_L; ++i) bq[i] = *reinterpret_cast<const uint4*>(wp[i] + woff);
    if (++f_tap == 9) { f_tap = 0; ++f_c; }
  };
  auto park = [&](int buf, const Regs& R) {
    const uint4 (&a)[A_L] = R.a; const uint4 (&bq)[B_L] = R.b; const unsigned z = R.z;
#pragma unroll
    for (int i = 0; i < A_L; ++i)
      *reinterpret_cast<uint4*>(&As[buf][(srow + i * ROWS_PER_PASS) * LDS_ROW + sseg * 8]) = ((z >> i) & 1u) ? make_uint4(0u, 0u, 0u, 0u) : a[i];
#pragma unroll
    for (int i = 0; i < B_L; ++i) {
      if constexpr (BTR) {
        const int e = tid + i * NT;
        *reinterpret_cast<uint4*>(&Bs[buf][(e / BSEG) * BROW + (e % BSEG) * 8]) = bq[i];
      } else {
        *reinterpret_cast<uint4*>(&Bs[buf][(srow + i * ROWS_PER_PASS) * LDS_ROW + sseg * 8]) = bq[i];
      }
    }
    if constexpr (!DGRAD) {
      // the gathered A tiles ARE the rows of the patch (im2col) matrix the weight gradient contracts with: the column-0 workgroups
      // write them out on the way (16 bytes per thread and row) and the backward needs no im2col pass
      if (cols && blockIdx.y == 0 && R.coff >= 0) {
#pragma unroll
        for (int i = 0; i < A_L; ++i)
          if (row_ok[i])
            *reinterpret_cast<uint4*>(cols + (m0 + srow + i * ROWS_PER_PASS) * (9LL * C) + R.coff) = ((z >> i) & 1u) ? make_uint4(0u, 0u, 0u, 0u) : a[i];
      }
    }
  };

  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;                // fragment row / k-half of this lane
  // BTR: fragment (8 consecutive k from k0 + 8 fh, column ncol0 + fr) out of the [k][n] tile by two transposing reads
  auto btr = [&](int buf, int k0, int ncol0) -> bf16x8 {
    typedef short s4 __attribute__((ext_vector_type(4)));
    const int li = lane & 15, grp = lane >> 4;
    const short* p = &Bs[buf][(k0 + 8 * fh + (li >> 2)) * BROW + ncol0 + 16 * (grp & 1) + 4 * (li & 3)];
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p + 4 * BROW));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  };
  auto compute = [&](int buf) {
    if constexpr (KSPLIT) {
      static_assert(BK / 16 == NT / 64, "one 16-wide K slice per wave");
      const int ko = wave * 16 + fh * 8;
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&As[buf][fr * LDS_ROW + ko]);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&As[buf][(32 + fr) * LDS_ROW + ko]);
      bf16x8 b0, b1;
      if constexpr (BTR) {
        b0 = btr(buf, wave * 16, 0);
        b1 = btr(buf, wave * 16, 32);
      } else {
        b0 = *reinterpret_cast<const bf16x8*>(&Bs[buf][fr * LDS_ROW + ko]);
        b1 = *reinterpret_cast<const bf16x8*>(&Bs[buf][(32 + fr) * LDS_ROW + ko]);
      }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[3], 0, 0, 0);
      return;
    }
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(&As[buf][(wm * 32 + fr) * LDS_ROW + kk * 16 + fh * 8]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        bf16x8 b;
        if constexpr (BTR) b = btr(buf, kk * 16, wn * WN + j * 32);
        else b = *reinterpret_cast<const bf16x8*>(&Bs[buf][(wn * WN + j * 32 + fr) * LDS_ROW + kk * 16 + fh * 8]);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
      }
    }
  };
  // Three register sets rotate (step s waits in set s % 3), two LDS buffers alternate (step s is parked into buffer s % 2);
  // ksteps = 9 * C / BK is a multiple of 3.  A load has three compute phases to come back: one workgroup alone on a CU measured
  // 0.65 us per K step with two sets (= half a load's round trip under load), and the layer3 launch has only 2.3 workgroups per CU.
  static_assert(NSETS == 3, "the k loop below is written for three register sets");
  Regs S2;
  fetch(S0);
  fetch(S1);
  fetch(S2);
  park(0, S0);
  fetch(S0);                                               // step 3
  __syncthreads();
  auto step = [&](int s_, Regs& nxt) {                     // nxt holds step s_ + 1
    park((s_ & 1) ^ 1, nxt);                               // that buffer was last read in step s_ - 1 (barrier since); past the end: unread
    fetch(nxt);                                            // step s_ + 4 (past the end: a clamped, unused load)
    if (s_ < ksteps) compute(s_ & 1);
    __syncthreads();
  };
  for (int ks = 0; ks < ksteps; ks += 3) {
    step(ks, S1);
    step(ks + 1, S2);
    step(ks + 2, S0);
  }
  if constexpr (KSPLIT) {
    // ---- sum the four waves' partial 64 x 64 tiles through LDS (the K loop ended with a barrier: the staging buffers are free).
    // A tile image is [MFMA tile t = 2 * (row / 32) + col / 32][register r][lane]: 4096 floats, conflict-free for its writer.
    float* red = reinterpret_cast<float*>(smem);
    static_assert(sizeof(smem) >= 2 * 4096 * sizeof(float), "two tile images");
    auto put = [&](float* d) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[(t * 16 + r) * 64 + lane] = acc[t][r];
    };
    auto add = [&](const float* d) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += d[(t * 16 + r) * 64 + lane];
    };
    if (wave >= 2) put(red + (wave - 2) * 4096);
    __syncthreads();
    if (wave < 2) add(red + wave * 4096);
    __syncthreads();
    if (wave == 1) put(red + 4096);
    __syncthreads();
    if (wave == 0) { add(red + 4096); put(red); }
    __syncthreads();
    // ---- epilogue by all 256 threads: thread = (tile row m, 16 consecutive columns) -> one 32-byte store
    const int m = tid >> 2, nq = (tid & 3) * 16;
    const long long row = m0 + m;
    if (row < g.M) {
      const int r = (m & 3) + 4 * ((m & 31) >> 3), h = ((m & 31) >> 2) & 1, t0 = (m >> 5) * 2 + (nq >> 5);
      const float* src = red + (t0 * 16 + r) * 64 + (nq & 31) + 32 * h;
      const int col0 = n0 + nq;
      if constexpr (SPLITK) {                                                   // fp32 partial sums of this K range, 64 bytes per thread
        float* dst = reinterpret_cast<float*>(y) + ((long long)blockIdx.z * g.M + row) * g.Cout + col0;
        if (col0 + 16 <= g.Cout && (g.Cout & 3) == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(dst)[q] = make_float4(src[4 * q], src[4 * q + 1], src[4 * q + 2], src[4 * q + 3]);
        } else {
          for (int j = 0; j < 16 && col0 + j < g.Cout; ++j) dst[j] = src[j];
        }
        return;
      }
      short o[16];
      float sc[16], bi[16];
      const bool full = col0 + 16 <= g.Cout;
      if (full) {                                                             // 16 consecutive floats each: 4 + 4 vector loads, issued together
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 a = scale ? reinterpret_cast<const float4*>(scale + col0)[q] : make_float4(1.f, 1.f, 1.f, 1.f);
          const float4 b = bias ? reinterpret_cast<const float4*>(bias + col0)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
          sc[4 * q] = a.x, sc[4 * q + 1] = a.y, sc[4 * q + 2] = a.z, sc[4 * q + 3] = a.w;
          bi[4 * q] = b.x, bi[4 * q + 1] = b.y, bi[4 * q + 2] = b.z, bi[4 * q + 3] = b.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int col = min(col0 + j, g.Cout - 1);
          sc[j] = scale ? scale[col] : 1.f;
          bi[j] = bias ? bias[col] : 0.f;
        }
      }
      bool keep[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) keep[j] = true;
      if (mask) {
        const __hip_bfloat16* mk = mask + row * g.Cout + col0;
        if (full && (g.Cout & 7) == 0) {
          const bf16x8 lo = *reinterpret_cast<const bf16x8*>(mk), hi = *reinterpret_cast<const bf16x8*>(mk + 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) { keep[j] = lo[j] > 0; keep[8 + j] = hi[j] > 0; }      // bf16 > 0 <=> its bits as a signed short > 0
        } else {
          for (int j = 0; j < 16 && col0 + j < g.Cout; ++j) keep[j] = reinterpret_cast<const short*>(mk)[j] > 0;
        }
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float v = src[j] * sc[j] + bi[j];                                     // frozen-BN affine / conv bias in the epilogue
        if (relu) v = fmaxf(v, 0.f);
        if (!keep[j]) v = 0.f;
        o[j] = (short)__bfloat16_as_ushort(__float2bfloat16(v));
      }
      __hip_bfloat16* dst = y + row * g.Cout + col0;
      if (col0 + 16 <= g.Cout && (g.Cout & 7) == 0) {
        bf16x8 lo, hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) { lo[j] = o[j]; hi[j] = o[8 + j]; }
        *reinterpret_cast<bf16x8*>(dst) = lo;
        *reinterpret_cast<bf16x8*>(dst + 8) = hi;
      } else {
        for (int j = 0; j < 16 && col0 + j < g.Cout; ++j) reinterpret_cast<short*>(dst)[j] = o[j];
      }
    }
    return;
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
      if (row < g.M) {
        if (mask && !(reinterpret_cast<const short*>(mask)[row * g.Cout + col] > 0)) v = 0.f;
        y[row * g.Cout + col] = __float2bfloat16(v);
      }
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

// A/B switch: stride-1 convolutions through the halo-staged kernel (csrc/conv3x3_halo.hip)
inline bool halo_variant() {
  static const bool on = [] { const char* e = std::getenv("OCPG_CONV3X3_HALO"); return e && e[0] == '1'; }();
  return on;
}

}  // namespace

// x [N,H,W,Cin] bf16 channels-last, w [Cout,3,3,Cin] bf16 -> y [N,Ho,Wo,Cout] bf16 = act(conv(x, w) * scale + bias); pad 1, stride 1 / 2;
// scale / bias fp32 [Cout] or NULL (1 / 0): the frozen-BN affine of the ResNet body, or a plain conv bias
// cols (may be NULL): [N*Ho*Wo, 9*Cin] bf16, the patch matrix of x in (ky, kx, ci) order = what ocpg_im2col3x3_nhwc would write
extern "C" int ocpg_conv3x3_mfma_fwd_cols(const void* x, const void* w, const float* scale, const float* bias, int relu, int N, int H, int W,
                                          int Cin, int Cout, int stride, void* y, void* cols, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1006;
  if ((stride != 1 && stride != 2) || Cin % BK != 0) return -2000;
  if (N == 0) return 0;
  if (!x) return -1001;
  if (!w) return -1002;
  if (!y) return -1010;
  if (stride == 1 && !cols && halo_variant() &&
      ocpg_halo::conv3x3_halo((const __hip_bfloat16*)x, (const __hip_bfloat16*)w, scale, bias, relu, 0, N, H, W, Cin, Cout, (__hip_bfloat16*)y,
                              (hipStream_t)stream)) {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
  }
  ConvGeom g;
  g.N = N; g.H = (H - 1) / stride + 1; g.W = (W - 1) / stride + 1; g.C = Cin; g.Hs = H; g.Ws = W; g.Cout = Cout; g.stride = stride;
  g.M = (long long)N * g.H * g.W;
  const unsigned mt = (unsigned)((g.M + BM - 1) / BM);
  if (narrow_tiles(mt, Cout))
    conv3x3_mfma<false, 64><<<dim3(mt, (unsigned)((Cout + 63) / 64)), NT, 0, (hipStream_t)stream>>>(
        (const __hip_bfloat16*)x, (const __hip_bfloat16*)w, scale, bias, relu, g, (__hip_bfloat16*)y, (__hip_bfloat16*)cols);
  else
    conv3x3_mfma<false, 128><<<dim3(mt, (unsigned)((Cout + 127) / 128)), NT, 0, (hipStream_t)stream>>>(
        (const __hip_bfloat16*)x, (const __hip_bfloat16*)w, scale, bias, relu, g, (__hip_bfloat16*)y, (__hip_bfloat16*)cols);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ocpg_conv3x3_mfma_fwd(const void* x, const void* w, const float* scale, const float* bias, int relu, int N, int H, int W,
                                     int Cin, int Cout, int stride, void* y, void* stream) {
  return ocpg_conv3x3_mfma_fwd_cols(x, w, scale, bias, relu, N, H, W, Cin, Cout, stride, y, nullptr, stream);
}

// dy [N,Ho,Wo,Cout] bf16, wT [Cin,3,3,Cout] bf16 (the weight with its channel axes swapped) -> dx [N,H,W,Cin] bf16 fully written;
// _masked: dx = conv_transpose(dy) * scale[ci] where mask_y[n,h,w,ci] > 0, else 0 (scale fp32 [Cin] or NULL = 1; mask_y like dx)
extern "C" int ocpg_conv3x3_mfma_dgrad_masked(const void* dy, const void* wT, const void* mask_y, const float* scale, int N, int H, int W, int Cin,
                                              int Cout, int stride, void* dx, void* stream);
extern "C" int ocpg_conv3x3_mfma_dgrad(const void* dy, const void* wT, int N, int H, int W, int Cin, int Cout, int stride, void* dx,
                                       void* stream) {
  return ocpg_conv3x3_mfma_dgrad_masked(dy, wT, nullptr, nullptr, N, H, W, Cin, Cout, stride, dx, stream);
}

extern "C" int ocpg_conv3x3_mfma_dgrad_masked(const void* dy, const void* wT, const void* mask_y, const float* scale, int N, int H, int W, int Cin,
                                              int Cout, int stride, void* dx, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1006;
  if ((stride != 1 && stride != 2) || Cout % BK != 0) return -2000;
  if (N == 0) return 0;
  if (!dy) return -1001;
  if (!wT) return -1002;
  if (!dx) return -1009;
  if (stride == 1 && !mask_y && !scale && halo_variant() &&
      ocpg_halo::conv3x3_halo((const __hip_bfloat16*)dy, (const __hip_bfloat16*)wT, nullptr, nullptr, 0, 1, N, H, W, Cout, Cin, (__hip_bfloat16*)dx,
                              (hipStream_t)stream)) {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
  }
  ConvGeom g;
  g.N = N; g.H = H; g.W = W; g.C = Cout; g.Hs = (H - 1) / stride + 1; g.Ws = (W - 1) / stride + 1; g.Cout = Cin; g.stride = stride;
  g.M = (long long)N * H * W;
  const unsigned mt = (unsigned)((g.M + BM - 1) / BM);
  if (narrow_tiles(mt, Cin))
    conv3x3_mfma<true, 64><<<dim3(mt, (unsigned)((Cin + 63) / 64)), NT, 0, (hipStream_t)stream>>>(
        (const __hip_bfloat16*)dy, (const __hip_bfloat16*)wT, scale, nullptr, 0, g, (__hip_bfloat16*)dx, nullptr, (const __hip_bfloat16*)mask_y);
  else
    conv3x3_mfma<true, 128><<<dim3(mt, (unsigned)((Cin + 127) / 128)), NT, 0, (hipStream_t)stream>>>(
        (const __hip_bfloat16*)dy, (const __hip_bfloat16*)wT, scale, nullptr, 0, g, (__hip_bfloat16*)dx, nullptr, (const __hip_bfloat16*)mask_y);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// The same from the convolution's OWN weight w [Cout, 3, 3, Cin] (no transposed copy; Cin % 8 == 0).
extern "C" int ocpg_conv3x3_mfma_dgrad_w(const void* dy, const void* w, const void* mask_y, const float* scale, int N, int H, int W, int Cin, int Cout,
                                         int stride, void* dx, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1006;
  if ((stride != 1 && stride != 2) || Cout % BK != 0 || Cin % 8 != 0) return -2000;
  if (N == 0) return 0;
  if (!dy) return -1001;
  if (!w) return -1002;
  if (!dx) return -1011;
  ConvGeom g;
  g.N = N; g.H = H; g.W = W; g.C = Cout; g.Hs = (H - 1) / stride + 1; g.Ws = (W - 1) / stride + 1; g.Cout = Cin; g.stride = stride;
  g.M = (long long)N * H * W;
  const unsigned mt = (unsigned)((g.M + BM - 1) / BM);
  if (narrow_tiles(mt, Cin))
    conv3x3_mfma<true, 64, false, true><<<dim3(mt, (unsigned)((Cin + 63) / 64)), NT, 0, (hipStream_t)stream>>>(
        (const __hip_bfloat16*)dy, (const __hip_bfloat16*)w, scale, nullptr, 0, g, (__hip_bfloat16*)dx, nullptr, (const __hip_bfloat16*)mask_y);
  else
    conv3x3_mfma<true, 128, false, true><<<dim3(mt, (unsigned)((Cin + 127) / 128)), NT, 0, (hipStream_t)stream>>>(
        (const __hip_bfloat16*)dy, (const __hip_bfloat16*)w, scale, nullptr, 0, g, (__hip_bfloat16*)dx, nullptr, (const __hip_bfloat16*)mask_y);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

namespace {

// out[m][c] = act((sum_z part[z][m][c]) * scale[c] + bias[c]), zeroed where mask[m][c] <= 0  (out_dt 0 fp32 / 1 bf16; scale / bias / mask may
// be NULL; act = ReLU when relu != 0): the epilogues of the un-split kernel (conv bias; frozen-BN affine + ReLU; the layer in front's
// BN + ReLU backward) applied by the summing pass
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ part, const float* __restrict__ bias, int splits, long long MC, int Cout,
                                                       void* __restrict__ out, int out_dt, const float* __restrict__ scale = nullptr, int relu = 0,
                                                       const __hip_bfloat16* __restrict__ mask = nullptr) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= MC) return;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = 0; z < splits; ++z) {
    const float4 v = *reinterpret_cast<const float4*>(part + (long long)z * MC + i);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  const int c = (int)(i % Cout);
  if (scale) {
    const float4 sv = *reinterpret_cast<const float4*>(scale + c);
    a.x *= sv.x; a.y *= sv.y; a.z *= sv.z; a.w *= sv.w;
  }
  if (bias) {
    const float4 bv = *reinterpret_cast<const float4*>(bias + c);
    a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w;
  }
  if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
  if (mask) {
    const ushort4 mk = *reinterpret_cast<const ushort4*>(reinterpret_cast<const unsigned short*>(mask) + i);      // bf16 > 0 <=> its bits as a signed short > 0
    if (!((short)mk.x > 0)) a.x = 0.f;
    if (!((short)mk.y > 0)) a.y = 0.f;
    if (!((short)mk.z > 0)) a.z = 0.f;
    if (!((short)mk.w > 0)) a.w = 0.f;
  }
  if (out_dt == 0) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + i) = a;
  } else {
    ushort4 o;
    o.x = __bfloat16_as_ushort(__float2bfloat16(a.x)); o.y = __bfloat16_as_ushort(__float2bfloat16(a.y));
    o.z = __bfloat16_as_ushort(__float2bfloat16(a.z)); o.w = __bfloat16_as_ushort(__float2bfloat16(a.w));
    *reinterpret_cast<ushort4*>(reinterpret_cast<unsigned short*>(out) + i) = o;
  }
}

}  // namespace

// number of K ranges ocpg_conv3x3_mfma_fwd_splitk will use for this shape (a divisor of Cin / 64; 1 = the split does not pay: use the plain entry)
extern "C" int ocpg_conv3x3_mfma_splits(int N, int H, int W, int Cin, int Cout, int stride) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (stride != 1 && stride != 2) || Cin % BK != 0 || Cout % 4 != 0) return 1;
  const long long M = (long long)N * ((H - 1) / stride + 1) * ((W - 1) / stride + 1);
  const long long tiles = ((M + BM - 1) / BM) * ((Cout + 63) / 64);
  const int chunks = Cin / BK;
  int best = 1;
  for (int sgl = 1; sgl <= chunks; ++sgl)           // the largest split that keeps >= 2 chunks (18 K steps) per workgroup and <= ~1024 workgroups
    if (chunks % sgl == 0 && chunks / sgl >= 2 && tiles * sgl <= 1024) best = sgl;
  return tiles < 256 ? best : 1;
}

// part [splits][N*Ho*Wo][Cout] fp32 (scratch, fully written), y [N,Ho,Wo,Cout] = conv(x, w) + bias in out_dt (0 fp32 / 1 bf16);
// cols as in ocpg_conv3x3_mfma_fwd_cols (may be NULL).  splits must be ocpg_conv3x3_mfma_splits(...) (> 1).
extern "C" int ocpg_conv3x3_mfma_fwd_splitk(const void* x, const void* w, const float* bias, int N, int H, int W, int Cin, int Cout, int stride,
                                            int splits, float* part, void* y, int out_dt, void* cols, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1006;
  if ((stride != 1 && stride != 2) || Cin % BK != 0 || Cout % 4 != 0 || splits < 1 || (Cin / BK) % splits != 0 || (out_dt != 0 && out_dt != 1)) return -2000;
  if (N == 0) return 0;
  if (!x) return -1001;
  if (!w) return -1002;
  if (!part) return -1011;
  if (!y) return -1012;
  ConvGeom g;
  g.N = N; g.H = (H - 1) / stride + 1; g.W = (W - 1) / stride + 1; g.C = Cin; g.Hs = H; g.Ws = W; g.Cout = Cout; g.stride = stride;
  g.M = (long long)N * g.H * g.W;
  const unsigned mt = (unsigned)((g.M + BM - 1) / BM);
  conv3x3_mfma<false, 64, true><<<dim3(mt, (unsigned)((Cout + 63) / 64), (unsigned)splits), NT, 0, (hipStream_t)stream>>>(
      (const __hip_bfloat16*)x, (const __hip_bfloat16*)w, nullptr, nullptr, 0, g, reinterpret_cast<__hip_bfloat16*>(part), (__hip_bfloat16*)cols);
  const long long MC = g.M * Cout;
  k_splitk_reduce<<<(unsigned)((MC / 4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(part, bias, splits, MC, Cout, y, out_dt);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ---- split K for the ResNet body (round 4): at 2 (1) clips per step a layer3 / layer4 convolution is 300-600 (150-300) workgroups each
// walking 36-72 K steps one after the other -- a latency chain (the launch takes the SAME 45 us at 1 and at 2 clips per step).  With the
// 64-channel chunks of K split over blockIdx.z the chains are 18 steps and three times as many workgroups overlap; the summing pass
// carries the epilogue.  splits: ocpg_conv3x3_mfma_body_splits (1 = the un-split kernel is the better one).
extern "C" int ocpg_conv3x3_mfma_body_splits(long long M, int ncols, int kchannels) {
  if (M <= 0 || ncols <= 0 || kchannels <= 0 || kchannels % BK != 0 || ncols % 4 != 0) return 1;
  static const int mode = [] { const char* e = std::getenv("OCPG_CONV3X3_SPLITK"); return e ? std::atoi(e) : -1; }();       // 0: never, n > 1: force n where it divides
  const long long tiles = ((M + BM - 1) / BM) * ((ncols + 63) / 64);
  const int chunks = kchannels / BK;
  if (mode == 0) return 1;
  if (mode > 1) return chunks % mode == 0 ? mode : 1;
  if (tiles > 640) return 1;
  int best = 1;
  for (int sgl = 2; sgl <= chunks; ++sgl)
    if (chunks % sgl == 0 && chunks / sgl >= 2 && tiles * sgl <= 1280) best = sgl;
  return best;
}

// y = act(conv(x, w) * scale + shift) as in ocpg_conv3x3_mfma_fwd, K split `splits` ways; part: fp32 scratch [splits][N*Ho*Wo][Cout]
extern "C" int ocpg_conv3x3_mfma_fwd_bn_splitk(const void* x, const void* w, const float* scale, const float* shift, int relu, int N, int H, int W, int Cin,
                                               int Cout, int stride, int splits, float* part, void* y, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1006;
  if ((stride != 1 && stride != 2) || Cin % BK != 0 || Cout % 4 != 0 || splits < 2 || (Cin / BK) % splits != 0) return -2000;
  if (N == 0) return 0;
  if (!x) return -1001;
  if (!w) return -1002;
  if (!part) return -1013;
  if (!y) return -1014;
  ConvGeom g;
  g.N = N; g.H = (H - 1) / stride + 1; g.W = (W - 1) / stride + 1; g.C = Cin; g.Hs = H; g.Ws = W; g.Cout = Cout; g.stride = stride;
  g.M = (long long)N * g.H * g.W;
  const unsigned mt = (unsigned)((g.M + BM - 1) / BM);
  conv3x3_mfma<false, 64, true><<<dim3(mt, (unsigned)((Cout + 63) / 64), (unsigned)splits), NT, 0, (hipStream_t)stream>>>(
      (const __hip_bfloat16*)x, (const __hip_bfloat16*)w, nullptr, nullptr, 0, g, reinterpret_cast<__hip_bfloat16*>(part), nullptr);
  const long long MC = g.M * Cout;
  k_splitk_reduce<<<(unsigned)((MC / 4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(part, shift, splits, MC, Cout, y, 1, scale, relu);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// dx as in ocpg_conv3x3_mfma_dgrad_w (the convolution's own weight; mask_y / scale as there), K (= the output channels) split `splits` ways;
// part: fp32 scratch [splits][N*H*W][Cin]
extern "C" int ocpg_conv3x3_mfma_dgrad_w_splitk(const void* dy, const void* w, const void* mask_y, const float* scale, int N, int H, int W, int Cin,
                                                int Cout, int stride, int splits, float* part, void* dx, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1006;
  if ((stride != 1 && stride != 2) || Cout % BK != 0 || Cin % 8 != 0 || splits < 2 || (Cout / BK) % splits != 0) return -2000;
  if (N == 0) return 0;
  if (!dy) return -1001;
  if (!w) return -1002;
  if (!part) return -1012;
  if (!dx) return -1013;
  ConvGeom g;
  g.N = N; g.H = H; g.W = W; g.C = Cout; g.Hs = (H - 1) / stride + 1; g.Ws = (W - 1) / stride + 1; g.Cout = Cin; g.stride = stride;
  g.M = (long long)N * H * W;
  const unsigned mt = (unsigned)((g.M + BM - 1) / BM);
  conv3x3_mfma<true, 64, true, true><<<dim3(mt, (unsigned)((Cin + 63) / 64), (unsigned)splits), NT, 0, (hipStream_t)stream>>>(
      (const __hip_bfloat16*)dy, (const __hip_bfloat16*)w, nullptr, nullptr, 0, g, reinterpret_cast<__hip_bfloat16*>(part), nullptr);
  const long long MC = g.M * Cin;
  k_splitk_reduce<<<(unsigned)((MC / 4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(part, nullptr, splits, MC, Cin, dx, 1, scale, 0, (const __hip_bfloat16*)mask_y);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
