// MSO mask refinement (reference models/decoder.py:14-46): every convolution of the block as hand-written MFMA kernels on
// channels-last maps.  All of MSO's convolutions are 3x3 / padding 1 with at most 16 output channels, so one tile shape serves them:
//
//   k_conv_n16   out[n,y,x,16 z..16 z+15] = sum_{tap,c} w[16 z + o][tap][c] * act(in[n, y+dy, x+dx, c])  (+ bias, + addend[n % NA],
//                masked by mask > 0, + residual).  One workgroup = an 8 x 16 pixel tile x one group of 16 output channels; the
//                10 x 18 halo tile is staged through LDS 64 (fp32: 32) input channels at a time together with the 16 x 9 x 64
//                weight slice (the next stage's global loads are in flight while the MFMAs of this one run); MFMA with A = weights
//                (rows = output channels), B = pixels (columns), so a lane ends up with 4 consecutive output channels of one pixel
//                = one 16-byte store.  v_mfma_f32_16x16x32 on the wide (feature) inputs, 16x16x16 on the 16-channel mask path, whose
//                variant keeps 12 KB of LDS so that 4 workgroups share a CU (the mask-path launches are latency-bound).
//                Serves the forward convolutions (528/272 -> 16 split into the layer-shared feature half and the 16 -> 16 mask half,
//                16 -> 16, 16 -> 1) AND their input gradients (the same convolution with flipped, transposed weights; z runs over
//                the groups of 16 input channels; mask = the forward input of the ReLU in front of the convolution).
//   k_wgrad_n16  gw[o][tap][c] = sum_{n,y,x} g[n,y,x,o] * act(x[n, y+dy, x+dx, c]): the reduction axis is the PIXEL, so the
//                fragments take 4 consecutive pixels of one channel (4 16-bit LDS reads); a workgroup owns (a band of rows of one
//                image) x (64 / 32 / 16 input channels), its 4 waves split (channel group) x (tile rows), keep their 9 x 16 x 16
//                tiles in registers over the whole band and add them once (float atomics: <= 2.2 M per launch) to the weight gradient
//                [o][tap][c] -- or write them as a partial product per band -- together with the band's sum of g (the bias gradient).
//   k_up_fwd/bwd bilinear resize (align_corners=False, reference decoder.py:38) of a channels-last fp32 map; the backward is a
//                gather over the <= 6 x 6 output pixels that can touch an input pixel (no atomics).
// Compute type = the autocast dtype (bf16 / fp16 operands, fp32 accumulation, as the reference's autocast convolutions) or fp32
// (v_mfma_f32_16x16x4_f32, exact fp32 products).  HBM-bound by design: the feature maps (78 MB at stride 4) are read once (+ halo).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/ocpg_hip.h"

namespace {

typedef short s4v __attribute__((ext_vector_type(4)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));
typedef __bf16 b2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

enum { F32 = 0, BF16 = 1, F16 = 2 };          // dtype codes of the C ABI (include/ocpg_hip.h)

__device__ __forceinline__ float bf16_to_f(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ float f16_to_f(unsigned short v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ unsigned pack_bf16(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f2v){a, b}, b2v)); }
__device__ __forceinline__ unsigned pack_f16(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f2v){a, b}, h2v)); }
__device__ __forceinline__ unsigned short f_to_bf16(float x) { return (unsigned short)(pack_bf16(x, 0.f) & 0xffffu); }
__device__ __forceinline__ unsigned short f_to_f16(float x) { return (unsigned short)(pack_f16(x, 0.f) & 0xffffu); }

// compute types: E = the LDS element; fragments are KS consecutive k (input channels / pixels) split over the 4 lane groups
template <int DT> struct Cmp;
template <> struct Cmp<BF16> {
  typedef unsigned short E;
  static __device__ __forceinline__ E cvt(float x) { return f_to_bf16(x); }
  static __device__ __forceinline__ unsigned pack(float a, float b) { return pack_bf16(a, b); }
  static __device__ __forceinline__ f4v mma16(s4v a, s4v b, f4v c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f4v mma32(uint4 a, uint4 b, f4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8v, a), __builtin_bit_cast(b8v, b), c, 0, 0, 0);
  }
};
template <> struct Cmp<F16> {
  typedef unsigned short E;
  static __device__ __forceinline__ E cvt(float x) { return f_to_f16(x); }
  static __device__ __forceinline__ unsigned pack(float a, float b) { return pack_f16(a, b); }
  static __device__ __forceinline__ f4v mma16(s4v a, s4v b, f4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4v, a), __builtin_bit_cast(h4v, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f4v mma32(uint4 a, uint4 b, f4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
  }
};
template <> struct Cmp<F32> {
  typedef float E;
  static __device__ __forceinline__ E cvt(float x) { return x; }
  static __device__ __forceinline__ unsigned pack(float, float) { return 0u; }
  // lane group g holds k = 4 g + j of the 16-wide step for BOTH operands: four 16x16x4 products, j-th element with j-th element
  static __device__ __forceinline__ f4v mma16(f4v a, f4v b, f4v c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
    return c;
  }
};

// one k-step of KS channels: a / b point at the lane's first element (KS / 4 consecutive elements per lane)
template <int DT, int KS> struct Step;
template <int DT> struct Step<DT, 32> {
  static __device__ __forceinline__ f4v run(const unsigned short* a, const unsigned short* b, f4v c) {
    return Cmp<DT>::mma32(*(const uint4*)a, *(const uint4*)b, c);
  }
};
template <int DT> struct Step<DT, 16> {
  static __device__ __forceinline__ f4v run(const unsigned short* a, const unsigned short* b, f4v c) {
    return Cmp<DT>::mma16(*(const s4v*)a, *(const s4v*)b, c);
  }
};
template <> struct Step<F32, 16> {
  static __device__ __forceinline__ f4v run(const float* a, const float* b, f4v c) { return Cmp<F32>::mma16(*(const f4v*)a, *(const f4v*)b, c); }
};

__device__ __forceinline__ float load1(const void* p, int dt, size_t i) {
  if (dt == F32) return ((const float*)p)[i];
  const unsigned short v = ((const unsigned short*)p)[i];
  return dt == BF16 ? bf16_to_f(v) : f16_to_f(v);
}

// ---- staging of 8 consecutive channels: global -> registers (raw) -> LDS (compute type) --------------------------------------
template <bool WIDE> struct Raw8;          // WIDE: 8 x fp32 (two 16-byte loads), else 8 x 16 bit (one)
template <> struct Raw8<true> {
  uint4 a, b;
};
template <> struct Raw8<false> {
  uint4 a;
};

__device__ __forceinline__ unsigned relu16x2(unsigned w) {            // two 16-bit floats: negative (sign bit set) -> +0
  const unsigned neg = (w >> 15) & 0x00010001u;
  return w & ~(neg * 0xffffu);
}

// vector load of 8 valid, aligned elements
template <bool WIDE> __device__ __forceinline__ Raw8<WIDE> load_raw8(const void* p, size_t i) {
  Raw8<WIDE> r;
  if constexpr (WIDE) {
    r.a = *(const uint4*)((const float*)p + i);
    r.b = *(const uint4*)((const float*)p + i + 4);
  } else {
    r.a = *(const uint4*)((const unsigned short*)p + i);
  }
  return r;
}
template <bool WIDE> __device__ __forceinline__ void zero_raw8(Raw8<WIDE>& r) {
  r.a = make_uint4(0u, 0u, 0u, 0u);
  if constexpr (WIDE) r.b = r.a;
}

// generic (scalar, any dtype pair, partial) path
__device__ __forceinline__ void load8_slow(const void* p, int dt, size_t i, int nvalid, float v[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = k < nvalid ? load1(p, dt, i + k) : 0.f;
}

template <int DT> __device__ __forceinline__ void store8_f(typename Cmp<DT>::E* d, const float v[8]) {
  if constexpr (DT == F32) {
    *(float4*)d = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    *(uint2*)d = make_uint2(Cmp<DT>::pack(v[0], v[1]), Cmp<DT>::pack(v[2], v[3]));
    *(uint2*)(d + 4) = make_uint2(Cmp<DT>::pack(v[4], v[5]), Cmp<DT>::pack(v[6], v[7]));
  }
}

// raw registers -> LDS in the compute type: a narrow source IS in the compute type, a wide one is fp32
template <int DT, bool WIDE> __device__ __forceinline__ void store_raw8(typename Cmp<DT>::E* d, const Raw8<WIDE>& r, bool relu) {
  if constexpr (!WIDE) {
    uint4 a = r.a;
    if (relu) a = make_uint4(relu16x2(a.x), relu16x2(a.y), relu16x2(a.z), relu16x2(a.w));
    *(uint2*)d = make_uint2(a.x, a.y);
    *(uint2*)(d + 4) = make_uint2(a.z, a.w);
  } else {
    float v[8] = {__uint_as_float(r.a.x), __uint_as_float(r.a.y), __uint_as_float(r.a.z), __uint_as_float(r.a.w),
                  __uint_as_float(r.b.x), __uint_as_float(r.b.y), __uint_as_float(r.b.z), __uint_as_float(r.b.w)};
    if (relu) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    store8_f<DT>(d, v);
  }
}

template <int DT> __device__ __forceinline__ void zero8(typename Cmp<DT>::E* d) {
  const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  store8_f<DT>(d, z);
}

constexpr int TR = 8, TC = 16;                 // output tile: rows x columns
constexpr int HR = TR + 2, HC = TC + 2;        // halo tile
constexpr int NPIX = HR * HC;                  // 180

struct ConvP {
  const void* in;          // [NB, H, W, C]
  const void* w;           // [co_total][9][C]
  const float* bias;       // [co_total] | null
  const float* addend;     // [NA, H, W, co_total] | null   (added to image n % NA)
  const void* mask;        // [NB, H, W, co_total] | null   (result kept where mask > 0)
  const float* residual;   // [NB, H, W, co_total] | null
  void* out;               // [NB, H, W, co_total]
  int in_dt, w_dt, mask_dt, out_dt, relu_in;
  int NB, H, W, C, co_total, NA, tiles_x, gpw;     // gpw: groups of 16 output channels per workgroup
};

// CK input channels per LDS stage, KS channels per MFMA step; XW: the input is fp32 (else: in the compute type) on the vector path;
// weights on the vector path are in the compute type.  Occupancy: the 16-channel variant is latency-bound (4 workgroups per CU: 128 VGPRs; 6 would spill),
// the wide one is held to 3 by its 45 KB of LDS.
template <int DT, int CK, int KS, bool XW> __global__ __launch_bounds__(256, (CK == 16 ? 4 : 3)) void k_conv_n16(const ConvP p) {
  typedef Cmp<DT> CT;
  typedef typename CT::E E;
  constexpr bool WW = DT == F32;
  constexpr int PAD = (DT != F32 && KS == 32) ? 8 : 4;      // fragment reads stay aligned and spread over the banks
  constexpr int XS = CK + PAD;                 // elements per pixel of the halo tile
  constexpr int WS = 9 * CK + PAD;             // elements per output channel of the weight slice
  constexpr int XI = (NPIX * (CK / 8) + 255) / 256, WI = (16 * 9 * (CK / 8) + 255) / 256;
  __shared__ __attribute__((aligned(16))) E xl[NPIX * XS];
  __shared__ __attribute__((aligned(16))) E wl[16 * WS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, g = lane >> 4;
  const int ty0 = (blockIdx.x / p.tiles_x) * TR, tx0 = (blockIdx.x % p.tiles_x) * TC;
  const int n = blockIdx.y;
  const int C = p.C;
  // the vector path: whole groups of 8 channels, sources in fp32 or in the compute type
  const bool xvec = (C % 8) == 0 && p.in_dt == (XW ? F32 : DT);
  const bool wvec = (C % 8) == 0 && p.w_dt == DT;
  const bool one_chunk = C <= CK;              // the halo tile is staged once and shared by the groups of this workgroup
  const size_t img = (size_t)n * p.H * p.W;
  const int ngroups = (p.co_total + 15) / 16;

  for (int zz = 0; zz < p.gpw; ++zz) {
    const int z = blockIdx.z * p.gpw + zz;
    if (z >= ngroups) break;
    f4v acc[2];
    acc[0] = (f4v){0.f, 0.f, 0.f, 0.f};
    acc[1] = acc[0];
    const bool stage_x = !(one_chunk && zz > 0);

    Raw8<XW> xr[XI];
    Raw8<WW> wr[WI];
    // staged width of a stage in units of 8 channels (a compile-time constant when a stage is one MFMA step)
    auto width8 = [&](int c0) { return CK == KS ? KS / 8 : min(CK / KS, (C - c0 + KS - 1) / KS) * (KS / 8); };
    // issue the global loads of one stage (vector paths only; the slow paths load in `commit`)
    auto fetch = [&](int c0) {
      const int cw8 = width8(c0);
      if (stage_x && xvec) {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
          const int it = tid + i * 256;
          const int px = it / cw8, c8 = (it % cw8) * 8;
          const int y = ty0 + px / HC - 1, x = tx0 + px % HC - 1;
          if (it < NPIX * cw8 && y >= 0 && y < p.H && x >= 0 && x < p.W && c0 + c8 < C)
            xr[i] = load_raw8<XW>(p.in, (img + (size_t)y * p.W + x) * C + c0 + c8);
          else
            zero_raw8(xr[i]);
        }
      }
      if (wvec) {
#pragma unroll
        for (int i = 0; i < WI; ++i) {
          const int it = tid + i * 256;
          const int c8 = (it % cw8) * 8, ot = it / cw8, o = ot / 9, tap = ot % 9;
          if (it < 16 * 9 * cw8 && z * 16 + o < p.co_total && c0 + c8 < C)
            wr[i] = load_raw8<WW>(p.w, ((size_t)(z * 16 + o) * 9 + tap) * C + c0 + c8);
          else
            zero_raw8(wr[i]);
        }
      }
    };
    // registers (or, slow paths, memory) -> LDS
    auto commit = [&](int c0) {
      const int cw8 = width8(c0);
      if (stage_x) {
        if (xvec) {
#pragma unroll
          for (int i = 0; i < XI; ++i) {
            const int it = tid + i * 256;
            if (it < NPIX * cw8) store_raw8<DT, XW>(xl + (it / cw8) * XS + (it % cw8) * 8, xr[i], p.relu_in != 0);
          }
        } else {
          for (int it = tid; it < NPIX * cw8; it += 256) {
            const int px = it / cw8, c8 = (it % cw8) * 8;
            const int y = ty0 + px / HC - 1, x = tx0 + px % HC - 1;
            const int nv = min(8, C - c0 - c8);
            float v[8];
            if (y >= 0 && y < p.H && x >= 0 && x < p.W && nv > 0) {
              load8_slow(p.in, p.in_dt, (img + (size_t)y * p.W + x) * C + c0 + c8, nv, v);
              if (p.relu_in) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
              }
              store8_f<DT>(xl + px * XS + c8, v);
            } else {
              zero8<DT>(xl + px * XS + c8);
            }
          }
        }
      }
      if (wvec) {
#pragma unroll
        for (int i = 0; i < WI; ++i) {
          const int it = tid + i * 256;
          if (it < 16 * 9 * cw8) {
            const int c8 = (it % cw8) * 8, ot = it / cw8;
            store_raw8<DT, WW>(wl + (ot / 9) * WS + (ot % 9) * CK + c8, wr[i], false);
          }
        }
      } else {
        for (int it = tid; it < 16 * 9 * cw8; it += 256) {
          const int c8 = (it % cw8) * 8, ot = it / cw8, o = ot / 9, tap = ot % 9;
          const int nv = min(8, C - c0 - c8);
          float v[8];
          if (z * 16 + o < p.co_total && nv > 0) {
            load8_slow(p.w, p.w_dt, ((size_t)(z * 16 + o) * 9 + tap) * C + c0 + c8, nv, v);
            store8_f<DT>(wl + o * WS + tap * CK + c8, v);
          } else {
            zero8<DT>(wl + o * WS + tap * CK + c8);
          }
        }
      }
    };

    fetch(0);
    for (int c0 = 0; c0 < C; c0 += CK) {
      __syncthreads();                         // the previous stage's fragments have been read
      commit(c0);
      __syncthreads();
      if (c0 + CK < C) fetch(c0 + CK);         // in flight while this stage's MFMAs run
      const int nks = min(CK / KS, (C - c0 + KS - 1) / KS);
      for (int ks = 0; ks < nks; ++ks) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int dy = tap / 3, dx = tap % 3;
          const E* a = wl + col * WS + tap * CK + ks * KS + g * (KS / 4);
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const E* b = xl + ((2 * wave + r + dy) * HC + col + dx) * XS + ks * KS + g * (KS / 4);
            acc[r] = Step<DT, KS>::run(a, b, acc[r]);
          }
        }
      }
    }

    // ---- epilogue: lane = (pixel column `col`, output channels 16 z + 4 g .. + 3) of rows 2 wave, 2 wave + 1
    const int x = tx0 + col;
    const int o0 = z * 16 + 4 * g;
    if (x < p.W && o0 < p.co_total) {
      const int no = min(4, p.co_total - o0);
      const bool vec4 = no == 4 && (p.co_total % 4) == 0;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int y = ty0 + 2 * wave + r;
        if (y >= p.H) continue;
        const size_t pix = img + (size_t)y * p.W + x;
        const size_t oi = pix * p.co_total + o0;
        float v[4] = {acc[r].x, acc[r].y, acc[r].z, acc[r].w};
        if (vec4) {
          if (p.bias) {
            const float4 t = *(const float4*)(p.bias + o0);
            v[0] += t.x, v[1] += t.y, v[2] += t.z, v[3] += t.w;
          }
          if (p.addend) {
            const float4 t = *(const float4*)(p.addend + (((size_t)(n % p.NA) * p.H + y) * p.W + x) * p.co_total + o0);
            v[0] += t.x, v[1] += t.y, v[2] += t.z, v[3] += t.w;
          }
          if (p.mask) {
            float m[4];
            if (p.mask_dt == F32) {
              const float4 t = *(const float4*)((const float*)p.mask + oi);
              m[0] = t.x, m[1] = t.y, m[2] = t.z, m[3] = t.w;
            } else {
              const uint2 t = *(const uint2*)((const unsigned short*)p.mask + oi);
              const unsigned short h[4] = {(unsigned short)(t.x & 0xffffu), (unsigned short)(t.x >> 16), (unsigned short)(t.y & 0xffffu),
                                           (unsigned short)(t.y >> 16)};
              for (int i = 0; i < 4; ++i) m[i] = p.mask_dt == BF16 ? bf16_to_f(h[i]) : f16_to_f(h[i]);
            }
            for (int i = 0; i < 4; ++i) v[i] = m[i] > 0.f ? v[i] : 0.f;
          }
          if (p.residual) {
            const float4 t = *(const float4*)(p.residual + oi);
            v[0] += t.x, v[1] += t.y, v[2] += t.z, v[3] += t.w;
          }
          if (p.out_dt == F32) *(float4*)((float*)p.out + oi) = make_float4(v[0], v[1], v[2], v[3]);
          else if (p.out_dt == BF16) *(uint2*)((unsigned short*)p.out + oi) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
          else *(uint2*)((unsigned short*)p.out + oi) = make_uint2(pack_f16(v[0], v[1]), pack_f16(v[2], v[3]));
        } else {
          const size_t ai = p.addend ? (((size_t)(n % p.NA) * p.H + y) * p.W + x) * p.co_total + o0 : 0;
          for (int i = 0; i < no; ++i) {
            if (p.bias) v[i] += p.bias[o0 + i];
            if (p.addend) v[i] += p.addend[ai + i];
            if (p.mask && !(load1(p.mask, p.mask_dt, oi + i) > 0.f)) v[i] = 0.f;
            if (p.residual) v[i] += p.residual[oi + i];
            if (p.out_dt == F32) ((float*)p.out)[oi + i] = v[i];
            else ((unsigned short*)p.out)[oi + i] = p.out_dt == BF16 ? f_to_bf16(v[i]) : f_to_f16(v[i]);
          }
        }
      }
    }
  }
}

struct WgP {
  const void* x;           // [NB, H, W, C]
  const float* g;          // [NB, H, W, co]
  float* part;             // [bands][co][9][C]; accumulate: [co][9][C], zeroed by the caller, float atomics
  float* part_b;           // [bands][16] (accumulate: [16]) | null: the band's sum of g per output channel
  int x_dt, relu_in, accumulate;
  int NB, H, W, C, co, RB, bands_per_img;
};

// 4 waves = NCG groups of 16 input channels x (4 / NCG) subsets of the tile's rows; XW: x is fp32 (else: in the compute type)
template <int DT, int NCG, bool XW> __global__ __launch_bounds__(256, (DT == F32 ? 2 : NCG == 4 ? 3 : 4)) void k_wgrad_n16(const WgP p) {
  typedef Cmp<DT> CT;
  typedef typename CT::E E;
  typedef typename std::conditional<DT == F32, f4v, s4v>::type Frag;
  constexpr int CW = 16 * NCG;                 // input channels per workgroup
  constexpr int RS = 4 / NCG;                  // row subsets
  constexpr int XS = CW + 4, GS = 16 + 4;
  constexpr int XI = (NPIX * (CW / 8) + 255) / 256;
  __shared__ __attribute__((aligned(16))) E xl[NPIX * XS];
  __shared__ __attribute__((aligned(16))) E gl[TR * TC * GS];
  __shared__ float red[RS > 1 ? NCG * 9 * 256 : 1];
  __shared__ float redb[16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, g = lane >> 4;
  const int cg = wave % NCG, rs = wave / NCG;
  const int band = blockIdx.x, n = band / p.bands_per_img, rb = band % p.bands_per_img;
  const int c0 = blockIdx.y * CW;
  const int C = p.C, co = p.co;
  const bool xvec = (C % 8) == 0 && p.x_dt == (XW ? F32 : DT);
  const bool gvec = co == 16;
  const size_t img = (size_t)n * p.H * p.W;
  const int y_lo = rb * p.RB, y_hi = min(p.H, y_lo + p.RB);
  const int tiles_x = (p.W + TC - 1) / TC, ntiles = ((y_hi - y_lo + TR - 1) / TR) * tiles_x;

  f4v acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f4v){0.f, 0.f, 0.f, 0.f};
  // the band's sum of g (fp32, as it comes): this thread's output channels are fixed -- 4 (tid & 3) .. + 3 on the vector path,
  // tid & 15 on the scalar one
  float bs[4] = {0.f, 0.f, 0.f, 0.f};

  Raw8<XW> xr[XI];
  float4 gr[2];
  auto fetch = [&](int t) {
    const int ty0 = y_lo + (t / tiles_x) * TR, tx0 = (t % tiles_x) * TC;
    if (xvec) {
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const int it = tid + i * 256;
        const int px = it / (CW / 8), c8 = (it % (CW / 8)) * 8;
        const int y = ty0 + px / HC - 1, x = tx0 + px % HC - 1;
        if (it < NPIX * (CW / 8) && y >= 0 && y < p.H && x >= 0 && x < p.W && c0 + c8 < C)
          xr[i] = load_raw8<XW>(p.x, (img + (size_t)y * p.W + x) * C + c0 + c8);
        else
          zero_raw8(xr[i]);
      }
    }
    if (gvec) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int it = tid + i * 256, px = it >> 2, o4 = (it & 3) * 4;
        const int y = ty0 + px / TC, x = tx0 + px % TC;
        gr[i] = (y < y_hi && x < p.W) ? *(const float4*)(p.g + (img + (size_t)y * p.W + x) * 16 + o4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto commit = [&](int t) {
    const int ty0 = y_lo + (t / tiles_x) * TR, tx0 = (t % tiles_x) * TC;
    if (xvec) {
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const int it = tid + i * 256;
        if (it < NPIX * (CW / 8)) store_raw8<DT, XW>(xl + (it / (CW / 8)) * XS + (it % (CW / 8)) * 8, xr[i], p.relu_in != 0);
      }
    } else {
      for (int it = tid; it < NPIX * (CW / 8); it += 256) {
        const int px = it / (CW / 8), c8 = (it % (CW / 8)) * 8;
        const int y = ty0 + px / HC - 1, x = tx0 + px % HC - 1;
        const int nv = min(8, C - c0 - c8);
        float v[8];
        if (y >= 0 && y < p.H && x >= 0 && x < p.W && nv > 0) {
          load8_slow(p.x, p.x_dt, (img + (size_t)y * p.W + x) * C + c0 + c8, nv, v);
          if (p.relu_in) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
          }
          store8_f<DT>(xl + px * XS + c8, v);
        } else {
          zero8<DT>(xl + px * XS + c8);
        }
      }
    }
    if (gvec) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int it = tid + i * 256, px = it >> 2, o4 = (it & 3) * 4;
        E* d = gl + px * GS + o4;
        bs[0] += gr[i].x, bs[1] += gr[i].y, bs[2] += gr[i].z, bs[3] += gr[i].w;
        if constexpr (DT == F32) *(float4*)d = gr[i];
        else *(uint2*)d = make_uint2(CT::pack(gr[i].x, gr[i].y), CT::pack(gr[i].z, gr[i].w));
      }
    } else {
      for (int it = tid; it < TR * TC * 16; it += 256) {
        const int o = it & 15, px = it >> 4;
        const int y = ty0 + px / TC, x = tx0 + px % TC;
        float v = 0.f;
        if (o < co && y < y_hi && x < p.W) v = p.g[(img + (size_t)y * p.W + x) * co + o];
        bs[0] += v;
        gl[px * GS + o] = CT::cvt(v);
      }
    }
  };

  if (ntiles > 0) fetch(0);
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    commit(t);
    __syncthreads();
    if (t + 1 < ntiles) fetch(t + 1);
#pragma unroll 1
    for (int r = rs; r < TR; r += RS) {
      Frag a, b;
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = gl[(r * TC + 4 * g + j) * GS + col];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = xl[((r + dy) * HC + 4 * g + j + dx) * XS + cg * 16 + col];
        acc[tap] = CT::mma16(a, b, acc[tap]);
      }
    }
  }

  // ---- the band's bias-gradient partial
  if (p.part_b && blockIdx.y == 0) {
    if (tid < 16) redb[tid] = 0.f;
    __syncthreads();
    if (gvec) {
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(&redb[(tid & 3) * 4 + k], bs[k]);
    } else {
      atomicAdd(&redb[tid & 15], bs[0]);
    }
    __syncthreads();
    if (tid < 16) {
      if (p.accumulate) unsafeAtomicAdd(p.part_b + tid, redb[tid]);
      else p.part_b[(size_t)band * 16 + tid] = redb[tid];
    }
  }
  const size_t pb = p.accumulate ? 0 : (size_t)band * co * 9 * C;

  // acc[tap][i] = gw[o = 4 g + i][tap][c = c0 + 16 cg + col] of this wave's rows
  if constexpr (RS > 1) {
    __syncthreads();
    for (int i = tid; i < NCG * 9 * 256; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float v[4] = {acc[tap].x, acc[tap].y, acc[tap].z, acc[tap].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) atomicAdd(&red[((cg * 9 + tap) * 16 + 4 * g + i) * 16 + col], v[i]);
    }
    __syncthreads();
    for (int i = tid; i < NCG * 9 * 256; i += 256) {
      const int cc = i & 15, o = (i >> 4) & 15, tap = (i >> 8) % 9, gcg = i / (9 * 256);
      const int c = c0 + gcg * 16 + cc;
      if (o < co && c < C) {
        float* d = p.part + pb + ((size_t)o * 9 + tap) * C + c;
        if (p.accumulate) unsafeAtomicAdd(d, red[i]);
        else *d = red[i];
      }
    }
  } else {
    const int c = c0 + cg * 16 + col;
    if (c < C) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float v[4] = {acc[tap].x, acc[tap].y, acc[tap].z, acc[tap].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int o = 4 * g + i;
          if (o < co) {
            float* d = p.part + pb + ((size_t)o * 9 + tap) * C + c;
            if (p.accumulate) unsafeAtomicAdd(d, v[i]);
            else *d = v[i];
          }
        }
      }
    }
  }
}

// ---- bilinear resize, align_corners = False, channels-last fp32 [NB, H, W, C] (C % 4 == 0) ------------------------------------
__device__ __forceinline__ void src_index(int o, float scale, int in_size, int& i0, int& i1, float& w1) {
  // at::native area_pixel_compute_source_index (align_corners = False): max(0, (o + 0.5) * scale - 0.5)
  const float s = fmaxf((o + 0.5f) * scale - 0.5f, 0.f);
  i0 = min((int)s, in_size - 1);
  i1 = min(i0 + 1, in_size - 1);
  w1 = s - (float)i0;
}

__global__ __launch_bounds__(256) void k_up_fwd(const float* __restrict__ in, int NB, int H, int W, int C, int HO, int WO, float sy, float sx,
                                                float* __restrict__ out) {
  const int c4 = C / 4;
  const size_t total = (size_t)NB * HO * WO * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % c4);
    size_t r = i / c4;
    const int ox = (int)(r % WO);
    r /= WO;
    const int oy = (int)(r % HO), n = (int)(r / HO);
    int y0, y1, x0, x1;
    float wy, wx;
    src_index(oy, sy, H, y0, y1, wy);
    src_index(ox, sx, W, x0, x1, wx);
    const float4* b = (const float4*)in + (size_t)n * H * W * c4 + c;
    const float4 v00 = b[((size_t)y0 * W + x0) * c4], v01 = b[((size_t)y0 * W + x1) * c4];
    const float4 v10 = b[((size_t)y1 * W + x0) * c4], v11 = b[((size_t)y1 * W + x1) * c4];
    const float a00 = (1.f - wy) * (1.f - wx), a01 = (1.f - wy) * wx, a10 = wy * (1.f - wx), a11 = wy * wx;
    float4 o;
    o.x = a00 * v00.x + a01 * v01.x + a10 * v10.x + a11 * v11.x;
    o.y = a00 * v00.y + a01 * v01.y + a10 * v10.y + a11 * v11.y;
    o.z = a00 * v00.z + a01 * v01.z + a10 * v10.z + a11 * v11.z;
    o.w = a00 * v00.w + a01 * v01.w + a10 * v10.w + a11 * v11.w;
    ((float4*)out)[i] = o;
  }
}

// gin[n, iy, ix, :] = sum over the output pixels whose two source rows / columns include (iy, ix)
__global__ __launch_bounds__(256) void k_up_bwd(const float* __restrict__ go, int NB, int H, int W, int C, int HO, int WO, float sy, float sx,
                                                float ry, float rx, float* __restrict__ gin) {
  const int c4 = C / 4;
  const size_t total = (size_t)NB * H * W * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % c4);
    size_t r = i / c4;
    const int ix = (int)(r % W);
    r /= W;
    const int iy = (int)(r % H), n = (int)(r / H);
    // output rows whose source coordinate lies in (iy - 1, iy + 1) (+ the clamped borders): a superset, each one re-checked below
    const int oy_lo = max(0, (int)floorf((iy - 1.f + 0.5f) * ry - 0.5f) - 1), oy_hi = min(HO - 1, (int)ceilf((iy + 1.f + 0.5f) * ry - 0.5f) + 1);
    const int ox_lo = max(0, (int)floorf((ix - 1.f + 0.5f) * rx - 0.5f) - 1), ox_hi = min(WO - 1, (int)ceilf((ix + 1.f + 0.5f) * rx - 0.5f) + 1);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* b = (const float4*)go + (size_t)n * HO * WO * c4 + c;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float wy;
      src_index(oy, sy, H, y0, y1, wy);
      const float ay = (y0 == iy ? 1.f - wy : 0.f) + (y1 == iy ? wy : 0.f);
      if (ay == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float wx;
        src_index(ox, sx, W, x0, x1, wx);
        const float ax = (x0 == ix ? 1.f - wx : 0.f) + (x1 == ix ? wx : 0.f);
        if (ax == 0.f) continue;
        const float4 v = b[((size_t)oy * WO + ox) * c4];
        const float a = ay * ax;
        s.x += a * v.x, s.y += a * v.y, s.z += a * v.z, s.w += a * v.w;
      }
    }
    ((float4*)gin)[i] = s;
  }
}

int check_dt(int dt) { return dt == F32 || dt == BF16 || dt == F16; }

int wgrad_cw(int C, int compute_dt) { return C >= 64 && compute_dt != F32 ? 64 : C >= 32 ? 32 : 16; }

}  // namespace

extern "C" int ocpg_mso_conv3x3(const void* in, int in_dt, int relu_in, const void* w, int w_dt, const float* bias, const float* addend, int NA,
                                const void* mask, int mask_dt, const float* residual, void* out, int out_dt, int NB, int H, int W, int C,
                                int co_total, int compute_dt, void* stream) {
  if (NB < 0 || H < 1 || W < 1 || C < 1 || co_total < 1) return -1002;
  if (NB == 0) return 0;
  if (!in) return -1001;
  if (!w) return -1004;
  if (!out) return -1012;
  if (!check_dt(in_dt) || !check_dt(w_dt) || !check_dt(out_dt) || !check_dt(compute_dt) || (mask && !check_dt(mask_dt))) return -1003;
  if (addend && (NA < 1 || NB % NA)) return -1008;
  if (NB > 65535 || (co_total + 15) / 16 > 65535) return -1002;
  ConvP p;
  p.in = in, p.w = w, p.bias = bias, p.addend = addend, p.mask = mask, p.residual = residual, p.out = out;
  p.in_dt = in_dt, p.w_dt = w_dt, p.mask_dt = mask_dt, p.out_dt = out_dt, p.relu_in = relu_in;
  p.NB = NB, p.H = H, p.W = W, p.C = C, p.co_total = co_total, p.NA = addend ? NA : 1;
  p.tiles_x = (W + TC - 1) / TC;
  const int ngroups = (co_total + 15) / 16;
  // input gradient of a feature half (16 -> 256 / 512 channels): one staged gradient tile serves 8 groups of output channels
  p.gpw = (C <= 16 && ngroups >= 8) ? 8 : 1;
  const dim3 grid((unsigned)(p.tiles_x * ((H + TR - 1) / TR)), (unsigned)NB, (unsigned)((ngroups + p.gpw - 1) / p.gpw));
  hipStream_t st = (hipStream_t)stream;
  const bool xw = in_dt == F32;       // fp32 input: the mask path and every gradient (a 16-bit input in another type than compute_dt: scalar path)
  if (C <= 16) {              // the mask path and every input gradient: 12 KB of LDS, 4 workgroups per CU
    if (compute_dt == BF16) xw ? k_conv_n16<BF16, 16, 16, true><<<grid, 256, 0, st>>>(p) : k_conv_n16<BF16, 16, 16, false><<<grid, 256, 0, st>>>(p);
    else if (compute_dt == F16) xw ? k_conv_n16<F16, 16, 16, true><<<grid, 256, 0, st>>>(p) : k_conv_n16<F16, 16, 16, false><<<grid, 256, 0, st>>>(p);
    else k_conv_n16<F32, 16, 16, true><<<grid, 256, 0, st>>>(p);
  } else {
    if (compute_dt == BF16) xw ? k_conv_n16<BF16, 64, 32, true><<<grid, 256, 0, st>>>(p) : k_conv_n16<BF16, 64, 32, false><<<grid, 256, 0, st>>>(p);
    else if (compute_dt == F16) xw ? k_conv_n16<F16, 64, 32, true><<<grid, 256, 0, st>>>(p) : k_conv_n16<F16, 64, 32, false><<<grid, 256, 0, st>>>(p);
    else k_conv_n16<F32, 32, 16, true><<<grid, 256, 0, st>>>(p);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// rows per band of the weight-gradient kernel (a multiple of 8): about 512 workgroups over the images
extern "C" int ocpg_mso_wgrad_rows(int NB, int H, int C, int compute_dt) {
  const int cw = wgrad_cw(C, compute_dt), chunks = (C + cw - 1) / cw;
  int rb = 8;
  while (rb < H && (long long)NB * ((H + rb - 1) / rb) * chunks > 640) rb += 8;
  return rb;
}

extern "C" int ocpg_mso_wgrad(const void* x, int x_dt, int relu_in, const float* g, float* part, float* part_bias, int accumulate, int NB, int H,
                              int W, int C, int co, int rows_per_band, int compute_dt, void* stream) {
  if (NB < 0 || H < 1 || W < 1 || C < 1 || co < 1 || co > 16 || rows_per_band < 8 || rows_per_band % 8) return -1002;
  if (NB == 0) return 0;
  if (!x) return -1001;
  if (!g) return -1004;
  if (!part) return -1005;
  if (!check_dt(x_dt) || !check_dt(compute_dt)) return -1003;
  WgP p;
  p.x = x, p.g = g, p.part = part, p.part_b = part_bias, p.x_dt = x_dt, p.relu_in = relu_in, p.accumulate = accumulate;
  p.NB = NB, p.H = H, p.W = W, p.C = C, p.co = co, p.RB = rows_per_band, p.bands_per_img = (H + rows_per_band - 1) / rows_per_band;
  const long long bands = (long long)NB * p.bands_per_img;
  if (bands > 0x7fffffffLL) return -1002;
  hipStream_t st = (hipStream_t)stream;
  const int cw = wgrad_cw(C, compute_dt);
  const dim3 grid((unsigned)bands, (unsigned)((C + cw - 1) / cw));
  const bool xw = x_dt == F32;
#define OCPG_WG(DT_, NCG_) (xw ? k_wgrad_n16<DT_, NCG_, true><<<grid, 256, 0, st>>>(p) : k_wgrad_n16<DT_, NCG_, false><<<grid, 256, 0, st>>>(p))
  if (cw == 64) {
    if (compute_dt == BF16) OCPG_WG(BF16, 4);
    else OCPG_WG(F16, 4);
  } else if (cw == 32) {
    if (compute_dt == BF16) OCPG_WG(BF16, 2);
    else if (compute_dt == F16) OCPG_WG(F16, 2);
    else k_wgrad_n16<F32, 2, true><<<grid, 256, 0, st>>>(p);
  } else {
    if (compute_dt == BF16) OCPG_WG(BF16, 1);
    else if (compute_dt == F16) OCPG_WG(F16, 1);
    else k_wgrad_n16<F32, 1, true><<<grid, 256, 0, st>>>(p);
  }
#undef OCPG_WG
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ocpg_bilinear_nhwc_fwd(const float* in, int NB, int H, int W, int C, int HO, int WO, float* out, void* stream) {
  if (NB < 0 || H < 1 || W < 1 || HO < 1 || WO < 1 || C < 4 || C % 4) return -1002;
  if (NB == 0) return 0;
  if (!in) return -1001;
  if (!out) return -1008;
  const size_t total = (size_t)NB * HO * WO * (C / 4);
  const unsigned blocks = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  k_up_fwd<<<blocks, 256, 0, (hipStream_t)stream>>>(in, NB, H, W, C, HO, WO, (float)H / HO, (float)W / WO, out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int ocpg_bilinear_nhwc_bwd(const float* gout, int NB, int H, int W, int C, int HO, int WO, float* gin, void* stream) {
  if (NB < 0 || H < 1 || W < 1 || HO < 1 || WO < 1 || C < 4 || C % 4) return -1002;
  if (NB == 0) return 0;
  if (!gout) return -1001;
  if (!gin) return -1008;
  const size_t total = (size_t)NB * H * W * (C / 4);
  const unsigned blocks = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  k_up_bwd<<<blocks, 256, 0, (hipStream_t)stream>>>(gout, NB, H, W, C, HO, WO, (float)H / HO, (float)W / WO, (float)HO / H, (float)WO / W, gin);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
