// Column-tile MSDeformAttn kernels (self-attention over the value's own pixels, Lq == S): shared declarations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ocpg_col {

constexpr int kMaxLevels = 4;
constexpr int kMarginLo = 4;   // gather kernels: window extends this many pixels below the tile's footprint ...
constexpr int kMarginHi = 5;   // ... and this many above (x0 + 1 is the far corner)
// Scatter kernels (round 4): 5 below.  At the model's initial ring offsets (<= 4 pixels) a query of a FINER level lands a quarter to
// three quarters of a pixel below its column's footprint at a coarser level, so with 4 below 1.07 % of the samples missed the window --
// and each miss is a wave-serial step with four global atomics: 30 of the kernel's 190 us (measured by cutting the kernel, tools/r4_exp7.sh).
constexpr int kScatterMarginLo = 5;

// Host-derived tiling of one (frame, head) problem into "pyramid columns": the finest level is cut into
// nty x ntx blocks of <= 8x8 pixels; a column owns, at EVERY level, the pixels whose index range scales to the same
// block (rows [ty*H_l/nty, (ty+1)*H_l/nty)).  Queries of a column sample -- at every destination level -- around the
// column's own footprint, so one small window per (column, level) catches them; anything outside the window takes a
// direct global path, so results never depend on locality.
struct ColGeom {
  int L, nty, ntx, ntiles;
  unsigned m_ntx, m_nty, m_ntiles, m_M, m_P;   // multiply-high reciprocals (no integer divide on the device)
  int mlo, mhi;   // window margins below / above the footprint, in pixels of the destination level
  int tmax;   // max queries of a column
  int wmax;   // max window pixels of a (column, level)
  int wrest;  // max window pixels of a column over levels 1..L-1
  int H[kMaxLevels], W[kMaxLevels], S0[kMaxLevels];
};

// false: shapes not supported by the column kernels (caller falls back to the row kernels)
// tile_h x tile_w: block of the finest level that defines a column (8x8 for the gather kernels, 8x16 for the scatter)
bool make_col_geom(const int64_t* shapes_host, int L, int S, int M, int P, int tile_h, int tile_w, ColGeom& g, int margin_lo = kMarginLo,
                   int margin_hi = kMarginHi);

bool scatter_supported(const ColGeom& g, int D, int P);

// ---- per-call path selection between the column scatter (path 0) and the output-tiled kernels (path 1, msda_tile.h) ---------------
// State of one CALL SITE (a module's backward): 8 ints in device memory, zero-initialised once by the caller and handed to every call.
// Every call launches both paths' kernels; the workgroups of the path that is not current leave at once (a few microseconds).  The
// active path counts the samples that miss its locality assumption and proposes the path of the NEXT call (hysteresis: two thresholds);
// a one-thread kernel at the end of the launch sequence commits the proposal.  No host round trip: the same launches replay inside a HIP graph.
enum { kSelFar = 0, kSelTotal = 1, kSelUnusedA = 2, kSelCur = 3, kSelNext = 4, kSelUnusedB = 5, kSelLastFar = 6, kSelLastTotal = 7 };   // [0..1]: the packed 64-bit report word

// The active path's (sampled) workgroups report (far samples, samples) with ONE fire-and-forget 64-bit atomic on the packed word at
// sel[kSelFar .. kSelTotal] (bits 0..31 samples, 32..63 far samples); nothing comes back, nobody waits.  The one-thread kernel launched
// after every kernel of the call (select_commit) reads the word -- complete at the kernel boundary --, turns it into the path of the
// NEXT call (two thresholds: hysteresis), makes it current and clears the word: no kernel of a call sees the state change under it.
// (Round 4, first half: two adds, a fence, a ticket, a fence and two read-backs per reporting workgroup at the END of the kernel -- on
// cold operands the reporting workgroups waited for their returning atomics behind the flush traffic and the launch ended 27 us later:
// 184-188 vs 158 us, tools/ab_gv_variants.sh GV_SELECT=1.  A returning atomic issued early and consumed late kept two registers alive through the sums
// of a kernel at its 80-register cap: 172 vs 160 us.)
__device__ __forceinline__ void sel_report(int* sel, int far, int total) {
  atomicAdd(reinterpret_cast<unsigned long long*>(sel + kSelFar), ((unsigned long long)(unsigned)far << 32) | (unsigned long long)(unsigned)total);
}

// last launch of a call that passed `sel`: this call's reports -> the next call's path (the column family moves to the tiled one above
// to_tile_pct percent far samples, the tiled family back below to_col_pct), which becomes current
void select_commit(int* sel, int to_tile_pct, int to_col_pct, hipStream_t st);

// true: the one-pass patch kernel (the only column kernel that takes part in the selection) serves this geometry
bool select_supported(const ColGeom& g, int D, int P);

// Each returns 1 (2: the launched kernel honours `sel`) when it launched, 0 when the shape is not supported (nothing launched).
int fwd_col(const float* value, const float* loc, const float* attn, int N, int S, int M, int D, int P, const ColGeom& g,
            float* out, hipStream_t st);
int bwd_scatter_col(const float* loc, const float* attn, const float* gout, int N, int S, int M, int D, int P, const ColGeom& g,
                    float* gvalue, hipStream_t st, int* sel = nullptr, int to_tile_pct = 0);

}  // namespace ocpg_col
