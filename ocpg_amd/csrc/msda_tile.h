// Output-tiled grad_value of the MSDeformAttn backward (self-attention, Lq == S): shared declarations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ocpg_tile {

constexpr int kLM = 4;        // levels at most
constexpr int kLA = 2;        // levels handled by the output-tiled kernel at most (the fine ones)
constexpr int kMaxH0 = 64, kMaxW0 = 128;     // finest level at most (rows, columns)
constexpr int kSumH = 128, kSumW = 256;      // sum over the levels of rows / of columns at most
constexpr int kMaxT = 16;                    // tiles per axis at most
constexpr int kMargin = 5;                   // a query is a CANDIDATE of every tile within this many pixels of its own position

// Host-built description of one (frame, head) problem; passed to the kernels BY VALUE (kernel argument, < 4 KB).
// Pixel rows of level l < LA are cut into nty bands, columns into ntx bands (the "footprint" of tile (ty, tx) at level l);
// every pixel of those levels belongs to exactly one tile.  Query (lq, qy, qx) -- queries ARE the pixels of all levels -- is a
// candidate of tile (ty, tx) at destination level l iff ty in [tlo_y, thi_y] and tx in [tlo_x, thi_x], the tiles touched by the
// pixel box (its own position mapped to level l) +- kMargin.  Both kernels read the SAME tables, so the split of the
// contributions between them is exact whatever the sampling offsets are:
//   candY/candX [l][lq][t]  = the query rows / columns of level lq that are candidates of tile band t      (tile kernel)
//   nearY/nearX [l][row of lq] = the pixel rows / columns of level l covered by the bands the query is a candidate of
//                                 (coarse kernel: a corner outside this box is "far" and goes to memory with atomics there)
struct TileTab {
  int L, LA, nty, ntx, ntiles, K;
  int H[kLM], W[kLM], S0[kLM];
  int ybase[kLM], xbase[kLM];
  int nbA, offB[kLM + 1], nbB;
  unsigned m_M, m_ntiles, m_ntx, m_K, m_W[kLM];
  uint8_t ry0[kLA][kMaxT + 1], cx0[kLA][kMaxT + 1];
  uint8_t nearY[kLA][kSumH][2], nearX[kLA][kSumW][2];
  uint8_t candY[kLA][kLM][kMaxT][2], candX[kLA][kLM][kMaxT][2];
};

// false: shape not served (the caller keeps the column-scatter kernel)
bool make_tile_tab(const int64_t* shapes_host, int N, int L, int S, int M, int P, int D, TileTab& t);

// grad_value (caller zero-fills it) of the self-attention backward; 1 = launched (two kernels), 0 = shape not served
// sel: the call site's path-selection state (msda_col.h) or null; with it the kernels run only when the state's current path is 1, and the
// second kernel proposes the next call's path (the caller launches ocpg_col::select_commit after both families)
int bwd_value_tile(const float* loc, const float* attn, const float* gout, const int64_t* shapes_host, int N, int S, int M, int D, int L,
                   int P, float* gvalue, hipStream_t st, int* sel = nullptr, int to_col_pct = 0);

// true: the shape is served (what bwd_value_tile checks before it launches)
bool tile_supported(const int64_t* shapes_host, int N, int L, int S, int M, int P, int D);

}  // namespace ocpg_tile
