// Pieces shared by the streaming reductions (stream_reduce.hip, ensemble.hip).
#pragma once

#include "common.hpp"

namespace wb2 {

// Epilogue of a wave that owns the 64*VEC columns [tile*64*VEC, ...): fold the
// per-column fp64 sums into every seg that intersects the tile with a wave64
// butterfly (no LDS, no barrier, fixed order => bit-reproducible) and store the
// K sums of each (seg, tile) entry.  `out` points at this (outer, chunk)'s
// [NWF][n_ts][K] block; see seg_eoff in include/wb2hip.h for the entry index.
template <int NWF, int VEC, int K>
__device__ __forceinline__ void fold_tile_to_segs(
    const double (&acc)[NWF][VEC][K], int lane, int tile, int col0, int n_col,
    const int* __restrict__ seg_col0, const int* __restrict__ seg_eoff,
    int n_seg, int n_ts, double* __restrict__ out) {
  constexpr int TILE = kWave * VEC;
  const int tile_c0 = tile * TILE;
  const int tile_c1 = min(tile_c0 + TILE, n_col);
  int s_lo = 0;
  while (seg_col0[s_lo + 1] <= tile_c0) ++s_lo;
  int s_hi = s_lo;
  while (s_hi + 1 < n_seg && seg_col0[s_hi + 1] < tile_c1) ++s_hi;
  for (int s = s_lo; s <= s_hi; ++s) {
    const int c0 = seg_col0[s], c1 = seg_col0[s + 1];
#pragma unroll
    for (int w = 0; w < NWF; ++w) {
      double v[K];
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = 0.0;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const bool in_seg = (col0 + e >= c0) && (col0 + e < c1);
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] += in_seg ? acc[w][e][k] : 0.0;
      }
      double mine = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const double tot = wave_allsum(v[k]);
        mine = (lane == k) ? tot : mine;
      }
      const int e = seg_eoff[s] + tile - c0 / TILE;
      if (lane < K) out[((long long)w * n_ts + e) * K + lane] = mine;
    }
  }
}

}  // namespace wb2
