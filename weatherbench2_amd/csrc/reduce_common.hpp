// Pieces shared by the streaming reductions (stream_reduce.hip, ensemble.hip).
#pragma once

#include "common.hpp"

#include <cstdint>

#ifndef WB2_ROTATE_CHUNKS
// 1: kernels whose workgroup index runs over the chunks of a slab first (grid
// x = chunk, y = slab: K1 without a 2-D field, the ensemble kernels) take chunk
// (x + slab) % n_chunk: workgroup x always lands on XCD x % 8 (n_chunk % 8 ==
// 0), chunks differ in size, and a FIXED chunk -> XCD map gives some XCDs the
// short chunks of every slab (profiles/r04_xcd_balance.md).
#define WB2_ROTATE_CHUNKS 1
#endif

namespace wb2 {

// wb2_det_combine with the slot count given (stream_reduce.hip)
int combine_slots(int mode, int skipna, int k_slots, const double* partials,
                  int64_t n_outer, int32_t n_chunk, int32_t nwf, int32_t n_seg,
                  const int32_t* seg_eoff, int32_t n_ts,
                  const int32_t* band_chunk0, int32_t n_band,
                  const double* coef_band, const double* coef_seg,
                  const int32_t* region_wf, const double* region_wsum,
                  int32_t n_region, double* sums, double* metrics,
                  void* stream);

// ---- many sums at once -------------------------------------------------------
// Sums N per-lane doubles v[0..N) across the 64 lanes of a wave with a HALVING
// tree: at every lane-bit the two halves of the wave exchange HALF of their
// values (the lower half keeps v[0..H), the upper half v[H..N), H = ceil(N/2))
// and add what they receive, so the number of live values per lane halves with
// every step -- N + O(log) exchanges in total instead of the 6 N of one
// xor-butterfly per value (a fold of 12 sums: 14 exchanges instead of 72).  When
// one value is left the remaining lane bits are a plain all-reduce.  On return
// v[0] of a lane holds the total of value `slot`; `real` > 0 says the slot is a
// real value (not padding) and `writer` picks one lane per slot.  The order of
// the additions is a fixed tree: bit-reproducible.
//
// The exchanges at lane bits 5 and 4 are gfx950's v_permlane32_swap /
// v_permlane16_swap: two 32-bit swaps move a double of each half to the other
// half at once (no LDS, no selects); bits 3..0 use ds_bpermute for the (few)
// halving steps left and DPP moves for the all-reduce tail.
__device__ __forceinline__ void swap_halves32(double& a, double& b) {
  const unsigned long long ua = __builtin_bit_cast(unsigned long long, a);
  const unsigned long long ub = __builtin_bit_cast(unsigned long long, b);
  const auto lo = __builtin_amdgcn_permlane32_swap(
      (unsigned)ua, (unsigned)ub, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(
      (unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  a = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]);
  b = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
}
__device__ __forceinline__ void swap_halves16(double& a, double& b) {
  const unsigned long long ua = __builtin_bit_cast(unsigned long long, a);
  const unsigned long long ub = __builtin_bit_cast(unsigned long long, b);
  const auto lo = __builtin_amdgcn_permlane16_swap(
      (unsigned)ua, (unsigned)ub, false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(
      (unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  a = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]);
  b = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
}

template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(
      0, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(
      0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// All-reduce of ONE value over the lane bits OFF, OFF / 2, ..., 1 -- in THAT
// order, like every halving step above it: each total is then the same binary
// tree over the lanes (bit 5 first, bit 0 last) whatever its position among the
// N values, i.e. bit-identical to a plain xor-butterfly of that value alone
// (MSE from a DET pass == MSE from a DET_ACC pass).  Bit 3: row_ror:8, bits 1
// and 0: quad permutes (DPP, no LDS); bit 2 has no DPP form: ds_swizzle.
template <int OFF>
__device__ __forceinline__ double allsum_low_bits(double v) {
  static_assert(OFF >= 1 && OFF <= 32, "lane bit");

  if constexpr (OFF >= 32) v += __shfl_xor(v, 32, kWave);
  if constexpr (OFF >= 16) v += __shfl_xor(v, 16, kWave);
  if constexpr (OFF >= 8) v += dpp_move<0x128>(v);     // row_ror:8 == xor 8
  if constexpr (OFF >= 4) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    // BITMODE swizzle: and 0x1f, or 0, xor 4
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)u,
                                                              0x101F);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_swizzle(
        (int)(unsigned)(u >> 32), 0x101F);
    v += __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  }
  if constexpr (OFF >= 2) v += dpp_move<0x4E>(v);      // xor 2 (quad_perm)
  v += dpp_move<0xB1>(v);                              // xor 1 (quad_perm)
  return v;
}

template <int N, int OFF, int NV>
__device__ __forceinline__ void wave_sum_many_step(double (&v)[NV], int lane,
                                                   int& slot, int& real,
                                                   bool& writer) {
  if constexpr (OFF == 0) {
    return;
  } else if constexpr (N == 1) {
    v[0] = allsum_low_bits<OFF>(v[0]);
    writer = writer && (lane & (2 * OFF - 1)) == 0;
  } else {
    constexpr int H = (N + 1) / 2;
    const bool up = (lane & OFF) != 0;
#pragma unroll
    for (int i = 0; i < H; ++i) {
      double lo = v[i];
      double hi = (H + i < N) ? v[H + i] : 0.0;
      if constexpr (OFF == 32) {
        swap_halves32(lo, hi);
        v[i] = lo + hi;
      } else if constexpr (OFF == 16) {
        swap_halves16(lo, hi);
        v[i] = lo + hi;
      } else {
        const double keep = up ? hi : lo, send = up ? lo : hi;
        v[i] = keep + __shfl_xor(send, OFF, kWave);
      }
    }
    slot += up ? H : 0;
    real = up ? (real > H ? real - H : 0) : (real < H ? real : H);
    wave_sum_many_step<H, OFF / 2, NV>(v, lane, slot, real, writer);
  }
}

template <int N>
__device__ __forceinline__ void wave_sum_many(double (&v)[N], int lane,
                                              int& slot, bool& writes) {
  int real = N;
  bool writer = true;
  slot = 0;
  wave_sum_many_step<N, 32, N>(v, lane, slot, real, writer);
  writes = writer && real > 0;
}

// Epilogue of a wave that owns the 64*VEC columns [tile*64*VEC, ...): fold the
// per-column fp64 sums into every seg that intersects the tile with a wave64
// butterfly (no LDS, no barrier, fixed order => bit-reproducible) and store the
// K sums of each (seg, tile) entry.  `out` points at this (outer, chunk)'s
// [NWF][n_ts][K] block; see seg_eoff in include/wb2hip.h for the entry index.
// The lane holds the columns col0 .. col0 + VEC - 1 and owns those >= own0
// (own0 > col0 only for a row-end lane that loaded shifted back).
template <int NWF, int VEC, int K>
__device__ __forceinline__ void fold_tile_to_segs(
    const double (&acc)[NWF][VEC][K], int lane, int tile, int col0, int own0,
    int n_col, const int* __restrict__ seg_col0, const int* __restrict__ seg_eoff,
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
    // the NWF * K sums of this (seg, tile) entry in ONE halving tree
    double v[NWF * K];
#pragma unroll
    for (int w = 0; w < NWF; ++w) {
#pragma unroll
      for (int k = 0; k < K; ++k) v[w * K + k] = 0.0;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const bool in_seg = (col0 + e >= max(c0, own0)) && (col0 + e < c1);
#pragma unroll
        for (int k = 0; k < K; ++k)
          v[w * K + k] += in_seg ? acc[w][e][k] : 0.0;
      }
    }
    int slot;
    bool writes;
    wave_sum_many<NWF * K>(v, lane, slot, writes);
    const int e = seg_eoff[s] + tile - c0 / TILE;
    if (writes) {
      const int w = slot / K, k = slot - w * K;
      out[((long long)w * n_ts + e) * K + k] = v[0];
    }
  }
}

}  // namespace wb2
