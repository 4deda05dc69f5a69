// K1/K2 of libwb2hip.so: the fused, region-aware, latitude-weighted streaming
// reduction behind MSE / RMSE / MAE / Bias / ACC / WindVector{MSE,RMSE}.
//
// Replaces (reference = /root/reference/weatherbench2):
//   metrics.py:141-163  _spatial_average     (einsum(x,w) / einsum(notnull(x),w))
//   metrics.py:166-172  _spatial_average_l2_norm
//   metrics.py:175-414  WindVectorMSE, RMSE, MSE, MAE, Bias, ACC elementwise temps
//   regions.py:57-158   Region.apply (slice / extra-tropics / land / combined)
//   evaluation.py:408-437  the metric x region loop
//
// HBM-bandwidth bound (12 B per grid point for f32 f,t,c; no reuse), so the
// design is a pure stream: every thread OWNS `VEC` adjacent columns of the slab
// (16-byte loads, a wavefront covers 1 KiB of a row) and walks down the rows of
// one row-chunk keeping one fp64 accumulator per (weight-field, owned column,
// slot) in VGPRs.  Region membership never enters the hot loop: columns with the
// same membership form a `seg`, rows a `band`; the per-column sums are folded
// into per-seg sums once per workgroup through LDS + a wave64 shuffle tree, and a
// tiny second kernel folds (band, seg) cells into every region.  No atomics, so
// results are bit-reproducible run to run.
//
// Elementwise arithmetic is done in the INPUT dtype, like numpy does under
// xarray (float32 stays float32; -ffp-contract=off keeps products unfused),
// and every sum is accumulated in fp64 like the promoted einsum.

#include "common.hpp"
#include "trace.hpp"
#include "reduce_common.hpp"
#include "gauss_math.hpp"
#include "suite_streams.hpp"
#include "wb2hip.h"

#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace wb2 {
namespace {

constexpr int kMaxIn = 4;

struct StreamParams {
  const void* in[kMaxIn];
  const long long* slab[kMaxIn];
  const double* w_row;
  const double* w_col;
  const void* wfield;  // [n_row][n_col] float64, or float32 (wfield_f32)
  const double* aux;   // mode-specific 2-D field [n_row][n_col] (SEEPS: p1)
  double scalar;       // mode-specific scalar (SEEPS: dry threshold)
  const int* chunk_row0;
  const int* chunk_nrow;
  const int* seg_col0;
  const int* seg_eoff;
  double* partials;
  long long n_outer;
  // bytes between consecutive slab numbers of an input: n_row * n_col *
  // sizeof(T) when `slab` holds slab numbers (or is NULL), 1 when it holds the
  // byte ADDRESS of every slab and in[] is NULL (wb2_stream_partials_addr: the
  // slabs of one launch live in different allocations -- the variables of a
  // chunk, consecutive chunks)
  long long slab_step_bytes;
  int n_row, n_col, n_chunk, n_ctile, n_seg, n_ts;
  // host only: rows (n_col * sizeof(T)) or input bases are not 16-byte aligned
  // -- the lon-lat layout's 721-column rows: selects the SGPR-addressed
  // instantiation where there is one
  int unaligned;
  int wfield_f32;  // the weight field is stored as float32 (exactly)
};

template <int MODE, bool SKIPNA>
struct ModeTraits;
template <bool S>
struct ModeTraits<WB2_MODE_DET, S> {
  static constexpr int NIN = 2, KQ = 3, K = KQ + (S ? 1 : 0);
};
template <bool S>
struct ModeTraits<WB2_MODE_DET_ACC, S> {
  static constexpr int NIN = 3, KQ = 6, K = KQ + (S ? 4 : 0);
};
template <bool S>
struct ModeTraits<WB2_MODE_WIND, S> {
  static constexpr int NIN = 4, KQ = 1, K = KQ + (S ? 1 : 0);
};
template <bool S>
struct ModeTraits<WB2_MODE_GAUSS, S> {
  static constexpr int NIN = 3, KQ = 2, K = KQ + (S ? 2 : 0);
};
template <bool S>
struct ModeTraits<WB2_MODE_GAUSS_THR, S> {
  static constexpr int NIN = 4, KQ = 3, K = KQ + (S ? 3 : 0);
};
template <bool S>
struct ModeTraits<WB2_MODE_SEEPS, S> {
  static constexpr int NIN = 3, KQ = 1, K = KQ + (S ? 1 : 0);
};

// One grid point: inputs -> the K values whose weighted sums we need.  With
// SKIPNA, NaN numerators become 0 and the trailing slots carry notnull() as
// 1.0/0.0 so that the same weighted accumulation yields xarray's sum_of_weights.
// skipna: `ok ? v : 0` for a float32 value as ONE v_mul_legacy_f32 with the
// flag as a 1.0f / 0.0f multiplier (DX9 rule: 0 * anything, NaN and Inf
// included, is 0; times 1.0 is exact) -- hipcc turns the select into a 64-bit
// one after the conversion otherwise (two v_cndmask per slot).
__device__ __forceinline__ float keep_if(float v, float flag) {
  float r;
  asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(flag));
  return r;
}
__device__ __forceinline__ double keep_if(double v, double flag) {
  return flag != 0.0 ? v : 0.0;
}

// The deterministic modes in two stages.  elementwise(): the arithmetic the
// reference does in the input dtype (metrics.py:195-197, 264, 284, 329, 358,
// 405-410), written on a value type V that is either T or a 2-vector of T --
// two adjacent columns of a lane's load go through v_pk_add_f32 / v_pk_mul_f32
// as ONE instruction each (same IEEE results per component).  slots(): one
// point's quantities -> the K float64 values whose weighted sums are needed.
template <int MODE>
struct PointOps;

template <>
struct PointOps<WB2_MODE_WIND> {
  static constexpr int NIN = 4, NQ = 1;
  template <typename V>
  static __device__ __forceinline__ void elementwise(const V (&in)[NIN],
                                                     V (&q)[NQ]) {
    const V du = in[0] - in[1];
    const V dv = in[2] - in[3];
    q[0] = du * du + dv * dv;  // metrics.py:195-197
  }
  template <bool SKIPNA, typename T>
  static __device__ __forceinline__ void slots(
      const T (&q)[NQ], double (&x)[ModeTraits<WB2_MODE_WIND, SKIPNA>::K]) {
    if constexpr (SKIPNA) {
      const bool ok = !is_nan(q[0]);
      x[0] = (double)(ok ? q[0] : (T)0);
      x[1] = ok ? 1.0 : 0.0;
    } else {
      x[0] = (double)q[0];
    }
  }
};

template <>
struct PointOps<WB2_MODE_DET> {
  static constexpr int NIN = 2, NQ = 2;
  template <typename V>
  static __device__ __forceinline__ void elementwise(const V (&in)[NIN],
                                                     V (&q)[NQ]) {
    const V d = in[0] - in[1];  // metrics.py:264, 284, 329, 358
    q[0] = d;
    q[1] = d * d;  // MSE / RMSE
  }
  // MAE: |d| converted == the converted d with its sign cleared, which the
  // float64 FMA takes as a source modifier (no instruction of its own).
  template <bool SKIPNA, typename T>
  static __device__ __forceinline__ void slots(
      const T (&q)[NQ], double (&x)[ModeTraits<WB2_MODE_DET, SKIPNA>::K]) {
    if constexpr (SKIPNA) {
      // the selects act on the input dtype (one v_cndmask for float32, not
      // the two of a float64 select); the conversion of 0 is exact
      const T okd = !is_nan(q[0]) ? (T)1 : (T)0;
      x[0] = (double)keep_if(q[0], okd);
      x[1] = __builtin_fabs(x[0]);
      x[2] = (double)keep_if(q[1], okd);
      x[3] = (double)okd;
    } else {
      x[0] = (double)q[0];
      x[1] = __builtin_fabs(x[0]);
      x[2] = (double)q[1];
    }
  }
};

template <>
struct PointOps<WB2_MODE_DET_ACC> {
  static constexpr int NIN = 3, NQ = 5;
  template <typename V>
  static __device__ __forceinline__ void elementwise(const V (&in)[NIN],
                                                     V (&q)[NQ]) {
    const V d = in[0] - in[1];
    const V fa = in[0] - in[2];  // metrics.py:405
    const V ta = in[1] - in[2];  // metrics.py:406
    q[0] = d;
    q[1] = d * d;
    q[2] = fa * ta;
    q[3] = fa * fa;
    q[4] = ta * ta;
  }
  template <bool SKIPNA, typename T>
  static __device__ __forceinline__ void slots(
      const T (&q)[NQ], double (&x)[ModeTraits<WB2_MODE_DET_ACC, SKIPNA>::K]) {
    if constexpr (SKIPNA) {
      const T okd = !is_nan(q[0]) ? (T)1 : (T)0,
              okp = !is_nan(q[2]) ? (T)1 : (T)0,
              okf = !is_nan(q[3]) ? (T)1 : (T)0,
              okt = !is_nan(q[4]) ? (T)1 : (T)0;
      x[0] = (double)keep_if(q[0], okd);
      x[1] = __builtin_fabs(x[0]);
      x[2] = (double)keep_if(q[1], okd);
      x[3] = (double)keep_if(q[2], okp);
      x[4] = (double)keep_if(q[3], okf);
      x[5] = (double)keep_if(q[4], okt);
      x[6] = (double)okd;
      x[7] = (double)okp;
      x[8] = (double)okf;
      x[9] = (double)okt;
    } else {
      x[0] = (double)q[0];
      x[1] = __builtin_fabs(x[0]);
#pragma unroll
      for (int j = 1; j < NQ; ++j) x[1 + j] = (double)q[j];
    }
  }
};

// Returns true when a Gaussian mode met |z| beyond the erfcx table and the
// point has to be repeated with ERFCX_FAR = true (gauss_math.hpp).
template <int MODE, bool SKIPNA, typename T, bool ERFCX_FAR = false>
__device__ __forceinline__ bool eval_slots(
    const T (&in)[ModeTraits<MODE, SKIPNA>::NIN],
    double (&x)[ModeTraits<MODE, SKIPNA>::K], double aux = 0.0,
    double scalar = 0.0, const double* erfcx_lds = nullptr) {
  bool far = false;
  if constexpr (MODE == WB2_MODE_SEEPS) {
    // metrics.py:444-507: in = (forecast, truth, wet threshold at valid time),
    // aux = climatological dry fraction p1 (NaN where masked out), scalar = dry
    // threshold.  Categories: dry x < dry; light dry < x < wet; heavy x >= wet
    // (x == dry is in none: zero contingency row, score 0); NaN stays NaN.
    const T f = in[0], y = in[1], wet = in[2], dry = (T)scalar;
    // branch-free on purpose: written as nested conditions the category logic
    // became a switch under exec masks (4 saveexec + 2 branches per point, 36
    // scalar instructions per point; profiles/r06_round_log.md)
    auto cat = [&](T v) {
      int c = v >= wet ? 2 : 3;
      c = ((v > dry) & (v < wet)) ? 1 : c;
      c = v < dry ? 0 : c;
      return c;
    };
    const int fc = cat(f), tc = cat(y);
    const unsigned cell = (unsigned)(fc * 4 + tc);  // [forecast_cat][truth_cat]
    const T p1 = (T)aux, one = (T)1;
    // 0.5 * scoring matrix [forecast_cat][truth_cat], in the dtype of p1:
    //   [0][1] 1 / (1 - p1)   [0][2] 4 / (1 - p1)   [1][0] 1 / p1
    //   [1][2] 3 / (1 - p1)   [2][1] 3 / (2 + p1)   [2][0] 1 / p1 + 3 / (2 + p1)
    // as TWO IEEE divisions per point -- numerator and denominator of the one
    // quotient a cell needs are selected first (the quotients themselves are
    // the reference's, bit for bit), 3 / (2 + p1) serves [2][1] and [2][0] --
    // instead of the five a select over ready-made entries evaluates.  Which
    // quotients a cell takes: two 16-bit tables indexed by the cell.
    const bool to_dry = tc == 0;  // [1][0], [2][0]: 1 / p1
    const T num = to_dry ? one
                         : (fc == 0 ? (tc == 1 ? one : (T)4) : (T)3);
    const T den = to_dry ? p1 : one - p1;
    const T d1 = num / den;
    const T d2 = (T)3 / ((T)2 + p1);
    // d1: cells [0][1] [0][2] [1][0] [1][2] [2][0]; d2: cells [2][0] [2][1]
    const T a = ((0x0156u >> cell) & 1u) ? d1 : (T)0;
    const T b = ((0x0300u >> cell) & 1u) ? d2 : (T)0;
    const T m = a + b;  // (x + 0 is x; [2][0]: d1 + d2 as before)
    // NaN in -> NaN out (the quotients themselves are finite: p1 is masked to
    // (min_p1, max_p1) or NaN), selected in the input dtype (one v_cndmask, not
    // the two of a float64 select)
    const T half_m = (T)0.5 * m;
    const bool bad = is_nan(f) | is_nan(y) | is_nan(aux) | is_nan(half_m);
    if constexpr (SKIPNA) {
      const T okf = bad ? (T)0 : (T)1;
      x[0] = (double)keep_if(half_m, okf);
      x[1] = (double)okf;
    } else {
      x[0] = (double)(bad ? (T)__builtin_nanf("") : half_m);
    }
  } else if constexpr (MODE == WB2_MODE_GAUSS_THR) {
    // metrics.py:975-1000 (Brier), :1043-1066 (ignorance), :1104-1121 (RPS
    // part): in = (mean, std, truth, threshold).  xr.where(truth > thr, 1., 0.)
    // maps a NaN truth to 0; the normalised threshold is formed in the input
    // dtype and scipy's norm.cdf promotes it to float64.
    const T mean = in[0], sd = in[1], y = in[2], thr = in[3];
    const T nt = (thr - mean) / sd;
    double cdf, pdf_unused;
    far = normal_cdf_pdf<ERFCX_FAR>((double)nt, erfcx_lds, cdf, pdf_unused);
    const bool above = y > thr, below = y < thr;
    const double tp = above ? 1.0 : 0.0, te = below ? 1.0 : 0.0;
    const double db = (1.0 - cdf) - tp;
    const double dr = cdf - te;
        const double v[3] = {db * db, -log_unit(above ? 1.0 - cdf : cdf), dr * dr};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if constexpr (SKIPNA) {
        const bool ok = !is_nan(v[k]);
        x[k] = ok ? v[k] : 0.0;
        x[3 + k] = ok ? 1.0 : 0.0;
      } else {
        x[k] = v[k];
      }
    }
  } else if constexpr (MODE == WB2_MODE_GAUSS) {
    // metrics.py:895-905, 925-927: in = (mean, std, truth).  The normalised
    // difference is formed in the input dtype; scipy's norm.cdf / norm.pdf
    // promote it to float64, and so does everything downstream.
    const T mean = in[0], sd = in[1], y = in[2];
    const T nd = (mean - y) / sd;
    const double z = (double)nd;
    double cdf, pdf;
    far = normal_cdf_pdf<ERFCX_FAR>(z, erfcx_lds, cdf, pdf);
    const double crps =
        (double)sd * (z * (2.0 * cdf - 1.0) + 2.0 * pdf - 0.56418958354775628695);
    const T var = sd * sd;
    if constexpr (SKIPNA) {
      const bool okc = !is_nan(crps), okv = !is_nan(var);
      x[0] = okc ? crps : 0.0;
      x[1] = okv ? (double)var : 0.0;
      x[2] = okc ? 1.0 : 0.0;
      x[3] = okv ? 1.0 : 0.0;
    } else {
      x[0] = crps;
      x[1] = (double)var;
    }
  } else {
    // DET / DET_ACC / WIND: the elementwise stage, then the slots
    T q[PointOps<MODE>::NQ];
    PointOps<MODE>::template elementwise<T>(in, q);
    PointOps<MODE>::template slots<SKIPNA, T>(q, x);
  }
  return far;
}

// Tuning knobs (defaults are the measured best; see profiles/NOTES.md).
#ifndef WB2_U_ROWS
#define WB2_U_ROWS 2
#endif
#ifndef WB2_U_ROWS_NARROW
#define WB2_U_ROWS_NARROW 8   // rows per batch when a lane's load is < 16 bytes
#endif
#ifndef WB2_F32_VEC
#define WB2_F32_VEC 4         // float32 columns per lane (16-byte loads)
#endif
#ifndef WB2_F32_VEC_HEAVY
// float32 columns per lane of the register-heavy DET_ACC instantiations (a 2-D
// weight field: two accumulator sets; skipna: 10 slots).  Round 3 first halved
// it (4 columns: 150-252 VGPRs = 2-3 waves per SIMD, 0.43-0.46 of the HBM peak;
// 2 columns: 0.51-0.57); with the elementwise stage packed, the fold a halving
// tree and |d| a source modifier the 4-column kernels need 122 / 157 VGPRs and
// are the faster ones again (same box, interleaved: weight field 0.455-0.462 ms
// against 0.463-0.500, skipna 0.445-0.447 against 0.449-0.461;
// profiles/r03_k1_ab7_summary.txt) -- 1 KB instead of 512 B per wave and row.
#define WB2_F32_VEC_HEAVY 4
#endif
#ifndef WB2_NT_LOADS
#define WB2_NT_LOADS 1
#endif
#ifndef WB2_K1_RING_DEFAULT
#define WB2_K1_RING_DEFAULT 0   // rows per wave in flight of the ring form
#endif
#ifndef WB2_GAUSS_GROUP
// Gaussian modes: points of a lane's load evaluated as one straight-line block
// (2: two dependent fp64 chains interleaved at 114 VGPRs; 4 costs a wave per SIMD)
#define WB2_GAUSS_GROUP 2
#endif
#ifndef WB2_PACK_PAIRS
// 1: float32 columns through the elementwise stage in pairs (v_pk_*_f32).
#define WB2_PACK_PAIRS 1
#endif

#if WB2_NT_LOADS
#define WB2_LOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define WB2_LOAD(ptr) (*(ptr))
#endif

#define WB2_GLOBAL __attribute__((address_space(1)))

template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const WB2_GLOBAL T* p, T (&v)[VEC]) {
  if constexpr (VEC == 1) {
    v[0] = WB2_LOAD(p);
  } else {
    // Element alignment only: rows of an odd length (721 latitudes last) start
    // anywhere; gfx950 takes the same global_load_dwordx4 either way.
    typedef T V0 __attribute__((ext_vector_type(VEC)));
    typedef V0 V __attribute__((aligned(sizeof(T))));
    const V0 x = WB2_LOAD(reinterpret_cast<const WB2_GLOBAL V*>(p));
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = x[e];
  }
}

// FT: element type of the 2-D weight field in memory.  A field whose values
// are float32 numbers (an ERA5 land-sea mask, a thresholded mask) is kept as
// float32: half the bytes through L2 and L1 per point, the same float64 value
// after the (exact) conversion -- the 16 official regions 0.706 -> 0.739 of
// the HBM peak on one box.
template <int VEC, typename FT>
__device__ __forceinline__ void load_wf(const WB2_GLOBAL FT* p,
                                        double (&v)[VEC]) {
  // The weight field is re-read by every outer slab: keep it cacheable.
  typedef FT V0 __attribute__((ext_vector_type(VEC)));
  typedef V0 V __attribute__((aligned(sizeof(FT))));
  const V0 x = *reinterpret_cast<const WB2_GLOBAL V*>(p);
#pragma unroll
  for (int e = 0; e < VEC; ++e) v[e] = (double)x[e];
}

// LDS-DMA (gfx950 `global_load_lds_dwordx4`): 16 bytes per lane straight from
// global memory into LDS at M0 + 16 * lane, no VGPR in between.  `row` is the
// wave-uniform row pointer (an SGPR pair), `voff` the lane's byte offset in the
// row, `lds_dst` the wave-uniform LDS byte address of the 1 KiB destination.
// hipcc does not count these loads (inline asm): the ring kernel waits for
// them itself (ring_wait).
template <bool NT>
__device__ __forceinline__ void glds16(unsigned long long row, unsigned voff,
                                       unsigned lds_dst) {
  unsigned keep;
  if constexpr (NT) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(row), "s"(lds_dst)
        : "memory");
  } else {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(row), "s"(lds_dst)
        : "memory");
  }
}
// s_waitcnt vmcnt(n) for a wave-uniform n in [0, 23]
__device__ __forceinline__ void ring_wait(int n) {
  switch (n) {
#define WB2_RING_WAIT(N) \
  case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    WB2_RING_WAIT(0) WB2_RING_WAIT(1) WB2_RING_WAIT(2) WB2_RING_WAIT(3)
    WB2_RING_WAIT(4) WB2_RING_WAIT(5) WB2_RING_WAIT(6) WB2_RING_WAIT(7)
    WB2_RING_WAIT(8) WB2_RING_WAIT(9) WB2_RING_WAIT(10) WB2_RING_WAIT(11)
    WB2_RING_WAIT(12) WB2_RING_WAIT(13) WB2_RING_WAIT(14) WB2_RING_WAIT(15)
    WB2_RING_WAIT(16) WB2_RING_WAIT(17) WB2_RING_WAIT(18) WB2_RING_WAIT(19)
    WB2_RING_WAIT(20) WB2_RING_WAIT(21) WB2_RING_WAIT(22) WB2_RING_WAIT(23)
#undef WB2_RING_WAIT
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

// Geometry.  Workgroups are dealt to the 8 XCDs round-robin by their linear
// index, and the chunks of a slab differ in size (bands cut them at region
// boundaries, the last one is short, the padding to a multiple of 8 is empty).
//   field-free instantiations: blockIdx.x = tile_block * n_chunk + chunk',
//     blockIdx.y/z = outer slab, chunk = (chunk' + slab) % n_chunk.  Without the
//     rotation chunk c of EVERY slab lands on XCD c % 8 (n_chunk % 8 == 0) and
//     some XCDs get all the short chunks: same box, bench.py --variants-only,
//     plain / rotated / slab-fastest: the benched launch 0.781 / 0.802 / 0.789
//     of the HBM peak, the lon-lat layout 0.678 / 0.757 / 0.711, wind 0.788 /
//     0.826 / 0.818 (profiles/r04_xcd_balance.md).
//   a 2-D weight field or the SEEPS p1 field: the outer SLAB is the fastest
//     grid dim, so the workgroups resident at any moment work on the SAME few
//     row chunks of different slabs and re-read the same ~1 MB of the field
//     from L2 (chunk-fastest: re-fetched through the fabric for every slab,
//     1.6 x the algorithmic bytes, profiles/r03_k1_variants.md); the XCDs each
//     take slabs of their own, balanced whatever the chunk sizes.
// Every WAVE
// of the workgroup owns one column tile of 64*VEC columns and is completely
// independent of the others (no LDS, no barrier): waves stream, then fold their
// own columns into the segs that intersect their tile with a wave64 butterfly.
//
// Rows of one chunk are processed U at a time so that U*NIN 16-byte loads per
// lane are in flight before the first one is consumed.  The prologue is written
// branch-free on purpose: every scalar (table / slab-index) load is issued
// before the first wait, instead of one dependent round trip per table.
//
// RING > 0 (float32, 4 columns per lane, point-op modes, no skipna): the rows
// of the chunk come through a ring of RING row stages per wave in LDS, filled
// by LDS-DMA -- row r + RING is requested when row r has been read out of its
// stage, so RING rows of NIN (+ 1: the float32 weight field) x 1 KiB per wave
// are in flight ALL the time and none of them holds a VGPR.  The batch form
// has its U rows in flight only while it waits for them (then it computes with
// nothing outstanding), and what it may keep in flight is bounded by registers:
// the land-mask instantiation (157 VGPRs, 3 waves per SIMD) has 72 KB per CU
// outstanding at best.  Same loads per lane, same arithmetic in the same order:
// the same bits.
template <typename T, int VEC, int MODE, bool SKIPNA, bool WF,
          bool SG = false, typename FT = double, int RING = 0>
__global__ void __launch_bounds__(512)
    stream_partials_kernel(const StreamParams p) {
  using M = ModeTraits<MODE, SKIPNA>;
  constexpr int NIN = M::NIN, K = M::K, NWF = WF ? 2 : 1;
  // rows per batch: the same bytes in flight per lane for 16- and 8-byte loads
  constexpr int U = (sizeof(T) * VEC >= 16)
                        ? (NIN >= 4 ? WB2_U_ROWS / 2 : WB2_U_ROWS)
                        : (sizeof(T) * VEC == 8 ? 2 * WB2_U_ROWS
                                                : WB2_U_ROWS_NARROW);
  constexpr int TILE = kWave * VEC;

  const int lane = threadIdx.x & (kWave - 1);
  // readfirstlane: tell the compiler the wave index is wave-uniform (SGPR).
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int nwave = blockDim.x / kWave;
  constexpr bool SLAB_FASTEST = WF || MODE == WB2_MODE_SEEPS;
  const unsigned bx = SLAB_FASTEST ? blockIdx.y : blockIdx.x;
  const unsigned tblk = bx / (unsigned)p.n_chunk;
  const long long o =
      SLAB_FASTEST ? (long long)blockIdx.z * gridDim.x + blockIdx.x
                   : (long long)blockIdx.z * gridDim.y + blockIdx.y;
  int chunk = (int)(bx - tblk * (unsigned)p.n_chunk);
  if (!SLAB_FASTEST && WB2_ROTATE_CHUNKS)
    chunk = (int)(((long long)chunk + o) % p.n_chunk);
  const int tile = (int)tblk * nwave + wave;

  // ---- branch-free prologue: issue every scalar load before any wait ----
  const int row0 = p.chunk_row0[chunk];
  const int nrow = p.chunk_nrow[chunk];
  long long slab_idx[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    // Identity when no table is given; the dummy read keeps the load
    // unconditional (chunk tables are >= 8 ints, i.e. >= 4 readable int64).
    const long long* tab =
        p.slab[i] ? p.slab[i] : reinterpret_cast<const long long*>(p.chunk_row0);
    const long long v = tab[(p.slab[i] && o < p.n_outer) ? o : 0];
    slab_idx[i] = p.slab[i] ? v : o;
  }
  const int col0 = tile * TILE + lane * VEC;  // first column the lane OWNS
  const bool active = tile < p.n_ctile && col0 < p.n_col;
  // A lane whose VEC columns would run past the end of the row (n_col % VEC
  // != 0) loads the row's LAST VEC columns instead; the columns below col0,
  // which its neighbour owns, are dropped when the sums are folded.
  const int colb = (VEC > 1 && col0 + VEC > p.n_col) ? p.n_col - VEC : col0;
  // Gaussian modes: the erfcx table (gauss_math.hpp) into LDS, by the whole
  // workgroup, before any wave leaves
  [[maybe_unused]] const double* erfcx_lds = nullptr;
  if constexpr (MODE == WB2_MODE_GAUSS || MODE == WB2_MODE_GAUSS_THR) {
    __shared__ __attribute__((aligned(16))) double erfcx_table[kErfcxDoubles];
    load_erfcx_table(erfcx_table);
    erfcx_lds = erfcx_table;
  }
  if (nrow <= 0 || tile >= p.n_ctile || o >= p.n_outer) return;  // wave-uniform

  double acc[NWF][VEC][K];
#pragma unroll
  for (int w = 0; w < NWF; ++w)
#pragma unroll
    for (int e = 0; e < VEC; ++e)
#pragma unroll
      for (int k = 0; k < K; ++k) acc[w][e][k] = 0.0;

  if (active) {
    // base[i] / wfp: wave-uniform row pointers (scalar adds advance them).  Two
    // ways to add the lane's columns (template parameter SG):
    //   SG = false  plain pointer arithmetic -- hipcc adds the row offset to a
    //               64-bit lane offset once and pays one v_lshl_add_u64 per
    //               load for its base: the faster form on 16-byte aligned rows;
    //   SG = true   the row pointer pinned to an SGPR pair and the lane's
    //               columns as ONE unsigned 32-bit byte offset, i.e. `global_load
    //               v_off, s[base:base+1]`: 4.5 % faster on the unaligned rows
    //               of the lon-lat layout, 2-20 % slower on aligned ones (each
    //               load waits for the scalar add that makes its base).
    const T* base[NIN];
#pragma unroll
    for (int i = 0; i < NIN; ++i)
      base[i] = reinterpret_cast<const T*>(
                    static_cast<const char*>(p.in[i]) +
                    slab_idx[i] * p.slab_step_bytes) +
                (long long)row0 * p.n_col;
    const FT* wfp =
        WF ? static_cast<const FT*>(p.wfield) + (long long)row0 * p.n_col
           : nullptr;
    const double* wrp = p.w_row + row0;
    unsigned lane_off = (unsigned)colb * (unsigned)sizeof(T);
    unsigned lane_off_wf = (unsigned)colb * (unsigned)sizeof(FT);
    auto at = [&](const T* row) {
      if constexpr (SG) {
        unsigned long long rp = reinterpret_cast<unsigned long long>(row);
        asm volatile("" : "+s"(rp));
        return reinterpret_cast<const WB2_GLOBAL T*>(
            reinterpret_cast<const WB2_GLOBAL char*>(rp) + lane_off);
      } else {
        return reinterpret_cast<const WB2_GLOBAL T*>(
            reinterpret_cast<unsigned long long>(row + colb));
      }
    };
    auto at_wf = [&](const FT* row) {
      if constexpr (SG) {
        unsigned long long rp = reinterpret_cast<unsigned long long>(row);
        asm volatile("" : "+s"(rp));
        return reinterpret_cast<const WB2_GLOBAL FT*>(
            reinterpret_cast<const WB2_GLOBAL char*>(rp) + lane_off_wf);
      } else {
        return reinterpret_cast<const WB2_GLOBAL FT*>(
            reinterpret_cast<unsigned long long>(row + colb));
      }
    };
    // SG: the lane offset opaque to the optimiser inside the loop -- hoisted,
    // its zero-extension becomes a loop-invariant 64-bit VGPR pair and
    // instruction selection (per block) no longer sees a 32-bit offset: back
    // to one v_lshl_add_u64 per load.
    auto pin_offsets = [&]() {
      if constexpr (SG) {
        asm volatile("" : "+v"(lane_off));
        if constexpr (WF) asm volatile("" : "+v"(lane_off_wf));
      }
    };

    // One point's K values into the accumulators (both weight sets).
    auto accumulate = [&](int e, const double (&x)[K], double wfe, double wr) {
#pragma unroll
      for (int k = 0; k < K; ++k)
        acc[0][e][k] = __builtin_fma(wr, x[k], acc[0][e][k]);
      if constexpr (WF) {
        // metrics.py:159-160: values where the weight is not > 0 become 0.
        // The field is >= 0 and finite (plan.py rejects anything else), so
        // outside w2 = wr * 0 = 0 and the slot only has to be FINITE to drop
        // out of the sum: clearing the high dword (sign, exponent, top of the
        // mantissa) of the float64 slot does that in ONE 32-bit AND instead
        // of the two v_cndmask of a 64-bit select.
#ifdef WB2_DIAG_WF_NOFMA   // timing diagnostic only: the field is loaded, not used
        acc[1][e][0] += wfe;
        return;
#endif
        const bool inside = wfe > 0.0;
        const unsigned long long keep = inside ? ~0ull : 0x00000000ffffffffull;
        const double w2 = wr * wfe;
        double xs[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if ((MODE == WB2_MODE_DET || MODE == WB2_MODE_DET_ACC) && k == 1) {
            xs[1] = __builtin_fabs(xs[0]);  // slot 1 is |slot 0| (PointOps)
          } else {
            xs[k] = __builtin_bit_cast(
                double, __builtin_bit_cast(unsigned long long, x[k]) & keep);
          }
          acc[1][e][k] = __builtin_fma(w2, xs[k], acc[1][e][k]);
        }
      }
    };
    constexpr bool POINT_OPS = MODE == WB2_MODE_DET ||
                               MODE == WB2_MODE_DET_ACC ||
                               MODE == WB2_MODE_WIND;
    // float32 columns in pairs: the elementwise stage as packed instructions
    constexpr bool PAIRS =
        WB2_PACK_PAIRS && POINT_OPS && sizeof(T) == 4 && VEC % 2 == 0;

    auto consume = [&](const T (&v)[NIN][VEC], const double (&wf)[VEC],
                       double wr, const double* auxrow) {
      if constexpr (PAIRS) {
        typedef T V2 __attribute__((ext_vector_type(2)));
        using Ops = PointOps<POINT_OPS ? MODE : WB2_MODE_DET>;
#pragma unroll
        for (int e = 0; e < VEC; e += 2) {
          V2 in2[NIN], q2[Ops::NQ];
#pragma unroll
          for (int i = 0; i < NIN; ++i) {
            in2[i][0] = v[i][e];
            in2[i][1] = v[i][e + 1];
          }
          Ops::template elementwise<V2>(in2, q2);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            T q[Ops::NQ];
#pragma unroll
            for (int j = 0; j < Ops::NQ; ++j) q[j] = q2[j][h];
            double x[K];
            Ops::template slots<SKIPNA, T>(q, x);
            accumulate(e + h, x, wf[e + h], wr);
          }
        }
      } else if constexpr (MODE == WB2_MODE_GAUSS ||
                           MODE == WB2_MODE_GAUSS_THR) {
        // the lane's points as ONE straight-line block (their dependent fp64
        // chains interleave); a point beyond the erfcx table is repeated with
        // the series behind a wave-uniform branch
        constexpr int G = VEC < WB2_GAUSS_GROUP ? VEC : WB2_GAUSS_GROUP;
#pragma unroll
        for (int e0 = 0; e0 < VEC; e0 += G) {
          double x[G][K];
          bool far[G], any = false;
#pragma unroll
          for (int g = 0; g < G; ++g) {
            T in[NIN];
#pragma unroll
            for (int i = 0; i < NIN; ++i) in[i] = v[i][e0 + g];
            far[g] = eval_slots<MODE, SKIPNA, T>(in, x[g], 0.0, 0.0, erfcx_lds);
            any = any || far[g];
          }
          if (__builtin_amdgcn_ballot_w64(any) != 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
              if (far[g]) {
                T in[NIN];
#pragma unroll
                for (int i = 0; i < NIN; ++i) in[i] = v[i][e0 + g];
                eval_slots<MODE, SKIPNA, T, true>(in, x[g], 0.0, 0.0,
                                                  erfcx_lds);
              }
            }
          }
#pragma unroll
          for (int g = 0; g < G; ++g)
            accumulate(e0 + g, x[g], wf[e0 + g], wr);
        }
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          T in[NIN];
#pragma unroll
          for (int i = 0; i < NIN; ++i) in[i] = v[i][e];
          double x[K];
          if constexpr (MODE == WB2_MODE_SEEPS) {
            eval_slots<MODE, SKIPNA, T>(in, x, auxrow[e], p.scalar);
          } else {
            eval_slots<MODE, SKIPNA, T>(in, x);
          }
          accumulate(e, x, wf[e], wr);
        }
      }
    };

    // A batch = U rows.  issue() puts the batch's loads in flight, eat() folds
    // it into the accumulators.
    struct Batch {
      T v[U][NIN][VEC];
      double wf[U][VEC];
      double wr[U];
      double ax[U][VEC];  // SEEPS: the p1 field of the row (with the batch's
    };                    // loads, as vectors: it was one 8-byte load per point)
    auto issue = [&](Batch& bt, int r) {
      pin_offsets();
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int i = 0; i < NIN; ++i)
          load_vec<T, VEC>(at(base[i] + (long long)(r + u) * p.n_col),
                           bt.v[u][i]);
        if constexpr (WF) {
#ifdef WB2_DIAG_WF_SAMEROW   // timing diagnostic only: every row reads field row 0
          load_wf<VEC, FT>(at_wf(wfp + (long long)(u) * p.n_col), bt.wf[u]);
#else
          load_wf<VEC, FT>(at_wf(wfp + (long long)(r + u) * p.n_col), bt.wf[u]);
#endif
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) bt.wf[u][e] = 1.0;
        }
        bt.wr[u] = wrp[r + u];
        if constexpr (MODE == WB2_MODE_SEEPS)
          load_wf<VEC, double>(
              reinterpret_cast<const WB2_GLOBAL double*>(
                  reinterpret_cast<unsigned long long>(
                      p.aux + (long long)(row0 + r + u) * p.n_col + colb)),
              bt.ax[u]);
      }
    };
    auto eat = [&](const Batch& bt, int r) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        consume(bt.v[u], bt.wf[u], bt.wr[u],
                MODE == WB2_MODE_SEEPS ? bt.ax[u] : nullptr);
    };
    int r = 0;
    if constexpr (RING > 0) {
      static_assert(sizeof(T) == 4 && VEC == 4 && POINT_OPS && !SKIPNA && !SG,
                    "ring kernel: float32 x 4 columns, point-op modes");
      static_assert(!WF || sizeof(FT) == 4, "ring kernel: float32 field");
      constexpr int NSLOT = NIN + (WF ? 1 : 0);
      static_assert((RING - 1) * NSLOT <= 23, "ring_wait covers 0..23");
      extern __shared__ __attribute__((aligned(16))) char ring_lds[];
      typedef __attribute__((address_space(3))) char* LdsPtr;
      typedef T V4 __attribute__((ext_vector_type(4)));
      const LdsPtr mine = (LdsPtr)ring_lds + wave * (RING * NSLOT * 1024);
      const unsigned mine_addr = (unsigned)(unsigned long long)mine;
      const unsigned voff = (unsigned)colb * 4u;
      auto request = [&](int row) {   // wave-uniform row < nrow
        const unsigned dst = mine_addr + (unsigned)(row % RING) * (NSLOT * 1024);
#pragma unroll
        for (int i = 0; i < NIN; ++i)
          glds16<WB2_NT_LOADS != 0>(
              reinterpret_cast<unsigned long long>(
                  base[i] + (long long)row * p.n_col),
              voff, dst + i * 1024);
        if constexpr (WF)
          glds16<false>(reinterpret_cast<unsigned long long>(
                            wfp + (long long)row * p.n_col),
                        voff, dst + NIN * 1024);
      };
      int requested = nrow < RING ? nrow : RING;
      for (int q = 0; q < requested; ++q) request(q);
#pragma clang loop unroll(disable)
      for (; r < nrow; ++r) {
        // rows r + 1 .. requested - 1 may still be on their way
        ring_wait((requested - r - 1) * NSLOT);
        const LdsPtr st = mine + (r % RING) * (NSLOT * 1024) + lane * 16;
        T v[NIN][VEC];
        double wf[VEC];
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
          const V4 x = *reinterpret_cast<
              const __attribute__((address_space(3))) V4*>(st + i * 1024);
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[i][e] = x[e];
        }
        if constexpr (WF) {
          const V4 x = *reinterpret_cast<
              const __attribute__((address_space(3))) V4*>(st + NIN * 1024);
#pragma unroll
          for (int e = 0; e < VEC; ++e) wf[e] = (double)x[e];
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) wf[e] = 1.0;
        }
        // the row weight by a scalar load of our own (behind the asm memory
        // clobbers hipcc turns wrp[r] into a VECTOR load and waits vmcnt(0)
        // for it: the ring would drain every row)
        double wr;
        asm volatile("s_load_dwordx2 %0, %1, 0x0"
                     : "=s"(wr)
                     : "s"(wrp + r)
                     : "memory");
        // the stage (and wr) in registers before its next row is requested
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(wr) : : "memory");
        if (requested < nrow) request(requested++);
        consume(v, wf, wr, nullptr);
      }
    }
#pragma clang loop unroll(disable)
    for (; r + U <= nrow; r += U) {
      Batch bt;
      issue(bt, r);
      // Keep every load of the batch in flight before the first use: without
      // this hipcc sinks half of them below the arithmetic (register heuristics).
      __builtin_amdgcn_sched_barrier(0);
      eat(bt, r);
    }
#pragma clang loop unroll(disable)
    for (; r < nrow; ++r) {
      T v[NIN][VEC];
      double wf[VEC];
      pin_offsets();
#pragma unroll
      for (int i = 0; i < NIN; ++i)
        load_vec<T, VEC>(at(base[i] + (long long)r * p.n_col), v[i]);
      if constexpr (WF) {
        load_wf<VEC, FT>(at_wf(wfp + (long long)r * p.n_col), wf);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) wf[e] = 1.0;
      }
      consume(v, wf, wrp[r],
              MODE == WB2_MODE_SEEPS
                  ? p.aux + (long long)(row0 + r) * p.n_col + colb
                  : nullptr);
    }
    if (p.w_col) {  // only when columns are latitudes (lon-lat layout)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const double wc = p.w_col[colb + e];
#pragma unroll
        for (int w = 0; w < NWF; ++w)
#pragma unroll
          for (int k = 0; k < K; ++k) acc[w][e][k] *= wc;
      }
    }
  }

  // ---- fold owned columns into the segs intersecting this wave's tile ----
  fold_tile_to_segs<NWF, VEC, K>(
      acc, lane, tile, colb, col0, p.n_col, p.seg_col0, p.seg_eoff, p.n_seg,
      p.n_ts,
      p.partials + (o * p.n_chunk + chunk) * (long long)(NWF * p.n_ts * K));
}

// ---------------------------------------------------------------------------
// K1p: the wind-vector pairs of a launch from the SAME read as their
// per-variable metrics.
//
// The reference forms diff = forecast - truth once; MSE.compute_chunk derives
// the per-variable numbers AND the wind-vector numbers (du^2 + dv^2, float32)
// from it (metrics.py:283-301 calling :194-201).  A second K1 launch in mode
// WIND re-read u and v: 233 MB per 1 067 MB official chunk.  Here the u slab
// and the v slab of a pair are streamed by the TWO waves of one workgroup, over
// the same row chunk and column tile: each wave runs the per-variable pass of
// its slab exactly as stream_partials_kernel does (same loads, same arithmetic,
// same accumulation order, same fold: the same bits); the u wave hands its
// float32 d^2 to the v wave through LDS (16 B per lane and row), and the v wave
// adds its own d^2 in float32 -- metrics.py:195-197 -- and keeps the wind slot
// the WIND launch kept.  One workgroup barrier per row batch, double-buffered.
// 13 float64 sums per point live in two waves of 6 and 7: the register budget
// of the per-variable kernel, with 4 columns per lane and the same halving
// tree (any narrower tile would change the order of the additions).
//
// Slabs: the launch's slab list ends with the u slabs, then the v slabs, of its
// n_pair pairs (pair k: slabs first + k and first + n_pair + k); wind partials
// are [n_pair][n_chunk][nwf][n_ts][KW].
#ifndef WB2_PAIR_U
#define WB2_PAIR_U WB2_U_ROWS   // rows per batch of the pair kernel
#endif
#ifndef WB2_PAIR_WAVES
#define WB2_PAIR_WAVES 0        // > 0: waves per SIMD the register budget is cut to
#endif
#ifndef WB2_PAIR_RELOAD
// 1: no LDS hand-off and no barrier -- the v wave loads the u slab's forecast
// and truth itself (the u wave of the same workgroup streams them at about the
// same time: L2 / L1 hits, no HBM traffic of their own if the caches hold) and
// forms du^2 with the u wave's arithmetic.  The two waves run free.
#define WB2_PAIR_RELOAD 0
#endif
#if WB2_PAIR_WAVES > 0
#define WB2_PAIR_OCCUPANCY \
  __attribute__((amdgpu_waves_per_eu(WB2_PAIR_WAVES, WB2_PAIR_WAVES)))
#else
#define WB2_PAIR_OCCUPANCY
#endif

struct PairParams {
  double* wind_partials;
  long long first;
  long long n_pair;
};

template <typename T, int VEC, bool ACC, bool SKIPNA, bool WF,
          typename FT = double>
__global__ void __launch_bounds__(128) WB2_PAIR_OCCUPANCY
    stream_pair_kernel(const StreamParams p, const PairParams pp) {
  constexpr int MODE = ACC ? WB2_MODE_DET_ACC : WB2_MODE_DET;
  using M = ModeTraits<MODE, SKIPNA>;
  using Ops = PointOps<MODE>;
  using MW = ModeTraits<WB2_MODE_WIND, SKIPNA>;
  constexpr int NIN = M::NIN, KD = M::K, KW = MW::K, NWF = WF ? 2 : 1;
  constexpr int U = WB2_PAIR_U;
  constexpr int TILE = kWave * VEC;
  constexpr bool PAIRS = WB2_PACK_PAIRS && sizeof(T) == 4 && VEC % 2 == 0;
  typedef T VT __attribute__((ext_vector_type(VEC)));
  // the u wave's d^2 of a row batch, [buffer][row][lane]
  __shared__ VT handoff[2][U][kWave];

  const int lane = threadIdx.x & (kWave - 1);
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const unsigned bx = WF ? blockIdx.y : blockIdx.x;
  const int tile = (int)(bx / (unsigned)p.n_chunk);
  int chunk = (int)(bx - (unsigned)tile * (unsigned)p.n_chunk);
  const long long pair = WF ? (long long)blockIdx.z * gridDim.x + blockIdx.x
                            : (long long)blockIdx.z * gridDim.y + blockIdx.y;
  if (!WF && WB2_ROTATE_CHUNKS)
    chunk = (int)(((long long)chunk + pair) % p.n_chunk);
  const bool live = pair < pp.n_pair;
  const long long o = pp.first + (live ? pair : 0) + role * pp.n_pair;

  const int row0 = p.chunk_row0[chunk];
  const int nrow = p.chunk_nrow[chunk];
  long long slab_idx[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) slab_idx[i] = p.slab[i] ? p.slab[i][o] : o;
  const int col0 = tile * TILE + lane * VEC;
  // every lane loads (the loop holds a barrier: it is walked by whole waves);
  // a lane past the row end reads the row's last VEC columns and its sums are
  // dropped by the fold (col0 >= n_col is in no seg)
  const int colb = col0 + VEC > p.n_col ? p.n_col - VEC : col0;
  if (nrow <= 0 || !live) return;  // both waves of the workgroup alike

  const T* base[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i)
    base[i] = reinterpret_cast<const T*>(static_cast<const char*>(p.in[i]) +
                                         slab_idx[i] * p.slab_step_bytes) +
              (long long)row0 * p.n_col;
#if WB2_PAIR_RELOAD
  // the u slab's forecast and truth, for the v wave
  const T* ubase[2];
  {
    const long long ou = pp.first + pair;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long su = p.slab[i] ? p.slab[i][ou] : ou;
      ubase[i] = reinterpret_cast<const T*>(
                     static_cast<const char*>(p.in[i]) +
                     su * p.slab_step_bytes) +
                 (long long)row0 * p.n_col;
    }
  }
#endif
  const FT* wfp = WF ? static_cast<const FT*>(p.wfield) +
                           (long long)row0 * p.n_col
                     : nullptr;
  const double* wrp = p.w_row + row0;
  auto at = [&](const T* row) {
    return reinterpret_cast<const WB2_GLOBAL T*>(
        reinterpret_cast<unsigned long long>(row + colb));
  };
  auto at_wf = [&](const FT* row) {
    return reinterpret_cast<const WB2_GLOBAL FT*>(
        reinterpret_cast<unsigned long long>(row + colb));
  };

  // one point's K values into an accumulator set (stream_partials_kernel's
  // `accumulate`, for the per-variable slots DET = true and the wind slot)
  auto accumulate = [&](auto& acc, auto is_det, int e, const auto& x,
                        double wfe, double wr) {
    constexpr int K = sizeof(x) / sizeof(double);
#pragma unroll
    for (int k = 0; k < K; ++k)
      acc[0][e][k] = __builtin_fma(wr, x[k], acc[0][e][k]);
    if constexpr (WF) {
      const bool inside = wfe > 0.0;
      const unsigned long long keep = inside ? ~0ull : 0x00000000ffffffffull;
      const double w2 = wr * wfe;
      double xs[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (decltype(is_det)::value && k == 1) {
          xs[1] = __builtin_fabs(xs[0]);
        } else {
          xs[k] = __builtin_bit_cast(
              double, __builtin_bit_cast(unsigned long long, x[k]) & keep);
        }
        acc[1][e][k] = __builtin_fma(w2, xs[k], acc[1][e][k]);
      }
    }
  };
  // The per-variable pass of one row: slots accumulated.  The u wave (WIND =
  // false) returns its d^2 in `sq`; the v wave (WIND = true) receives the u
  // wave's there and adds the wind slot of the same point right behind its own
  // slots (nothing but the loads is live across the barrier).
  auto consume = [&](auto wind_c, auto& acc, auto& accw,
                     const T (&v)[NIN][VEC], const double (&wf)[VEC], double wr,
                     T (&sq)[VEC]) {
    constexpr bool WIND = decltype(wind_c)::value;
    auto point = [&](int e, const T (&q)[Ops::NQ]) {
      double x[KD];
      Ops::template slots<SKIPNA, T>(q, x);
      accumulate(acc, std::true_type{}, e, x, wf[e], wr);
      if constexpr (WIND) {
        const T qw[1] = {sq[e] + q[1]};  // du^2 + dv^2 in the input dtype
        double xw[KW];
        PointOps<WB2_MODE_WIND>::template slots<SKIPNA, T>(qw, xw);
        accumulate(accw, std::false_type{}, e, xw, wf[e], wr);
      } else {
        sq[e] = q[1];
      }
    };
    if constexpr (PAIRS) {
      typedef T V2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int e = 0; e < VEC; e += 2) {
        V2 in2[NIN], q2[Ops::NQ];
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
          in2[i][0] = v[i][e];
          in2[i][1] = v[i][e + 1];
        }
        Ops::template elementwise<V2>(in2, q2);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          T q[Ops::NQ];
#pragma unroll
          for (int j = 0; j < Ops::NQ; ++j) q[j] = q2[j][h];
          point(e + h, q);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        T in[NIN], q[Ops::NQ];
#pragma unroll
        for (int i = 0; i < NIN; ++i) in[i] = v[i][e];
        Ops::template elementwise<T>(in, q);
        point(e, q);
      }
    }
  };
  auto load_row = [&](int r, T (&v)[NIN][VEC], double (&wf)[VEC], double& wr) {
#pragma unroll
    for (int i = 0; i < NIN; ++i)
      load_vec<T, VEC>(at(base[i] + (long long)r * p.n_col), v[i]);
    if constexpr (WF) {
      load_wf<VEC, FT>(at_wf(wfp + (long long)r * p.n_col), wf);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) wf[e] = 1.0;
    }
    wr = wrp[r];
  };

  double acc[NWF][VEC][KD];
  double accw[NWF][VEC][KW];
#pragma unroll
  for (int w = 0; w < NWF; ++w)
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
#pragma unroll
      for (int k = 0; k < KD; ++k) acc[w][e][k] = 0.0;
#pragma unroll
      for (int k = 0; k < KW; ++k) accw[w][e][k] = 0.0;
    }

  // Rows [r, r + n) of the chunk, n <= U, loads in flight together.  The u
  // wave consumes its batch, stores d^2 and meets the barrier; the v wave meets
  // the barrier with its loads still in flight, then consumes -- the u wave is
  // already loading its next batch.  Buffer `it & 1`: the u wave writes it
  // again two barriers later, after the v wave has read it.
  auto batch_u = [&](auto n_c, int r, int buf) {
    constexpr int N = decltype(n_c)::value;
    T v[N][NIN][VEC];
    double wf[N][VEC], wr[N];
#pragma unroll
    for (int u = 0; u < N; ++u) load_row(r + u, v[u], wf[u], wr[u]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < N; ++u) {
      T sq[VEC];
      consume(std::false_type{}, acc, accw, v[u], wf[u], wr[u], sq);
#if !WB2_PAIR_RELOAD
      VT out;
#pragma unroll
      for (int e = 0; e < VEC; ++e) out[e] = sq[e];
      handoff[buf][u][lane] = out;
#endif
    }
#if !WB2_PAIR_RELOAD
    __syncthreads();
#endif
  };
  auto batch_v = [&](auto n_c, int r, int buf) {
    constexpr int N = decltype(n_c)::value;
    T v[N][NIN][VEC];
    double wf[N][VEC], wr[N];
#pragma unroll
    for (int u = 0; u < N; ++u) load_row(r + u, v[u], wf[u], wr[u]);
#if WB2_PAIR_RELOAD
    T uu[N][2][VEC];
#pragma unroll
    for (int u = 0; u < N; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        load_vec<T, VEC>(at(ubase[i] + (long long)(r + u) * p.n_col),
                         uu[u][i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < N; ++u) {
      T sq[VEC];
      if constexpr (PAIRS) {   // the u wave's own arithmetic (packed)
        typedef T V2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int e = 0; e < VEC; e += 2) {
          V2 a, b;
          a[0] = uu[u][0][e];
          a[1] = uu[u][0][e + 1];
          b[0] = uu[u][1][e];
          b[1] = uu[u][1][e + 1];
          const V2 d = a - b;
          const V2 q = d * d;
          sq[e] = q[0];
          sq[e + 1] = q[1];
        }
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const T d = uu[u][0][e] - uu[u][1][e];
          sq[e] = d * d;
        }
      }
      consume(std::true_type{}, acc, accw, v[u], wf[u], wr[u], sq);
    }
#else
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < N; ++u) {
      const VT theirs = handoff[buf][u][lane];
      T sq[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) sq[e] = theirs[e];
      consume(std::true_type{}, acc, accw, v[u], wf[u], wr[u], sq);
    }
#endif
  };
  int r = 0, it = 0;
  if (role == 0) {
#pragma clang loop unroll(disable)
    for (; r + U <= nrow; r += U, ++it)
      batch_u(std::integral_constant<int, U>{}, r, it & 1);
#pragma clang loop unroll(disable)
    for (; r < nrow; ++r, ++it)
      batch_u(std::integral_constant<int, 1>{}, r, it & 1);
  } else {
#pragma clang loop unroll(disable)
    for (; r + U <= nrow; r += U, ++it)
      batch_v(std::integral_constant<int, U>{}, r, it & 1);
#pragma clang loop unroll(disable)
    for (; r < nrow; ++r, ++it)
      batch_v(std::integral_constant<int, 1>{}, r, it & 1);
  }

  if (p.w_col) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const double wc = p.w_col[colb + e];
#pragma unroll
      for (int w = 0; w < NWF; ++w) {
#pragma unroll
        for (int k = 0; k < KD; ++k) acc[w][e][k] *= wc;
#pragma unroll
        for (int k = 0; k < KW; ++k) accw[w][e][k] *= wc;
      }
    }
  }
  fold_tile_to_segs<NWF, VEC, KD>(
      acc, lane, tile, colb, col0, p.n_col, p.seg_col0, p.seg_eoff, p.n_seg,
      p.n_ts,
      p.partials + (o * p.n_chunk + chunk) * (long long)(NWF * p.n_ts * KD));
  if (role == 1)
    fold_tile_to_segs<NWF, VEC, KW>(
        accw, lane, tile, colb, col0, p.n_col, p.seg_col0, p.seg_eoff, p.n_seg,
        p.n_ts,
        pp.wind_partials +
            (pair * p.n_chunk + chunk) * (long long)(NWF * p.n_ts * KW));
}

// ---------------------------------------------------------------------------
// K2: (chunk, col-tile) partials -> band sums (LDS) -> region sums -> metrics
// ---------------------------------------------------------------------------
struct CombineParams {
  const double* partials;
  const int* seg_eoff;
  const int* band_chunk0;
  const double* coef_band;
  const double* coef_seg;
  const int* region_wf;
  const double* region_wsum;
  double* sums;
  double* metrics;
  long long n_outer;
  int n_chunk, nwf, n_seg, n_ts, n_band, n_region, K, mode, skipna;
  int group;  // lanes per (band, weight field, seg, slot) cell: 1, 2, ..., 64
};

__device__ __forceinline__ double nan_if_zero(double d) {
  return d != 0.0 ? d : __builtin_nan("");
}

__device__ __forceinline__ void det_combine_body(const CombineParams& p,
                                                 const long long o,
                                                 double* lds) {
  const int K = p.K;
  const int cell = p.nwf * p.n_seg * K;
  double* bandsum = lds;                     // [n_band][nwf][n_seg][K]
  double* rsum = lds + p.n_band * cell;      // [n_region][K]
  const int tid = threadIdx.x;
  // the region coefficients are read in dependent loops below: from LDS (a
  // global load per iteration cost ~10 us per slab, one workgroup per slab)
  double* cseg = rsum + p.n_region * K * (1 + p.n_band);  // [n_region][n_seg]
  double* cband = cseg + p.n_region * p.n_seg;            // [n_region][n_band]
  for (int i = tid; i < p.n_region * p.n_seg; i += blockDim.x)
    cseg[i] = p.coef_seg[i];
  for (int i = tid; i < p.n_region * p.n_band; i += blockDim.x)
    cband[i] = p.coef_band[i];
  const long long chunk_stride = (long long)p.nwf * p.n_ts * K;
  const double* part = p.partials + o * p.n_chunk * chunk_stride;

  // Few cells (one global region: ONE band, ONE seg -- K cells of n_chunk x
  // n_tile terms each): `group` lanes share a cell, lane g sums the terms g, g +
  // group, ... in order, a fixed xor-tree adds the lanes.  With one thread per
  // cell the 6 sums of a 50-member slab (152 chunks x 23 tiles) took 135 us --
  // a third of the K3 launch they follow.  `group` depends on the region
  // decomposition only (not on the mode's K): sums common to two modes (MSE of
  // DET and of DET_ACC) stay bit-identical.
  if (p.group > 1) {
    const int G = p.group;
    for (int idx = tid; idx < p.n_band * cell * G; idx += blockDim.x) {
      const int ci = idx / G, g = idx - ci * G;
      const int b = ci / cell, j = ci - b * cell;
      const int w = j / (p.n_seg * K), sk = j - w * (p.n_seg * K);
      const int s = sk / K, k = sk - s * K;
      const int e0 = p.seg_eoff[s], ne = p.seg_eoff[s + 1] - e0;
      const int c0 = p.band_chunk0[b], n = (p.band_chunk0[b + 1] - c0) * ne;
      const double* base = part + ((long long)w * p.n_ts) * K + k +
                           c0 * chunk_stride + (long long)e0 * K;
      int c = g / ne, e = g - c * ne;
      const int dq = G / ne, dr = G - dq * ne;
      auto next = [&]() {
        const double* r = base + c * chunk_stride + (long long)e * K;
        c += dq;
        e += dr;
        if (e >= ne) {
          e -= ne;
          ++c;
        }
        return r;
      };
      double v = 0.0;
      int i = g;
      for (; i + 7 * G < n; i += 8 * G) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *next();
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
      }
      for (; i < n; i += G) v += *next();
      for (int off = G >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
      if (g == 0) bandsum[ci] = v;
    }
  } else
  // One thread per (band, weight field, seg, slot).  Its (chunk, entry) terms
  // are independent loads; the sum order is fixed (deterministic).
  for (int idx = tid; idx < p.n_band * cell; idx += blockDim.x) {
    const int b = idx / cell, j = idx - b * cell;
    const int w = j / (p.n_seg * K), sk = j - w * (p.n_seg * K);
    const int s = sk / K, k = sk - s * K;
    const int e0 = p.seg_eoff[s], ne = p.seg_eoff[s + 1] - e0;
    const int c0 = p.band_chunk0[b], n = (p.band_chunk0[b + 1] - c0) * ne;
    const double* base = part + ((long long)w * p.n_ts) * K + k;
    // The terms (chunk c, entry e) are walked with two counters -- no division
    // per term: with i / ne and i % ne the kernel spent 27 us per ensemble slab
    // on integer division alone, one workgroup per slab -- and loaded up to 32
    // at a time before the adds, which stay in (chunk, entry) order.
    const double* q = base + c0 * chunk_stride + (long long)e0 * K;
    const long long wrap = chunk_stride - (long long)ne * K;
    int e = 0;
    auto next = [&]() {  // pointer increments only: this loop is instruction-bound
      const double* r = q;
      q += K;
      if (++e == ne) {
        e = 0;
        q += wrap;
      }
      return r;
    };
    double v = 0.0;
    int i = 0;
    for (; i + 32 <= n; i += 32) {
      double t[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) t[u] = *next();
#pragma unroll
      for (int u = 0; u < 32; ++u) v += t[u];
    }
    for (; i + 16 <= n; i += 16) {
      double t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = *next();
#pragma unroll
      for (int u = 0; u < 16; ++u) v += t[u];
    }
    for (; i + 4 <= n; i += 4) {
      double t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = *next();
#pragma unroll
      for (int u = 0; u < 4; ++u) v += t[u];
    }
    for (; i < n; ++i) v += *next();
    bandsum[idx] = v;
  }
  __syncthreads();
  // Region sums in two short steps (fixed order => deterministic):
  //   (1) one thread per (region, band, slot): sum over the segs of the band,
  //   (2) one thread per (region, slot): sum over bands.
  // Cells whose coefficient is 0 are never read: they may legitimately hold
  // NaN/Inf (metrics.py:159-160).
  double* rb = rsum + p.n_region * K;  // [n_region][n_band][K]
  for (int idx = tid; idx < p.n_region * p.n_band * K; idx += blockDim.x) {
    const int r = idx / (p.n_band * K), bk = idx - r * (p.n_band * K);
    const int b = bk / K, k = bk - b * K;
    const int wf = p.region_wf[r];
    const double cb = cband[r * p.n_band + b];
    double v = 0.0;
    if (cb != 0.0) {
      const double* row = bandsum + ((b * p.nwf + wf) * p.n_seg) * K + k;
      for (int s = 0; s < p.n_seg; ++s) {
        const double cs = cseg[r * p.n_seg + s];
        if (cs != 0.0) v += (cb * cs) * row[s * K];
      }
    }
    rb[idx] = v;
  }
  __syncthreads();
  for (int idx = tid; idx < p.n_region * K; idx += blockDim.x) {
    const int r = idx / K, k = idx - r * K;
    double v = 0.0;
    for (int b = 0; b < p.n_band; ++b)
      if (cband[r * p.n_band + b] != 0.0)
        v += rb[(r * p.n_band + b) * K + k];
    rsum[idx] = v;
    if (p.sums) p.sums[(o * p.n_region + r) * K + k] = v;
  }
  __syncthreads();
  if (!p.metrics) return;
  for (int r = tid; r < p.n_region; r += blockDim.x) {
    const double* s = rsum + r * K;
    const double nan = __builtin_nan("");
    const double wsum = nan_if_zero(p.region_wsum[r]);
    double mse = nan, rmse = nan, mae = nan, bias = nan, acc = nan;
    if (p.mode == WB2_MODE_ENS) {
      const double n_skill = p.skipna ? nan_if_zero(s[6]) : wsum;
      const double n_spread = p.skipna ? nan_if_zero(s[7]) : wsum;
      const double n_var = p.skipna ? nan_if_zero(s[8]) : wsum;
      const double n_deb = p.skipna ? nan_if_zero(s[9]) : wsum;
      const double skill = s[0] / n_skill, spread = s[1] / n_spread;
      const double emse = s[2] / n_skill, var = s[3] / n_var;
      const long long stride = (long long)p.n_region * p.n_outer;
      double* m = p.metrics + (long long)r * p.n_outer + o;
      m[WB2_ENS_CRPS * stride] = skill - 0.5 * spread;  // metrics.py:665-675
      m[WB2_ENS_CRPS_SPREAD * stride] = spread;
      m[WB2_ENS_CRPS_SKILL * stride] = skill;
      m[WB2_ENS_MEAN_MSE * stride] = emse;
      m[WB2_ENS_MEAN_RMSE * stride] = sqrt(emse);        // :1302-1307
      m[WB2_ENS_VARIANCE * stride] = var;
      m[WB2_ENS_STDDEV * stride] = sqrt(s[4] / n_var);   // :1205-1210
      m[WB2_ENS_DEBIASED_MSE * stride] = s[5] / n_deb;
      continue;
    }
    if (p.mode >= WB2_MODE_GAUSS) {
      // generic modes: slots [q_0..q_{KQ-1} | n_0..n_{KQ-1}] -> KQ plain means
      const int kq = p.skipna ? K / 2 : K;
      const long long stride = (long long)p.n_region * p.n_outer;
      double* m = p.metrics + (long long)r * p.n_outer + o;
      for (int i = 0; i < kq; ++i)
        m[i * stride] = s[i] / (p.skipna ? nan_if_zero(s[kq + i]) : wsum);
      continue;
    }
    if (p.mode == WB2_MODE_WIND) {
      const double den = p.skipna ? nan_if_zero(s[1]) : wsum;
      mse = s[0] / den;
      rmse = sqrt(mse);
    } else {
      const int kq = p.mode == WB2_MODE_DET_ACC ? 6 : 3;
      const double den_d = p.skipna ? nan_if_zero(s[kq]) : wsum;
      bias = s[0] / den_d;
      mae = s[1] / den_d;
      mse = s[2] / den_d;
      rmse = sqrt(mse);
      if (p.mode == WB2_MODE_DET_ACC) {
        const double den_p = p.skipna ? nan_if_zero(s[7]) : wsum;
        const double den_f = p.skipna ? nan_if_zero(s[8]) : wsum;
        const double den_t = p.skipna ? nan_if_zero(s[9]) : wsum;
        // metrics.py:407-414
        acc = (s[3] / den_p) / sqrt((s[4] / den_f) * (s[5] / den_t));
      }
    }
    const long long stride = (long long)p.n_region * p.n_outer;
    double* m = p.metrics + (long long)r * p.n_outer + o;
    m[WB2_METRIC_MSE * stride] = mse;
    m[WB2_METRIC_RMSE * stride] = rmse;
    m[WB2_METRIC_MAE * stride] = mae;
    m[WB2_METRIC_BIAS * stride] = bias;
    m[WB2_METRIC_ACC * stride] = acc;
  }
}

__global__ void __launch_bounds__(1024) det_combine_kernel(const CombineParams p) {
  extern __shared__ double lds[];
  det_combine_body(p, blockIdx.x, lds);
}

// Two folds in ONE launch: blocks [0, a.n_outer) fold `a`, the rest fold `b` --
// the per-variable slabs of a chunk and its wind-vector pairs
// (wb2_det_wind_suite_step).  Every block runs the code of the single launch
// on the parameters of its fold: the same bits, one launch (and one queue
// barrier) less per chunk.
__global__ void __launch_bounds__(1024)
    det_combine2_kernel(const CombineParams a, const CombineParams b) {
  extern __shared__ double lds[];
  if ((long long)blockIdx.x < a.n_outer)
    det_combine_body(a, blockIdx.x, lds);
  else
    det_combine_body(b, (long long)blockIdx.x - a.n_outer, lds);
}

// The running temporal mean (xbeam.Mean's (sum, count) combiner,
// evaluation.py:735-744).  The sum continues from the accumulator, value by
// value in time order: how many time steps one call brings (one chunk, or k
// chunks evaluated as one batch) does not change a bit of the result.  `dst`
// (optional) sends element idx of the [n_lead][n_tail] result to accumulator
// element dst[idx] (lead-time blocks of chunks that split the lead dim).
template <typename T>
__global__ void __launch_bounds__(256)
    time_accumulate_kernel(const T* __restrict__ values, long long n_lead,
                           long long n_time, long long n_tail, int skipna,
                           const long long* __restrict__ dst, long long run,
                           double* __restrict__ sum,
                           double* __restrict__ count) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_lead * n_tail) return;
  const long long l = idx / n_tail, j = idx - l * n_tail;
  // `run` consecutive result elements go to consecutive accumulator elements:
  // one table entry per run (maps: one per slab, not one per grid point)
  long long out = idx;
  if (dst) {
    const long long r = run > 1 ? idx / run : idx;
    out = dst[r] + (idx - r * run);
  }
  double s = sum[out], c = count ? count[out] : 0.0;
  const T* base = values + l * n_time * n_tail + j;
  auto add = [&](double v) {  // float32 values widen exactly
    const bool keep = !(skipna && is_nan(v));
    s += keep ? v : 0.0;
    c += keep ? 1.0 : 0.0;
  };
  long long t = 0;
  for (; t + 4 <= n_time; t += 4) {  // four independent loads per wait
    const T v0 = base[t * n_tail], v1 = base[(t + 1) * n_tail],
            v2 = base[(t + 2) * n_tail], v3 = base[(t + 3) * n_tail];
    add(v0);
    add(v1);
    add(v2);
    add(v3);
  }
  for (; t < n_time; ++t) add(base[t * n_tail]);
  sum[out] = s;
  if (count) count[out] = c;
}

// The running temporal mean of a whole chunk RESULT in one launch: output
// element e (of every variable of the result Dataset alike) is the sum over the
// chunk's time steps of arena[src[e][t]] -- the K2 outputs of the chunk's
// launches, side by side in `arena` --, rounded to float32 first where the
// reference's result dtype is float32, added to the accumulator at the device
// ADDRESS sum_addr[e] / count_addr[e] (the accumulators of the variables are
// separate allocations).  src < 0: the element is a NaN fill (a metric that
// lacks the variable, evaluation.py:424-437).  Same arithmetic, value by value
// in time order, as time_accumulate_kernel.
// `rows` (optional; wb2_gather_accumulate_rows): element e lands rows[sel[e]] *
// rsz8[e] bytes behind sum_addr[e] / count_addr[e] -- a sink that KEEPS the
// time steps (temporal_mean=False, evaluation.py:735) files every chunk under
// the row of its (time, lead) labels; the chunk brings its few row numbers,
// which entry takes which of them and how long a row is are structural.
__global__ void __launch_bounds__(256)
    gather_accumulate_kernel(const double* __restrict__ arena,
                             const int* __restrict__ src,
                             const unsigned char* __restrict__ round32,
                             long long n_out, long long n_time, int skipna,
                             const long long* __restrict__ sum_addr,
                             const long long* __restrict__ count_addr,
                             const long long* __restrict__ rows,
                             const int* __restrict__ sel,
                             const long long* __restrict__ rsz8) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_out) return;
  const long long shift = rows ? rows[sel[e]] * rsz8[e] : 0;
  double* sp = reinterpret_cast<double*>(sum_addr[e] + shift);
  double* cp = reinterpret_cast<double*>(count_addr[e] + shift);
  double s = *sp, c = *cp;
  const bool r32 = round32[e] != 0;
  const int* q = src + e * n_time;
  for (long long t = 0; t < n_time; ++t) {
    const int k = q[t];
    double v = k < 0 ? __builtin_nan("") : arena[k];
    if (r32) v = (double)(float)v;
    const bool keep = !(skipna && is_nan(v));
    s += keep ? v : 0.0;
    c += keep ? 1.0 : 0.0;
  }
  *sp = s;
  *cp = c;
}

// ---------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------
// float32 weight fields: instantiated for float32 inputs of the deterministic
// modes (the production passes)
template <typename T, int MODE>
constexpr bool field_f32_supported() {
  return std::is_same<T, float>::value &&
         (MODE == WB2_MODE_DET || MODE == WB2_MODE_DET_ACC ||
          MODE == WB2_MODE_WIND);
}
bool field_f32_supported(int dtype, int mode) {
  return dtype == WB2_F32 && (mode == WB2_MODE_DET || mode == WB2_MODE_DET_ACC ||
                              mode == WB2_MODE_WIND);
}

// The ring form of K1 (stream_partials_kernel<..., RING>): WB2HIP_K1_RING = rows
// per wave in flight (0 = the batch form), WB2HIP_K1_RING_WAVES = waves per
// workgroup (0 = as the batch form).
// (read at every launch: tests and A/B runs switch inside one process)
int ring_depth() {
  const char* e = getenv("WB2HIP_K1_RING");
  const int v = e ? atoi(e) : WB2_K1_RING_DEFAULT;
  return v < 2 ? 0 : (v > 5 ? 5 : v);
}
int ring_waves() {
  const char* e = getenv("WB2HIP_K1_RING_WAVES");
  const int v = e ? atoi(e) : 0;
  return v < 1 ? 0 : (v > 8 ? 8 : v);
}

template <typename T, int VEC, int MODE, bool SKIPNA, bool WF, typename FT,
          int RING>
int launch_ring(const StreamParams& p, dim3 grid, int threads,
                hipStream_t stream) {
  constexpr int NSLOT = ModeTraits<MODE, SKIPNA>::NIN + (WF ? 1 : 0);
  const size_t lds = (size_t)(threads / kWave) * RING * NSLOT * 1024;
  auto kern = stream_partials_kernel<T, VEC, MODE, SKIPNA, WF, false, FT, RING>;
  static const hipError_t attr = hipFuncSetAttribute(
      reinterpret_cast<const void*>(kern),
      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  WB2_HIP_OK(attr);
  hipLaunchKernelGGL(kern, grid, dim3(threads), lds, stream, p);
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

template <typename T, int VEC, int MODE, bool SKIPNA, bool WF>
int launch_stream(const StreamParams& p, int threads, hipStream_t stream) {
  constexpr bool HAS_RING = std::is_same<T, float>::value && VEC == 4 &&
                            !SKIPNA &&
                            (MODE == WB2_MODE_DET || MODE == WB2_MODE_DET_ACC);
  const bool ring = HAS_RING && ring_depth() > 0 && !p.unaligned &&
                    (!WF || p.wfield_f32);
  if (ring && ring_waves() > 0) {
    const int tiles = p.n_ctile;
    threads = (ring_waves() < tiles ? ring_waves() : tiles) * kWave;
  }
  const int nwave = threads / kWave;
  const int n_tblk = (p.n_ctile + nwave - 1) / nwave;
  const long long gy = p.n_outer < 32768 ? p.n_outer : 32768;
  const long long gz = (p.n_outer + gy - 1) / gy;  // kernel guards o < n_outer
  const dim3 grid =
      (WF || MODE == WB2_MODE_SEEPS)
          ? dim3((unsigned)gy, (unsigned)(p.n_chunk * n_tblk), (unsigned)gz)
          : dim3((unsigned)(p.n_chunk * n_tblk), (unsigned)gy, (unsigned)gz);
  // unaligned float32 rows at 4 columns per lane: SGPR row bases are worth 4 %
  // there (profiles/r03_k1_ab5_summary.txt), and cost 2-20 % everywhere else
  constexpr bool HAS_SG = std::is_same<T, float>::value && VEC == 4 &&
                          !SKIPNA && !WF &&
                          (MODE == WB2_MODE_DET || MODE == WB2_MODE_DET_ACC ||
                           MODE == WB2_MODE_WIND);
  if constexpr (HAS_SG) {
    static const bool sg_on = [] {  // WB2HIP_SGPR_UNALIGNED=0: A/B runs
      const char* e = getenv("WB2HIP_SGPR_UNALIGNED");
      return !(e && e[0] == '0');
    }();
    if (p.unaligned && sg_on) {
      hipLaunchKernelGGL(
          (stream_partials_kernel<T, VEC, MODE, SKIPNA, WF, true>), grid,
          dim3(threads), 0, stream, p);
      WB2_HIP_OK(hipGetLastError());
      return 0;
    }
  }
  if constexpr (HAS_RING) {
    if (ring) {
      using FT = typename std::conditional<WF, float, double>::type;
      switch (ring_depth()) {
        case 2:
          return launch_ring<T, VEC, MODE, SKIPNA, WF, FT, 2>(p, grid, threads,
                                                              stream);
        case 3:
          return launch_ring<T, VEC, MODE, SKIPNA, WF, FT, 3>(p, grid, threads,
                                                              stream);
        case 4:
          return launch_ring<T, VEC, MODE, SKIPNA, WF, FT, 4>(p, grid, threads,
                                                              stream);
        default:
          return launch_ring<T, VEC, MODE, SKIPNA, WF, FT, 5>(p, grid, threads,
                                                              stream);
      }
    }
  }
  if constexpr (field_f32_supported<T, MODE>() && WF) {
    if (p.wfield_f32) {
      hipLaunchKernelGGL(
          (stream_partials_kernel<T, VEC, MODE, SKIPNA, WF, false, float>),
          grid, dim3(threads), 0, stream, p);
      WB2_HIP_OK(hipGetLastError());
      return 0;
    }
  }
  hipLaunchKernelGGL((stream_partials_kernel<T, VEC, MODE, SKIPNA, WF>), grid,
                     dim3(threads), 0, stream, p);
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

template <typename T, int VEC, int MODE>
int launch_stream_flags(const StreamParams& p, bool skipna, bool wf,
                        int threads, hipStream_t stream) {
  if (skipna) {
    return wf ? launch_stream<T, VEC, MODE, true, true>(p, threads, stream)
              : launch_stream<T, VEC, MODE, true, false>(p, threads, stream);
  }
  return wf ? launch_stream<T, VEC, MODE, false, true>(p, threads, stream)
            : launch_stream<T, VEC, MODE, false, false>(p, threads, stream);
}

template <typename T, int VECW>
int launch_stream_mode(const StreamParams& p, int mode, int vec, bool skipna,
                       bool wf, int threads, hipStream_t stream) {
  // the narrower vector of the register-heavy instantiations (vec_width)
  if (std::is_same<T, float>::value && mode == WB2_MODE_DET_ACC &&
      vec == WB2_F32_VEC_HEAVY && WB2_F32_VEC_HEAVY != VECW && vec > 1)
    return launch_stream_flags<T, WB2_F32_VEC_HEAVY, WB2_MODE_DET_ACC>(
        p, skipna, wf, threads, stream);
#define WB2_MODE_CASE(M)                                                     \
  case M:                                                                    \
    return vec > 1 ? launch_stream_flags<T, VECW, M>(p, skipna, wf, threads, \
                                                     stream)                 \
                   : launch_stream_flags<T, 1, M>(p, skipna, wf, threads,    \
                                                  stream);
  switch (mode) {
    WB2_MODE_CASE(WB2_MODE_DET)
    WB2_MODE_CASE(WB2_MODE_DET_ACC)
    WB2_MODE_CASE(WB2_MODE_WIND)
    WB2_MODE_CASE(WB2_MODE_GAUSS)
    WB2_MODE_CASE(WB2_MODE_GAUSS_THR)
    WB2_MODE_CASE(WB2_MODE_SEEPS)
  }
#undef WB2_MODE_CASE
  return fail("unknown mode %d", mode);
}

// Columns per lane.  Wide loads need neither 16-byte alignment nor n_col % w ==
// 0 (the kernel's loads are element-aligned, its last lane shifts back): rows of
// 721 latitudes -- the lon-lat layout of the WeatherBench 2 Zarr stores -- run
// at the same width as rows of 1440 longitudes.  WB2HIP_UNALIGNED_VEC=0 brings
// back the round-1..3 rule (one column per lane unless everything is aligned),
// for A/B measurements.
int vec_width(int mode, int dtype, bool skipna, bool wf, int n_col,
              bool aligned16) {
  static const bool unaligned_ok = [] {
    const char* e = getenv("WB2HIP_UNALIGNED_VEC");
    return !(e && e[0] == '0');
  }();
  int w = dtype == WB2_F32 ? WB2_F32_VEC : 2;
  if (dtype == WB2_F32 && mode == WB2_MODE_DET_ACC && (skipna || wf))
    w = WB2_F32_VEC_HEAVY;
  if (unaligned_ok) return n_col >= w ? w : 1;
  return (aligned16 && n_col % w == 0) ? w : 1;
}

// SpatialSEEPS (metrics.py:418-509): the per-point score of mode SEEPS as a
// float64 map, no spatial reduction.  One lane per grid point.
struct SeepsMapParams {
  const void* in[3];           // forecast, truth, wet threshold
  const long long* slab[3];
  const double* aux;           // p1 [n_point], NaN where masked
  double scalar;               // dry threshold
  double* out;                 // [n_outer][n_point]
  long long n_outer, n_point;
  bool by_addr;                // slab[i][o] = byte address of the slab
};

// A thread owns one grid point of kSeepsMapSlabs consecutive outer slabs: p1
// (8 bytes per point, the same for every slab) is read once per group instead
// of once per slab, and the group's 3 x kSeepsMapSlabs loads are in flight
// together.  grid: x = point blocks, (y, z) = slab groups.
constexpr int kSeepsMapSlabs = 8;

template <typename T>
__global__ void __launch_bounds__(256) seeps_map_kernel(const SeepsMapParams p) {
  const long long pt = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long o0 =
      (blockIdx.y + (long long)blockIdx.z * gridDim.y) * kSeepsMapSlabs;
  if (pt >= p.n_point || o0 >= p.n_outer) return;
  const double p1 = p.aux[pt];
  T in[kSeepsMapSlabs][3];
#pragma unroll
  for (int s = 0; s < kSeepsMapSlabs; ++s) {
    // slabs past the end read the group's first one again (dropped below)
    const long long o = o0 + s < p.n_outer ? o0 + s : o0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const long long sl = p.slab[i] ? p.slab[i][o] : o;
      // (o is the same for the whole workgroup: a scalar select)
      const T* src = p.by_addr ? reinterpret_cast<const T*>(sl)
                               : static_cast<const T*>(p.in[i]) + sl * p.n_point;
      in[s][i] = __builtin_nontemporal_load(src + pt);
    }
  }
#pragma unroll
  for (int s = 0; s < kSeepsMapSlabs; ++s) {
    if (o0 + s < p.n_outer) {
      double x[1];
      eval_slots<WB2_MODE_SEEPS, false, T>(in[s], x, p1, p.scalar);
      __builtin_nontemporal_store(x[0], p.out + (o0 + s) * p.n_point + pt);
    }
  }
}

// One wave per column tile; up to 8 tiles share a workgroup.
#ifndef WB2_MAX_WG_WAVES
// Waves per workgroup.  Every wave is independent (no LDS, no barrier), so the
// workgroup is only a scheduling unit: with 2 waves instead of 6-8 the headline
// kernel gains 4 % (0.435 -> 0.418 ms), the weight-field and skipna
// instantiations 8-11 % (same box, interleaved: profiles/r03_k1_variants.md) --
// a wave that finishes its chunk frees its slot without waiting for five others.
#define WB2_MAX_WG_WAVES 2
#endif

int threads_for(int n_col, int vec) {
  const int lanes = (n_col + vec - 1) / vec;
  const int tiles = (lanes + kWave - 1) / kWave;
  if (tiles <= WB2_MAX_WG_WAVES) return tiles * kWave;
  // more tiles than the waves of a workgroup: the widest workgroup that
  // divides them leaves no idle wave (1440 columns, 2 per lane: 12 tiles =
  // 2 x 6 waves)
  for (int w = WB2_MAX_WG_WAVES; w >= 3; --w)
    if (tiles % w == 0) return w * kWave;
  return WB2_MAX_WG_WAVES * kWave;
}

int mode_nin(int mode) {
  return mode == WB2_MODE_DET ? 2
         : (mode == WB2_MODE_WIND || mode == WB2_MODE_GAUSS_THR) ? 4 : 3;
}

}  // namespace
}  // namespace wb2

namespace wb2 {
namespace {

// The one body behind wb2_stream_partials[_ex] (in[] + slab NUMBERS) and
// wb2_stream_partials_addr (slab ADDRESSES, in == nullptr).
int stream_partials_impl(int mode, int dtype, int skipna, const void* const* in,
                         const int64_t* const* slab, int addr_aligned16,
                         int64_t n_outer, int32_t n_row, int32_t n_col,
                         const double* w_row, const double* w_col,
                         const void* wfield, int wfield_dtype,
                         const double* aux, double scalar,
                         const int32_t* chunk_row0, const int32_t* chunk_nrow,
                         int32_t n_chunk, int32_t n_ctile,
                         const int32_t* seg_col0, const int32_t* seg_eoff,
                         int32_t n_seg, int32_t n_ts, double* partials,
                         void* stream) {
  const bool by_addr = in == nullptr;
  WB2_REQUIRE(mode == WB2_MODE_DET || mode == WB2_MODE_DET_ACC ||
                  mode == WB2_MODE_WIND || mode == WB2_MODE_GAUSS ||
                  mode == WB2_MODE_GAUSS_THR || mode == WB2_MODE_SEEPS,
              "unknown mode %d", mode);
  WB2_REQUIRE(mode != WB2_MODE_SEEPS || aux != nullptr,
              "WB2_MODE_SEEPS needs the p1 field in `aux`");
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_REQUIRE(!wfield || wfield_dtype == WB2_F64 ||
                  (wfield_dtype == WB2_F32 && field_f32_supported(dtype, mode)),
              "a float32 weight field goes with float32 inputs of the modes "
              "DET / DET_ACC / WIND (wfield_dtype=%d dtype=%d mode=%d)",
              wfield_dtype, dtype, mode);
  WB2_EMPTY_OK(n_outer);
  WB2_REQUIRE((in || slab) && w_row && chunk_row0 && chunk_nrow && seg_col0 &&
                  seg_eoff && partials,
              "null pointer argument");
  WB2_REQUIRE(n_outer >= 0 && n_row > 0 && n_col > 0 && n_chunk > 0 &&
                  n_seg > 0,
              "bad sizes: n_outer=%lld n_row=%d n_col=%d n_chunk=%d n_seg=%d",
              (long long)n_outer, n_row, n_col, n_chunk, n_seg);
  if (n_outer == 0) return 0;
  WB2_REQUIRE(n_chunk % 8 == 0, "n_chunk=%d must be a multiple of 8", n_chunk);
  WB2_REQUIRE(n_outer < (1ll << 31), "n_outer=%lld too large",
              (long long)n_outer);
  StreamParams p{};
  const int nin = mode_nin(mode);
  const long long elem = dtype == WB2_F32 ? 4 : 8;
  bool aligned = by_addr ? addr_aligned16 != 0 : true;
  for (int i = 0; i < nin; ++i) {
    if (by_addr) {
      WB2_REQUIRE(slab[i] != nullptr, "address table %d is null", i);
      p.in[i] = nullptr;
      p.slab[i] = reinterpret_cast<const long long*>(slab[i]);
      continue;
    }
    WB2_REQUIRE(in[i] != nullptr, "input %d is null", i);
    p.in[i] = in[i];
    p.slab[i] = slab ? reinterpret_cast<const long long*>(slab[i]) : nullptr;
    aligned = aligned && (reinterpret_cast<uintptr_t>(in[i]) % 16 == 0);
  }
  p.slab_step_bytes = by_addr ? 1 : (long long)n_row * n_col * elem;
  if (wfield) aligned = aligned && reinterpret_cast<uintptr_t>(wfield) % 16 == 0;
  const int vec = vec_width(mode, dtype, skipna != 0, wfield != nullptr, n_col,
                            aligned);
  const int threads = threads_for(n_col, vec);
  p.unaligned = !aligned || ((long long)n_col * elem) % 16 != 0;
  p.w_row = w_row;
  p.w_col = w_col;
  p.wfield = wfield;
  p.wfield_f32 = wfield && wfield_dtype == WB2_F32;
  p.aux = aux;
  p.scalar = scalar;
  p.chunk_row0 = chunk_row0;
  p.chunk_nrow = chunk_nrow;
  p.seg_col0 = seg_col0;
  p.seg_eoff = seg_eoff;
  p.n_ts = n_ts;
  p.partials = partials;
  p.n_outer = n_outer;
  p.n_row = n_row;
  p.n_col = n_col;
  p.n_chunk = n_chunk;
  p.n_ctile = (n_col + kWave * vec - 1) / (kWave * vec);
  WB2_REQUIRE(p.n_ctile == n_ctile,
              "n_ctile=%d does not match the launch geometry (%d): inputs "
              "must be 16-byte aligned iff wb2_tile_cols_ex() was asked so",
              n_ctile, p.n_ctile);
  p.n_seg = n_seg;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == WB2_F32)
    return launch_stream_mode<float, WB2_F32_VEC>(p, mode, vec, skipna != 0,
                                        wfield != nullptr, threads, s);
  return launch_stream_mode<double, 2>(p, mode, vec, skipna != 0,
                                       wfield != nullptr, threads, s);
}


// K1p launch: the pairs of a launch (stream_pair_kernel).  `p` is filled as for
// the per-variable kernel over the same tables (stream_partials_impl).
template <typename T, int VEC, bool ACC>
int launch_pairs_flags(const StreamParams& p, const PairParams& pp, bool skipna,
                       bool wf, hipStream_t stream) {
  const long long gy = pp.n_pair < 32768 ? pp.n_pair : 32768;
  const long long gz = (pp.n_pair + gy - 1) / gy;
  const unsigned per_pair = (unsigned)(p.n_chunk * p.n_ctile);
  const dim3 grid = wf ? dim3((unsigned)gy, per_pair, (unsigned)gz)
                       : dim3(per_pair, (unsigned)gy, (unsigned)gz);
#define WB2_PAIR_LAUNCH(S, W, FT)                                            \
  hipLaunchKernelGGL((stream_pair_kernel<T, VEC, ACC, S, W, FT>), grid,      \
                     dim3(2 * kWave), 0, stream, p, pp)
  if (wf) {
    if constexpr (std::is_same<T, float>::value) {
      if (p.wfield_f32) {
        if (skipna) WB2_PAIR_LAUNCH(true, true, float);
        else WB2_PAIR_LAUNCH(false, true, float);
        WB2_HIP_OK(hipGetLastError());
        return 0;
      }
    }
    if (skipna) WB2_PAIR_LAUNCH(true, true, double);
    else WB2_PAIR_LAUNCH(false, true, double);
  } else {
    if (skipna) WB2_PAIR_LAUNCH(true, false, double);
    else WB2_PAIR_LAUNCH(false, false, double);
  }
#undef WB2_PAIR_LAUNCH
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

bool pairs_supported(int mode, int dtype, bool skipna, bool wf, int n_col,
                     bool aligned16) {
  if (mode != WB2_MODE_DET && mode != WB2_MODE_DET_ACC) return false;
  if (dtype != WB2_F32 && dtype != WB2_F64) return false;
  // the pair kernel exists at the per-variable kernel's full width only (its
  // partials share the tile geometry with the per-variable and WIND passes)
  const int w = dtype == WB2_F32 ? 4 : 2;
  return vec_width(mode, dtype, skipna, wf, n_col, aligned16) == w &&
         vec_width(WB2_MODE_WIND, dtype, skipna, wf, n_col, aligned16) == w;
}

int stream_pairs_impl(int mode, int dtype, int skipna, const void* const* in,
                      const int64_t* const* slab, int addr_aligned16,
                      int64_t n_outer, int64_t n_pair, int32_t n_row,
                      int32_t n_col, const double* w_row, const double* w_col,
                      const void* wfield, int wfield_dtype,
                      const int32_t* chunk_row0, const int32_t* chunk_nrow,
                      int32_t n_chunk, int32_t n_ctile, const int32_t* seg_col0,
                      const int32_t* seg_eoff, int32_t n_seg, int32_t n_ts,
                      double* partials, double* wind_partials, void* stream,
                      void* pair_stream = nullptr, void* join_event = nullptr) {
  WB2_REQUIRE(n_pair >= 0 && 2 * n_pair <= n_outer,
              "n_pair=%lld does not fit n_outer=%lld", (long long)n_pair,
              (long long)n_outer);
  const int64_t n_single = n_outer - 2 * n_pair;
  // the slabs outside the pairs: the per-variable kernel, as ever
  int rc = stream_partials_impl(mode, dtype, skipna, in, slab, addr_aligned16,
                                n_single, n_row, n_col, w_row, w_col, wfield,
                                wfield_dtype, nullptr, 0.0, chunk_row0,
                                chunk_nrow, n_chunk, n_ctile, seg_col0, seg_eoff,
                                n_seg, n_ts, partials, stream);
  if (rc != 0 || n_pair == 0) return rc;
  WB2_REQUIRE(mode == WB2_MODE_DET || mode == WB2_MODE_DET_ACC,
              "wind-vector pairs ride on WB2_MODE_DET / WB2_MODE_DET_ACC "
              "(mode=%d)", mode);
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_REQUIRE((in || slab) && w_row && chunk_row0 && chunk_nrow && seg_col0 &&
                  seg_eoff && partials && wind_partials,
              "null pointer argument");
  WB2_REQUIRE(n_row > 0 && n_col > 0 && n_chunk > 0 && n_chunk % 8 == 0 &&
                  n_seg > 0 && n_outer < (1ll << 31),
              "bad sizes");
  const bool by_addr = in == nullptr;
  const int nin = mode_nin(mode);
  const long long elem = dtype == WB2_F32 ? 4 : 8;
  bool aligned = by_addr ? addr_aligned16 != 0 : true;
  StreamParams p{};
  for (int i = 0; i < nin; ++i) {
    if (by_addr) {
      WB2_REQUIRE(slab[i] != nullptr, "address table %d is null", i);
      p.slab[i] = reinterpret_cast<const long long*>(slab[i]);
      continue;
    }
    WB2_REQUIRE(in[i] != nullptr, "input %d is null", i);
    p.in[i] = in[i];
    p.slab[i] = slab ? reinterpret_cast<const long long*>(slab[i]) : nullptr;
    aligned = aligned && (reinterpret_cast<uintptr_t>(in[i]) % 16 == 0);
  }
  if (wfield) aligned = aligned && reinterpret_cast<uintptr_t>(wfield) % 16 == 0;
  WB2_REQUIRE(pairs_supported(mode, dtype, skipna != 0, wfield != nullptr, n_col,
                              aligned),
              "no pair kernel for this launch (n_col=%d too narrow for the "
              "wide loads): ask wb2_pairs_supported first", n_col);
  const int vec = dtype == WB2_F32 ? 4 : 2;
  p.slab_step_bytes = by_addr ? 1 : (long long)n_row * n_col * elem;
  p.w_row = w_row;
  p.w_col = w_col;
  p.wfield = wfield;
  p.wfield_f32 = wfield && wfield_dtype == WB2_F32;
  WB2_REQUIRE(!wfield || wfield_dtype == WB2_F64 ||
                  (wfield_dtype == WB2_F32 && dtype == WB2_F32),
              "a float32 weight field goes with float32 inputs");
  p.chunk_row0 = chunk_row0;
  p.chunk_nrow = chunk_nrow;
  p.seg_col0 = seg_col0;
  p.seg_eoff = seg_eoff;
  p.n_ts = n_ts;
  p.partials = partials;
  p.n_outer = n_outer;
  p.n_row = n_row;
  p.n_col = n_col;
  p.n_chunk = n_chunk;
  p.n_ctile = (n_col + kWave * vec - 1) / (kWave * vec);
  WB2_REQUIRE(p.n_ctile == n_ctile,
              "n_ctile=%d does not match the launch geometry (%d)", n_ctile,
              p.n_ctile);
  p.n_seg = n_seg;
  PairParams pp{};
  pp.wind_partials = wind_partials;
  pp.first = n_single;
  pp.n_pair = n_pair;
  // `pair_stream`: the pair kernel beside the per-variable kernel (their
  // tails overlap); the caller has ordered pair_stream behind whatever made
  // the inputs, `stream` waits for `join_event` before anything reads the
  // pairs' partials
  hipStream_t s = static_cast<hipStream_t>(pair_stream ? pair_stream : stream);
  const bool acc = mode == WB2_MODE_DET_ACC;
  if (dtype == WB2_F32)
    rc = acc ? launch_pairs_flags<float, 4, true>(p, pp, skipna != 0,
                                                  wfield != nullptr, s)
             : launch_pairs_flags<float, 4, false>(p, pp, skipna != 0,
                                                   wfield != nullptr, s);
  else
    rc = acc ? launch_pairs_flags<double, 2, true>(p, pp, skipna != 0,
                                                   wfield != nullptr, s)
             : launch_pairs_flags<double, 2, false>(p, pp, skipna != 0,
                                                    wfield != nullptr, s);
  if (rc != 0 || !pair_stream) return rc;
  WB2_REQUIRE(join_event != nullptr, "a pair stream needs a join event");
  WB2_HIP_OK(hipEventRecord(static_cast<hipEvent_t>(join_event), s));
  WB2_HIP_OK(hipStreamWaitEvent(static_cast<hipStream_t>(stream),
                                static_cast<hipEvent_t>(join_event), 0));
  return 0;
}

}  // namespace
}  // namespace wb2

extern "C" {

int wb2_num_slots(int mode, int skipna) {
  switch (mode) {
    case WB2_MODE_DET: return skipna ? 4 : 3;
    case WB2_MODE_DET_ACC: return skipna ? 10 : 6;
    case WB2_MODE_WIND: return skipna ? 2 : 1;
    case WB2_MODE_GAUSS: return skipna ? 4 : 2;
    case WB2_MODE_GAUSS_THR: return skipna ? 6 : 3;
    case WB2_MODE_ENS_THR: return skipna ? 8 : 4;
    case WB2_MODE_SEEPS: return skipna ? 2 : 1;
  }
  return wb2::fail("unknown mode %d", mode);
}

int wb2_tile_cols(int dtype, int n_col, int aligned16) {
  return wb2_tile_cols_ex(WB2_MODE_DET, dtype, 0, 0, n_col, aligned16);
}

int wb2_tile_cols_ex(int mode, int dtype, int skipna, int has_wfield, int n_col,
                     int aligned16) {
  if (dtype != WB2_F32 && dtype != WB2_F64) return wb2::fail("bad dtype");
  return wb2::kWave * wb2::vec_width(mode, dtype, skipna != 0, has_wfield != 0,
                                     n_col, aligned16 != 0);
}

int wb2_stream_partials(int mode, int dtype, int skipna,
                        const void* const* in, const int64_t* const* slab,
                        int64_t n_outer, int32_t n_row, int32_t n_col,
                        const double* w_row, const double* w_col,
                        const double* wfield, const int32_t* chunk_row0,
                        const int32_t* chunk_nrow, int32_t n_chunk,
                        int32_t n_ctile, const int32_t* seg_col0,
                        const int32_t* seg_eoff, int32_t n_seg, int32_t n_ts,
                        double* partials, void* stream) {
  WB2_TRACE();
  return wb2_stream_partials_ex(mode, dtype, skipna, in, slab, n_outer, n_row,
                                n_col, w_row, w_col, wfield, WB2_F64, nullptr, 0.0,
                                chunk_row0, chunk_nrow, n_chunk, n_ctile,
                                seg_col0, seg_eoff, n_seg, n_ts, partials,
                                stream);
}


int wb2_stream_partials_ex(int mode, int dtype, int skipna,
                           const void* const* in, const int64_t* const* slab,
                           int64_t n_outer, int32_t n_row, int32_t n_col,
                           const double* w_row, const double* w_col,
                           const void* wfield, int wfield_dtype,
                           const double* aux,
                           double scalar, const int32_t* chunk_row0,
                           const int32_t* chunk_nrow, int32_t n_chunk,
                           int32_t n_ctile, const int32_t* seg_col0,
                           const int32_t* seg_eoff, int32_t n_seg,
                           int32_t n_ts, double* partials, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(in != nullptr, "null pointer argument");
  return stream_partials_impl(mode, dtype, skipna, in, slab, 0, n_outer, n_row,
                              n_col, w_row, w_col, wfield, wfield_dtype, aux,
                              scalar,
                              chunk_row0, chunk_nrow, n_chunk, n_ctile,
                              seg_col0, seg_eoff, n_seg, n_ts, partials, stream);
}

int wb2_stream_partials_addr(int mode, int dtype, int skipna,
                             const int64_t* const* slab_addr, int aligned16,
                             int64_t n_outer, int32_t n_row, int32_t n_col,
                             const double* w_row, const double* w_col,
                             const void* wfield, int wfield_dtype,
                             const double* aux,
                             double scalar, const int32_t* chunk_row0,
                             const int32_t* chunk_nrow, int32_t n_chunk,
                             int32_t n_ctile, const int32_t* seg_col0,
                             const int32_t* seg_eoff, int32_t n_seg,
                             int32_t n_ts, double* partials, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(slab_addr != nullptr, "null pointer argument");
  return stream_partials_impl(mode, dtype, skipna, nullptr, slab_addr,
                              aligned16, n_outer, n_row, n_col, w_row, w_col,
                              wfield, wfield_dtype, aux, scalar, chunk_row0,
                              chunk_nrow,
                              n_chunk, n_ctile, seg_col0, seg_eoff, n_seg, n_ts,
                              partials, stream);
}

int wb2_pairs_supported(int mode, int dtype, int skipna, int has_wfield,
                        int n_col, int aligned16) {
  return wb2::pairs_supported(mode, dtype, skipna != 0, has_wfield != 0, n_col,
                              aligned16 != 0)
             ? 1
             : 0;
}

int wb2_stream_partials_pairs(int mode, int dtype, int skipna,
                              const void* const* in,
                              const int64_t* const* slab, int aligned16,
                              int64_t n_outer, int64_t n_pair, int32_t n_row,
                              int32_t n_col, const double* w_row,
                              const double* w_col, const void* wfield,
                              int wfield_dtype, const int32_t* chunk_row0,
                              const int32_t* chunk_nrow, int32_t n_chunk,
                              int32_t n_ctile, const int32_t* seg_col0,
                              const int32_t* seg_eoff, int32_t n_seg,
                              int32_t n_ts, double* partials,
                              double* wind_partials, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_EMPTY_OK(n_outer);
  WB2_REQUIRE(in != nullptr || slab != nullptr, "null pointer argument");
  return stream_pairs_impl(mode, dtype, skipna, in, slab, aligned16, n_outer,
                           n_pair, n_row, n_col, w_row, w_col, wfield,
                           wfield_dtype, chunk_row0, chunk_nrow, n_chunk,
                           n_ctile, seg_col0, seg_eoff, n_seg, n_ts, partials,
                           wind_partials, stream);
}

int wb2_det_combine(int mode, int skipna, const double* partials,
                    int64_t n_outer, int32_t n_chunk, int32_t nwf,
                    int32_t n_seg, const int32_t* seg_eoff, int32_t n_ts,
                    const int32_t* band_chunk0, int32_t n_band,
                    const double* coef_band,
                    const double* coef_seg, const int32_t* region_wf,
                    const double* region_wsum, int32_t n_region, double* sums,
                    double* metrics, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(mode >= 0 && mode <= WB2_MODE_SEEPS, "unknown mode %d", mode);
  const int k = mode == WB2_MODE_ENS ? wb2_ens_num_slots(skipna)
                                     : wb2_num_slots(mode, skipna);
  return combine_slots(mode, skipna, k, partials, n_outer, n_chunk, nwf, n_seg,
                       seg_eoff, n_ts, band_chunk0, n_band, coef_band, coef_seg,
                       region_wf, region_wsum, n_region, sums, metrics, stream);
}

}  // extern "C"

namespace wb2 {
// The body of wb2_det_combine with the slot count given (the energy-score pass
// folds 2 x block member sums per virtual slab: a generic mode whose slot
// count is not a function of the mode alone).
namespace {
CombineParams combine_params(int mode, int skipna, int k_slots,
                             const double* partials, int64_t n_outer,
                             int32_t n_chunk, int32_t nwf, int32_t n_seg,
                             const int32_t* seg_eoff, int32_t n_ts,
                             const int32_t* band_chunk0, int32_t n_band,
                             const double* coef_band, const double* coef_seg,
                             const int32_t* region_wf,
                             const double* region_wsum, int32_t n_region,
                             double* sums, double* metrics, size_t* lds) {
  CombineParams p{};
  p.partials = partials;
  p.seg_eoff = seg_eoff;
  p.band_chunk0 = band_chunk0;
  p.coef_band = coef_band;
  p.coef_seg = coef_seg;
  p.region_wf = region_wf;
  p.region_wsum = region_wsum;
  p.sums = sums;
  p.metrics = metrics;
  p.n_outer = n_outer;
  p.n_chunk = n_chunk;
  p.nwf = nwf;
  p.n_seg = n_seg;
  p.n_ts = n_ts;
  p.n_band = n_band;
  p.n_region = n_region;
  p.K = k_slots;
  p.mode = mode;
  p.skipna = skipna != 0;
  // lanes per cell: as many as keep the 1024 threads busy at a nominal K of 8
  p.group = 1;
  while (p.group < kWave &&
         (long long)n_band * nwf * n_seg * 8 * (2 * p.group) <= 1024)
    p.group *= 2;
  *lds = ((size_t)n_band * nwf * n_seg * p.K +
          (size_t)n_region * p.K * (1 + (size_t)n_band) +
          (size_t)n_region * ((size_t)n_seg + n_band)) *
         sizeof(double);
  return p;
}
}  // namespace

// The folds of a launch with wind-vector pairs -- wb2_det_combine over the
// per-variable partials and wb2_det_combine(WB2_MODE_WIND) over the pairs' -- as
// ONE launch (det_combine2_kernel) when both fit the default dynamic LDS;
// the two launches otherwise.
int combine_det_and_wind(const wb2_plan_tables& t, int mode, int skipna,
                         const double* partials, int64_t n_outer,
                         const double* wind_partials, int64_t n_pair,
                         double* metrics, double* wind_metrics, void* stream) {
  const int nwf = t.wfield ? 2 : 1;
  size_t lds_a = 0, lds_b = 0;
  const CombineParams a = combine_params(
      mode, skipna, wb2_num_slots(mode, skipna), partials, n_outer, t.n_chunk,
      nwf, t.n_seg, t.seg_eoff, t.n_ts, t.band_chunk0, t.n_band, t.coef_band,
      t.coef_seg, t.region_wf, t.region_wsum, t.n_region, nullptr, metrics,
      &lds_a);
  const CombineParams b = combine_params(
      WB2_MODE_WIND, skipna, wb2_num_slots(WB2_MODE_WIND, skipna),
      wind_partials, n_pair, t.n_chunk, nwf, t.n_seg, t.seg_eoff, t.n_ts,
      t.band_chunk0, t.n_band, t.coef_band, t.coef_seg, t.region_wf,
      t.region_wsum, t.n_region, nullptr, wind_metrics, &lds_b);
  const size_t lds = lds_a > lds_b ? lds_a : lds_b;
  static const bool fused = [] {  // WB2HIP_FUSED_FOLDS=0: A/B runs
    const char* e = getenv("WB2HIP_FUSED_FOLDS");
    return !(e && e[0] == '0');
  }();
  if (!fused || lds > 64 * 1024 || n_outer <= 0 || n_pair <= 0 ||
      n_outer + n_pair >= (1ll << 31) || !partials || !wind_partials ||
      !metrics || !wind_metrics || !t.seg_eoff || !t.band_chunk0 ||
      !t.coef_band || !t.coef_seg || !t.region_wf || !t.region_wsum ||
      t.n_chunk <= 0 || t.n_seg <= 0 || t.n_ts < t.n_seg || t.n_band <= 0 ||
      t.n_region <= 0) {  // (the two calls below report what is wrong)
    int rc = wb2_det_combine(mode, skipna, partials, n_outer, t.n_chunk, nwf,
                             t.n_seg, t.seg_eoff, t.n_ts, t.band_chunk0,
                             t.n_band, t.coef_band, t.coef_seg, t.region_wf,
                             t.region_wsum, t.n_region, nullptr, metrics,
                             stream);
    if (rc != 0 || n_pair == 0) return rc;
    return wb2_det_combine(WB2_MODE_WIND, skipna, wind_partials, n_pair,
                           t.n_chunk, nwf, t.n_seg, t.seg_eoff, t.n_ts,
                           t.band_chunk0, t.n_band, t.coef_band, t.coef_seg,
                           t.region_wf, t.region_wsum, t.n_region, nullptr,
                           wind_metrics, stream);
  }
  hipLaunchKernelGGL(det_combine2_kernel, dim3((unsigned)(n_outer + n_pair)),
                     dim3(1024), lds, static_cast<hipStream_t>(stream), a, b);
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

int combine_slots(int mode, int skipna, int k_slots, const double* partials,
                  int64_t n_outer, int32_t n_chunk, int32_t nwf, int32_t n_seg,
                  const int32_t* seg_eoff, int32_t n_ts,
                  const int32_t* band_chunk0, int32_t n_band,
                  const double* coef_band, const double* coef_seg,
                  const int32_t* region_wf, const double* region_wsum,
                  int32_t n_region, double* sums, double* metrics,
                  void* stream) {
  WB2_EMPTY_OK(n_outer);
  WB2_REQUIRE(partials && seg_eoff && band_chunk0 && coef_band && coef_seg &&
                  region_wf && region_wsum,
              "null pointer argument");
  WB2_REQUIRE(n_outer >= 0 && n_chunk > 0 && n_ts >= n_seg &&
                  (nwf == 1 || nwf == 2) && n_seg > 0 && n_band > 0 &&
                  n_region > 0,
              "bad sizes");
  if (n_outer == 0) return 0;
  size_t lds = 0;
  const CombineParams p = combine_params(
      mode, skipna, k_slots, partials, n_outer, n_chunk, nwf, n_seg, seg_eoff,
      n_ts, band_chunk0, n_band, coef_band, coef_seg, region_wf, region_wsum,
      n_region, sums, metrics, &lds);
  // gfx950: a workgroup may take all 160 KiB of a CU's LDS; beyond the default
  // 64 KiB of dynamic LDS the kernel has to be told once
  WB2_REQUIRE(lds <= 160 * 1024,
              "region decomposition too fine for the combine kernel's LDS "
              "(%zu bytes > 160 KiB): n_band=%d n_seg=%d slots=%d",
              lds, n_band, n_seg, p.K);
  if (lds > 64 * 1024) {
    static std::atomic<size_t> allowed{64 * 1024};
    if (lds > allowed.load()) {
      WB2_HIP_OK(hipFuncSetAttribute(
          reinterpret_cast<const void*>(det_combine_kernel),
          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      allowed.store(160 * 1024);
    }
  }
  hipLaunchKernelGGL(det_combine_kernel, dim3((unsigned)n_outer), dim3(1024),
                     lds, static_cast<hipStream_t>(stream), p);
  WB2_HIP_OK(hipGetLastError());
  return 0;
}
}  // namespace wb2

extern "C" {

int wb2_ens_combine(int skipna, const double* partials, int64_t n_outer,
                    int32_t n_chunk, int32_t nwf, int32_t n_seg,
                    const int32_t* seg_eoff, int32_t n_ts,
                    const int32_t* band_chunk0, int32_t n_band,
                    const double* coef_band, const double* coef_seg,
                    const int32_t* region_wf, const double* region_wsum,
                    int32_t n_region, double* sums, double* metrics,
                    void* stream) {
  WB2_TRACE();
  return wb2_det_combine(WB2_MODE_ENS, skipna, partials, n_outer, n_chunk, nwf,
                         n_seg, seg_eoff, n_ts, band_chunk0, n_band, coef_band,
                         coef_seg, region_wf, region_wsum, n_region, sums,
                         metrics, stream);
}

int wb2_time_accumulate(const double* values, int64_t n_lead, int64_t n_time,
                        int64_t n_tail, int skipna, double* sum, double* count,
                        void* stream) {
  WB2_TRACE();
  return wb2_time_accumulate_scatter(WB2_F64, values, n_lead, n_time, n_tail,
                                     skipna, nullptr, sum, count, stream);
}

int wb2_time_accumulate_scatter(int dtype, const void* values, int64_t n_lead,
                                int64_t n_time, int64_t n_tail, int skipna,
                                const int64_t* dst, double* sum, double* count,
                                void* stream) {
  WB2_TRACE();
  return wb2_time_accumulate_runs(dtype, values, n_lead, n_time, n_tail, skipna,
                                  dst, 1, sum, count, stream);
}

int wb2_time_accumulate_runs(int dtype, const void* values, int64_t n_lead,
                             int64_t n_time, int64_t n_tail, int skipna,
                             const int64_t* dst, int64_t run, double* sum,
                             double* count, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_REQUIRE(run >= 1, "run=%lld", (long long)run);
  WB2_EMPTY_OK(n_lead);
  WB2_EMPTY_OK(n_time);
  WB2_EMPTY_OK(n_tail);
  // count == NULL without skipna: the caller counts the time steps itself
  WB2_REQUIRE(values && sum && (count || !skipna), "null pointer argument");
  const long long n = n_lead * n_tail;
  if (n == 0 || n_time == 0) return 0;
  WB2_REQUIRE(!dst || n % run == 0, "run=%lld does not divide %lld elements",
              (long long)run, n);
  const dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long long* d = reinterpret_cast<const long long*>(dst);
  if (dtype == WB2_F32)
    hipLaunchKernelGGL(time_accumulate_kernel<float>, grid, dim3(256), 0, s,
                       static_cast<const float*>(values), (long long)n_lead,
                       (long long)n_time, (long long)n_tail, skipna, d,
                       (long long)run, sum, count);
  else
    hipLaunchKernelGGL(time_accumulate_kernel<double>, grid, dim3(256), 0, s,
                       static_cast<const double*>(values), (long long)n_lead,
                       (long long)n_time, (long long)n_tail, skipna, d,
                       (long long)run, sum, count);
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

int wb2_det_suite_step(const wb2_plan_tables* plan, int mode, int dtype,
                       int skipna, const void* const* in,
                       const int64_t* const* slab, int aligned16,
                       int64_t n_outer, double* partials, double* metrics,
                       int64_t acc_lead, int64_t acc_time, int64_t acc_tail,
                       int acc_skipna, const int64_t* dst, double* sum,
                       double* count, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(plan != nullptr, "null plan");
  WB2_EMPTY_OK(n_outer);
  WB2_REQUIRE(in != nullptr || slab != nullptr, "null pointer argument");
  WB2_REQUIRE(metrics != nullptr, "metrics is null");
  const wb2_plan_tables& t = *plan;
  const int nm = mode >= WB2_MODE_GAUSS
                     ? (mode == WB2_MODE_GAUSS ? 2
                        : mode == WB2_MODE_GAUSS_THR ? 3
                        : mode == WB2_MODE_ENS_THR ? 4 : 1)
                     : WB2_NMETRIC;
  if (sum != nullptr || count != nullptr) {
    WB2_REQUIRE(sum && count, "sum and count go together");
    WB2_REQUIRE(acc_lead > 0 && acc_time > 0 && acc_tail > 0 &&
                    acc_lead * acc_time * acc_tail ==
                        (long long)nm * t.n_region * n_outer,
                "accumulate view [%lld][%lld][%lld] does not cover metrics "
                "[%d][%d][%lld]",
                (long long)acc_lead, (long long)acc_time, (long long)acc_tail,
                nm, t.n_region, (long long)n_outer);
  }
  int rc = stream_partials_impl(
      mode, dtype, skipna, in, slab, aligned16, n_outer, t.n_row, t.n_col,
      t.w_row, t.w_col, t.wfield, t.wfield_dtype, t.aux, t.scalar, t.chunk_row0,
      t.chunk_nrow, t.n_chunk, t.n_ctile, t.seg_col0, t.seg_eoff, t.n_seg,
      t.n_ts, partials, stream);
  if (rc != 0) return rc;
  rc = wb2_det_combine(mode, skipna, partials, n_outer, t.n_chunk,
                       t.wfield ? 2 : 1, t.n_seg, t.seg_eoff, t.n_ts,
                       t.band_chunk0, t.n_band, t.coef_band, t.coef_seg,
                       t.region_wf, t.region_wsum, t.n_region, nullptr, metrics,
                       stream);
  if (rc != 0 || sum == nullptr) return rc;
  return wb2_time_accumulate_scatter(WB2_F64, metrics, acc_lead, acc_time,
                                     acc_tail, acc_skipna, dst, sum, count,
                                     stream);
}

int wb2_det_wind_suite_step(const wb2_plan_tables* plan, int mode, int dtype,
                            int skipna, const void* const* in,
                            const int64_t* const* slab, int aligned16,
                            int64_t n_outer, int64_t n_pair, double* partials,
                            double* wind_partials, double* metrics,
                            double* wind_metrics, void* stream) {
  WB2_TRACE();
  return wb2::det_wind_suite_step_streams(
      plan, mode, dtype, skipna, in, slab, aligned16, n_outer, n_pair, partials,
      wind_partials, metrics, wind_metrics, stream, nullptr, nullptr);
}

}  // extern "C"

namespace wb2 {
int det_wind_suite_step_streams(const wb2_plan_tables* plan, int mode,
                                int dtype, int skipna, const void* const* in,
                                const int64_t* const* slab, int aligned16,
                                int64_t n_outer, int64_t n_pair,
                                double* partials, double* wind_partials,
                                double* metrics, double* wind_metrics,
                                void* stream, void* pair_stream,
                                void* join_event) {
  WB2_REQUIRE(plan != nullptr, "null plan");
  WB2_EMPTY_OK(n_outer);
  WB2_REQUIRE(in != nullptr || slab != nullptr, "null pointer argument");
  WB2_REQUIRE(metrics != nullptr && (n_pair == 0 || wind_metrics != nullptr),
              "metrics is null");
  const wb2_plan_tables& t = *plan;
  int rc = stream_pairs_impl(mode, dtype, skipna, in, slab, aligned16, n_outer,
                             n_pair, t.n_row, t.n_col, t.w_row, t.w_col,
                             t.wfield, t.wfield_dtype, t.chunk_row0,
                             t.chunk_nrow, t.n_chunk, t.n_ctile, t.seg_col0,
                             t.seg_eoff, t.n_seg, t.n_ts, partials,
                             wind_partials, stream, pair_stream, join_event);
  if (rc != 0) return rc;
  return combine_det_and_wind(t, mode, skipna, partials, n_outer, wind_partials,
                              n_pair, metrics, wind_metrics, stream);
}
}  // namespace wb2

extern "C" {

int wb2_gather_accumulate(const double* arena, const int32_t* src,
                          const uint8_t* round32, int64_t n_out, int64_t n_time,
                          int skipna, const int64_t* sum_addr,
                          const int64_t* count_addr, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_EMPTY_OK(n_out);
  WB2_EMPTY_OK(n_time);
  WB2_REQUIRE(arena && src && round32 && sum_addr && count_addr,
              "null pointer argument");
  hipLaunchKernelGGL(gather_accumulate_kernel,
                     dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), arena, src, round32,
                     (long long)n_out, (long long)n_time, skipna,
                     reinterpret_cast<const long long*>(sum_addr),
                     reinterpret_cast<const long long*>(count_addr),
                     (const long long*)nullptr, (const int*)nullptr,
                     (const long long*)nullptr);
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

int wb2_gather_accumulate_rows(const double* arena, const int32_t* src,
                               const uint8_t* round32, int64_t n_out,
                               int64_t n_time, int skipna,
                               const int64_t* sum_addr,
                               const int64_t* count_addr, const int64_t* rows,
                               const int32_t* sel, const int64_t* rsz8,
                               void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_EMPTY_OK(n_out);
  WB2_EMPTY_OK(n_time);
  WB2_REQUIRE(arena && src && round32 && sum_addr && count_addr && rows &&
                  sel && rsz8,
              "null pointer argument");
  hipLaunchKernelGGL(gather_accumulate_kernel,
                     dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), arena, src, round32,
                     (long long)n_out, (long long)n_time, skipna,
                     reinterpret_cast<const long long*>(sum_addr),
                     reinterpret_cast<const long long*>(count_addr),
                     reinterpret_cast<const long long*>(rows), sel,
                     reinterpret_cast<const long long*>(rsz8));
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

}  // extern "C"

namespace wb2 {
namespace {
int seeps_map_impl(int dtype, const void* const* in,
                   const int64_t* const* slab, bool by_addr, int64_t n_outer,
                   int64_t n_point, const double* aux, double scalar,
                   double* out, void* stream) {
  if (n_outer == 0 || n_point == 0) return 0;
  SeepsMapParams p{};
  p.by_addr = by_addr;
  for (int i = 0; i < 3; ++i) {
    p.in[i] = in ? in[i] : nullptr;
    p.slab[i] = slab ? reinterpret_cast<const long long*>(slab[i]) : nullptr;
  }
  p.aux = aux;
  p.scalar = scalar;
  p.out = out;
  p.n_outer = n_outer;
  p.n_point = n_point;
  const long long n_group = (n_outer + kSeepsMapSlabs - 1) / kSeepsMapSlabs;
  const long long gy = n_group < 32768 ? n_group : 32768;
  const long long gz = (n_group + gy - 1) / gy;
  WB2_REQUIRE(gz <= 65535, "n_outer=%lld too large", (long long)n_outer);
  const dim3 grid((unsigned)((n_point + 255) / 256), (unsigned)gy, (unsigned)gz);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == WB2_F32)
    hipLaunchKernelGGL(seeps_map_kernel<float>, grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(seeps_map_kernel<double>, grid, dim3(256), 0, s, p);
  WB2_HIP_OK(hipGetLastError());
  return 0;
}
}  // namespace
}  // namespace wb2

extern "C" {

int wb2_seeps_map(int dtype, const void* const* in, const int64_t* const* slab,
                  int64_t n_outer, int64_t n_point, const double* aux,
                  double scalar, double* out, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_EMPTY_OK(n_outer);
  WB2_EMPTY_OK(n_point);
  WB2_REQUIRE(in && in[0] && in[1] && in[2] && aux && out,
              "null pointer argument");
  return seeps_map_impl(dtype, in, slab, false, n_outer, n_point, aux, scalar,
                        out, stream);
}

int wb2_seeps_map_addr(int dtype, const int64_t* const* addr, int64_t n_outer,
                       int64_t n_point, const double* aux, double scalar,
                       double* out, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_EMPTY_OK(n_outer);
  WB2_EMPTY_OK(n_point);
  WB2_REQUIRE(addr && addr[0] && addr[1] && addr[2] && aux && out,
              "null pointer argument");
  return seeps_map_impl(dtype, nullptr, addr, true, n_outer, n_point, aux,
                        scalar, out, stream);
}

}  // extern "C"
