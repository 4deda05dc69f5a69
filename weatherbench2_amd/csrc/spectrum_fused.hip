// K4f of libwb2hip.so: the zonal energy spectrum as ONE kernel.
//
// Same arithmetic contract as spectrum.hip (ZonalEnergySpectrum.compute,
// /root/reference/weatherbench2/derived_variables.py:592-626), but the real FFT
// along longitude is done here, in LDS, so each input value is read once and
// each spectrum value written once (8 B per grid point instead of the 16 B of
// the rocFFT + epilogue pipeline).  Used for float32 rows whose length N is even
// with N/2 in the instantiated set (0.25-degree N = 1440 included); everything
// else takes the rocFFT path.
//
// One WAVE transforms one latitude row of N/2 complex points
// z[m] = x[2m] + i x[2m+1] in (at most) three Stockham passes (fft_core.hpp:
// 720 = 12 x 12 x 5, i.e. 60, 60 and 3 x 48 busy lanes):
//   * pass 0 reads its inputs straight from HBM (8-byte loads, consecutive lanes
//     = consecutive complex points) and writes each butterfly's R outputs to the
//     wave's LDS slab as one contiguous run;
//   * the later passes read R strided points per butterfly from the slab,
//     multiply by the inter-pass twiddles -- which depend on the lane only and
//     therefore live in VGPRs for the whole kernel -- run a radix-R butterfly in
//     registers (composite radices 6/8/10/12 are Cooley-Tukey inside the lane,
//     their twiddles compile-time constants) and write back in place (LDS is
//     in-order per wave: one slab, no barrier);
//   * the real-FFT recombination X[k] = E[k] + W^k O[k] is evaluated for the
//     bin pair (k, N/2 - k) from one pair of LDS reads, scaled (x{1,2}, x
//     circumference in fp64) and stored -- or, with TIME, summed in registers
//     over the wave's n_time rows (the time mean of
//     scripts/compute_zonal_energy_spectrum.py:234) and stored once.
// Twiddles come from two small tables in the row dtype (evaluated in fp64,
// rounded once) that the plan owns.  float64 rows run the same code on
// complex128 points (no packed arithmetic: twice the VALU instructions, twice
// the LDS).

#include "common.hpp"

// 8-byte LDS stores stay single ds_write_b64 (17 cycles per instruction and
// SIMD, tools/valu_rate.hip) instead of the ds_write2_b64 pairs hipcc's
// load/store optimiser makes of them (44): volatile stores in address space 3
#if defined(__HIP_DEVICE_COMPILE__)
namespace wb2 {
template <typename C>
__device__ __forceinline__ void fused_slab_store(C* p, C v) {
  *(__attribute__((address_space(3))) volatile C*)p = v;
}
}  // namespace wb2
#define WB2_FFT_SLAB_STORE(ptr, value) ::wb2::fused_slab_store(ptr, value)
#endif
#include "fft_core.hpp"
#include "wb2hip.h"

#ifndef WB2_FFT_MAX_BLOCKS
#define WB2_FFT_MAX_BLOCKS 2048   // persistent workgroups (4 waves each)
#endif
#ifndef WB2_FFT_NWAVE
#define WB2_FFT_NWAVE 4   // waves (= rows in flight) per workgroup
#endif

// input rows are read once: non-temporal loads.  The spectra are written once
// too, but plain stores (write-back through L2) are the faster ones:
// MATERIALISE 0.279 / 0.304 ms per 16 units against 0.303 / 0.331 with
// non-temporal stores (profiles/r03_k4_ab11_summary.txt).
#define WB2_FFT_LOAD(p) __builtin_nontemporal_load(p)
#define WB2_FFT_STORE(v, p) (*(p) = (v))

#include <cstdlib>
#include <type_traits>

#ifndef WB2_FFT_PAIRED_DEFAULT
#define WB2_FFT_PAIRED_DEFAULT 0
#endif

namespace wb2 {
namespace fused {

using namespace fftcore;

// ---- inter-pass twiddle multiplies --------------------------------------------
// a * w as v_pk_mul_f32 + v_pk_fma_f32 with the rotation (-w.y, w.x) expressed
// through op_sel / neg_lo (hipcc materialises it with two extra VALU moves per
// twiddle, or keeps a second register pair per twiddle alive).  gfx950 needs one
// wait state between a packed-fp32 write and a dependent VALU read; the compiler
// cannot see inside the asm, so the multiplies of a block are interleaved and the
// block ends with the wait state for whatever consumes the last result.
#define WB2_PK_MUL(t, a, w) "v_pk_mul_f32 " t ", " a ", " w " op_sel_hi:[0,1]\n\t"
#define WB2_PK_FMA(t, a, w)                                       \
  "v_pk_fma_f32 " t ", " a ", " w ", " t                          \
  " op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
__device__ __forceinline__ void cmul1_asm(cf& a0, cf w0) {
  cf t0;
  asm(WB2_PK_MUL("%0", "%1", "%2") "s_nop 0\n\t" WB2_PK_FMA("%0", "%1", "%2")
      "s_nop 0"
      : "=&v"(t0)
      : "v"(a0), "v"(w0));
  a0 = t0;
}
__device__ __forceinline__ void cmul2_asm(cf& a0, cf w0, cf& a1, cf w1) {
  cf t0, t1;
  asm(WB2_PK_MUL("%0", "%2", "%3") WB2_PK_MUL("%1", "%4", "%5")
      WB2_PK_FMA("%0", "%2", "%3") WB2_PK_FMA("%1", "%4", "%5") "s_nop 0"
      : "=&v"(t0), "=&v"(t1)
      : "v"(a0), "v"(w0), "v"(a1), "v"(w1));
  a0 = t0;
  a1 = t1;
}
__device__ __forceinline__ void cmul3_asm(cf& a0, cf w0, cf& a1, cf w1, cf& a2,
                                          cf w2) {
  cf t0, t1, t2;
  asm(WB2_PK_MUL("%0", "%3", "%4") WB2_PK_MUL("%1", "%5", "%6")
      WB2_PK_MUL("%2", "%7", "%8") WB2_PK_FMA("%0", "%3", "%4")
      WB2_PK_FMA("%1", "%5", "%6") WB2_PK_FMA("%2", "%7", "%8") "s_nop 0"
      : "=&v"(t0), "=&v"(t1), "=&v"(t2)
      : "v"(a0), "v"(w0), "v"(a1), "v"(w1), "v"(a2), "v"(w2));
  a0 = t0;
  a1 = t1;
  a2 = t2;
}
#undef WB2_PK_MUL
#undef WB2_PK_FMA

template <int N>
__device__ __forceinline__ void twiddle_block(cf* v, const cf* tw) {
  // v[0..N) *= tw[0..N)
  if constexpr (N >= 3) {
    cmul3_asm(v[0], tw[0], v[1], tw[1], v[2], tw[2]);
    twiddle_block<N - 3>(v + 3, tw + 3);
  } else if constexpr (N == 2) {
    cmul2_asm(v[0], tw[0], v[1], tw[1]);
  } else if constexpr (N == 1) {
    cmul1_asm(v[0], tw[0]);
  }
}

// float64 rows: plain complex multiplies (v_mul_f64 / v_fma_f64; there is no
// packed float64 arithmetic to hand-schedule)
template <int N>
__device__ __forceinline__ void twiddle_block(cd* v, const cd* tw) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = cmul(v[i], tw[i]);
}

// One read of a complex point from the wave's LDS slab (or the shared twiddle
// table): 8 bytes for float32 rows, 16 for float64 ones.
template <typename C>
__device__ __forceinline__ C lds_read(const C* p) {
  // its own ds_read_b64: hipcc's load/store optimiser otherwise pairs the
  // 8-byte slab reads into ds_read2_b64 / ds_read2st64_b64, which gfx950
  // services at 44 cycles per wave-instruction against 6.4 for a ds_read_b64
  // (tools/valu_rate.hip); volatile in address space 3 is the one thing the
  // optimiser does not merge
  typedef __attribute__((address_space(3))) const volatile C* lds_ptr;
  return *(lds_ptr)p;
}

struct FusedParams {
  const void* x;     // float or double rows
  const void* twz;   // [N2]      exp(-2 pi i j / N2), complex of the row dtype
  const void* twq;   // [N2/2+1]  exp(-2 pi i k / N) * (-i) * (0.5 / N)
  const double* circ;  // MATERIALISE / TIME: circumference[n_lat];
                       // LATSEG: row weight[n_lat] = circumference x latitude weight
  double* out;         // MATERIALISE / TIME: spectra; LATSEG: partial[n_field][n_seg][N2+1]
  long long n_rows;   // input rows
  long long n_time;   // TIME: input rows are [n_time][n_rows / n_time]
  int n_lat;
  int n_seg;          // LATSEG: latitude segments per field
  int skipna;
};

// A later pass (NS > 1) over the wave's slab: strided reads, twiddle multiplies
// (the twiddles of a lane depend on the lane only: VGPR-resident for the whole
// kernel), butterflies, in-place writes.
template <typename P, int R, typename C>
__device__ __forceinline__ void lds_pass(C* __restrict__ z, int lane,
                                         C (&tw)[P::ROUNDS][R - 1]) {
  C v[P::ROUNDS][R];
  P::load([&](int i) { return lds_read(z + i); }, lane, v);
#pragma unroll
  for (int rd = 0; rd < P::ROUNDS; ++rd)
    twiddle_block<R - 1>(&v[rd][1], &tw[rd][0]);
  P::butterflies(v);
  // every read of this pass precedes every write (program order; the DS
  // operations of one wave execute in order) -- the fence only restrains the
  // compiler
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  P::store(z, lane, v);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// What a wave does with the spectra of its rows:
//   MATERIALISE  stores every row's spectrum (ZonalEnergySpectrum.compute);
//   TIME         owns one OUTPUT row, transforms its n_time input rows in time
//                order with the bin powers summed in registers (fp64, NaN spectra
//                skipped with skipna like xbeam.Mean) and stores the mean once:
//                the time mean of scripts/compute_zonal_energy_spectrum.py:234
//                fused -- 4 B read per grid point and almost nothing written;
//   LATSEG       owns one (field, latitude segment): sums weight[lat] x spectrum
//                over the segment's consecutive rows in registers and stores one
//                partial spectrum (BASELINE configs[3]: the area-weighted latitude
//                mean of the spectrum without materialising it; the partials of a
//                field are added in segment order by latseg_combine_kernel).
enum { MATERIALISE = 0, TIME_MEAN = 1, LATSEG = 2 };

template <int N2, int MODE, typename S = float>
__global__ void __launch_bounds__(64 * WB2_FFT_NWAVE)
    fused_spectrum_kernel(const FusedParams p) {
  typedef cx<S> C;    // one complex point
  typedef cx2<S> C2;  // two adjacent ones
  using PL = Plan<N2>;
  constexpr int R0 = PL::R0, R1 = PL::R1, R2 = PL::R2;
  using P0 = Pass<N2, R0, 1, 1, 0, PL::PAD0>;
  using P1 = Pass<N2, R1, R0, R0, PL::PAD0, PL::PAD1>;
  using P2 = Pass<N2, R2, R0 * R1, R0 * R1, PL::PAD1, 0>;
  constexpr int N = 2 * N2, NB = N2 + 1, NWAVE = WB2_FFT_NWAVE;
  constexpr int NH = N2 / 2 + 1;  // bin pairs (k, N2 - k), k = 0..N2/2
  constexpr int NIT = (NH + kWave - 1) / kWave;
  constexpr bool REDUCE = MODE != MATERIALISE;
  __shared__ __attribute__((aligned(16))) C s_twq[NH + 1];
  __shared__ __attribute__((aligned(16))) C s_z[NWAVE][slab_slots<N2>()];
  const C* g_twz = static_cast<const C*>(p.twz);
  const C* g_twq = static_cast<const C*>(p.twq);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  for (int i = threadIdx.x; i <= NH; i += blockDim.x)
    s_twq[i] = g_twq[i < NH ? i : NH - 1];
  // inter-pass twiddles: functions of the lane only, resident in VGPRs
  // float32: resident in VGPRs for the whole kernel.  float64: twice the
  // registers (92 for 720 points) -- they are fetched again for every row,
  // right before the pass that uses them (the 11.5 KB table stays in L1 / L2),
  // so that a pass's twiddles and its butterfly inputs share the registers
  constexpr bool RESIDENT_TW = sizeof(S) == 4;
  C tw1[P1::ROUNDS][P1::NTW], tw2[P2::ROUNDS][P2::NTW];
  if constexpr (RESIDENT_TW) {
    P1::load_twiddles(g_twz, lane, tw1);
    if constexpr (R2 > 1) P2::load_twiddles(g_twz, lane, tw2);
  }
  __syncthreads();
  C* z = s_z[wave];
  const S half_inv_n = (S)0.5 / (S)N;
  // output rows (MATERIALISE: = input rows; TIME: n_rows / n_time; LATSEG:
  // (field, segment) pairs) and the input rows each one reduces
  long long rows_out;
  if constexpr (MODE == TIME_MEAN) rows_out = p.n_rows / p.n_time;
  else if constexpr (MODE == LATSEG) rows_out = p.n_rows / p.n_lat * p.n_seg;
  else rows_out = p.n_rows;
  const long long stride = (long long)gridDim.x * NWAVE;
  // the input rows of output row o: row0 + t * row_step, t = 0 .. nt - 1
  struct Task {
    long long row0, nt, row_step;
    int lat0;
  };
  auto task_of = [&](long long o) {
    Task k{o, 1, 0, 0};
    if constexpr (MODE == TIME_MEAN) {
      k.nt = p.n_time;
      k.row_step = rows_out;
    } else if constexpr (MODE == LATSEG) {
      const long long field = o / p.n_seg;
      const int seg = (int)(o - field * p.n_seg);
      k.lat0 = (int)((long long)seg * p.n_lat / p.n_seg);  // balanced split
      k.nt = (long long)(seg + 1) * p.n_lat / p.n_seg - k.lat0;
      k.row0 = field * p.n_lat + k.lat0;
      k.row_step = 1;
    }
    return k;
  };
  const long long orow_first = (long long)blockIdx.x * NWAVE + wave;
  Task next_task = task_of(orow_first < rows_out ? orow_first : 0);
  for (long long orow_i = orow_first; orow_i < rows_out; orow_i += stride) {
    double sum1[NIT], sum2[NIT];
    int cnt[NIT];  // TIME + skipna: valid spectra, bin k (low half) / N2 - k
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      sum1[i] = sum2[i] = 0.0;
      cnt[i] = 0;
    }
    const Task task = next_task;
    if (orow_i + stride < rows_out)
      next_task = task_of(orow_i + stride);  // once per task
    const long long nt = task.nt, row0 = task.row0, row_step = task.row_step;
    const int lat0 = task.lat0;
    double c = 0.0;
    if constexpr (MODE != LATSEG) c = p.circ[(unsigned)(orow_i % p.n_lat)];
    double* orow = p.out + orow_i * NB;
    for (long long t = 0; t < nt; ++t) {
      if constexpr (MODE == LATSEG) c = p.circ[lat0 + t * row_step];
      const double c2 = 2.0 * c;
      {  // ---- pass 0: HBM -> butterflies -> contiguous runs in the slab
        C v[P0::ROUNDS][R0];
        const C* src = reinterpret_cast<const C*>(
            static_cast<const S*>(p.x) + (row0 + t * row_step) * N);
        P0::load([&](int i) { return WB2_FFT_LOAD(src + i); },
                 lane, v);
        P0::butterflies(v);
        P0::store(z, lane, v);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
      if constexpr (!RESIDENT_TW) {
        const C* tz = g_twz;
        asm volatile("" : "+s"(tz));  // per row: not hoisted out of the loop
        P1::load_twiddles(tz, lane, tw1);
      }
      lds_pass<P1, R1>(z, lane, tw1);
      if constexpr (R2 > 1) {
        if constexpr (!RESIDENT_TW) {
          const C* tz = g_twz;
          asm volatile("" : "+s"(tz));
          P2::load_twiddles(tz, lane, tw2);
        }
        lds_pass<P2, R2>(z, lane, tw2);
      }
      if constexpr (!REDUCE) {
        // ---- materialising kernel: a lane owns the ADJACENT bins k0, k0 + 1
        // (and their mirrors N2 - k0, N2 - k0 - 1), so the fp64 spectrum leaves
        // in 16-byte stores (1 KiB per wave instruction instead of 512 B)
        static_assert(N2 % 4 == 0, "adjacent-bin epilogue needs N2/2 even");
        typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
        constexpr int NIT2 = (NH + 2 * kWave - 1) / (2 * kWave);
#pragma unroll
        for (int i = 0; i < NIT2; ++i) {
          const int k0 = 2 * lane + i * 2 * kWave;
          if (k0 <= N2 / 2) {
            const C2 a01 = *reinterpret_cast<const C2*>(z + k0);
            const C a0 = {a01.x, a01.y}, a1 = {a01.z, a01.w};
            const C b0 = lds_read(z + ((i == 0 && k0 == 0) ? 0 : N2 - k0));
            const C b1 = lds_read(z + (N2 - k0 - 1));
            const C2 w01 = *reinterpret_cast<const C2*>(s_twq + k0);
            S p1a, p2a, p1b, p2b;
            recombine_pair(a0, b0, C{w01.x, w01.y}, half_inv_n, p1a, p2a);
            recombine_pair(a1, b1, C{w01.z, w01.w}, half_inv_n, p1b, p2b);
            // derived_variables.py:600: every bin but 0 is doubled (Nyquist too)
            const double v1a = (double)p1a * ((i == 0 && k0 == 0) ? c : c2);
            const double v1b = (double)p1b * c2;
            const double v2a = (double)p2a * c2, v2b = (double)p2b * c2;
            if (k0 < N2 / 2) {
              WB2_FFT_STORE((d2{v1a, v1b}),
                                          reinterpret_cast<d2*>(orow + k0));
              WB2_FFT_STORE(
                  (d2{v2b, v2a}), reinterpret_cast<d2*>(orow + N2 - k0 - 1));
            } else {  // k0 == N2 / 2: its own mirror; bin k0 + 1 belongs to k0 - 2
              WB2_FFT_STORE(v1a, orow + k0);
            }
          }
        }
      } else {
        // ---- recombination + power for the bin pairs (k, N2 - k)
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int k = lane + i * kWave;
          if ((i + 1) * kWave <= NH || k < NH) {
            const C a = lds_read(z + k);
            const C b = lds_read(z + ((i == 0 && k == 0) ? 0 : N2 - k));
            S p1, p2;
            recombine_pair(a, b, lds_read(s_twq + k), half_inv_n, p1, p2);
            // derived_variables.py:600: every bin but 0 is doubled (Nyquist too)
            const double v1 = (double)p1 * ((i == 0 && k == 0) ? c : c2);
            const double v2 = (double)p2 * c2;
            if constexpr (MODE == TIME_MEAN) {
              const bool k1 = !(p.skipna && is_nan(v1));
              const bool k2 = !(p.skipna && is_nan(v2));
              sum1[i] += k1 ? v1 : 0.0;
              sum2[i] += k2 ? v2 : 0.0;
              cnt[i] += (k1 ? 1 : 0) + (k2 ? 0x10000 : 0);
            } else if constexpr (MODE == LATSEG) {
              sum1[i] += v1;
              sum2[i] += v2;
            } else {
              WB2_FFT_STORE(v1, orow + k);
              if (2 * k != N2) WB2_FFT_STORE(v2, orow + N2 - k);
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }  // reduced rows
    if constexpr (REDUCE) {
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int k = lane + i * kWave;
        if (k < NH) {
          double o1 = sum1[i], o2 = sum2[i];
          if constexpr (MODE == TIME_MEAN) {
            o1 /= (double)(cnt[i] & 0xffff);
            o2 /= (double)(cnt[i] >> 16);
          }
          WB2_FFT_STORE(o1, orow + k);
          if (2 * k != N2) WB2_FFT_STORE(o2, orow + N2 - k);
        }
      }
    }
  }
}

// The reducing modes with the last pass PAIRED in the lane (fft_core.hpp:
// PairedLast, PairedPlan: 1440 = 2 x 20 x 6 x 6).  Passes 0 and 1 as above;
// the last pass runs the butterflies j and T - j of a lane side by side, the
// recombination takes its bin pairs from registers: no store of the last pass,
// no load of the recombination -- the row crosses LDS twice instead of three
// times and a wave's dependent LDS round trips per row drop from three to two
// (profiles/r03_k4_stalls.md section 4: the kernel waits on that chain, not on
// a saturated unit).  Each lane keeps the 12 bins of its two butterflies.
#ifndef WB2_FFT_PAIRED_WAVES
#define WB2_FFT_PAIRED_WAVES 0   // > 0: waves per SIMD the registers are cut to
#endif
#if WB2_FFT_PAIRED_WAVES > 0
#define WB2_FFT_PAIRED_OCC \
  __attribute__((amdgpu_waves_per_eu(WB2_FFT_PAIRED_WAVES, WB2_FFT_PAIRED_WAVES)))
#else
#define WB2_FFT_PAIRED_OCC
#endif
template <int N2, int MODE, typename S = float>
__global__ void __launch_bounds__(64 * WB2_FFT_NWAVE) WB2_FFT_PAIRED_OCC
    fused_spectrum_paired_kernel(const FusedParams p) {
  static_assert(MODE == TIME_MEAN || MODE == LATSEG, "reducing modes");
  typedef cx<S> C;
  using PL = PairedPlan<N2>;
  constexpr int R0 = PL::R0, R1 = PL::R1, R2 = PL::R2;
  using P0 = Pass<N2, R0, 1, 1, 0, PL::PAD0>;
  using P1 = Pass<N2, R1, R0, R0, PL::PAD0, PL::PAD1>;
  using PP = PairedLast<N2, R2, PL::PAD1>;
  constexpr int N = 2 * N2, NB = N2 + 1, NWAVE = WB2_FFT_NWAVE, H = PP::H;
  __shared__ __attribute__((aligned(16))) C s_z[NWAVE][slab_slots<N2, PL>()];
  const C* g_twz = static_cast<const C*>(p.twz);
  const C* g_twq = static_cast<const C*>(p.twq);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  // twiddles of passes 1 and 2 and of the recombination: functions of the lane
  C tw1[P1::ROUNDS][P1::NTW], tw2[2][R2 - 1], wq[2][H];
  P1::load_twiddles(g_twz, lane, tw1);
  PP::load_twiddles(g_twz, lane, tw2);
  PP::load_recombination(g_twq, lane, wq);
  C* z = s_z[wave];
  const S half_inv_n = (S)0.5 / (S)N;
  long long rows_out;
  if constexpr (MODE == TIME_MEAN) rows_out = p.n_rows / p.n_time;
  else rows_out = p.n_rows / p.n_lat * p.n_seg;
  const long long stride = (long long)gridDim.x * NWAVE;
  struct Task {
    long long row0, nt, row_step;
    int lat0;
  };
  auto task_of = [&](long long o) {
    Task k{o, 1, 0, 0};
    if constexpr (MODE == TIME_MEAN) {
      k.nt = p.n_time;
      k.row_step = rows_out;
    } else {
      const long long field = o / p.n_seg;
      const int seg = (int)(o - field * p.n_seg);
      k.lat0 = (int)((long long)seg * p.n_lat / p.n_seg);
      k.nt = (long long)(seg + 1) * p.n_lat / p.n_seg - k.lat0;
      k.row0 = field * p.n_lat + k.lat0;
      k.row_step = 1;
    }
    return k;
  };
  const long long orow_first = (long long)blockIdx.x * NWAVE + wave;
  Task next_task = task_of(orow_first < rows_out ? orow_first : 0);
  for (long long orow_i = orow_first; orow_i < rows_out; orow_i += stride) {
    double sum[2][H][2];
    int cnt[2][H];  // TIME + skipna: valid spectra of the pair's two bins
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int u = 0; u < H; ++u) {
        sum[s][u][0] = sum[s][u][1] = 0.0;
        cnt[s][u] = 0;
      }
    const Task task = next_task;
    if (orow_i + stride < rows_out) next_task = task_of(orow_i + stride);
    const long long nt = task.nt, row0 = task.row0, row_step = task.row_step;
    const int lat0 = task.lat0;
    double c = 0.0;
    if constexpr (MODE != LATSEG) c = p.circ[(unsigned)(orow_i % p.n_lat)];
    double* orow = p.out + orow_i * NB;
    for (long long t = 0; t < nt; ++t) {
      if constexpr (MODE == LATSEG) c = p.circ[lat0 + t * row_step];
      const double c2 = 2.0 * c;
      {  // ---- pass 0: HBM -> butterflies -> contiguous runs in the slab
        C v[P0::ROUNDS][R0];
        const C* src = reinterpret_cast<const C*>(
            static_cast<const S*>(p.x) + (row0 + t * row_step) * N);
        P0::load([&](int i) { return WB2_FFT_LOAD(src + i); }, lane, v);
        P0::butterflies(v);
        P0::store(z, lane, v);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
      lds_pass<P1, R1>(z, lane, tw1);
      {  // ---- the paired last pass + recombination, in registers
        C v[2][R2];
        PP::load([&](int i) { return lds_read(z + i); }, lane, v);
        twiddle_block<R2 - 1>(&v[0][1], &tw2[0][0]);
        twiddle_block<R2 - 1>(&v[1][1], &tw2[1][0]);
        Radix<R2>::run(v[0]);
        Radix<R2>::run(v[1]);
        PP::fix_lane0(lane, v);
        S pw[2][H][2];
        PP::recombine(v, wq, half_inv_n, pw);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int u = 0; u < H; ++u) {
            // derived_variables.py:600: every bin but 0 is doubled (Nyquist too)
            const bool first = s == 0 && u == 0 && lane == 0;
            const double v1 = (double)pw[s][u][0] * (first ? c : c2);
            const double v2 = (double)pw[s][u][1] * c2;
            if constexpr (MODE == TIME_MEAN) {
              const bool k1 = !(p.skipna && is_nan(v1));
              const bool k2 = !(p.skipna && is_nan(v2));
              sum[s][u][0] += k1 ? v1 : 0.0;
              sum[s][u][1] += k2 ? v2 : 0.0;
              cnt[s][u] += (k1 ? 1 : 0) + (k2 ? 0x10000 : 0);
            } else {
              sum[s][u][0] += v1;
              sum[s][u][1] += v2;
            }
          }
      }
      // (the next row's pass 0 overwrites the slab: after this row's reads)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int u = 0; u < H; ++u) {
        double o1 = sum[s][u][0], o2 = sum[s][u][1];
        if constexpr (MODE == TIME_MEAN) {
          o1 /= (double)(cnt[s][u] & 0xffff);
          o2 /= (double)(cnt[s][u] >> 16);
        }
        // (which of its 2 x H x 2 results the lane keeps, and where they go:
        // worked out here, once per output row, not held in registers)
        const int low = PP::low_bin(lane, s, u);
        if (PP::keeps(lane, s, u, 0)) WB2_FFT_STORE(o1, orow + low);
        if (PP::keeps(lane, s, u, 1)) WB2_FFT_STORE(o2, orow + (N2 - low));
      }
  }
}

// LATSEG second step: out[field][k] = scale * sum_seg partial[field][seg][k],
// segments added in order (deterministic).  Eight loads are in flight before the
// first add: with ~75 k threads the kernel is latency-bound otherwise (14.6 us
// for 19 MB of partials when every add waited for its own load).
__global__ void latseg_combine_kernel(const double* __restrict__ partial,
                                      long long n_field, int n_seg, int n_bins,
                                      double scale, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_field * n_bins) return;
  const long long f = i / n_bins;
  const int k = (int)(i - f * n_bins);
  const double* src = partial + (long long)f * n_seg * n_bins + k;
  double s = 0.0;
  int g = 0;
  for (; g + 8 <= n_seg; g += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      v[u] = WB2_FFT_LOAD(src + (long long)(g + u) * n_bins);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; g < n_seg; ++g) s += src[(long long)g * n_bins];
  out[i] = s * scale;
}

template <typename C>
__global__ void fused_twiddle_kernel(C* twz, C* twq, int n2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double sn, cs;
  if (i < n2) {
    sincospi(2.0 * (double)i / (double)n2, &sn, &cs);
    table_entry_z(i, n2, cs, sn, twz[i]);
  }
  if (i <= n2 / 2) {
    sincospi((double)i / (double)n2, &sn, &cs);
    table_entry_q(n2, cs, sn, twq[i]);
  }
}

// Resident workgroups of a kernel on this device (occupancy x CUs), cached.
template <typename K>
int resident_blocks(K kernel) {
  static int cached = 0;  // benign race: every thread computes the same value
  if (cached > 0) return cached;
  int dev = 0, per_cu = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) !=
          hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel,
                                                   64 * WB2_FFT_NWAVE, 0) !=
          hipSuccess ||
      per_cu <= 0 || cus <= 0)
    return 768;
  cached = per_cu * cus;
  return cached;
}

template <int N2, typename S>
int launch(const FusedParams& p, int mode, hipStream_t s) {
  long long rows_out = p.n_rows;
  if (mode == TIME_MEAN) rows_out = p.n_rows / p.n_time;
  if (mode == LATSEG) rows_out = p.n_rows / p.n_lat * p.n_seg;
  long long blocks = (rows_out + WB2_FFT_NWAVE - 1) / WB2_FFT_NWAVE;
  WB2_REQUIRE(p.n_time < 65536, "fused time mean: n_time=%lld exceeds 65535",
              p.n_time);
  // row-strided waves beyond that (the cap counts 4-wave workgroups)
  if (blocks > WB2_FFT_MAX_BLOCKS * 4 / WB2_FFT_NWAVE)
    blocks = WB2_FFT_MAX_BLOCKS * 4 / WB2_FFT_NWAVE;
  if constexpr (N2 == 720 && std::is_same<S, float>::value) {
    // the reducing modes with the last pass paired in the lane
    // (WB2HIP_FFT_PAIRED=0 | 1: A/B runs; the default is the measured winner)
    static const bool paired = [] {
      const char* e = getenv("WB2HIP_FFT_PAIRED");
      return e ? e[0] == '1' : WB2_FFT_PAIRED_DEFAULT != 0;
    }();
    if (paired && mode != MATERIALISE) {
      if (mode == TIME_MEAN)
        hipLaunchKernelGGL((fused_spectrum_paired_kernel<N2, TIME_MEAN, S>),
                           dim3((unsigned)blocks), dim3(64 * WB2_FFT_NWAVE), 0,
                           s, p);
      else
        hipLaunchKernelGGL((fused_spectrum_paired_kernel<N2, LATSEG, S>),
                           dim3((unsigned)blocks), dim3(64 * WB2_FFT_NWAVE), 0,
                           s, p);
      WB2_HIP_OK(hipGetLastError());
      return 0;
    }
  }
  if (mode == TIME_MEAN)
    hipLaunchKernelGGL((fused_spectrum_kernel<N2, TIME_MEAN, S>),
                       dim3((unsigned)blocks), dim3(64 * WB2_FFT_NWAVE), 0, s, p);
  else if (mode == LATSEG)
    hipLaunchKernelGGL((fused_spectrum_kernel<N2, LATSEG, S>),
                       dim3((unsigned)blocks), dim3(64 * WB2_FFT_NWAVE), 0, s, p);
  else
    hipLaunchKernelGGL((fused_spectrum_kernel<N2, MATERIALISE, S>),
                       dim3((unsigned)blocks), dim3(64 * WB2_FFT_NWAVE), 0, s, p);
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

// Latitude segments per field such that all (field, segment) tasks are resident
// at once (one task per wave, no second round).
template <int N2, typename S>
int latseg_segments(long long n_field, int n_lat) {
  const long long waves = (long long)WB2_FFT_NWAVE *
                          resident_blocks(fused_spectrum_kernel<N2, LATSEG, S>);
  // WB2HIP_LATSEG_ROUNDS (A/B runs): tasks = that many times the resident waves
  static const double rounds = [] {
    const char* e = getenv("WB2HIP_LATSEG_ROUNDS");
    const double v = e ? atof(e) : 1.0;
    return v > 0.0 ? v : 1.0;
  }();
  long long n_seg =
      (long long)((double)waves * rounds) / (n_field > 0 ? n_field : 1);
  if (n_seg < 1) n_seg = 1;
  if (n_seg > n_lat) n_seg = n_lat;
  return (int)n_seg;
}

}  // namespace fused

// Entry points used by spectrum.hip ------------------------------------------
#define WB2_FUSED_SIZES(X)                                                    \
  X(32) X(64) X(120) X(128) X(180) X(256) X(360) X(512) X(720) X(48)           \
  X(144) X(160) X(192) X(240) X(320) X(384) X(640) X(900) X(1024)       \
  X(1280) X(1440) X(1800)

bool fused_spectrum_supported(int dtype, int n_lon) {
  if ((dtype != WB2_F32 && dtype != WB2_F64) || n_lon % 2) return false;
  switch (n_lon / 2) {
#define WB2_CASE(N2) case N2:
    WB2_FUSED_SIZES(WB2_CASE)
#undef WB2_CASE
      return true;
  }
  return false;
}

size_t fused_spectrum_table_bytes(int dtype, int n_lon) {
  return (size_t)(n_lon / 2 + n_lon / 4 + 1) *
         (dtype == WB2_F32 ? sizeof(fused::cf) : sizeof(fused::cd));
}

// Fills the two twiddle tables (fused_spectrum_table_bytes(dtype, n_lon) bytes
// at `tables`, complex of the row dtype, evaluated in fp64); the plan does this
// once at creation and owns the memory.
int fused_spectrum_tables(void* tables, int dtype, int n_lon, hipStream_t s) {
  using namespace fused;
  const int n2 = n_lon / 2;
  const dim3 grid((unsigned)((n2 + 255) / 256));
  if (dtype == WB2_F32) {
    cf* twz = static_cast<cf*>(tables);
    hipLaunchKernelGGL(fused_twiddle_kernel<cf>, grid, dim3(256), 0, s, twz,
                       twz + n2, n2);
  } else {
    cd* twz = static_cast<cd*>(tables);
    hipLaunchKernelGGL(fused_twiddle_kernel<cd>, grid, dim3(256), 0, s, twz,
                       twz + n2, n2);
  }
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

namespace fused {
// (twz, twq) inside the plan's table block
inline void table_pointers(void* tables, int dtype, int n2, const void** twz,
                           const void** twq) {
  *twz = tables;
  *twq = static_cast<char*>(tables) +
         (size_t)n2 * (dtype == WB2_F32 ? sizeof(cf) : sizeof(cd));
}

template <typename S>
int launch_size(const FusedParams& p, int n2, int mode, hipStream_t s) {
  switch (n2) {
#define WB2_CASE(N2) case N2: return launch<N2, S>(p, mode, s);
    WB2_FUSED_SIZES(WB2_CASE)
#undef WB2_CASE
  }
  return fail("fused spectrum: n_lon=%d is not instantiated", 2 * n2);
}
}  // namespace fused

int fused_spectrum_run(const void* x, int dtype, long long n_rows, int n_lon,
                       const double* circ, int n_lat, long long n_time,
                       int skipna, double* out, void* tables, hipStream_t s) {
  using namespace fused;
  const int n2 = n_lon / 2;
  FusedParams p{x, nullptr, nullptr, circ, out, n_rows, n_time, n_lat, 0,
                skipna};
  table_pointers(tables, dtype, n2, &p.twz, &p.twq);
  const int mode = n_time > 0 ? TIME_MEAN : MATERIALISE;
  return dtype == WB2_F32 ? launch_size<float>(p, n2, mode, s)
                          : launch_size<double>(p, n2, mode, s);
}

int fused_spectrum_latmean_segments(int dtype, long long n_rows, int n_lon,
                                    int n_lat) {
  using namespace fused;
  switch (n_lon / 2) {
#define WB2_CASE(N2)                                                      \
  case N2:                                                                \
    return dtype == WB2_F32                                               \
               ? latseg_segments<N2, float>(n_rows / n_lat, n_lat)       \
               : latseg_segments<N2, double>(n_rows / n_lat, n_lat);
    WB2_FUSED_SIZES(WB2_CASE)
#undef WB2_CASE
  }
  return 1;
}

// out[field][bin] = scale * sum_lat row_weight[lat] * spectrum[field][lat][bin]
// (row_weight = circumference x latitude weight), partial[n_field][n_seg][bins]
// is scratch.
int fused_spectrum_latmean(const void* x, int dtype, long long n_rows,
                           int n_lon, const double* row_weight, int n_lat,
                           int n_seg, double scale, double* partial,
                           double* out, void* tables, hipStream_t s) {
  using namespace fused;
  const int n2 = n_lon / 2;
  FusedParams p{x, nullptr, nullptr, row_weight, partial, n_rows, 0, n_lat,
                n_seg, 0};
  table_pointers(tables, dtype, n2, &p.twz, &p.twq);
  const int rc = dtype == WB2_F32 ? launch_size<float>(p, n2, LATSEG, s)
                                  : launch_size<double>(p, n2, LATSEG, s);
  if (rc != 0) return rc;
  const long long n_field = n_rows / n_lat, n = n_field * (n2 + 1);
  hipLaunchKernelGGL(latseg_combine_kernel, dim3((unsigned)((n + 255) / 256)),
                     dim3(256), 0, s, partial, n_field, n_seg, n2 + 1, scale,
                     out);
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace wb2
