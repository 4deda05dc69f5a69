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
// Twiddles come from two small fp32 tables (evaluated in fp64, rounded once)
// that the plan owns.

#include "common.hpp"
#include "fft_core.hpp"
#include "wb2hip.h"

#ifndef WB2_FFT_MAX_BLOCKS
#define WB2_FFT_MAX_BLOCKS 2048   // persistent workgroups (4 waves each)
#endif
#ifndef WB2_FFT_MIN_WAVES
#define WB2_FFT_MIN_WAVES 1
#endif
#ifndef WB2_FFT_ASM_CMUL
#define WB2_FFT_ASM_CMUL 1   // 0: let hipcc build (-w.y, w.x) per twiddle (2 extra VALU)
#endif
#ifndef WB2_FFT_PREFETCH
// issue the next row's HBM loads before pass 1 of the current row (2 R0 extra
// VGPRs): bit 0 = materialising kernel, bit 1 = TIME kernel.  Measured round 2:
// no gain (the waves of a CU already overlap each other's loads), so off.
#define WB2_FFT_PREFETCH 0
#endif
#ifndef WB2_FFT_WIDE_STORE
#define WB2_FFT_WIDE_STORE 1  // materialising kernel: 16-byte stores of adjacent bins
#endif
#ifndef WB2_FFT_DYNAMIC
#define WB2_FFT_DYNAMIC 0   // 1: rows handed out through per-XCD atomic counters
                            // (measured round 2: slower, see profiles/r02_k4_notes.md)
#endif
#ifndef WB2_FFT_DIAG
// timing diagnostics only (wrong results): 1 / 2 skip LDS pass 1 / 2, 4 replace
// the recombination epilogue by a token read, 8 skip pass 0's butterflies and
// stores, 16 no global stores in the materialising kernel
#define WB2_FFT_DIAG 0
#endif
#ifndef WB2_FFT_TW_LDS
// inter-pass twiddles from an LDS table instead of VGPRs: bit 0 / 1 = pass 1 / 2
// of the materialising kernel, bit 2 / 3 = pass 1 / 2 of the TIME kernel
#define WB2_FFT_TW_LDS 0
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

struct FusedParams {
  const float* x;
  const cf* twz;   // [N2]      exp(-2 pi i j / N2)
  const cf* twq;   // [N2/2+1]  exp(-2 pi i k / N) * (-i) * (0.5 / N)
  const double* circ;
  double* out;
  unsigned* sched;    // dynamic row scheduling: 8 counters, 64 B apart, zeroed
  long long n_rows;   // input rows
  long long n_time;   // TIME: input rows are [n_time][n_rows / n_time]
  int n_lat;
  int skipna;
};

// Output rows are handed out dynamically in units of kUnitRows consecutive rows
// (TIME: one output row = n_time transforms): a wave that finishes early pulls
// the next unit instead of idling -- with ~3 output rows per resident wave a
// static split loses a quarter of the machine to rounding.  One counter per XCD
// (workgroup b runs on XCD b mod 8; only speed depends on that), counter c hands
// out the units c, c + 8, ...; the next unit is requested one unit ahead so the
// atomic's latency never shows.
constexpr int kSchedStride = 16;  // uints between counters (64 B)
template <bool TIME>
constexpr int unit_rows() { return TIME ? 1 : 8; }

// A later pass (NS > 1) over the wave's slab: strided reads, twiddle multiplies,
// butterflies, in-place writes.
template <typename P, int R>
__device__ __forceinline__ void lds_pass(cf* __restrict__ z, int lane,
                                         const cf (&tw)[P::ROUNDS][P::NTW]) {
  cf v[P::ROUNDS][R];
  P::load([&](int i) { return z[i]; }, lane, v);
#if WB2_FFT_ASM_CMUL
#pragma unroll
  for (int rd = 0; rd < P::ROUNDS; ++rd) twiddle_block<R - 1>(&v[rd][1], &tw[rd][0]);
#else
  P::twiddle(v, tw);
#endif
  P::butterflies(v);
  // every read of this pass precedes every write (program order; the DS
  // operations of one wave execute in order) -- the fence only restrains the
  // compiler
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  P::store(z, lane, v);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// The inter-pass twiddles of a lane depend on the lane only.  TWLDS = false:
// they live in VGPRs for the whole kernel (2 (R - 1) registers per round);
// TWLDS = true: they are re-read from a compact LDS table for every row (cheap:
// 8-byte conflict-free reads) to keep the register count of the TIME variant at
// three waves per SIMD.
template <typename P, int R, bool TWLDS>
struct PassTwiddles {
  cf reg[P::ROUNDS][P::NTW];
  int row[P::ROUNDS];
  __device__ __forceinline__ void init(const cf* twz, int lane) {
    if constexpr (TWLDS) {
#pragma unroll
      for (int rd = 0; rd < P::ROUNDS; ++rd) row[rd] = P::table_row(lane, rd);
    } else {
      P::load_twiddles(twz, lane, reg);
    }
  }
  __device__ __forceinline__ void run(cf* z, int lane, const cf* tbl) {
    if constexpr (TWLDS) {
      int rr[P::ROUNDS];
#pragma unroll
      for (int rd = 0; rd < P::ROUNDS; ++rd) {
        rr[rd] = row[rd];
        asm volatile("" : "+v"(rr[rd]));  // keep the reads inside the row loop
      }
      cf tw[P::ROUNDS][P::NTW];
      P::load_twiddles_table(tbl, rr, tw);
      lds_pass<P, R>(z, lane, tw);
    } else {
      lds_pass<P, R>(z, lane, reg);
    }
  }
};

// TIME: the mean over the leading time axis is fused: a wave owns one OUTPUT
// row, transforms its n_time input rows in time order with the bin powers summed
// in registers (fp64, NaN spectra skipped with skipna like xbeam.Mean) and stores
// the mean once -- 4 B read per grid point and almost nothing written.
template <int N2, bool TIME>
__global__ void __launch_bounds__(256, WB2_FFT_MIN_WAVES)
    fused_spectrum_kernel(const FusedParams p) {
  using PL = Plan<N2>;
  constexpr int R0 = PL::R0, R1 = PL::R1, R2 = PL::R2;
  using P0 = Pass<N2, R0, 1, 1, 0, PL::PAD0>;
  using P1 = Pass<N2, R1, R0, R0, PL::PAD0, PL::PAD1>;
  using P2 = Pass<N2, R2, R0 * R1, R0 * R1, PL::PAD1, 0>;
  constexpr int N = 2 * N2, NB = N2 + 1, NWAVE = 4;
  constexpr int NH = N2 / 2 + 1;  // bin pairs (k, N2 - k), k = 0..N2/2
  constexpr int NIT = (NH + kWave - 1) / kWave;
  constexpr bool PF = (WB2_FFT_PREFETCH & (TIME ? 2 : 1)) != 0;
  constexpr bool TW1_LDS = (WB2_FFT_TW_LDS & (TIME ? 4 : 1)) != 0;
  constexpr bool TW2_LDS = R2 > 1 && (WB2_FFT_TW_LDS & (TIME ? 8 : 2)) != 0;
  __shared__ __attribute__((aligned(16))) cf s_twq[NH + 1];
  __shared__ cf s_tw1[TW1_LDS ? (R1 - 1) * P1::KP : 1];
  __shared__ cf s_tw2[TW2_LDS ? (R2 - 1) * P2::KP : 1];
  __shared__ __attribute__((aligned(16))) cf s_z[NWAVE][slab_slots<N2>()];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  for (int i = threadIdx.x; i <= NH; i += blockDim.x)
    s_twq[i] = p.twq[i < NH ? i : NH - 1];
  if constexpr (TW1_LDS) P1::fill_table(p.twz, s_tw1, threadIdx.x, blockDim.x);
  if constexpr (TW2_LDS) P2::fill_table(p.twz, s_tw2, threadIdx.x, blockDim.x);
  PassTwiddles<P1, R1, TW1_LDS> t1;
  PassTwiddles<P2, R2, TW2_LDS> t2;
  t1.init(p.twz, lane);
  if constexpr (R2 > 1) t2.init(p.twz, lane);
  __syncthreads();
  cf* z = s_z[wave];
  const float half_inv_n = 0.5f / (float)N;
  const long long nt = TIME ? p.n_time : 1;
  const long long rows_out = p.n_rows / nt;
  long long orow_i;
#if WB2_FFT_DYNAMIC
  constexpr int UNIT = unit_rows<TIME>();
  const int xcd = blockIdx.x & 7;
  unsigned tok = 0;
  auto request = [&]() {
    if (lane == 0) tok = atomicAdd(p.sched + xcd * kSchedStride, 1u);
  };
  auto granted = [&]() -> long long {  // first row of the unit just granted
    return ((long long)__builtin_amdgcn_readfirstlane(tok) * 8 + xcd) * UNIT;
  };
  request();
  orow_i = granted();
  if (orow_i >= rows_out) return;
  long long unit_end = orow_i + UNIT < rows_out ? orow_i + UNIT : rows_out;
  request();
#else
  const long long stride = (long long)gridDim.x * NWAVE;
  orow_i = (long long)blockIdx.x * NWAVE + wave;
  if (orow_i >= rows_out) return;
#endif
  auto fetch = [&](long long row, cf (&v)[P0::ROUNDS][R0]) {
    const cf* src = reinterpret_cast<const cf*>(p.x + row * N);
    P0::load([&](int i) { return __builtin_nontemporal_load(src + i); }, lane,
             v);
  };
  // PF: the HBM loads of row i + 1 are in flight while row i goes through its
  // LDS passes (2 R0 extra VGPRs)
  cf pf[PF ? P0::ROUNDS : 1][PF ? R0 : 1];
  if constexpr (PF) fetch(orow_i, pf);
  while (true) {
    long long onext;  // the output row after this one (>= rows_out: none)
#if WB2_FFT_DYNAMIC
    if (orow_i + 1 < unit_end) {
      onext = orow_i + 1;
    } else {
      onext = granted();
      unit_end = onext + UNIT < rows_out ? onext + UNIT : rows_out;
      if (onext < rows_out) request();
    }
#else
    onext = orow_i + stride;
#endif
    double sum1[NIT], sum2[NIT];
    int cnt[NIT];  // TIME + skipna: valid spectra, bin k (low half) / N2 - k
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      sum1[i] = sum2[i] = 0.0;
      cnt[i] = 0;
    }
    const double c = p.circ[(unsigned)(orow_i % p.n_lat)];
    const double c2 = 2.0 * c;
    double* orow = p.out + orow_i * NB;
    for (long long t = 0; t < nt; ++t) {
      {  // ---- pass 0: HBM -> butterflies -> contiguous runs in the slab
        cf v[P0::ROUNDS][R0];
        if constexpr (PF) {
#pragma unroll
          for (int rd = 0; rd < P0::ROUNDS; ++rd)
#pragma unroll
            for (int r = 0; r < R0; ++r) v[rd][r] = pf[rd][r];
        } else {
          fetch(t * rows_out + orow_i, v);
        }
#if WB2_FFT_DIAG & 8
        sum1[0] += (double)(v[0][0].x + v[0][R0 - 1].y);
#else
        P0::butterflies(v);
        P0::store(z, lane, v);
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
      if constexpr (PF) {
        long long nrow = t * rows_out + orow_i;  // no next row: harmless re-read
        if (t + 1 < nt) nrow += rows_out;
        else if (onext < rows_out) nrow = onext;
        fetch(nrow, pf);
      }
#if !(WB2_FFT_DIAG & 1)
      t1.run(z, lane, s_tw1);
#endif
#if !(WB2_FFT_DIAG & 2)
      if constexpr (R2 > 1) t2.run(z, lane, s_tw2);
#endif
#if WB2_FFT_DIAG & 4
      {
        const cf a = z[lane];
        sum1[0] += (double)a.x * c2;
        if (!TIME && a.x == 1.2345f) orow[lane] = sum1[0];
      }
#else
      if constexpr (!TIME && WB2_FFT_WIDE_STORE) {
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
            const f4 a01 = *reinterpret_cast<const f4*>(z + k0);
            const cf a0 = {a01.x, a01.y}, a1 = {a01.z, a01.w};
            const cf b0 = z[(i == 0 && k0 == 0) ? 0 : N2 - k0];
            const cf b1 = z[N2 - k0 - 1];
            const f4 w01 = *reinterpret_cast<const f4*>(s_twq + k0);
            float p1a, p2a, p1b, p2b;
            recombine_pair(a0, b0, cf{w01.x, w01.y}, half_inv_n, p1a, p2a);
            recombine_pair(a1, b1, cf{w01.z, w01.w}, half_inv_n, p1b, p2b);
            // derived_variables.py:600: every bin but 0 is doubled (Nyquist too)
            const double v1a = (double)p1a * ((i == 0 && k0 == 0) ? c : c2);
            const double v1b = (double)p1b * c2;
            const double v2a = (double)p2a * c2, v2b = (double)p2b * c2;
#if WB2_FFT_DIAG & 16
            if (p1a == 1.2345f) orow[k0] = v1a + v1b + v2a + v2b;
#else
            if (k0 < N2 / 2) {
              __builtin_nontemporal_store(d2{v1a, v1b},
                                          reinterpret_cast<d2*>(orow + k0));
              __builtin_nontemporal_store(
                  d2{v2b, v2a}, reinterpret_cast<d2*>(orow + N2 - k0 - 1));
            } else {  // k0 == N2 / 2: its own mirror; bin k0 + 1 belongs to k0 - 2
              __builtin_nontemporal_store(v1a, orow + k0);
            }
#endif
          }
        }
      } else {
      // ---- recombination + power for the bin pairs (k, N2 - k)
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int k = lane + i * kWave;
        if ((i + 1) * kWave <= NH || k < NH) {
          const cf a = z[k];
          const cf b = z[(i == 0 && k == 0) ? 0 : N2 - k];
          float p1, p2;
          recombine_pair(a, b, s_twq[k], half_inv_n, p1, p2);
          // derived_variables.py:600: every bin but 0 is doubled (Nyquist too)
          const double v1 = (double)p1 * ((i == 0 && k == 0) ? c : c2);
          const double v2 = (double)p2 * c2;
          if constexpr (TIME) {
            const bool k1 = !(p.skipna && is_nan(v1));
            const bool k2 = !(p.skipna && is_nan(v2));
            sum1[i] += k1 ? v1 : 0.0;
            sum2[i] += k2 ? v2 : 0.0;
            cnt[i] += (k1 ? 1 : 0) + (k2 ? 0x10000 : 0);
          } else {
#if WB2_FFT_DIAG & 16
            if (p1 == 1.2345f) orow[k] = v1 + v2;
#else
            __builtin_nontemporal_store(v1, orow + k);
            if (2 * k != N2) __builtin_nontemporal_store(v2, orow + N2 - k);
#endif
          }
        }
      }
      }
#endif
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }  // time
    if constexpr (TIME) {
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int k = lane + i * kWave;
        if (k < NH) {
          __builtin_nontemporal_store(sum1[i] / (double)(cnt[i] & 0xffff),
                                      orow + k);
          if (2 * k != N2)
            __builtin_nontemporal_store(sum2[i] / (double)(cnt[i] >> 16),
                                        orow + N2 - k);
        }
      }
    }
    if (onext >= rows_out) break;
    orow_i = onext;
  }
}

__global__ void fused_twiddle_kernel(cf* twz, cf* twq, int n2) {
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
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) !=
          hipSuccess ||
      per_cu <= 0 || cus <= 0)
    return WB2_FFT_MAX_BLOCKS;
  cached = per_cu * cus;
  return cached;
}

template <int N2>
int launch(const FusedParams& p, hipStream_t s) {
  const long long rows_out = p.n_time > 0 ? p.n_rows / p.n_time : p.n_rows;
  long long blocks = (rows_out + 3) / 4;
  WB2_REQUIRE(p.n_time < 65536, "fused time mean: n_time=%lld exceeds 65535",
              p.n_time);
#if WB2_FFT_DYNAMIC
  // exactly the resident set (a multiple of 8: one share per XCD counter); every
  // wave keeps pulling units until its counter runs dry
  const long long cap = p.n_time > 0
                            ? resident_blocks(fused_spectrum_kernel<N2, true>)
                            : resident_blocks(fused_spectrum_kernel<N2, false>);
  const int unit = p.n_time > 0 ? unit_rows<true>() : unit_rows<false>();
  blocks = (rows_out + 4 * unit - 1) / (4 * unit);
  if (blocks > cap) blocks = cap;
  blocks = (blocks + 7) / 8 * 8;
  WB2_HIP_OK(hipMemsetAsync(p.sched, 0, 8 * kSchedStride * sizeof(unsigned), s));
#else
  if (blocks > WB2_FFT_MAX_BLOCKS) blocks = WB2_FFT_MAX_BLOCKS;  // row-strided waves beyond that
#endif
  if (p.n_time > 0)
    hipLaunchKernelGGL((fused_spectrum_kernel<N2, true>),
                       dim3((unsigned)blocks), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((fused_spectrum_kernel<N2, false>),
                       dim3((unsigned)blocks), dim3(256), 0, s, p);
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace fused

// Entry points used by spectrum.hip ------------------------------------------
bool fused_spectrum_supported(int dtype, int n_lon) {
  if (dtype != WB2_F32 || n_lon % 2) return false;
  switch (n_lon / 2) {
    case 32: case 64: case 120: case 128: case 180: case 256: case 360:
    case 512: case 720:
      return true;
  }
  return false;
}

size_t fused_spectrum_table_bytes(int n_lon) {
  return (size_t)(n_lon / 2 + n_lon / 4 + 1) * sizeof(fused::cf);
}

// Fills the two twiddle tables (fused_spectrum_table_bytes(n_lon) bytes at
// `tables`); the plan does this once at creation and owns the memory.
int fused_spectrum_tables(void* tables, int n_lon, hipStream_t s) {
  using namespace fused;
  const int n2 = n_lon / 2;
  cf* twz = static_cast<cf*>(tables);
  hipLaunchKernelGGL(fused_twiddle_kernel, dim3((unsigned)((n2 + 255) / 256)),
                     dim3(256), 0, s, twz, twz + n2, n2);
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

size_t fused_spectrum_sched_bytes() {
  return 8 * fused::kSchedStride * sizeof(unsigned);
}

int fused_spectrum_run(const float* x, long long n_rows, int n_lon,
                       const double* circ, int n_lat, long long n_time,
                       int skipna, double* out, void* tables, void* sched,
                       hipStream_t s) {
  using namespace fused;
  const int n2 = n_lon / 2;
  cf* twz = static_cast<cf*>(tables);  // filled once by fused_spectrum_tables
  cf* twq = twz + n2;
  FusedParams p{x,      twz,    twq,   circ, out, static_cast<unsigned*>(sched),
                n_rows, n_time, n_lat, skipna};
  switch (n2) {
#define WB2_CASE(N2) case N2: return launch<N2>(p, s);
    WB2_CASE(32) WB2_CASE(64) WB2_CASE(120) WB2_CASE(128) WB2_CASE(180)
    WB2_CASE(256) WB2_CASE(360) WB2_CASE(512) WB2_CASE(720)
#undef WB2_CASE
  }
  return fail("fused spectrum: n_lon=%d is not instantiated", n_lon);
}

}  // namespace wb2
