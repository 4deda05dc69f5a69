// K6 of libwb2hip.so: the rank histogram of an ensemble forecast.
//
// Replaces (reference = /root/reference/weatherbench2/metrics.py):
//   RankHistogram.compute_chunk        :1982-2042
//     combined = concat([truth, forecast], ensemble_dim)          :2001-2014
//     _perturb_by_min_ensemble_diff (random tie breaking)         :1955-1980
//     ranks = argmin(argsort(combined))  -> truth's position      :2024-2027
//     _bin_ranks: rank // ((M + 1) // num_bins)                   :1941-1953
//     np.eye(num_bins)[ranks]  (float64 one-hot, `bins` last)     :2031-2038
//   and its temporal mean (Metric.compute :117-138).
//
// The sort is not needed: truth's position among the members is
//   lo = #{x_m < t},  eq = #{x_m == t},  rank in [lo, lo + eq].
// Without ties (eq == 0) this IS the reference's rank, bit for bit: its
// perturbation is at most a quarter of the smallest gap, so it cannot reorder
// distinct values.  With ties the reference draws NumPy-RNG perturbations,
// which makes the truth's place among the eq + 1 equal values uniform; here
// that uniform draw comes from a counter-based hash of (seed, sample index),
// statistically equivalent and reproducible, but not NumPy's bit stream.
// With a SEED the reference's draw itself is reproduced (the `pcg` fields):
// `np.random.default_rng(seed).uniform(size=da.shape, low=-s/2, high=s/2)` is
// PCG64 consumed one 64-bit output per element in C order of the concatenated
// array, so element k needs the generator advanced by k + 1 steps -- an O(log k)
// jump of the 128-bit LCG -- and nothing else: every sample whose rank the
// perturbation can change (a member equal to the truth, or a NaN / repeated
// infinity, for which the reference falls back to a perturbation of +-1/4)
// recomputes the reference's perturbation size s (half the smallest positive
// gap among the M + 1 values, in the data dtype), perturbs its M + 1 values in
// float64 exactly like `da + perturbation` and counts the members below the
// truth.  All other samples are untouched by the perturbation (it is at most a
// quarter of the smallest gap).  The element order of the concatenated array is
// the host's business (`ref_*`: xarray's concat puts the ensemble dim after the
// truth's dims, see metrics.RankHistogram).
// break_ties=0 puts the truth first among equals (the reference leaves that
// case to an unstable argsort).  NaN members rank above everything; a NaN truth
// ranks above every non-NaN member (:1909-1912).
//
// One lane per grid point: M + 1 coalesced loads, integer compares.  Output:
//   one-hot   (acc_row == null)  each wave writes its 64 x n_bins doubles as
//             one contiguous, coalesced run (bins fetched across lanes).
//   counts    (acc_row != null)  one fp64 atomic add of 1.0 per sample into the
//             time-collapsed histogram; integer-valued sums are exact in any
//             order, so the result is deterministic.
// HBM bound: (M + 1) * sizeof(T) read + n_bins * 8 written per sample.

#include "common.hpp"
#include "trace.hpp"
#include "wb2hip.h"

namespace wb2 {
namespace {

struct RankParams {
  const void* ens;
  const void* truth;
  const long long* ens_slab;
  const long long* truth_slab;
  const long long* acc_row;
  double* out;
  long long member_stride, n_outer, n_point;
  long long n_lead, n_time, n_tail;  // rank_histogram_mean_kernel only
  int mean;                          // 1: counts / n_time
  unsigned long long seed;
  int n_member, n_bins, factor, break_ties;
  // PCG64 emulation of the reference's seeded perturbation (ref_outer_off ==
  // null: off).  Element (outer o, row r, col c, j) of the reference's
  // concatenated array (j = 0: truth, j >= 1: member j - 1) has the C-order index
  //   ref_outer_off[o] + r ref_row_stride + c ref_col_stride + j ref_member_stride.
  const long long* ref_outer_off;
  long long ref_row_stride, ref_col_stride, ref_member_stride;
  int n_col;
  unsigned long long pcg_state_hi, pcg_state_lo, pcg_inc_hi, pcg_inc_lo;
};

// ---- NumPy's PCG64 (PCG XSL RR 128/64): state = state * MULT + inc, output of
// the NEW state; next_double = (next_uint64 >> 11) * 2^-53 --------------------
typedef unsigned __int128 u128;
__device__ __forceinline__ u128 make_u128(unsigned long long hi,
                                          unsigned long long lo) {
  return ((u128)hi << 64) | (u128)lo;
}
__device__ __forceinline__ u128 pcg_mult() {
  return make_u128(0x2360ED051FC65DA4ull, 0x4385DF649FCCF645ull);
}
// state after n steps (Brown's O(log n) LCG jump, as in pcg64_advance)
__device__ u128 pcg_advance(u128 state, u128 inc, unsigned long long n) {
  u128 acc_mult = 1, acc_plus = 0, cur_mult = pcg_mult(), cur_plus = inc;
  while (n > 0) {
    if (n & 1) {
      acc_mult *= cur_mult;
      acc_plus = acc_plus * cur_mult + cur_plus;
    }
    cur_plus = (cur_mult + 1) * cur_plus;
    cur_mult *= cur_mult;
    n >>= 1;
  }
  return acc_mult * state + acc_plus;
}
__device__ __forceinline__ double pcg_double(u128 state) {
  const unsigned long long hi = (unsigned long long)(state >> 64);
  const unsigned long long lo = (unsigned long long)state;
  const unsigned long long x = hi ^ lo;
  const unsigned rot = (unsigned)(hi >> 58);
  const unsigned long long r = (x >> rot) | (x << ((64 - rot) & 63));
  return (double)(r >> 11) * (1.0 / 9007199254740992.0);
}

// The reference's rank of the truth among its M members after
// _perturb_by_min_ensemble_diff (metrics.py:1955-1980) + argsort/argmin
// (:2024-2027), for ONE sample whose first element has index `elem0` in the
// reference's stream.
template <typename T>
__device__ int perturbed_rank(const T* xb, long long member_stride, int M, T t,
                              int nn, const RankParams& p, long long elem0) {
  const T inf = __builtin_huge_val();
  // perturbation size: half the smallest positive difference of the sorted
  // values (= of any two values), NaN (-> 1) if the sorted differences contain
  // a NaN: a NaN value, or inf - inf of two equal infinities
  bool nan_rule = is_nan(t) || nn != M;
  int n_pinf = t == inf ? 1 : 0, n_ninf = t == -inf ? 1 : 0;
  T mn = inf;
  for (int a = 0; a <= M; ++a) {
    const T va = a == 0 ? t : xb[(a - 1) * member_stride];
    if (a > 0) {
      n_pinf += va == inf ? 1 : 0;
      n_ninf += va == -inf ? 1 : 0;
    }
    for (int b = a + 1; b <= M; ++b) {
      const T vb = xb[(b - 1) * member_stride];
      const T d = abs_of(va - vb);
      if (d > (T)0 && d < mn) mn = d;
    }
  }
  nan_rule = nan_rule || n_pinf >= 2 || n_ninf >= 2;
  const T size = (!nan_rule && mn < inf) ? mn / (T)2 : (T)1;
  const T low = -size / (T)2, high = size / (T)2;
  const double dlow = (double)low, range = (double)high - (double)low;
  const u128 s0 = make_u128(p.pcg_state_hi, p.pcg_state_lo);
  const u128 inc = make_u128(p.pcg_inc_hi, p.pcg_inc_lo);
  u128 s = pcg_advance(s0, inc, (unsigned long long)elem0 + 1);
  const double v0 = (double)t + (dlow + range * pcg_double(s));
  int rank = 0;
  for (int j = 1; j <= M; ++j) {
    if (p.ref_member_stride == 1)
      s = s * pcg_mult() + inc;
    else
      s = pcg_advance(s0, inc,
                      (unsigned long long)(elem0 + j * p.ref_member_stride) + 1);
    const double vj =
        (double)xb[(j - 1) * member_stride] + (dlow + range * pcg_double(s));
    rank += vj < v0 ? 1 : 0;
  }
  // NaN sorts last (np.argsort): a NaN truth ranks above every non-NaN member
  return is_nan(t) ? nn : rank;
}

// splitmix64 finaliser: a counter-based stream, one draw per sample
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Bin of sample (outer slab o, grid point pt): the truth's rank among the
// members, ties broken as configured, divided by the bin factor.
template <typename T>
__device__ __forceinline__ int sample_bin(const RankParams& p, long long o,
                                          long long pt) {
  const long long es = p.ens_slab ? p.ens_slab[o] : o;
  const long long ts = p.truth_slab ? p.truth_slab[o] : o;
  const int M = p.n_member;
  const T* xb = static_cast<const T*>(p.ens) + es * p.n_point + pt;
  const T t = __builtin_nontemporal_load(
      static_cast<const T*>(p.truth) + ts * p.n_point + pt);
  int lo = 0, eq = 0, nn = 0, ninf = 0;
  auto take = [&](T x) {
    lo += x < t ? 1 : 0;
    eq += x == t ? 1 : 0;
    nn += is_nan(x) ? 0 : 1;
    ninf += abs_of(x) == (T)__builtin_huge_val() ? 1 : 0;
  };
  int m = 0;
  // sixteen member loads in flight per lane (4 KB per wave), then four
  for (; m + 16 <= M; m += 16) {
    T x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u)
      x[u] = __builtin_nontemporal_load(xb + (m + u) * p.member_stride);
#pragma unroll
    for (int u = 0; u < 16; ++u) take(x[u]);
  }
  for (; m + 4 <= M; m += 4) {
    T x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      x[u] = __builtin_nontemporal_load(xb + (m + u) * p.member_stride);
#pragma unroll
    for (int u = 0; u < 4; ++u) take(x[u]);
  }
  for (; m < M; ++m)
    take(__builtin_nontemporal_load(xb + m * p.member_stride));
  int rank = is_nan(t) ? nn : lo;
  if (p.break_ties && p.ref_outer_off) {
    // seeded: the reference's own perturbation wherever it can matter
    const bool special = eq > 0 || nn != M || is_nan(t) || ninf > 0 ||
                         abs_of(t) == (T)__builtin_huge_val();
    if (special) {
      const long long row = pt / p.n_col, col = pt - row * p.n_col;
      const long long elem0 = p.ref_outer_off[o] + row * p.ref_row_stride +
                              col * p.ref_col_stride;
      rank = perturbed_rank<T>(xb, p.member_stride, M, t, nn, p, elem0);
    }
  } else if (eq > 0 && p.break_ties) {
    const unsigned long long h =
        mix64(p.seed ^ mix64((unsigned long long)(o * p.n_point + pt)));
    // uniform integer in [0, eq]: high bits of a 32x32 multiply
    rank += (int)(((h >> 32) * (unsigned long long)(eq + 1)) >> 32);
  }
  return rank / p.factor;
}

template <typename T>
__global__ void __launch_bounds__(256) rank_histogram_kernel(const RankParams p) {
  const int lane = threadIdx.x & (kWave - 1);
  const long long wave_pt0 =
      ((long long)blockIdx.x * blockDim.x + threadIdx.x) - lane;
  const long long pt = wave_pt0 + lane;
  const long long o = blockIdx.y + (long long)blockIdx.z * gridDim.y;
  if (o >= p.n_outer || wave_pt0 >= p.n_point) return;  // wave-uniform
  const bool active = pt < p.n_point;
  const int bin = active ? sample_bin<T>(p, o, pt) : 0;

  if (p.acc_row) {
    if (active)
      atomicAdd(p.out + (p.acc_row[o] * p.n_point + pt) * p.n_bins + bin, 1.0);
    return;
  }
  // one-hot: this wave owns out[o][wave_pt0 .. +64)[0 .. n_bins), contiguous
  const long long npt = p.n_point - wave_pt0 < kWave ? p.n_point - wave_pt0
                                                     : kWave;
  double* dst = p.out + (o * p.n_point + wave_pt0) * p.n_bins;
  const int total = (int)npt * p.n_bins;
  const int dq = kWave / p.n_bins, dr = kWave % p.n_bins;
  int src = lane / p.n_bins, b = lane % p.n_bins;
  for (int i = lane; i - lane < total; i += kWave) {  // wave-uniform trip count
    const int their = __shfl(bin, src & (kWave - 1), kWave);
    if (i < total) __builtin_nontemporal_store(their == b ? 1.0 : 0.0, dst + i);
    src += dq;
    b += dr;
    if (b >= p.n_bins) {
      b -= p.n_bins;
      ++src;
    }
  }
}


// The histogram summed over one axis of the outer index -- o = (l * n_time + t)
// * n_tail + j, result row l * n_tail + j -- without atomics and without the
// per-sample one-hots: ONE wave owns 64 points of one result row, walks the
// n_time samples of each point with its counts in LDS ([point][bin], a lane
// only ever touches its own row), and writes the row's 64 x n_bins float64
// values once, as one contiguous run (counts, or counts / n_time: the mean of
// the one-hots, Metric.compute :117-138).  HBM: the members once + the result
// once -- the atomic form re-fetches a 128-byte line of the result per sample.
template <typename T>
__global__ void __launch_bounds__(kWave)
    rank_histogram_mean_kernel(const RankParams p) {
  extern __shared__ unsigned counts[];  // [kWave][n_bins]
  const int lane = threadIdx.x;
  const long long wave_pt0 = (long long)blockIdx.x * kWave;
  const long long row = blockIdx.y + (long long)blockIdx.z * gridDim.y;
  if (row >= p.n_lead * p.n_tail) return;  // wave-uniform
  const long long l = row / p.n_tail, j = row - l * p.n_tail;
  const long long pt = wave_pt0 + lane;
  const bool active = pt < p.n_point;
  const int nb = p.n_bins;
  for (int i = lane; i < kWave * nb; i += kWave) counts[i] = 0;
  __syncthreads();
  if (active) {
    for (long long t = 0; t < p.n_time; ++t) {
      const long long o = (l * p.n_time + t) * p.n_tail + j;
      counts[lane * nb + sample_bin<T>(p, o, pt)] += 1;
    }
  }
  __syncthreads();
  const long long npt = p.n_point - wave_pt0 < kWave ? p.n_point - wave_pt0
                                                     : kWave;
  double* dst = p.out + (row * p.n_point + wave_pt0) * nb;
  const int total = (int)npt * nb;
  const double n = (double)p.n_time;
  for (int i = lane; i < total; i += kWave) {
    const double c = (double)counts[i];
    __builtin_nontemporal_store(p.mean ? c / n : c, dst + i);
  }
}

}  // namespace
}  // namespace wb2

static int rank_histogram_impl(int dtype, const void* ens,
                               const int64_t* ens_slab, const void* truth,
                               const int64_t* truth_slab, int32_t n_member,
                               int64_t member_stride, int64_t n_outer,
                               int64_t n_point, int32_t n_bins, int break_ties,
                               uint64_t seed, const int64_t* acc_row,
                               double* out, const int64_t* ref_outer_off,
                               const int64_t* ref_strides, int32_t n_col,
                               const uint64_t* pcg_state_inc, void* stream,
                               int64_t n_lead = 0, int64_t n_time = 0,
                               int64_t n_tail = 0, int mean = 0);

extern "C" int wb2_rank_histogram_seeded(
    int dtype, const void* ens, const int64_t* ens_slab, const void* truth,
    const int64_t* truth_slab, int32_t n_member, int64_t member_stride,
    int64_t n_outer, int64_t n_point, int32_t n_col, int32_t n_bins,
    const uint64_t* pcg_state_inc, const int64_t* ref_outer_off,
    const int64_t* ref_strides, const int64_t* acc_row, double* out,
    void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_EMPTY_OK(n_outer);
  WB2_EMPTY_OK(n_point);
  WB2_REQUIRE(pcg_state_inc && ref_outer_off && ref_strides,
              "null pointer argument");
  WB2_REQUIRE(n_col >= 1 && n_point % n_col == 0, "n_point=%lld n_col=%d",
              (long long)n_point, n_col);
  return rank_histogram_impl(dtype, ens, ens_slab, truth, truth_slab, n_member,
                             member_stride, n_outer, n_point, n_bins, 1, 0,
                             acc_row, out, ref_outer_off, ref_strides, n_col,
                             pcg_state_inc, stream);
}

extern "C" int wb2_rank_histogram_mean(
    int dtype, const void* ens, const int64_t* ens_slab, const void* truth,
    const int64_t* truth_slab, int32_t n_member, int64_t member_stride,
    int64_t n_lead, int64_t n_time, int64_t n_tail, int64_t n_point,
    int32_t n_col, int32_t n_bins, int break_ties, uint64_t seed,
    const uint64_t* pcg_state_inc, const int64_t* ref_outer_off,
    const int64_t* ref_strides, int mean, double* out, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_EMPTY_OK(n_lead);
  WB2_EMPTY_OK(n_tail);
  WB2_EMPTY_OK(n_point);
  WB2_REQUIRE(n_lead >= 0 && n_time >= 1 && n_tail >= 0,
              "bad sizes: n_lead=%lld n_time=%lld n_tail=%lld",
              (long long)n_lead, (long long)n_time, (long long)n_tail);
  const bool seeded = pcg_state_inc != nullptr;
  WB2_REQUIRE(!seeded || (ref_outer_off && ref_strides && n_col >= 1 &&
                          n_point % n_col == 0),
              "seeded ties need ref_outer_off, ref_strides and n_col");
  if (n_lead == 0 || n_tail == 0) return 0;
  return rank_histogram_impl(
      dtype, ens, ens_slab, truth, truth_slab, n_member, member_stride,
      n_lead * n_time * n_tail, n_point, n_bins, seeded ? 1 : break_ties,
      seeded ? 0 : seed, nullptr, out, seeded ? ref_outer_off : nullptr,
      seeded ? ref_strides : nullptr, seeded ? n_col : 1,
      seeded ? pcg_state_inc : nullptr, stream, n_lead, n_time, n_tail,
      mean != 0);
}

extern "C" int wb2_rank_histogram(int dtype, const void* ens,
                                  const int64_t* ens_slab, const void* truth,
                                  const int64_t* truth_slab, int32_t n_member,
                                  int64_t member_stride, int64_t n_outer,
                                  int64_t n_point, int32_t n_bins,
                                  int break_ties, uint64_t seed,
                                  const int64_t* acc_row, double* out,
                                  void* stream) {
  WB2_TRACE();
  return rank_histogram_impl(dtype, ens, ens_slab, truth, truth_slab, n_member,
                             member_stride, n_outer, n_point, n_bins,
                             break_ties, seed, acc_row, out, nullptr, nullptr, 1,
                             nullptr, stream);
}

static int rank_histogram_impl(int dtype, const void* ens,
                               const int64_t* ens_slab, const void* truth,
                               const int64_t* truth_slab, int32_t n_member,
                               int64_t member_stride, int64_t n_outer,
                               int64_t n_point, int32_t n_bins, int break_ties,
                               uint64_t seed, const int64_t* acc_row,
                               double* out, const int64_t* ref_outer_off,
                               const int64_t* ref_strides, int32_t n_col,
                               const uint64_t* pcg_state_inc, void* stream,
                               int64_t n_lead, int64_t n_time, int64_t n_tail,
                               int mean) {
  using namespace wb2;
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64,
              "dtype must be WB2_F32 or WB2_F64, got %d", dtype);
  WB2_EMPTY_OK(n_outer);
  WB2_EMPTY_OK(n_point);
  WB2_REQUIRE(ens && truth && out, "ens, truth and out must not be null");
  WB2_REQUIRE(n_member >= 1 && n_outer >= 0 && n_point >= 0,
              "bad sizes: n_member=%d n_outer=%lld n_point=%lld", n_member,
              (long long)n_outer, (long long)n_point);
  // metrics.py:1933-1938
  WB2_REQUIRE(n_bins >= 1 && (n_member + 1) % n_bins == 0,
              "Cannot bin data with ensemble_size=%d into %d bins", n_member,
              n_bins);
  WB2_REQUIRE((long long)n_point * n_bins < (1ll << 31) * kWave,
              "n_point * n_bins too large");
  if (n_outer == 0 || n_point == 0) return 0;
  RankParams p{};
  p.ens = ens;
  p.truth = truth;
  p.ens_slab = reinterpret_cast<const long long*>(ens_slab);
  p.truth_slab = reinterpret_cast<const long long*>(truth_slab);
  p.acc_row = reinterpret_cast<const long long*>(acc_row);
  p.out = out;
  p.member_stride = member_stride;
  p.n_outer = n_outer;
  p.n_point = n_point;
  p.seed = seed;
  p.n_member = n_member;
  p.n_bins = n_bins;
  p.factor = (n_member + 1) / n_bins;
  p.break_ties = break_ties;
  p.ref_outer_off = reinterpret_cast<const long long*>(ref_outer_off);
  p.ref_row_stride = ref_strides ? ref_strides[0] : 0;
  p.ref_col_stride = ref_strides ? ref_strides[1] : 0;
  p.ref_member_stride = ref_strides ? ref_strides[2] : 0;
  p.n_col = n_col;
  p.pcg_state_hi = pcg_state_inc ? pcg_state_inc[0] : 0;
  p.pcg_state_lo = pcg_state_inc ? pcg_state_inc[1] : 0;
  p.pcg_inc_hi = pcg_state_inc ? pcg_state_inc[2] : 0;
  p.pcg_inc_lo = pcg_state_inc ? pcg_state_inc[3] : 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_time > 0) {  // summed over the middle axis of (n_lead, n_time, n_tail)
    p.n_lead = n_lead;
    p.n_time = n_time;
    p.n_tail = n_tail;
    p.mean = mean;
    const long long n_row = n_lead * n_tail;
    const long long ry = n_row < 32768 ? n_row : 32768;
    const long long rz = (n_row + ry - 1) / ry;
    WB2_REQUIRE(rz <= 65535, "%lld result rows: too many", n_row);
    const size_t lds = (size_t)kWave * n_bins * sizeof(unsigned);
    WB2_REQUIRE(lds <= 64 * 1024, "n_bins=%d: more than 256 bins", n_bins);
    const dim3 grid((unsigned)((n_point + kWave - 1) / kWave), (unsigned)ry,
                    (unsigned)rz);
    if (dtype == WB2_F32)
      hipLaunchKernelGGL(rank_histogram_mean_kernel<float>, grid, dim3(kWave),
                         lds, s, p);
    else
      hipLaunchKernelGGL(rank_histogram_mean_kernel<double>, grid, dim3(kWave),
                         lds, s, p);
    WB2_HIP_OK(hipGetLastError());
    return 0;
  }
  const long long gy = n_outer < 32768 ? n_outer : 32768;
  const long long gz = (n_outer + gy - 1) / gy;
  WB2_REQUIRE(gz <= 65535, "n_outer=%lld too large", (long long)n_outer);
  const dim3 grid((unsigned)((n_point + 255) / 256), (unsigned)gy, (unsigned)gz);
  if (dtype == WB2_F32)
    hipLaunchKernelGGL(rank_histogram_kernel<float>, grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(rank_histogram_kernel<double>, grid, dim3(256), 0, s, p);
  WB2_HIP_OK(hipGetLastError());
  return 0;
}
