// K3's device code (templates): included by ensemble.hip (the runtime-M and
// exact-50 instantiations + the host entry points) and by the per-size
// objects of ensemble_exact.hip (one per compile-time member count).
#pragma once

#include <cstdlib>

#include "common.hpp"
#include "trace.hpp"
#include "reduce_common.hpp"
#include "sort_networks.inc"
#include "sort3_network_50.inc"
#include "sort3_networks.inc"
#include "wb2hip.h"

#include <limits>
#include <type_traits>

namespace wb2 {

struct EnsParams {
  const void* ens;
  const void* truth;
  const long long* ens_slab;
  const long long* truth_slab;
  // gathered ensembles (wb2_ens_partials_gather): device ADDRESS of the slab of
  // member m at outer index o in member_ptr[o * n_member + m]; NULL = members
  // at a constant stride from `ens`.  Runtime-M register-sort kernels only.
  const long long* member_ptr;
  const double* w_row;
  const double* w_col;
  const double* wfield;
  const int* chunk_row0;
  const int* chunk_nrow;
  const int* seg_col0;
  const int* seg_eoff;
  double* partials;
  double* maps;  // optional [6][n_outer][n_row*n_col]: the pointwise values
  long long member_stride;
  long long n_outer;
  int n_member, n_row, n_col, n_chunk, n_ctile, n_seg, n_ts;
  // bytes per unit of ens_slab / truth_slab entries (K3 kernels of this
  // header: one slab, or 1 for tables of addresses with a NULL base)
  long long ens_scale, truth_scale;
};

// Exact float32 instantiations live in translation units of their own
// (ensemble_exact.hip compiled once per member count of WB2_SORT3_SIZES,
// sort3_networks.inc: the list tools/gen_sort3_network.py generates programs
// for): X(member count, padded register count).
#define WB2_ENS_EXACT_SIZES(X) WB2_SORT3_SIZES(X)
#define WB2_ENS_DECLARE(M, NPAD)                                       \
  int launch_ens_exact_f32_##M(const EnsParams& p, bool skipna, bool wf, \
                               hipStream_t stream);
WB2_ENS_EXACT_SIZES(WB2_ENS_DECLARE)
#undef WB2_ENS_DECLARE
// The same programs as HOSTS of smaller ensembles: a runtime member count
// m <= M runs in the M-member program with the slots m..M-1 held at +inf (they
// sort behind every live member and carry no rank weight); statistics over the
// first m members.  float32; strided or gathered members; with NaN skipping the
// dead slots are NaN members of the general exact code.
#define WB2_ENS_DECLARE_HOSTED(M, NPAD)                                 \
  int launch_ens_hosted_f32_##M(const EnsParams& p, bool skipna, bool wf, \
                                hipStream_t stream);
WB2_ENS_EXACT_SIZES(WB2_ENS_DECLARE_HOSTED)
#undef WB2_ENS_DECLARE_HOSTED

namespace {

// v_min_f32 / v_max_f32 (NaNs never reach the network: they are replaced first
// or the result is overridden).
// The instructions themselves: fminf/fmaxf make hipcc canonicalise every loaded
// value first (a v_max_f32 x, x, x per member) because of signalling NaNs,
// which this kernel never looks at.
__device__ __forceinline__ float vmin(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// max(|d|, 0): |d|, or 0 when d is NaN (v_max returns the non-NaN operand).
__device__ __forceinline__ float abs_or_zero(float d) {
  float r;
  asm("v_max_f32 %0, |%1|, 0" : "=v"(r) : "v"(d));
  return r;
}
__device__ __forceinline__ double abs_or_zero(double d) {
  double r;
  asm("v_max_f64 %0, |%1|, 0" : "=v"(r) : "v"(d));
  return r;
}

// Member loads: the M addresses of a grid point differ by a wave-uniform
// stride, so a raw buffer load (SGPR base per member, one constant per-lane byte
// offset) needs no vector address arithmetic at all, where a global load costs
// a 64-bit VALU add per member.
// ONCE: the caller reads every member exactly once (the register-sort kernel):
// non-temporal loads, +3.5 % on BASELINE configs[2] (0.488 -> 0.471 ms,
// profiles/r03_k3_ab8_summary.txt); the multi-pass streaming form re-reads its
// members from the caches and keeps them cacheable.
// buffer-load cache policy of the read-once loads (gfx940+ aux bits: 1 sc0,
// 2 nt, 16 sc1)
constexpr int kNtPolicy = 2;

// `num_records` (bytes addressable through the descriptor): 0 turns the load
// into a no-op that returns 0 -- the range check happens before any memory
// access -- which is how the hosted kernels skip their dead member slots with
// ONE scalar select that is not part of the address chain.
template <typename T, bool ONCE = false>
__device__ __forceinline__ T member_load(const T* uniform_base, int lane_bytes,
                                         int num_records = 0x7fffffff) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(uniform_base), 0, num_records, 0x00020000);
  // cache policy (gfx940+): bit 1 = nt -- the members are read once
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(
        T, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_bytes, 0,
                                                ONCE ? kNtPolicy : 0));
  } else {
    return __builtin_bit_cast(
        T, __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane_bytes, 0,
                                                ONCE ? kNtPolicy : 0));
  }
}

// Sorts x[0..LIMIT) (slots >= LIMIT hold +inf padding and never move: every
// comparator that touches one of them is a no-op and is pruned at compile time,
// e.g. 403 instead of 543 comparators for 50 members in the 64-network).
template <int NPAD, int LIMIT, typename T>
__device__ __forceinline__ void sort_network(T (&x)[NPAD]) {
#define WB2_CE(i, j)                    \
  if constexpr ((j) < LIMIT) {          \
    const T lo_ = vmin(x[i], x[j]);     \
    const T hi_ = vmax(x[i], x[j]);     \
    x[i] = lo_;                         \
    x[j] = hi_;                         \
  }
  if constexpr (NPAD == 2) { WB2_SORT_NETWORK_2 }
  if constexpr (NPAD == 4) { WB2_SORT_NETWORK_4 }
  if constexpr (NPAD == 8) { WB2_SORT_NETWORK_8 }
  if constexpr (NPAD == 16) { WB2_SORT_NETWORK_16 }
  if constexpr (NPAD == 32) { WB2_SORT_NETWORK_32 }
  if constexpr (NPAD == 64) { WB2_SORT_NETWORK_64 }
  if constexpr (NPAD == 128) { WB2_SORT_NETWORK_128 }
#undef WB2_CE
}

#ifndef WB2_ENS_PAIRED_SPREAD
#define WB2_ENS_PAIRED_SPREAD 1  // 50 float32 members: rank-weighted sum over (hi, lo) pairs
#endif
__device__ __forceinline__ void sort3_asm(float& a, float& b, float& c) {
  float lo, mid, hi;
  asm("v_min3_f32 %0, %3, %4, %5\n\t"
      "v_med3_f32 %1, %3, %4, %5\n\t"
      "v_max3_f32 %2, %3, %4, %5"
      : "=&v"(lo), "=&v"(mid), "=&v"(hi)
      : "v"(a), "v"(b), "v"(c));
  a = lo;
  b = mid;
  c = hi;
}
// Sorting program for exactly 50 float32 values from 2-sorters and 3-sorters
// (v_min3 / v_med3 / v_max3: three instructions order three values, where three
// compare-exchanges cost six): two 27-sorters by 3-way odd-even merge sort + one
// 2-way odd-even merge, pruned for the +inf padding -- 677 instructions instead
// of the 806 of the pruned Batcher network.  Generated and verified (0-1
// principle, exhaustively for the 27-sorter and every merge) by
// tools/gen_sort3_network.py.  The values are never moved: rank r ends up in
// register kSort3Order50[r].
constexpr int kSort3Order50[50] = {WB2_SORT3_ORDER_50};
// Which register a wire of the program lives in is free (the inputs are a set):
// choose it so that the ranks the paired rank-weighted sum combines sit in
// (even, odd) register pairs -- ranks (1,2), (3,4), ..., (47,48) in registers
// (0,1), (2,3), ..., (46,47), ranks 0 and 49 in (48,49) -- and that sum runs on
// v_pk_add_f32 / v_pk_fma_f32.  of_rank[r]: register of rank r after the sort;
// reg[w]: register of wire w.
struct Sort3Layout50 {
  int reg[50];
  int of_rank[50];
  constexpr Sort3Layout50() : reg{}, of_rank{} {
    for (int r = 0; r < 50; ++r) {
      const int q = kSort3Order50[r];
      of_rank[r] = q;
      reg[kSort3Order50[r]] = q;
    }
  }
};
constexpr Sort3Layout50 kSort3Layout50{};
__device__ __forceinline__ void sort3_network_50(float (&x)[64]) {
#define WB2_R(i) x[kSort3Layout50.reg[i]]
#define WB2_S2(i, j)                            \
  {                                             \
    const float lo_ = vmin(WB2_R(i), WB2_R(j)); \
    const float hi_ = vmax(WB2_R(i), WB2_R(j)); \
    WB2_R(i) = lo_;                             \
    WB2_R(j) = hi_;                             \
  }
#define WB2_S3(i, j, k) sort3_asm(WB2_R(i), WB2_R(j), WB2_R(k));
  WB2_SORT3_NETWORK_50
#undef WB2_S2
#undef WB2_S3
#undef WB2_R
}

// The same kind of program for the other member counts K3 instantiates exactly
// (sort3_networks.inc, tools/gen_sort3_network.py --emit-exact): float32 only
// (there is no v_min3_f64); rank r ends up in register Sort3<M>::order[r].
template <int M>
struct Sort3 {
  static constexpr bool has = false;
};
#define WB2_SORT3_DEFINE(M)                                                  \
  template <>                                                                \
  struct Sort3<M> {                                                          \
    static constexpr bool has = true;                                        \
    static constexpr int order[M] = {WB2_SORT3_ORDER_##M};                   \
    template <int NPAD>                                                      \
    static __device__ __forceinline__ void run(float (&x)[NPAD]) {           \
      WB2_SORT3_NETWORK_##M                                                  \
    }                                                                        \
  };
#define WB2_S2(i, j)                      \
  {                                       \
    const float lo_ = vmin(x[i], x[j]);   \
    const float hi_ = vmax(x[i], x[j]);   \
    x[i] = lo_;                           \
    x[j] = hi_;                           \
  }
#define WB2_S3(i, j, k) sort3_asm(x[i], x[j], x[k]);
#define WB2_SORT3_DEFINE2(M, NPAD) WB2_SORT3_DEFINE(M)
WB2_SORT3_SIZES(WB2_SORT3_DEFINE2)
#undef WB2_SORT3_DEFINE2
#undef WB2_S2
#undef WB2_S3
#undef WB2_SORT3_DEFINE

// v * flag for flag in {0, 1} with 0 * inf = 0 * NaN = 0 (v_mul_legacy_f32's
// DX9 rule) -- a select without a lane mask; float64: a plain select.
__device__ __forceinline__ float times_flag(float v, float flag) {
  float r;
  asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(flag));
  return r;
}
__device__ __forceinline__ double times_flag(double v, double flag) {
  return flag != 0.0 ? v : 0.0;
}
// min(max(a, 0), 1) as one v_med3
__device__ __forceinline__ float clamp01(float a) {
  float r;
  asm("v_med3_f32 %0, %1, 0, 1.0" : "=v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ double clamp01(double a) {
  return a < 0.0 ? 0.0 : (a > 1.0 ? 1.0 : a);
}

template <typename T>
__device__ __forceinline__ T sqrt_of(T x);
template <>
__device__ __forceinline__ float sqrt_of(float x) { return __builtin_sqrtf(x); }
template <>
__device__ __forceinline__ double sqrt_of(double x) { return __builtin_sqrt(x); }

// x / C for a compile-time integer C in float32: Markstein's sequence, one
// multiply and two FMAs instead of the ten instructions of the IEEE division,
// with the IDENTICAL (correctly rounded) result: tools/check_div_const.py checks
// all 2^23 mantissas in integer arithmetic (C = 49, 50 and others).  The proof
// needs e = x - C q0 exact, i.e. not underflowed; whenever e is not a normal
// number or zero (tiny x, inf, NaN) the IEEE division runs instead -- behind a
// real branch (the empty asm keeps hipcc from speculating it).
template <int C, typename T>
__device__ __forceinline__ T div_const(T x) {
  if constexpr (sizeof(T) == 4) {
    constexpr float r = 1.0f / (float)C;
    const float q0 = x * r;
    const float e = __builtin_fmaf(-(float)C, q0, x);
    // classes: -normal (8), -0 (32), +0 (64), +normal (256)
    if (__builtin_expect(__builtin_amdgcn_classf(e, 8 | 32 | 64 | 256), 1))
      return __builtin_fmaf(e, r, q0);
    asm volatile("" ::: "memory");
    return x / (float)C;
  } else {
    return x / (T)C;
  }
}

// One grid point -> the K slot values (see header comment).  MS > 0: the member
// count is the compile-time constant MS (exact network, no selects); MS == 0:
// runtime M <= NPAD, slots >= M are neutralised with selects (straight-line code
// on purpose: per-member branches wreck hipcc's register allocation).
// REFCHAIN (MS > 0, !SKIPNA): the NaN-free fast path of a SKIPNA kernel -- the
// rank-weighted sum as the reference's fp64 chain and the final divisions as
// the skipna form writes them, so that a point's value does not depend on
// whether its wave took the fast path.
// RTM (MS > 0, SKIPNA): the MS slots hold a RUNTIME member count Mrt <= MS --
// the slots beyond it arrive as NaN and drop out like NaN members do (they are
// not counted, sort last, carry no rank weight), only the ensemble size in the
// rank weights and the debiasing term is the run-time one (hosted counts).
template <typename T, int NPAD, int MS, bool SKIPNA, bool REFCHAIN = false,
          bool RTM = false>
__device__ __forceinline__ void ens_point(T (&x)[NPAD], const T t, const int Mrt,
                                          double (&out)[SKIPNA ? 10 : 6]) {
  static_assert(!RTM || (MS > 0 && SKIPNA && !REFCHAIN),
                "a run-time count inside an exact program: NaN skipping only");
  const T nan = std::numeric_limits<T>::quiet_NaN();
  const T inf = std::numeric_limits<T>::infinity();
  constexpr int NM = MS > 0 ? MS : NPAD;  // slots visited
  const int M = (MS > 0 && !RTM) ? MS : Mrt;
  auto live = [&](int m) { return MS > 0 ? true : m < M; };
  T sum = 0, sk = 0;
  int n = 0;          // valid members (SKIPNA)
  bool bad = false;   // any NaN member (!SKIPNA)
  // Exact member count WITH NaN skipping: the per-member NaN masks are used
  // where they are made (sum, |t - x|, count) and nowhere else -- kept for the
  // variance and the sort they cost an SGPR pair per member and phase (spilled
  // and reloaded through VGPR lanes: 264 v_readlane / v_writelane per row).  The
  // later phases get mask-free forms with the same values: max(d * d, 0) is d * d
  // for a valid member and 0 for a NaN one (v_max returns the non-NaN
  // operand), min(x, +inf) turns a NaN member into the +inf the sort wants.
  // The one case where d * d is NaN for a VALID member is an infinite member
  // (inf - inf): then, as before, the sum of squares is NaN.
  constexpr bool SKIPNA_LEAN = SKIPNA && MS > 0;
  bool inf_member = false;
  T sq = 0;
  T mean;
  if constexpr (MS > 0 && !SKIPNA) {
    // exact member count without NaN skipping: one unordered compare per member
    // PAIR finds the NaNs (an odd count's last member alone)
#pragma unroll
    for (int m = 0; m + 1 < NM; m += 2) {
      const T d0 = t - x[m], d1 = t - x[m + 1];
      sum += x[m];
      sk += abs_of(d0);
      sum += x[m + 1];
      sk += abs_of(d1);
      // (REFCHAIN: the caller has found the whole wave free of NaNs)
      if constexpr (!REFCHAIN)
        bad = bad || __builtin_isunordered(x[m], x[m + 1]);
    }
    if constexpr (NM % 2 == 1) {
      sum += x[NM - 1];
      sk += abs_of(t - x[NM - 1]);
      if constexpr (!REFCHAIN) bad = bad || is_nan(x[NM - 1]);
    }
    // the flag is first needed after the sort: without this pin hipcc sinks the
    // compares down there and keeps the unsorted ensemble alive next to the
    // sorted one (131 instead of 87 VGPRs)
    int pinned = bad ? 1 : 0;
    asm volatile("" : "+v"(pinned));
    bad = pinned != 0;
  } else {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const bool isn = is_nan(x[m]);
      const bool use = live(m) && (SKIPNA ? !isn : true);
      sum += use ? x[m] : (T)0;
      sk += use ? abs_of(t - x[m]) : (T)0;
      n += use ? 1 : 0;
      bad = bad || (live(m) && isn);
      if constexpr (SKIPNA_LEAN) inf_member = inf_member || abs_of(x[m]) == inf;
    }
    if constexpr (SKIPNA_LEAN) {
      // as for `bad` above: everything the sort does not need is finished (and
      // pinned) before it, or hipcc sinks it below the sort and keeps the
      // unsorted ensemble alive beside the sorted one
      int pinned = inf_member ? 1 : 0;
      asm volatile("" : "+v"(sum), "+v"(sk), "+v"(n), "+v"(pinned));
      inf_member = pinned != 0;
    }
  }
  const int cnt = SKIPNA ? n : M;
  // metrics.py:562-565 / :824 -- numpy mean / var(ddof=1) / mean(abs) over the
  // leading (member) axis: sequential, in the input dtype (nan* variants reduce
  // over the valid members only).
  {
    if constexpr (MS > 0 && !SKIPNA) mean = div_const<MS>(sum);
    else mean = sum / (T)cnt;
    {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const T d = x[m] - mean;
        if constexpr (SKIPNA_LEAN) {
          sq += abs_or_zero(d * d);
        } else {
          const bool use = live(m) && (SKIPNA ? !is_nan(x[m]) : true);
          sq += use ? d * d : (T)0;
        }
      }
      if (SKIPNA_LEAN && inf_member) sq = nan;
      if constexpr (SKIPNA_LEAN) asm volatile("" : "+v"(sq));
    }
    // everything that reads the members in MEMBER order is finished before the
    // sort starts (or the unsorted ensemble stays alive beside the sorted one)
    if constexpr (REFCHAIN) asm volatile("" : "+v"(sq), "+v"(sum), "+v"(sk));
  }
  T var;
  if constexpr (MS > 1 && !SKIPNA) var = div_const<MS - 1>(sq);
  else var = sq / (T)(cnt - 1);
  if (SKIPNA && cnt <= 1) var = nan;
  const T sd = sqrt_of(var);
  const T err = t - mean;
  const T mse = err * err;
  T deb, skill;
  if constexpr (MS > 0 && !RTM) deb = mse - div_const<MS>(var);
  else deb = mse - var / (T)M;
  if constexpr (MS > 0 && !SKIPNA) skill = div_const<MS>(sk);
  else skill = sk / (T)cnt;
  if (SKIPNA && is_nan(t)) skill = nan;
  // metrics.py:804-813: 2 * mean_m((2 r_m - M - 1) x_m) / (M - 1) in fp64; ranks
  // come from the FULL ensemble with NaN last (np.argsort), so sort with
  // NaN -> +inf and weight the i-th smallest by 2(i+1) - M - 1.
  double spread = 0.0;
  if (M >= 2) {
#pragma unroll
    for (int m = 0; m < NPAD; ++m) {
      if (m >= NM) {
        x[m] = inf;
      } else if constexpr (SKIPNA_LEAN) {
        x[m] = vmin(x[m], inf);  // NaN -> +inf, everything else unchanged
      } else {
        x[m] = (!live(m) || (SKIPNA && is_nan(x[m]))) ? inf : x[m];
      }
    }
    double s = 0.0;
    // NaN skipping: rank m counts while m < n -- as the flag min(max(n - m, 0), 1) and a legacy multiply
    // (0 * inf = 0) instead of a compare + select per member (51 lane masks
    // at once do not fit the SGPRs and spill through VGPR lanes)
    const T nf = (T)n;
    auto ranked = [&](int m, T xm) {
      if constexpr (SKIPNA) return times_flag(xm, clamp01(nf - (T)m));
      else return xm;
    };
    if constexpr (MS == 50 && NPAD == 64 && sizeof(T) == 4) {
      sort3_network_50(x);  // rank m lives in register kSort3Order50[m]
      if constexpr (WB2_ENS_PAIRED_SPREAD && !SKIPNA && !REFCHAIN) {
        // Ranks H + j and H + 1 - j (H = M / 2) carry the weights +-(2 j - 1):
        //   sum_r (2 r - M - 1) x_(r) = sum_j (2 j - 1) (x_(H+j) - x_(H+1-j)),
        // a sum of NON-NEGATIVE terms.  The differences are taken in float32
        // (exact whenever the two members are within a factor of two, Sterbenz)
        // and accumulated innermost pair first -- ascending magnitudes -- in two
        // float32 FMA chains that meet in fp64: 53 instructions instead of the
        // 100 of the fp64 form below, at most 1.1e-7 (rms 2.5e-8) away from it
        // on ERA5-like, normal and log-normal ensembles -- the accuracy of one
        // float32 rounding; the reference evaluates this sum in fp64
        // (int64 x float32, metrics.py:806-812) and the parity tolerance for
        // float32 ensembles is 1e-6.  inf / NaN members behave as there:
        // inf - finite = inf, and inf - inf = NaN exactly when an infinity
        // receives a non-positive weight.
        constexpr int H = MS / 2;
        constexpr auto& R = kSort3Layout50.of_rank;
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int j = 1; j <= H; ++j) {
          const float g = x[R[H + j - 1]] - x[R[H - j]];
          if (j & 1) s1 = __builtin_fmaf((float)(2 * j - 1), g, s1);
          else s0 = __builtin_fmaf((float)(2 * j - 1), g, s0);
        }
        s = (double)s0 + (double)s1;
      } else {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          s = __builtin_fma((double)(2 * (m + 1) - M - 1),
                            (double)ranked(m, x[kSort3Layout50.of_rank[m]]), s);
        }
      }
    } else if constexpr (Sort3<MS>::has && sizeof(T) == 4) {
      Sort3<MS>::template run<NPAD>(x);  // rank m: register Sort3<MS>::order[m]
      if constexpr (RTM) {
        // run-time ensemble size: the weight steps by 2 from 1 - M (exact in
        // fp64) -- converted per rank, hipcc makes all the weights first
        double w = (double)(1 - M);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          s = __builtin_fma(w, (double)ranked(m, x[Sort3<MS>::order[m]]), s);
          w += 2.0;
        }
      } else {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          s = __builtin_fma((double)(2 * (m + 1) - M - 1),
                            (double)ranked(m, x[Sort3<MS>::order[m]]), s);
        }
      }
    } else {
      sort_network<NPAD, NM>(x);
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        if constexpr (SKIPNA) {  // n <= M: dead slots are beyond rank n too
          s = __builtin_fma((double)(2 * (m + 1) - M - 1),
                            (double)ranked(m, x[m]), s);
        } else {
          s = __builtin_fma((double)(2 * (m + 1) - M - 1),
                            live(m) ? (double)x[m] : 0.0, s);
        }
      }
    }
    if constexpr (MS > 0 && !SKIPNA && !REFCHAIN) {
      // compile-time member count: 2 / (M (M - 1)) is one constant (<= 2 ulp of
      // fp64 away from the two divisions, far inside the summation noise)
      spread = s * (2.0 / ((double)MS * (double)(MS - 1)));
    } else {
      spread = 2.0 * (s / (double)cnt) / (double)(M - 1);
    }
    if (!SKIPNA && bad) spread = (double)nan;  // a NaN member poisons the mean
  }
  if constexpr (!SKIPNA) {
    out[0] = (double)skill;
    out[1] = spread;
    out[2] = (double)mse;
    out[3] = (double)var;
    out[4] = (double)(sd * sd);
    out[5] = (double)deb;
  } else {
    const bool ok_skill = !is_nan(skill), ok_spread = !is_nan(spread),
               ok_mse = !is_nan(mse), ok_var = !is_nan(var),
               ok_deb = !is_nan(deb);
    out[0] = ok_skill ? (double)skill : 0.0;
    out[1] = ok_spread ? spread : 0.0;
    out[2] = ok_mse ? (double)mse : 0.0;
    out[3] = ok_var ? (double)var : 0.0;
    out[4] = ok_var ? (double)(sd * sd) : 0.0;
    out[5] = ok_deb ? (double)deb : 0.0;
    out[6] = ok_skill ? 1.0 : 0.0;  // == ok_mse (same NaN pattern)
    out[7] = ok_spread ? 1.0 : 0.0;
    out[8] = ok_var ? 1.0 : 0.0;
    out[9] = ok_deb ? 1.0 : 0.0;
  }
}


// Runtime member count WITHOUT NaN skipping, lean form.  The caller has set the
// slots >= M to +inf; the dead slots are a wave-uniform SUFFIX, so the
// statistics walk the members in groups of four behind wave-uniform branches
// -- a group inside [0, M) runs without a single select, the group that
// straddles M takes per-member branches, groups beyond M are skipped -- and
// the padded network sorts the +inf to the end, where the rank-weighted sum
// stops.  Same operations in the same order as ens_point's generic path for
// the M live members (the select-per-member form cost six v_cndmask per member
// and kept a lane mask per member and phase in SGPRs: 0.26-0.50 of the HBM
// peak against 0.76 for the exact-50 kernel).
template <typename T, int NPAD>
__device__ __forceinline__ void ens_point_runtime(T (&x)[NPAD], const T t,
                                                  const int M,
                                                  double (&out)[6]) {
  const T nan = std::numeric_limits<T>::quiet_NaN();
  T sum = 0, sk = 0, sq = 0;
  bool bad = false;
  constexpr int G = 4;
  static_assert(NPAD % G == 0, "padded sizes are multiples of 4");
#pragma unroll
  for (int g = 0; g < NPAD; g += G) {
    if (g < M) {  // wave-uniform
      if (g + G <= M) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
          sum += x[g + u];
          sk += abs_of(t - x[g + u]);
        }
        bad = bad || __builtin_isunordered(x[g], x[g + 1]) ||
              __builtin_isunordered(x[g + 2], x[g + 3]);
      } else {
#pragma unroll
        for (int u = 0; u < G; ++u) {
          if (g + u < M) {
            sum += x[g + u];
            sk += abs_of(t - x[g + u]);
            bad = bad || is_nan(x[g + u]);
          }
        }
      }
    }
  }
  // metrics.py:562-565 / :824 -- numpy mean / var(ddof=1) / mean(abs) over the
  // leading (member) axis: sequential, in the input dtype
  const T mean = sum / (T)M;
#pragma unroll
  for (int g = 0; g < NPAD; g += G) {
    if (g < M) {
      if (g + G <= M) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
          const T d = x[g + u] - mean;
          sq += d * d;
        }
      } else {
#pragma unroll
        for (int u = 0; u < G; ++u) {
          if (g + u < M) {
            const T d = x[g + u] - mean;
            sq += d * d;
          }
        }
      }
    }
  }
  const T var = sq / (T)(M - 1);
  const T sd = sqrt_of(var);
  const T err = t - mean;
  const T mse = err * err;
  const T deb = mse - var / (T)M;
  const T skill = sk / (T)M;
  // metrics.py:804-813: ranks from the full ensemble; the +inf padding sorts
  // behind every live member (a NaN member poisons the result: `bad`)
  double spread = 0.0;
  if (M >= 2) {
    sort_network<NPAD, NPAD>(x);
    double s = 0.0;
    const int c0 = -M - 1;  // weight of rank r (0-based): 2 (r + 1) - M - 1
#pragma unroll
    for (int g = 0; g < NPAD; g += G) {
      if (g < M) {
        if (g + G <= M) {
#pragma unroll
          for (int u = 0; u < G; ++u)
            s = __builtin_fma((double)(2 * (g + u + 1) + c0), (double)x[g + u],
                              s);
        } else {
#pragma unroll
          for (int u = 0; u < G; ++u)
            if (g + u < M)
              s = __builtin_fma((double)(2 * (g + u + 1) + c0),
                                (double)x[g + u], s);
        }
      }
    }
    spread = 2.0 * (s / (double)M) / (double)(M - 1);
    if (bad) spread = (double)nan;  // a NaN member poisons the mean
  }
  out[0] = (double)skill;
  out[1] = spread;
  out[2] = (double)mse;
  out[3] = (double)var;
  out[4] = (double)(sd * sd);
  out[5] = (double)deb;
}


// Runtime member count M <= MS inside the exact MS-member sorting program
// (the 2- / 3-sorter programs of sort3_networks.inc): what ens_point_runtime
// does with a padded Batcher network -- the statistics over the first M members
// in groups of four behind wave-uniform branches, the dead slots (+inf, set by
// the caller) sorted to the end -- at the cost of the host program, not of the
// next power of two (44 members: the 51-member program's 585 instructions
// instead of the 64-network's 1 086).  Same operations in the same order as
// ens_point's generic path for the M live members.
// v_max_f32 / v_mul_legacy_f32 with the wave-uniform operand read straight
// from an SGPR (src0 of a VOP2): no v_mov, no lane mask per member.  (The
// operand goes in as an int: hipcc gives a FLOAT "s" operand a VGPR -- and
// builds it with a v_cndmask under a 64-bit lane mask, one SGPR pair per
// member.)
__device__ __forceinline__ float vmax_s(float x, float uniform) {
  float r;
  asm("v_max_f32 %0, %1, %2"
      : "=v"(r)
      : "s"(__builtin_bit_cast(int, uniform)), "v"(x));
  return r;
}
__device__ __forceinline__ float add_s(float x, float uniform) {
  float r;  // x + (0 | NaN): a live slot keeps its value, a dead one turns NaN
  asm("v_add_f32 %0, %1, %2"
      : "=v"(r)
      : "s"(__builtin_bit_cast(int, uniform)), "v"(x));
  return r;
}
__device__ __forceinline__ float times_sflag(float v, float uniform_flag) {
  float r;
  asm("v_mul_legacy_f32 %0, %1, %2"
      : "=v"(r)
      : "s"(__builtin_bit_cast(int, uniform_flag)), "v"(v));
  return r;
}

// Runtime member count M <= MS inside the exact MS-member sorting program
// (the 2- / 3-sorter programs of sort3_networks.inc) instead of a padded
// power-of-two network (44 members: the 45-member program, not the
// 64-network's 1 086 instructions).  The caller hands over the RAW loads: a dead
// slot holds 0 (its load was switched off through the buffer descriptor).
// Groups of eight members behind wave-uniform branches: a group that lies
// inside [0, M) runs the exact kernels' plain code, a group beyond M is skipped
// (its slots become +inf for the sort: they end up behind every live member
// and carry no rank weight), and only the ONE group that straddles M pays for
// flags -- f_m = (m < M) ? 1 : 0 in an SGPR, v_max_f32 against -+inf for the
// fill and v_mul_legacy_f32 (0 * inf = 0) for the sums, no lane masks.  (A
// straight-line form with flags on every member cost four VALU instructions
// per member and ~130 scalar spills per row; measured 5-10 % slower.)  Same
// operations in the same order as ens_point's generic path for the M live
// members: bit-identical to the padded runtime networks (tested).
template <int NPAD, int MS>
__device__ __forceinline__ void ens_point_hosted(float (&x)[NPAD], const float t,
                                                 const int M,
                                                 double (&out)[6]) {
  static_assert(Sort3<MS>::has, "no sorting program for this member count");
  using T = float;
  constexpr int G = 8;
  const T nan = std::numeric_limits<T>::quiet_NaN();
  const T inf = std::numeric_limits<T>::infinity();
  bool bad = false;
  T sum = 0, sk = 0, sq = 0;
#pragma unroll
  for (int g = 0; g < MS; g += G) {
    constexpr int dummy = 0;
    (void)dummy;
    const int hi = g + G < MS ? g + G : MS;  // compile time after unrolling
    if (hi <= M) {                            // wave-uniform: all live
      // member pairs, written as the exact kernels write them (separate
      // chains for the sum and the |t - x| sum: merged into packed adds by the
      // vectoriser they cost register pairs and copies)
#pragma unroll
      for (int m = g; m + 1 < hi; m += 2) {
        const T d0 = t - x[m], d1 = t - x[m + 1];
        sum += x[m];
        sk += abs_of(d0);
        sum += x[m + 1];
        sk += abs_of(d1);
        bad = bad || __builtin_isunordered(x[m], x[m + 1]);
      }
      if ((hi - g) % 2 == 1) {
        sum += x[hi - 1];
        sk += abs_of(t - x[hi - 1]);
        bad = bad || is_nan(x[hi - 1]);
      }
    } else if (g < M) {                       // the straddling group
#pragma unroll
      for (int m = g; m < hi; ++m) {
        const T f = m < M ? (T)1 : (T)0;
        const T floor_m = __builtin_bit_cast(
            T, (__builtin_bit_cast(unsigned, f) << 8) | 0x7f800000u);
        bad = bad || is_nan(x[m]);            // a dead slot holds 0
        x[m] = vmax_s(x[m], floor_m);         // dead -> +inf
        sum += times_sflag(x[m], f);
        sk += times_sflag(abs_of(t - x[m]), f);
      }
    } else {                                  // all dead
#pragma unroll
      for (int m = g; m < hi; ++m) x[m] = inf;
    }
  }
  const T mean = sum / (T)M;
#pragma unroll
  for (int g = 0; g < MS; g += G) {
    const int hi = g + G < MS ? g + G : MS;
    if (hi <= M) {
#pragma unroll
      for (int m = g; m < hi; ++m) {
        const T d = x[m] - mean;
        sq += d * d;
      }
    } else if (g < M) {
#pragma unroll
      for (int m = g; m < hi; ++m) {
        const T f = m < M ? (T)1 : (T)0;
        const T d = x[m] - mean;
        sq += times_sflag(d * d, f);
      }
    }
  }
  {
    int pinned = bad ? 1 : 0;
    asm volatile("" : "+v"(sq), "+v"(sum), "+v"(sk), "+v"(pinned));
    bad = pinned != 0;
  }
  const T var = sq / (T)(M - 1);
  const T sd = sqrt_of(var);
  const T err = t - mean;
  const T mse = err * err;
  const T deb = mse - var / (T)M;
  const T skill = sk / (T)M;
  double spread = 0.0;
  if (M >= 2) {
    Sort3<MS>::template run<NPAD>(x);  // rank r: register Sort3<MS>::order[r]
    double s = 0.0;
    // weight of rank r (0-based): 2 (r + 1) - M - 1, stepped by 2 (exact in
    // fp64) from 1 - M: one add per rank and ONE live value -- converted from
    // the integer per rank, hipcc makes all the weights first (2 VGPRs each)
    double w = (double)(1 - M);
#pragma unroll
    for (int g = 0; g < MS; g += G) {
      const int hi = g + G < MS ? g + G : MS;
      if (hi <= M) {
#pragma unroll
        for (int m = g; m < hi; ++m) {
          s = __builtin_fma(w, (double)x[Sort3<MS>::order[m]], s);
          w += 2.0;
        }
      } else if (g < M) {
#pragma unroll
        for (int m = g; m < hi; ++m) {
          const T f = m < M ? (T)1 : (T)0;
          s = __builtin_fma(w, (double)times_sflag(x[Sort3<MS>::order[m]], f),
                            s);
          w += 2.0;
        }
      }
    }
    spread = 2.0 * (s / (double)M) / (double)(M - 1);
  }
  // a NaN member poisons the mean and with it every value
  out[0] = bad ? (double)nan : (double)skill;
  out[1] = bad ? (double)nan : spread;
  out[2] = bad ? (double)nan : (double)mse;
  out[3] = bad ? (double)nan : (double)var;
  out[4] = bad ? (double)nan : (double)(sd * sd);
  out[5] = bad ? (double)nan : (double)deb;
}


// Ensembles too large for the register sort (M > 128 float32 / 64 float64):
// the same six values from three streaming passes over the members (cache
// resident after the first) -- no sort at all.  The rank-weighted sum is
//   sum_i (2 r_i - M - 1) x_i = P - (M - n) * sum_valid x_i,
//   P = sum_{i<j valid} |x_i - x_j|     (r: ranks in the full ensemble, NaN last)
// and P is accumulated blockwise: 32 members in VGPRs against every later
// member.  max(|d|, 0) drops the pairs that involve a NaN member (v_max returns
// the non-NaN operand) and the padding of the last block alike; without skipna
// a NaN member makes the result NaN, as in ens_point.
template <typename T, bool SKIPNA>
__device__ __forceinline__ void ens_point_large(
    const T* xrow, long long member_stride, int lane_bytes, const int M,
    const T t, double (&out)[SKIPNA ? 10 : 6]) {
  constexpr int B = 32;
  const T nan = std::numeric_limits<T>::quiet_NaN();
  T sum = 0, sk = 0;
  double sx = 0.0;  // fp64 sum of the valid members (rank correction term)
  int n = 0;
  bool bad = false;
#pragma unroll 4
  for (int m = 0; m < M; ++m) {
    const T x = member_load<T>(xrow + m * member_stride, lane_bytes);
    const bool isn = is_nan(x);
    const bool use = SKIPNA ? !isn : true;
    sum += use ? x : (T)0;
    sk += use ? abs_of(t - x) : (T)0;
    if constexpr (SKIPNA) sx += use ? (double)x : 0.0;
    n += use ? 1 : 0;
    bad = bad || isn;
  }
  const int cnt = SKIPNA ? n : M;
  const T mean = sum / (T)cnt;
  T sq = 0;
#pragma unroll 4
  for (int m = 0; m < M; ++m) {
    const T x = member_load<T>(xrow + m * member_stride, lane_bytes);
    const bool use = SKIPNA ? !is_nan(x) : true;
    const T d = x - mean;
    sq += use ? d * d : (T)0;
  }
  T var = sq / (T)(cnt - 1);
  if (SKIPNA && cnt <= 1) var = nan;
  const T sd = sqrt_of(var);
  const T err = t - mean;
  const T mse = err * err;
  const T deb = mse - var / (T)M;
  T skill = sk / (T)cnt;
  if (SKIPNA && is_nan(t)) skill = nan;

  double pairs = 0.0;  // P
  for (int a0 = 0; a0 < M; a0 += B) {
    T xa[B];
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int m = a0 + i < M ? a0 + i : M - 1;  // wave-uniform clamp
      const T x = member_load<T>(xrow + m * member_stride, lane_bytes);
      xa[i] = a0 + i < M ? x : nan;
    }
    // pairs inside the block: the full B x B table counts each one twice
    double own = 0.0;
    const int a1 = a0 + B < M ? a0 + B : M;
    for (int b = a0; b < a1; ++b) {
      const T xb = member_load<T>(xrow + b * member_stride, lane_bytes);
      T acc = 0;
#pragma unroll
      for (int i = 0; i < B; ++i) acc += abs_or_zero(xa[i] - xb);
      own += (double)acc;
    }
    double cross = 0.0;
    for (int b = a1; b < M; ++b) {
      const T xb = member_load<T>(xrow + b * member_stride, lane_bytes);
      T acc = 0;
#pragma unroll
      for (int i = 0; i < B; ++i) acc += abs_or_zero(xa[i] - xb);
      cross += (double)acc;
    }
    pairs += 0.5 * own + cross;
  }
  double spread = 0.0;
  if (M >= 2) {
    const double s = pairs - (SKIPNA ? (double)(M - n) * sx : 0.0);
    spread = 2.0 * (s / (double)cnt) / (double)(M - 1);
    if (!SKIPNA && bad) spread = (double)nan;
  }
  if constexpr (!SKIPNA) {
    out[0] = (double)skill;
    out[1] = spread;
    out[2] = (double)mse;
    out[3] = (double)var;
    out[4] = (double)(sd * sd);
    out[5] = (double)deb;
  } else {
    const bool ok_skill = !is_nan(skill), ok_spread = !is_nan(spread),
               ok_mse = !is_nan(mse), ok_var = !is_nan(var),
               ok_deb = !is_nan(deb);
    out[0] = ok_skill ? (double)skill : 0.0;
    out[1] = ok_spread ? spread : 0.0;
    out[2] = ok_mse ? (double)mse : 0.0;
    out[3] = ok_var ? (double)var : 0.0;
    out[4] = ok_var ? (double)(sd * sd) : 0.0;
    out[5] = ok_deb ? (double)deb : 0.0;
    out[6] = ok_skill ? 1.0 : 0.0;
    out[7] = ok_spread ? 1.0 : 0.0;
    out[8] = ok_var ? 1.0 : 0.0;
    out[9] = ok_deb ? 1.0 : 0.0;
  }
}

#ifndef WB2_ENS_WG_WAVES
// waves per workgroup (independent waves: the workgroup is only a scheduling
// unit).  2 instead of 4: +1 % with non-temporal loads (r03_k3_ab10_summary.txt)
#define WB2_ENS_WG_WAVES 2
#endif


// HOSTED: the member count is p.n_member <= MS at run time inside the exact
// MS-member program (ens_point_hosted); RT = the member count is a run-time
// value (members addressed one after the other, strided or gathered).
// GATHER (HOSTED only; the MS == 0 kernels decide at run time): the members are
// addressed through p.member_ptr -- a compile-time flag, because a run-time
// branch per member turns the member bases into per-lane values (two
// v_cndmask + a branch per member).
template <typename T, int NPAD, int MS, bool SKIPNA, bool WF,
          bool HOSTED = false, bool GATHER = false>
__global__ void __launch_bounds__(256)
    ens_partials_kernel(const EnsParams p) {
  static_assert(!HOSTED || (MS > 0 && sizeof(T) == 4),
                "hosted member counts: float32 programs");
  static_assert(HOSTED || !GATHER, "GATHER is a flag of the hosted kernels");
  constexpr bool RT = MS == 0 || HOSTED;
  constexpr int K = SKIPNA ? 10 : 6, NWF = WF ? 2 : 1;
  constexpr int NM = MS > 0 ? MS : NPAD;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int nwave = blockDim.x / kWave;
  const unsigned bx = blockIdx.x;
  const unsigned tblk = bx / (unsigned)p.n_chunk;
  const long long o = (long long)blockIdx.z * gridDim.y + blockIdx.y;
  // chunks rotated by the slab number: a fixed chunk -> XCD map (workgroup x
  // lands on XCD x % 8) would give some XCDs the short chunks of every slab
  const int chunk = (int)(((long long)(bx - tblk * (unsigned)p.n_chunk) +
                           (WB2_ROTATE_CHUNKS ? o : 0)) % p.n_chunk);
  const int tile = (int)tblk * nwave + wave;

  const int row0 = p.chunk_row0[chunk];
  const int nrow = p.chunk_nrow[chunk];
  const long long* dummy = reinterpret_cast<const long long*>(p.chunk_row0);
  const bool o_ok = o < p.n_outer;  // the (y, z) grid may overshoot n_outer
  const long long es_v =
      (p.ens_slab ? p.ens_slab : dummy)[(p.ens_slab && o_ok) ? o : 0];
  const long long ts_v =
      (p.truth_slab ? p.truth_slab : dummy)[(p.truth_slab && o_ok) ? o : 0];
  const long long es = p.ens_slab ? es_v : o, ts = p.truth_slab ? ts_v : o;
  const int col0 = tile * kWave + lane;
  const bool active = tile < p.n_ctile && col0 < p.n_col;
  if (nrow <= 0 || tile >= p.n_ctile || !o_ok) return;
  const int M = RT ? p.n_member : MS;

  double acc[NWF][1][K];
#pragma unroll
  for (int w = 0; w < NWF; ++w)
#pragma unroll
    for (int k = 0; k < K; ++k) acc[w][0][k] = 0.0;

  // gathered members: lane l keeps the slab address of member j * 64 + l in
  // a VGPR pair (ONE vector load per wave; indices >= M repeat the last
  // member, like the strided form); the row loop takes member m's address
  // out of lane m with two v_readlane -- scalar loads inside the loop would
  // each be a dependent round trip (measured: 6 x slower).  Loaded by EVERY
  // lane (in a row-end tile lane m may own no column; the row loop below runs
  // in wave-uniform control flow for the same reason).
  constexpr bool MAY_GATHER = HOSTED ? GATHER : (MS == 0 && NPAD > 0);
  constexpr int NMP = MAY_GATHER ? (NPAD + kWave - 1) / kWave : 1;
  unsigned mp_lo[NMP], mp_hi[NMP];
  const bool gathered = HOSTED ? GATHER : (MAY_GATHER && p.member_ptr != nullptr);
  if constexpr (MAY_GATHER) {
#pragma unroll
    for (int j = 0; j < NMP; ++j) {
      mp_lo[j] = mp_hi[j] = 0;
      if (gathered) {
        const int mi = j * kWave + lane;
        const unsigned long long a = (unsigned long long)
            p.member_ptr[o * (long long)M + (mi < M ? mi : M - 1)];
        mp_lo[j] = (unsigned)a;
        mp_hi[j] = (unsigned)(a >> 32);
      }
      // (side-effecting: keeps the load from being sunk into the branch below)
      asm volatile("" : "+v"(mp_lo[j]), "+v"(mp_hi[j]));
    }
  }
  // EVERY lane of the wave walks the rows: a lane past the end of the row (the
  // last column tile of a row whose length is no multiple of 64) reads the
  // row's last column again and its sums are dropped afterwards.  Control flow
  // stays wave-uniform, so the v_readlane of the gathered member addresses
  // (which ignore EXEC) can never meet a register copy made under a partial
  // EXEC mask.
  const int colc = active ? col0 : p.n_col - 1;
  {
    const long long slab_elems = (long long)p.n_row * p.n_col;
    // wave-uniform row base (SGPRs) + this lane's byte offset inside the row
    // (ens == NULL / truth == NULL: the tables hold byte ADDRESSES of the
    // slabs -- member 0's for the ensemble -- instead of slab numbers: the
    // variables of a chunk are separate allocations, wb2_ens_partials_addr;
    // wave-uniform)
    // table entry -> slab: entry * scale BYTES from the base.  scale = one
    // slab for tables of slab numbers; scale = 1 with a NULL base for tables
    // of byte addresses (the variables of a chunk are separate allocations:
    // wb2_ens_partials_addr) -- no branch either way
    const T* xrow0 = reinterpret_cast<const T*>(
                         static_cast<const char*>(p.ens) + es * p.ens_scale) +
                     (long long)row0 * p.n_col;
    const int lane_bytes = colc * (int)sizeof(T);
    const T* tb = reinterpret_cast<const T*>(
                      static_cast<const char*>(p.truth) + ts * p.truth_scale) +
                  (long long)row0 * p.n_col + colc;
    const double* wfp = WF ? p.wfield + (long long)row0 * p.n_col + colc
                           : nullptr;
    // One row of the chunk.  PASS 0: the whole job (every kernel but the exact
    // skipna ones).  Exact member count WITH NaN skipping: PASS 1 does the row
    // when the wave holds no NaN at all -- the select-free code of the
    // no-skipna kernel (fp64 rank chain, the skipna form's divisions: the same
    // bits as the general code gives for NaN-free points) -- and returns true
    // otherwise; PASS 2 does such a row with the general code.  NaNs come in
    // patches (a masked variable, a missing field): most waves hold none.
    auto row = [&](const int r, auto pass_tag) -> bool {
      constexpr int PASS = decltype(pass_tag)::value;
      const long long off = (long long)r * p.n_col;
      const T* xrow = xrow0 + off;
      const T t = __builtin_nontemporal_load(tb + off);
      const double wr = p.w_row[row0 + r];
      double wf = 1.0;
      if constexpr (WF) wf = wfp[off];
      double v[K];
      if constexpr (NPAD == 0) {  // any M: streaming passes, no sort
        ens_point_large<T, SKIPNA>(xrow, p.member_stride, lane_bytes, M, t, v);
      } else {
        T x[NPAD];
        // Runtime M: everything about the member bases is row-invariant, and
        // hipcc hoists all NPAD of them out of the row loop (64-bit pairs in
        // SGPRs: 192-468 dwords of SGPR spills).  The stride / the address
        // lanes are made opaque once per row instead, and the base advances
        // member by member.
        long long stride_r = p.member_stride;
        int Mr = M;  // runtime M, opaque per row: the per-member `m < M` lane
                     // masks are recomputed (scalar compares) instead of being
                     // kept in SGPR pairs across the whole row loop
        if constexpr (RT) {
          asm volatile("" : "+s"(Mr));
          asm volatile("" : "+s"(stride_r));
          if constexpr (MAY_GATHER) {
#pragma unroll
            for (int j = 0; j < NMP; ++j)
              asm volatile("" : "+v"(mp_lo[j]), "+v"(mp_hi[j]));
          }
        }
        const T* mb = xrow;
        // hosted, strided: the row base as two SGPRs for certain (readfirstlane
        // once per row), the member bases as a scalar add chain beside it
        unsigned long long hb0 = 0, hcur = 0, hstep = 0;
        if constexpr (HOSTED && !GATHER) {
          const unsigned long long a = reinterpret_cast<unsigned long long>(xrow);
          const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
          const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
          hb0 = hcur = ((unsigned long long)hi << 32) | lo;
          hstep = (unsigned long long)stride_r * sizeof(T);
        }
#pragma unroll
        for (int m = 0; m < NPAD; ++m) {
          if (m < NM) {
            const T* mrow;
            if constexpr (!RT) {
              mrow = xrow + m * p.member_stride;
            } else {
              // slots >= M read member 0 again (cache hit; replaced by +inf or
              // ignored); the base itself advances unconditionally -- a select
              // inside the chain made every load wait for 5 dependent scalar
              // instructions per member before it
              if constexpr (HOSTED && !GATHER) {
                // the base advances unconditionally; a dead slot's load is
                // switched off through its descriptor (below)
                mrow = reinterpret_cast<const T*>(hcur);
                hcur += hstep;
              } else if constexpr (!HOSTED) {
                mrow = m < Mr ? mb : xrow;
              }
              if constexpr (MAY_GATHER) {
                if (gathered) {
                  const unsigned lo = (unsigned)__builtin_amdgcn_readlane(
                      (int)mp_lo[m / kWave], m % kWave);
                  const unsigned hi = (unsigned)__builtin_amdgcn_readlane(
                      (int)mp_hi[m / kWave], m % kWave);
                  mrow = reinterpret_cast<const T*>(
                             ((unsigned long long)hi << 32) | lo) +
                         (long long)(row0 + r) * p.n_col;
                }
              }
              mb += stride_r;
            }
            if constexpr (HOSTED)
              x[m] = member_load<T, true>(mrow, lane_bytes,
                                          m < Mr ? 0x7fffffff : 0);
            else
              x[m] = member_load<T, true>(mrow, lane_bytes);
            // a fence every eight loads: the bases are made next to their
            // loads (hipcc would compute all of them first and spill them)
            if constexpr (HOSTED)
              if (m % 8 == 7) __builtin_amdgcn_sched_barrier(0);
          } else {
            x[m] = (T)0;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (HOSTED && SKIPNA) {
         if constexpr (PASS == 1) {
          // NaN skipping, a wave without a NaN (the dead slots hold 0): the
          // no-skipna hosted code -- the same operations in the same order as
          // the general code performs for NaN-free points, so a point's value
          // does not depend on which pass its wave took (tested pointwise)
          bool dirty = is_nan(t);
#pragma unroll
          for (int m = 0; m + 1 < MS; m += 2)
            dirty = dirty || __builtin_isunordered(x[m], x[m + 1]);
          if constexpr (MS % 2 == 1) dirty = dirty || is_nan(x[MS - 1]);
          if (__builtin_amdgcn_ballot_w64(dirty) != 0) return true;
          double q[6];
          ens_point_hosted<NPAD, MS>(x, t, Mr, q);
          const bool ok_skill = !is_nan(q[0]), ok_spread = !is_nan(q[1]),
                     ok_mse = !is_nan(q[2]), ok_var = !is_nan(q[3]),
                     ok_deb = !is_nan(q[5]);
          v[0] = ok_skill ? q[0] : 0.0;
          v[1] = ok_spread ? q[1] : 0.0;
          v[2] = ok_mse ? q[2] : 0.0;
          v[3] = ok_var ? q[3] : 0.0;
          v[4] = ok_var ? q[4] : 0.0;
          v[5] = ok_deb ? q[5] : 0.0;
          v[K - 4] = ok_skill ? 1.0 : 0.0;
          v[K - 3] = ok_spread ? 1.0 : 0.0;
          v[K - 2] = ok_var ? 1.0 : 0.0;
          v[K - 1] = ok_deb ? 1.0 : 0.0;
         } else {
          // a wave that holds a NaN: the dead slots (their loads returned 0)
          // become NaN and are skipped like NaN members by the general code
          const T nanv = std::numeric_limits<T>::quiet_NaN();
#pragma unroll
          for (int m = 0; m < MS; ++m) {
            x[m] = add_s(x[m], m < Mr ? (T)0 : nanv);
            if ((m & 7) == 7) __builtin_amdgcn_sched_barrier(0);
          }
          ens_point<T, NPAD, MS, true, false, true>(x, t, Mr, v);
         }
        } else if constexpr (HOSTED) {
          ens_point_hosted<NPAD, MS>(x, t, Mr, v);
        } else if constexpr (MS == 0 && !SKIPNA) {
          // dead slots (a wave-uniform suffix) become +inf ONCE, here
#pragma unroll
          for (int m = 0; m < NPAD; ++m)
            x[m] = m < Mr ? x[m] : std::numeric_limits<T>::infinity();
          ens_point_runtime<T, NPAD>(x, t, Mr, v);
        } else if constexpr (PASS == 1) {
          bool dirty = is_nan(t);
#pragma unroll
          for (int m = 0; m + 1 < NM; m += 2)
            dirty = dirty || __builtin_isunordered(x[m], x[m + 1]);
          if constexpr (NM % 2 == 1) dirty = dirty || is_nan(x[NM - 1]);
          if (__builtin_amdgcn_ballot_w64(dirty) != 0) return true;
          double q[6];
          ens_point<T, NPAD, MS, false, true>(x, t, Mr, q);
          const bool ok_skill = !is_nan(q[0]), ok_spread = !is_nan(q[1]),
                     ok_mse = !is_nan(q[2]), ok_var = !is_nan(q[3]),
                     ok_deb = !is_nan(q[5]);
          v[0] = ok_skill ? q[0] : 0.0;
          v[1] = ok_spread ? q[1] : 0.0;
          v[2] = ok_mse ? q[2] : 0.0;
          v[3] = ok_var ? q[3] : 0.0;
          v[4] = ok_var ? q[4] : 0.0;
          v[5] = ok_deb ? q[5] : 0.0;
          v[K - 4] = ok_skill ? 1.0 : 0.0;
          v[K - 3] = ok_spread ? 1.0 : 0.0;
          v[K - 2] = ok_var ? 1.0 : 0.0;
          v[K - 1] = ok_deb ? 1.0 : 0.0;
        } else {
          ens_point<T, NPAD, MS, SKIPNA>(x, t, Mr, v);
        }
      }
      if (p.maps && active) {
        // Spatial* metrics (metrics.py:718-772, 1244-1266, 1366-1399): the
        // pointwise values themselves; with SKIPNA slots 6.. flag the NaNs.
        const long long at = o * slab_elems + (long long)(row0 + r) * p.n_col +
                             col0;
        const long long plane = p.n_outer * slab_elems;
        const double qnan = __builtin_nan("");
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          double out_v = v[k];
          if constexpr (SKIPNA) {
            constexpr int flag[6] = {6, 7, 6, 8, 8, 9};
            out_v = v[flag[k]] != 0.0 ? v[k] : qnan;
          }
          __builtin_nontemporal_store(out_v, p.maps + k * plane + at);
        }
      }
#pragma unroll
      for (int k = 0; k < K; ++k)
        acc[0][0][k] = __builtin_fma(wr, v[k], acc[0][0][k]);
      if constexpr (WF) {
        const bool inside = wf > 0.0;
        const double w2 = inside ? wr * wf : 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k)
          acc[1][0][k] = __builtin_fma(w2, inside ? v[k] : 0.0, acc[1][0][k]);
      }
      return false;
    };
    constexpr bool TWO_PASS = SKIPNA && MS > 1 && NPAD > 0;
    if constexpr (TWO_PASS) {
      // rows with a NaN somewhere in the wave wait for a loop of their own:
      // two loops, two register allocations -- the NaN-free rows are not held
      // to the general code's register needs inside one loop body (and the
      // general code of PASS 0 is not instantiated at all: it would set the
      // kernel's register count).  64 rows at a time: one bit per row.
      for (int r0 = 0; r0 < nrow; r0 += 64) {
        const int r1 = r0 + 64 < nrow ? r0 + 64 : nrow;
        unsigned long long todo = 0;
#pragma clang loop unroll(disable)
        for (int r = r0; r < r1; ++r)
          if (row(r, std::integral_constant<int, 1>{}))
            todo |= 1ull << (r - r0);
#pragma clang loop unroll(disable)
        for (int r = r0; r < r1; ++r)
          if ((todo >> (r - r0)) & 1) row(r, std::integral_constant<int, 2>{});
      }
    } else {
#pragma clang loop unroll(disable)
      for (int r = 0; r < nrow; ++r) row(r, std::integral_constant<int, 0>{});
    }
    if (p.w_col) {
      const double wc = p.w_col[colc];
#pragma unroll
      for (int w = 0; w < NWF; ++w)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[w][0][k] *= wc;
    }
    if (!active) {
#pragma unroll
      for (int w = 0; w < NWF; ++w)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[w][0][k] = 0.0;
    }
  }
  fold_tile_to_segs<NWF, 1, K>(
      acc, lane, tile, col0, col0, p.n_col, p.seg_col0, p.seg_eoff, p.n_seg,
      p.n_ts,
      p.partials + (o * p.n_chunk + chunk) * (long long)(NWF * p.n_ts * K));
}

template <typename T, int NPAD, int MS>
int launch_ens(const EnsParams& p, bool skipna, bool wf, hipStream_t stream) {
  int nwave = p.n_ctile < WB2_ENS_WG_WAVES ? p.n_ctile : WB2_ENS_WG_WAVES;
  const int n_tblk = (p.n_ctile + nwave - 1) / nwave;
  const long long gy = p.n_outer < 32768 ? p.n_outer : 32768;
  const long long gz = (p.n_outer + gy - 1) / gy;
  const dim3 grid((unsigned)(p.n_chunk * n_tblk), (unsigned)gy, (unsigned)gz);
  const dim3 block(nwave * kWave);
#define WB2_L(S, W)                                                            \
  hipLaunchKernelGGL((ens_partials_kernel<T, NPAD, MS, S, W>), grid, block, 0, \
                     stream, p)
  if (skipna) {
    if (wf) WB2_L(true, true); else WB2_L(true, false);
  } else {
    if (wf) WB2_L(false, true); else WB2_L(false, false);
  }
#undef WB2_L
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

template <int NPAD, int MS>
int launch_ens_hosted(const EnsParams& p, bool skipna, bool wf,
                      hipStream_t stream) {
  int nwave = p.n_ctile < WB2_ENS_WG_WAVES ? p.n_ctile : WB2_ENS_WG_WAVES;
  const int n_tblk = (p.n_ctile + nwave - 1) / nwave;
  const long long gy = p.n_outer < 32768 ? p.n_outer : 32768;
  const long long gz = (p.n_outer + gy - 1) / gy;
  const dim3 grid((unsigned)(p.n_chunk * n_tblk), (unsigned)gy, (unsigned)gz);
  const dim3 block(nwave * kWave);
#define WB2_L(S, W, G)                                                  \
  hipLaunchKernelGGL(                                                   \
      (ens_partials_kernel<float, NPAD, MS, S, W, true, G>), grid, block, 0, \
      stream, p)
#define WB2_LS(S)                                            \
  if (p.member_ptr) {                                        \
    if (wf) WB2_L(S, true, true); else WB2_L(S, false, true);   \
  } else {                                                   \
    if (wf) WB2_L(S, true, false); else WB2_L(S, false, false); \
  }
  if (skipna) {
    WB2_LS(true)
  } else {
    WB2_LS(false)
  }
#undef WB2_LS
#undef WB2_L
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace
}  // namespace wb2
