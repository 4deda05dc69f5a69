// K7 of libwb2hip.so: means / raw moments along one (merged) axis.
//
// Replaces the compute core of the reference's averaging pipelines
// (reference = /root/reference/scripts):
//   compute_ensemble_mean.py:111-141        xbeam.Mean(realization, skipna)
//   compute_averages.py:125-167             v * lat_weights, then
//                                           xbeam.Mean(averaging_dims, skipna)
//   compute_statistical_moments.py:52-80    mean of notnull(x), x, x**2 over
//                                           (latitude, longitude), then time
// all of which are  sum_r w_r * f(x[l, r, t]) / count  over a reduced axis r of
// a [n_lead][n_red][n_tail] view (the host merges / permutes dims into that
// shape).  One read of x (HBM bound, sizeof(T) bytes per element), fp64 sums,
// deterministic: the reduced axis is cut into `n_split` slices whose partial
// sums are combined in slice order by a second tiny kernel (no atomics).
//
//   n_tail > 1  a thread owns one tail element and walks its slice of r
//               (loads coalesced along t);
//   n_tail == 1 a workgroup owns one (lead, slice): lanes stride over r
//               (coalesced), then a fixed-order wave + LDS tree.

#include "common.hpp"
#include "trace.hpp"
#include "wb2hip.h"

namespace wb2 {
namespace {

struct AxisParams {
  const void* x;
  const double* w_red;  // [n_red / w_repeat] or null
  double* part;         // [3][n_split][n_lead * n_tail]  (sum, sumsq, count)
  double* sum;          // final outputs, written directly when n_split == 1
  double* sumsq;
  double* count;
  long long n_lead, n_red, n_tail, slice;  // slice = elements of r per split
  long long w_repeat;   // weight j applies to r in [j * w_repeat, (j+1) * ...)
  int n_split, skipna, want_sq;
};

template <typename T, bool SKIPNA, bool SQ>
__device__ __forceinline__ void take(T x, double w, double& s, double& q,
                                     double& c) {
  if constexpr (SKIPNA) {
    const bool ok = !is_nan(x);
    s += ok ? w * (double)x : 0.0;
    if constexpr (SQ) q += ok ? w * (double)(x * x) : 0.0;  // np.square in T
    c += ok ? 1.0 : 0.0;
  } else {
    s = __builtin_fma(w, (double)x, s);
    if constexpr (SQ) q = __builtin_fma(w, (double)(x * x), q);
    // count is n_red: filled in by the combine step
  }
}

template <typename T, int VEC>
__device__ __forceinline__ void load_n(const T* p, T (&v)[VEC]) {
  if constexpr (VEC == 1) {
    v[0] = __builtin_nontemporal_load(p);
  } else {
    typedef T V __attribute__((ext_vector_type(VEC)));
    const V x = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = x[e];
  }
}

// n_tail > 1: a thread owns VEC consecutive tail elements (16-byte loads when
// the layout allows) and walks its slice of r, two rows in flight.
// grid: x = split * n_tail_blocks + tail block, (y, z) = lead
template <typename T, int VEC, bool SKIPNA, bool SQ>
__global__ void __launch_bounds__(256) axis_strided_kernel(const AxisParams p) {
  const long long per_blk = 256ll * VEC;
  const unsigned n_tblk = (unsigned)((p.n_tail + per_blk - 1) / per_blk);
  const int sp = (int)(blockIdx.x / n_tblk);
  const long long t =
      ((long long)(blockIdx.x - sp * n_tblk) * blockDim.x + threadIdx.x) * VEC;
  const long long l = blockIdx.y + (long long)blockIdx.z * gridDim.y;
  if (t >= p.n_tail || l >= p.n_lead) return;
  const long long r0 = (long long)sp * p.slice;
  const long long r1 = r0 + p.slice < p.n_red ? r0 + p.slice : p.n_red;
  const T* base = static_cast<const T*>(p.x) + (l * p.n_red) * p.n_tail + t;
  double s[VEC], q[VEC], c[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s[e] = q[e] = c[e] = 0.0;
  auto weight = [&](long long r) {
    return p.w_red ? p.w_red[r / p.w_repeat] : 1.0;
  };
  long long r = r0;
  constexpr int U = 8;  // rows in flight
  for (; r + U <= r1; r += U) {
    T v[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) load_n<T, VEC>(base + (r + u) * p.n_tail, v[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double w = weight(r + u);
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        take<T, SKIPNA, SQ>(v[u][e], w, s[e], q[e], c[e]);
    }
  }
  for (; r < r1; ++r) {
    T v[VEC];
    load_n<T, VEC>(base + r * p.n_tail, v);
    const double w = weight(r);
#pragma unroll
    for (int e = 0; e < VEC; ++e)
      take<T, SKIPNA, SQ>(v[e], w, s[e], q[e], c[e]);
  }
  const long long n_out = p.n_lead * p.n_tail;
  const long long at = (long long)sp * n_out + l * p.n_tail + t;
  if (p.n_split == 1) {  // no second stage: these are the results
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      p.sum[at + e] = s[e];
      if constexpr (SQ) p.sumsq[at + e] = q[e];
      p.count[at + e] = SKIPNA ? c[e] : (double)p.n_red;
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    p.part[at + e] = s[e];
    p.part[(long long)p.n_split * n_out + at + e] = q[e];
    p.part[2ll * p.n_split * n_out + at + e] = c[e];
  }
}

// n_tail == 1: a workgroup owns one (lead, slice); the slice is walked as runs
// of `w_repeat` elements that share one weight (e.g. a latitude row of a
// merged (latitude, longitude) axis), lanes striding over a run with VEC-wide
// loads; then a fixed-order wave + LDS tree.  grid: x = split, (y, z) = lead
template <typename T, int VEC, bool SKIPNA, bool SQ>
__global__ void __launch_bounds__(256) axis_contig_kernel(const AxisParams p) {
  __shared__ double lds[3][4];
  const long long l = blockIdx.y + (long long)blockIdx.z * gridDim.y;
  const int sp = blockIdx.x;
  if (l >= p.n_lead) return;  // workgroup-uniform
  const long long r0 = (long long)sp * p.slice;
  const long long r1 = r0 + p.slice < p.n_red ? r0 + p.slice : p.n_red;
  const T* base = static_cast<const T*>(p.x) + l * p.n_red;
  double s = 0.0, q = 0.0, c = 0.0;
  // Runs never straddle a weight boundary (`slice` is a multiple of w_repeat);
  // every wave takes every n-th run and strides over it with its 64 lanes.
  const int lane_ = threadIdx.x & (kWave - 1);
  const int wave_ = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const long long nw_ = blockDim.x / kWave;
  const long long step = (long long)kWave * VEC;
  for (long long run = r0 + wave_ * p.w_repeat; run < r1;
       run += nw_ * p.w_repeat) {
    const double w = p.w_red ? p.w_red[run / p.w_repeat] : 1.0;
    const long long end = run + p.w_repeat < r1 ? run + p.w_repeat : r1;
    long long r = run + (long long)lane_ * VEC;
    for (; r + 3 * step + VEC <= end; r += 4 * step) {  // four loads in flight
      T v[4][VEC];
#pragma unroll
      for (int u = 0; u < 4; ++u) load_n<T, VEC>(base + r + u * step, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          take<T, SKIPNA, SQ>(v[u][e], w, s, q, c);
    }
    for (; r + VEC <= end; r += step) {
      T v[VEC];
      load_n<T, VEC>(base + r, v);
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        take<T, SKIPNA, SQ>(v[e], w, s, q, c);
    }
    if constexpr (VEC > 1) {  // ragged end of the run
      for (; r < end; ++r)
        take<T, SKIPNA, SQ>(__builtin_nontemporal_load(base + r), w, s, q, c);
    }
  }
  s = wave_sum(s);
  q = wave_sum(q);
  c = wave_sum(c);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0) {
    lds[0][wave] = s;
    lds[1][wave] = q;
    lds[2][wave] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long n_out = p.n_lead;
    const long long at = (long long)sp * n_out + l;
    const int nw = blockDim.x / kWave;
    double a = 0.0, b = 0.0, d = 0.0;
    for (int w = 0; w < nw; ++w) {
      a += lds[0][w];
      b += lds[1][w];
      d += lds[2][w];
    }
    if (p.n_split == 1) {
      p.sum[at] = a;
      if constexpr (SQ) p.sumsq[at] = b;
      p.count[at] = SKIPNA ? d : (double)p.n_red;
    } else {
      p.part[at] = a;
      p.part[(long long)p.n_split * n_out + at] = b;
      p.part[2ll * p.n_split * n_out + at] = d;
    }
  }
}

__global__ void __launch_bounds__(256)
    axis_combine_kernel(const double* part, long long n_out, int n_split,
                        double n_red_if_all_valid, double* sum, double* sumsq,
                        double* count) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  double s = 0.0, q = 0.0, c = 0.0;
  for (int sp = 0; sp < n_split; ++sp) {  // fixed order
    s += part[(long long)sp * n_out + i];
    q += part[(long long)n_split * n_out + (long long)sp * n_out + i];
    c += part[2ll * n_split * n_out + (long long)sp * n_out + i];
  }
  sum[i] = s;
  if (sumsq) sumsq[i] = q;
  count[i] = n_red_if_all_valid >= 0.0 ? n_red_if_all_valid : c;
}

}  // namespace
}  // namespace wb2

// (dtype, vector width, skipna, want_sq) -> template instance of WB2_K
#define WB2_AXIS_L(T, V, S, Q) \
  hipLaunchKernelGGL((WB2_K<T, V, S, Q>), grid, dim3(256), 0, s, p)
#define WB2_AXIS_FLAGS(T, V)                              \
  do {                                                    \
    if (skipna) {                                         \
      if (sumsq) WB2_AXIS_L(T, V, true, true);            \
      else WB2_AXIS_L(T, V, true, false);                 \
    } else {                                              \
      if (sumsq) WB2_AXIS_L(T, V, false, true);           \
      else WB2_AXIS_L(T, V, false, false);                \
    }                                                     \
  } while (0)
#define WB2_AXIS_DISPATCH                                 \
  if (dtype == WB2_F32) {                                 \
    if (vec) WB2_AXIS_FLAGS(float, 4);                    \
    else WB2_AXIS_FLAGS(float, 1);                        \
  } else {                                                \
    if (vec) WB2_AXIS_FLAGS(double, 2);                   \
    else WB2_AXIS_FLAGS(double, 1);                       \
  }

extern "C" {

int wb2_axis_moments_splits(int64_t n_lead, int64_t n_red, int64_t n_tail,
                            int64_t w_repeat) {
  // enough slices of the reduced axis for ~2 workgroups per CU... of work, each
  // at least 64 rows (strided) / 4096 elements (contiguous) long and a whole
  // number of weight runs
  if (n_lead <= 0 || n_red <= 0 || n_tail <= 0) return 1;
  if (w_repeat < 1) w_repeat = 1;
  const long long min_slice = n_tail > 1 ? 64 : 4096;
  const long long work = n_tail > 1 ? n_lead * ((n_tail + 1023) / 1024) : n_lead;
  long long want = (2048 + work - 1) / work;
  long long max_split = (n_red + min_slice - 1) / min_slice;
  const long long runs = (n_red + w_repeat - 1) / w_repeat;
  if (max_split > runs) max_split = runs;
  if (want > max_split) want = max_split;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  return (int)want;
}

int wb2_axis_moments(int dtype, const void* x, int64_t n_lead, int64_t n_red,
                     int64_t n_tail, const double* w_red, int64_t w_repeat,
                     int skipna, int n_split, double* workspace, double* sum,
                     double* sumsq, double* count, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_REQUIRE(n_lead >= 0 && n_red >= 0 && n_tail >= 0, "bad sizes");
  const long long n_out = n_lead * n_tail;
  if (n_out == 0) return 0;
  WB2_REQUIRE(x || n_red == 0, "x is null");
  WB2_REQUIRE(sum && count && workspace, "null pointer argument");
  WB2_REQUIRE(n_split >= 1 && n_split <= 65535, "n_split=%d", n_split);
  if (!w_red || w_repeat < 1) w_repeat = w_red ? 1 : (n_red > 0 ? n_red : 1);
  WB2_REQUIRE(n_red % w_repeat == 0 || !w_red,
              "n_red=%lld is not a multiple of w_repeat=%lld",
              (long long)n_red, (long long)w_repeat);
  const long long gy = n_lead < 32768 ? n_lead : 32768;
  const long long gz = (n_lead + gy - 1) / gy;
  WB2_REQUIRE(gz <= 65535, "n_lead=%lld too large", (long long)n_lead);
  hipStream_t s = static_cast<hipStream_t>(stream);
  AxisParams p{};
  p.x = x;
  p.w_red = w_red;
  p.part = workspace;
  p.sum = sum;
  p.sumsq = sumsq;
  p.count = count;
  p.n_lead = n_lead;
  p.n_red = n_red;
  p.n_tail = n_tail;
  p.n_split = n_split;
  p.w_repeat = w_repeat;
  // slices are whole weight runs (unweighted: one run = the slice itself)
  {
    // unweighted: any partition into runs will do; 4096 elements per run keep
    // all waves of a workgroup busy and every run start 16-byte aligned
    const long long unit = w_red ? w_repeat : (n_tail > 1 ? 1 : 4096);
    const long long units = n_red > 0 ? (n_red + unit - 1) / unit : 1;
    p.slice = ((units + n_split - 1) / n_split) * unit;
    if (!w_red) p.w_repeat = n_tail > 1 ? (p.slice > 0 ? p.slice : 1) : unit;
  }
  p.skipna = skipna;
  p.want_sq = sumsq != nullptr;
  const int elt = dtype == WB2_F32 ? 4 : 8;
  const int wide = 16 / elt;
  const bool aligned = reinterpret_cast<uintptr_t>(x) % 16 == 0;
  if (n_tail > 1) {
    const bool vec = aligned && n_tail % wide == 0;
    const long long per_blk = 256ll * (vec ? wide : 1);
    const long long gx = ((n_tail + per_blk - 1) / per_blk) * n_split;
    WB2_REQUIRE(gx < (1ll << 31), "n_tail=%lld too large", (long long)n_tail);
    const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)gz);
#define WB2_K axis_strided_kernel
    WB2_AXIS_DISPATCH
#undef WB2_K
  } else {
    // every run start must be 16-byte aligned for the wide loads
    const bool vec = aligned && n_red % wide == 0 && p.w_repeat % wide == 0;
    const dim3 grid((unsigned)n_split, (unsigned)gy, (unsigned)gz);
#define WB2_K axis_contig_kernel
    WB2_AXIS_DISPATCH
#undef WB2_K
  }
  WB2_HIP_OK(hipGetLastError());
  if (n_split == 1) return 0;
  hipLaunchKernelGGL(axis_combine_kernel, dim3((unsigned)((n_out + 255) / 256)),
                     dim3(256), 0, s, workspace, n_out, n_split,
                     skipna ? -1.0 : (double)n_red, sum, sumsq, count);
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
