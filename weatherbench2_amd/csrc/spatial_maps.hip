// K5 of libwb2hip.so: the Spatial* metrics (no spatial reduction).
//
// Replaces (reference = /root/reference/weatherbench2/metrics.py):
//   SpatialMSE  :304-316   (forecast - truth) ** 2
//   SpatialMAE  :333-345   abs(forecast - truth)
//   SpatialBias :362-374   forecast - truth
// and their temporal mean, Metric.compute :117-138 / xbeam.Mean
// (evaluation.py:740-744), which is where the time goes in the
// `deterministic_spatial` configuration (scripts/evaluate.py:471-478): the
// reference materialises three full-size maps per time step and then averages.
//
// wb2_spatial_maps        one pass, up to three maps written in the input dtype
// wb2_spatial_accumulate  a thread owns 16 bytes of one (rest, point) location,
//                         walks the chunk's time steps in order with the sums
//                         in registers (fp64) and touches the (sum, count)
//                         accumulators once per launch: 8 B per point-time read,
//                         no per-time map traffic at all.
// Pure streams (HBM bound); elementwise arithmetic in the input dtype.

#include "common.hpp"
#include "trace.hpp"
#include "wb2hip.h"

namespace wb2 {
namespace {

struct SpatialParams {
  const void* f;
  const void* t;
  const long long* f_slab;
  const long long* t_slab;
  void* out[3];     // bias, mse, mae maps (wb2_spatial_maps), any may be null
  double* sum;      // [3][n_rest][n_point]
  double* count;    // [3][n_rest][n_point] (skipna only)
  long long n_time, n_rest, n_point;
};

#ifndef WB2_MAPS_NT_STORES
#define WB2_MAPS_NT_STORES 1
#endif
#if WB2_MAPS_NT_STORES
#define WB2_MAPS_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define WB2_MAPS_STORE(v, p) (*(p) = (v))
#endif

#ifndef WB2_SPATIAL_IN_NT
#define WB2_SPATIAL_IN_NT 1
#endif
template <typename T, int VEC>
__device__ __forceinline__ void load_v(const T* p, T (&v)[VEC]) {
  if constexpr (VEC == 1) {
    v[0] = __builtin_nontemporal_load(p);
  } else {
    typedef T V __attribute__((ext_vector_type(VEC)));
#if WB2_SPATIAL_IN_NT
    const V x = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
#else
    const V x = *reinterpret_cast<const V*>(p);
#endif
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = x[e];
  }
}

template <typename T, int VEC>
__device__ __forceinline__ void store_v(T* p, const T (&v)[VEC]) {
  if constexpr (VEC == 1) {
    WB2_MAPS_STORE(v[0], p);
  } else {
    typedef T V __attribute__((ext_vector_type(VEC)));
    V x;
#pragma unroll
    for (int e = 0; e < VEC; ++e) x[e] = v[e];
    WB2_MAPS_STORE(x, reinterpret_cast<V*>(p));
  }
}

// grid: x = point blocks, y = slab (time * n_rest + rest)
template <typename T, int VEC>
__global__ void __launch_bounds__(256) spatial_maps_kernel(const SpatialParams p) {
  const long long q = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (q >= p.n_point) return;
  const long long o = blockIdx.y + (long long)blockIdx.z * gridDim.y;
  if (o >= p.n_time) return;  // n_time carries n_outer here
  const long long fs = p.f_slab ? p.f_slab[o] : o;
  const long long ts = p.t_slab ? p.t_slab[o] : o;
  T f[VEC], t[VEC], d[VEC], d2[VEC], ad[VEC];
  load_v<T, VEC>(static_cast<const T*>(p.f) + fs * p.n_point + q, f);
  load_v<T, VEC>(static_cast<const T*>(p.t) + ts * p.n_point + q, t);
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    d[e] = f[e] - t[e];
    d2[e] = d[e] * d[e];
    ad[e] = abs_of(d[e]);
  }
  const long long off = o * p.n_point + q;
  if (p.out[0]) store_v<T, VEC>(static_cast<T*>(p.out[0]) + off, d);
  if (p.out[1]) store_v<T, VEC>(static_cast<T*>(p.out[1]) + off, d2);
  if (p.out[2]) store_v<T, VEC>(static_cast<T*>(p.out[2]) + off, ad);
}

// grid: x = point blocks, y = rest index
template <typename T, int VEC, bool SKIPNA>
__global__ void __launch_bounds__(256)
    spatial_accumulate_kernel(const SpatialParams p) {
  const long long q = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (q >= p.n_point) return;
  const long long j = blockIdx.y + (long long)blockIdx.z * gridDim.y;
  if (j >= p.n_rest) return;
  double s[3][VEC], c[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    s[0][e] = s[1][e] = s[2][e] = 0.0;
    c[e] = 0.0;
  }
  constexpr int U = 4;  // time steps in flight (8 x 16 B per thread)
  long long i = 0;
  auto body = [&](const T (&f)[VEC], const T (&t)[VEC]) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const T d = f[e] - t[e];
      const T d2 = d * d;
      const T ad = abs_of(d);
      const bool keep = !(SKIPNA && is_nan(d));
      s[0][e] += keep ? (double)d : 0.0;
      s[1][e] += keep ? (double)d2 : 0.0;
      s[2][e] += keep ? (double)ad : 0.0;
      if constexpr (SKIPNA) c[e] += keep ? 1.0 : 0.0;
    }
  };
  auto slab = [&](const long long* tab, long long it) {
    const long long o = it * p.n_rest + j;
    return tab ? tab[o] : o;
  };
  for (; i + U <= p.n_time; i += U) {
    T f[U][VEC], t[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      load_v<T, VEC>(static_cast<const T*>(p.f) +
                         slab(p.f_slab, i + u) * p.n_point + q, f[u]);
      load_v<T, VEC>(static_cast<const T*>(p.t) +
                         slab(p.t_slab, i + u) * p.n_point + q, t[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) body(f[u], t[u]);
  }
  for (; i < p.n_time; ++i) {
    T f[VEC], t[VEC];
    load_v<T, VEC>(static_cast<const T*>(p.f) +
                       slab(p.f_slab, i) * p.n_point + q, f);
    load_v<T, VEC>(static_cast<const T*>(p.t) +
                       slab(p.t_slab, i) * p.n_point + q, t);
    body(f, t);
  }
  const long long plane = p.n_rest * p.n_point;
  const long long off = j * p.n_point + q;
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int e = 0; e < VEC; ++e) p.sum[m * plane + off + e] += s[m][e];
  if constexpr (SKIPNA) {
#pragma unroll
    for (int m = 0; m < 3; ++m)  // same NaN pattern for d, d^2, |d|
#pragma unroll
      for (int e = 0; e < VEC; ++e) p.count[m * plane + off + e] += c[e];
  }
}

// the running sums of wb2_spatial_accumulate_addr are touched once per chunk
// (4+ GB of other traffic in between): streamed past the caches or not
#ifndef WB2_SPATIAL_ACC_NT
#define WB2_SPATIAL_ACC_NT 0
#endif
__device__ __forceinline__ double acc_load(const double* p) {
#if WB2_SPATIAL_ACC_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void acc_store(double v, double* p) {
#if WB2_SPATIAL_ACC_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

struct SpatialAddrParams {
  const long long* f_addr;      // [n_time][n_dst]
  const long long* t_addr;
  const long long* sum_addr;    // [3][n_dst], 0 = not wanted
  const long long* count_addr;  // [3][n_dst] (skipna only)
  long long n_time, n_dst, n_point;
};

// grid: x = point blocks, y = destination.  The thread owns VEC points of one
// destination: it reads the (up to three) running sums, adds the chunk's time
// steps value by value and writes them back.
template <typename T, int VEC, bool SKIPNA>
__global__ void __launch_bounds__(256)
    spatial_accumulate_addr_kernel(const SpatialAddrParams p) {
  const long long q = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (q >= p.n_point) return;
  const long long j = blockIdx.y + (long long)blockIdx.z * gridDim.y;
  if (j >= p.n_dst) return;
  double* sp[3];
  double* cp[3];
  double s[3][VEC], c[3][VEC];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    sp[m] = reinterpret_cast<double*>(p.sum_addr[m * p.n_dst + j]);
    cp[m] = SKIPNA ? reinterpret_cast<double*>(p.count_addr[m * p.n_dst + j])
                   : nullptr;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      s[m][e] = sp[m] ? acc_load(sp[m] + q + e) : 0.0;
      c[m][e] = (SKIPNA && cp[m]) ? acc_load(cp[m] + q + e) : 0.0;
    }
  }
  auto body = [&](const T (&f)[VEC], const T (&t)[VEC]) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const T d = f[e] - t[e];
      const T d2 = d * d;
      const T ad = abs_of(d);
      const bool keep = !(SKIPNA && is_nan(d));
      s[0][e] += keep ? (double)d : 0.0;
      s[1][e] += keep ? (double)d2 : 0.0;
      s[2][e] += keep ? (double)ad : 0.0;
      if constexpr (SKIPNA) {
#pragma unroll
        for (int m = 0; m < 3; ++m) c[m][e] += keep ? 1.0 : 0.0;
      }
    }
  };
  constexpr int U = 4;  // time steps in flight
  long long i = 0;
  for (; i + U <= p.n_time; i += U) {
    T f[U][VEC], t[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long o = (i + u) * p.n_dst + j;
      load_v<T, VEC>(reinterpret_cast<const T*>(p.f_addr[o]) + q, f[u]);
      load_v<T, VEC>(reinterpret_cast<const T*>(p.t_addr[o]) + q, t[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) body(f[u], t[u]);
  }
  for (; i < p.n_time; ++i) {
    const long long o = i * p.n_dst + j;
    T f[VEC], t[VEC];
    load_v<T, VEC>(reinterpret_cast<const T*>(p.f_addr[o]) + q, f);
    load_v<T, VEC>(reinterpret_cast<const T*>(p.t_addr[o]) + q, t);
    body(f, t);
  }
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    if (sp[m]) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc_store(s[m][e], sp[m] + q + e);
    }
    if constexpr (SKIPNA) {
      if (cp[m]) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc_store(c[m][e], cp[m] + q + e);
      }
    }
  }
}

int pick_vec(int dtype, long long n_point, const void* a, const void* b,
             void* const* outs) {
  const int w = dtype == WB2_F32 ? 4 : 2;
  bool ok = n_point % w == 0 && reinterpret_cast<uintptr_t>(a) % 16 == 0 &&
            reinterpret_cast<uintptr_t>(b) % 16 == 0;
  for (int i = 0; outs && i < 3; ++i)
    ok = ok && reinterpret_cast<uintptr_t>(outs[i]) % 16 == 0;
  return ok ? w : 1;
}

dim3 grid_for(long long n_point, int vec, long long n_y) {
  const long long gx = (n_point / vec + 255) / 256;
  const long long gy = n_y < 32768 ? n_y : 32768;
  return dim3((unsigned)gx, (unsigned)gy, (unsigned)((n_y + gy - 1) / gy));
}

}  // namespace
}  // namespace wb2

extern "C" {

int wb2_spatial_maps(int dtype, const void* forecast, const int64_t* f_slab,
                     const void* truth, const int64_t* t_slab, int64_t n_outer,
                     int64_t n_point, void* bias, void* mse, void* mae,
                     void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_EMPTY_OK(n_outer);
  WB2_REQUIRE(forecast && truth, "null pointer argument");
  WB2_REQUIRE(n_point > 0, "bad sizes");
  SpatialParams p{};
  p.f = forecast;
  p.t = truth;
  p.f_slab = reinterpret_cast<const long long*>(f_slab);
  p.t_slab = reinterpret_cast<const long long*>(t_slab);
  p.out[0] = bias;
  p.out[1] = mse;
  p.out[2] = mae;
  p.n_point = n_point;
  p.n_time = n_outer;
  const int vec = pick_vec(dtype, n_point, forecast, truth, p.out);
  const dim3 grid = grid_for(n_point, vec, n_outer);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define WB2_L(T, V) \
  hipLaunchKernelGGL((spatial_maps_kernel<T, V>), grid, dim3(256), 0, s, p)
  if (dtype == WB2_F32) { if (vec > 1) WB2_L(float, 4); else WB2_L(float, 1); }
  else { if (vec > 1) WB2_L(double, 2); else WB2_L(double, 1); }
#undef WB2_L
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

int wb2_spatial_accumulate(int dtype, int skipna, const void* forecast,
                           const int64_t* f_slab, const void* truth,
                           const int64_t* t_slab, int64_t n_time,
                           int64_t n_rest, int64_t n_point, double* sum,
                           double* count, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_EMPTY_OK(n_time);
  WB2_EMPTY_OK(n_rest);
  WB2_REQUIRE(forecast && truth && sum && (count || !skipna),
              "null pointer argument");
  WB2_REQUIRE(n_point > 0, "bad sizes");
  SpatialParams p{};
  p.f = forecast;
  p.t = truth;
  p.f_slab = reinterpret_cast<const long long*>(f_slab);
  p.t_slab = reinterpret_cast<const long long*>(t_slab);
  p.sum = sum;
  p.count = count;
  p.n_time = n_time;
  p.n_rest = n_rest;
  p.n_point = n_point;
  const int vec = pick_vec(dtype, n_point, forecast, truth, nullptr);
  const dim3 grid = grid_for(n_point, vec, n_rest);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define WB2_L(T, V)                                                          \
  do {                                                                       \
    if (skipna)                                                              \
      hipLaunchKernelGGL((spatial_accumulate_kernel<T, V, true>), grid,      \
                         dim3(256), 0, s, p);                                \
    else                                                                     \
      hipLaunchKernelGGL((spatial_accumulate_kernel<T, V, false>), grid,     \
                         dim3(256), 0, s, p);                                \
  } while (0)
  if (dtype == WB2_F32) { if (vec > 1) WB2_L(float, 4); else WB2_L(float, 1); }
  else { if (vec > 1) WB2_L(double, 2); else WB2_L(double, 1); }
#undef WB2_L
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

int wb2_spatial_accumulate_addr(int dtype, int skipna, int aligned16,
                                const int64_t* f_addr, const int64_t* t_addr,
                                int64_t n_time, int64_t n_dst, int64_t n_point,
                                const int64_t* sum_addr,
                                const int64_t* count_addr, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_EMPTY_OK(n_time);
  WB2_EMPTY_OK(n_dst);
  WB2_REQUIRE(f_addr && t_addr && sum_addr && (count_addr || !skipna),
              "null pointer argument");
  WB2_REQUIRE(n_point > 0, "bad sizes");
  if (n_time == 0 || n_dst == 0) return 0;
  SpatialAddrParams p{};
  p.f_addr = reinterpret_cast<const long long*>(f_addr);
  p.t_addr = reinterpret_cast<const long long*>(t_addr);
  p.sum_addr = reinterpret_cast<const long long*>(sum_addr);
  p.count_addr = reinterpret_cast<const long long*>(count_addr);
  p.n_time = n_time;
  p.n_dst = n_dst;
  p.n_point = n_point;
  const int w = dtype == WB2_F32 ? 4 : 2;
  const int vec = (aligned16 && n_point % w == 0) ? w : 1;
  const dim3 grid = grid_for(n_point, vec, n_dst);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define WB2_L(T, V)                                                          \
  do {                                                                       \
    if (skipna)                                                              \
      hipLaunchKernelGGL((spatial_accumulate_addr_kernel<T, V, true>), grid, \
                         dim3(256), 0, s, p);                                \
    else                                                                     \
      hipLaunchKernelGGL((spatial_accumulate_addr_kernel<T, V, false>),      \
                         grid, dim3(256), 0, s, p);                          \
  } while (0)
  if (dtype == WB2_F32) { if (vec > 1) WB2_L(float, 4); else WB2_L(float, 1); }
  else { if (vec > 1) WB2_L(double, 2); else WB2_L(double, 1); }
#undef WB2_L
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
