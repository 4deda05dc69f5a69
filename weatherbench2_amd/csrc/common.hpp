// Shared helpers for libwb2hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

namespace wb2 {

// Thread-local last-error message (C ABI: wb2_last_error()).
char* error_buffer();
int fail(const char* fmt, ...);

#define WB2_HIP_OK(expr)                                                     \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess)                                                    \
      return ::wb2::fail("%s failed: %s (%s:%d)", #expr,                     \
                         hipGetErrorString(_e), __FILE__, __LINE__);         \
  } while (0)

#define WB2_REQUIRE(cond, ...)                       \
  do {                                               \
    if (!(cond)) return ::wb2::fail(__VA_ARGS__);    \
  } while (0)

// A call over zero units (an empty chunk: no times, no levels, ...) is a legal
// no-op whatever the data pointers are -- empty buffers have no address --, so
// every entry point returns here BEFORE its null-pointer checks; a negative
// count is an error.
#define WB2_EMPTY_OK(n)                                             \
  do {                                                              \
    if ((n) < 0) return ::wb2::fail(#n "=%lld is negative", (long long)(n)); \
    if ((n) == 0) return 0;                                         \
  } while (0)

// A wavefront is 64 lanes on gfx950.
constexpr int kWave = 64;

// Deterministic 64-lane tree sum; the result is valid in lane 0.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}

// Deterministic 64-lane butterfly sum; every lane gets the result.
__device__ __forceinline__ double wave_allsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

__device__ __forceinline__ bool is_nan(float x) { return x != x; }
__device__ __forceinline__ bool is_nan(double x) { return x != x; }
__device__ __forceinline__ float abs_of(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double abs_of(double x) { return __builtin_fabs(x); }

}  // namespace wb2
