// Access-pattern probe for K1: 3 arrays of [rows][1440] f32, 384-thread blocks (one float4 per
// lane per row, 360 active lanes), items of `rpi` consecutive rows.
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void __launch_bounds__(384) probe(const f4* __restrict__ a, const f4* __restrict__ b,
                                             const f4* __restrict__ c, int n_items, int rpi, int mode,
                                             int* counter, float* out) {
  extern __shared__ int dyn_lds[];  // only to limit WGs per CU
  __shared__ int lds[2];
  const int tid = threadIdx.x;
  const bool active = tid < 360;
  float acc = 0.f;
  int item = blockIdx.x;
  if (mode == 2) {
    if (tid == 0) lds[0] = atomicAdd(counter, 1);
    __syncthreads();
    item = lds[0];
    __syncthreads();
  }
  while (item < n_items) {
    if (active) {
      const long long base = (long long)item * rpi * 360 + tid;
      for (int r = 0; r < rpi; r += U) {
        f4 x[U], y[U], z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long long i = base + (long long)(r + u) * 360;
          x[u] = NT ? __builtin_nontemporal_load(a + i) : a[i];
          y[u] = NT ? __builtin_nontemporal_load(b + i) : b[i];
          z[u] = NT ? __builtin_nontemporal_load(c + i) : c[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          acc += (x[u].x - y[u].x) * (x[u].y - z[u].y) + (x[u].z - y[u].z) * (x[u].w - z[u].w) + y[u].y + z[u].z;
      }
    }
    if (mode == 0) break;
    if (mode == 1) item += gridDim.x;
    if (mode == 2) {
      if (tid == 0) lds[0] = atomicAdd(counter, 1);
      __syncthreads();
      item = lds[0];
      __syncthreads();
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

extern "C" int probe_launch(const void* a, const void* b, const void* c, int n_items, int rpi, int mode,
                            int U, int nt, int grid, int lds_bytes, void* counter, void* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (mode == 2) (void)hipMemsetAsync(counter, 0, 4, s);
#define L(UU, T) hipLaunchKernelGGL((probe<UU, T>), dim3(grid), dim3(384), lds_bytes, s, (const f4*)a, \
    (const f4*)b, (const f4*)c, n_items, rpi, mode, (int*)counter, (float*)out)
  if (nt) { if (U == 1) L(1, true); else if (U == 2) L(2, true); else L(4, true); }
  else    { if (U == 1) L(1, false); else if (U == 2) L(2, false); else L(4, false); }
  return (int)hipGetLastError();
}
