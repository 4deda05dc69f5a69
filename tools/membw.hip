// Calibration kernels: what does a pure streaming READ reach on this box?
#include <hip/hip_runtime.h>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NARR, bool NT>
__global__ void __launch_bounds__(256) read_sum(const f4* a, const f4* b, const f4* c,
                                                long long n4, float* out) {
  float acc = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f4 x = NT ? __builtin_nontemporal_load(a + i) : a[i];
    acc += x.x + x.y + x.z + x.w;
    if (NARR > 1) { f4 y = NT ? __builtin_nontemporal_load(b + i) : b[i]; acc += y.x + y.y + y.z + y.w; }
    if (NARR > 2) { f4 z = NT ? __builtin_nontemporal_load(c + i) : c[i]; acc += z.x + z.y + z.z + z.w; }
  }
  if (acc == 12345.678f) out[0] = acc;
}

extern "C" int membw_read(const void* a, const void* b, const void* c, long long n4,
                          int narr, int nt, int blocks, void* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
#define L(N, T) hipLaunchKernelGGL((read_sum<N, T>), dim3(blocks), dim3(256), 0, s, \
    (const f4*)a, (const f4*)b, (const f4*)c, n4, (float*)out)
  if (narr == 1) { if (nt) L(1, true); else L(1, false); }
  else if (narr == 2) { if (nt) L(2, true); else L(2, false); }
  else { if (nt) L(3, true); else L(3, false); }
  return (int)hipGetLastError();
}
