// K4f of libwb2hip.so: the zonal energy spectrum as ONE kernel.
//
// Same arithmetic contract as spectrum.hip (ZonalEnergySpectrum.compute,
// /root/reference/weatherbench2/derived_variables.py:592-626), but the real FFT
// along longitude is done here, in LDS, so each input value is read once and
// each spectrum value written once (8 B per grid point instead of the 16 B of
// the rocFFT + epilogue pipeline).  Used for float32 rows whose length N is even
// with N/2 = 2^a 3^b 5^c in the instantiated set (0.25-degree N = 1440 included);
// everything else takes the rocFFT path.
//
// One WAVE transforms one latitude row:
//   * the row is read with 16-byte loads straight into the wave's LDS slab as
//     N/2 complex points z[m] = x[2m] + i x[2m+1];
//   * Stockham passes of radix 4/2/3/5: every lane reads the inputs of all its
//     butterflies into VGPRs, then writes the outputs back in place (LDS is
//     in-order per wave, so one slab suffices and no barrier is needed);
//   * the real-FFT recombination X[k] = E[k] + W^k O[k] is evaluated for the
//     bin pair (k, N/2 - k) from one pair of loads (|E + WO|^2, |E - WO|^2),
//     then scaled (1/N in fp32, x{1,2}, x circumference in fp64) and stored.
// Twiddles come from two small fp32 tables (evaluated in fp64, rounded once)
// that the caller keeps in the workspace; a workgroup copies them to LDS once.

#include "common.hpp"
#include "wb2hip.h"

#ifndef WB2_FFT_TW_POWERS
#define WB2_FFT_TW_POWERS 1
#endif
#ifndef WB2_FFT_TW_GLOBAL
#define WB2_FFT_TW_GLOBAL 0   // 1: read the pass twiddles from global/L1, not LDS
#endif
#ifndef WB2_FFT_MAX_BLOCKS
#define WB2_FFT_MAX_BLOCKS 2048   // persistent workgroups (4 waves each)
#endif
#ifndef WB2_FFT_FIRST_FROM_GLOBAL
#define WB2_FFT_FIRST_FROM_GLOBAL 1  // 0: stage the row in LDS, then all passes
#endif

namespace wb2 {
namespace fused {

typedef float cf __attribute__((ext_vector_type(2)));   // (re, im)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ cf cmul(cf a, cf b) {
  cf r;
  r.x = a.x * b.x - a.y * b.y;
  r.y = a.x * b.y + a.y * b.x;
  return r;
}
__device__ __forceinline__ cf mul_neg_i(cf a) {  // a * (-i)
  cf r;
  r.x = a.y;
  r.y = -a.x;
  return r;
}

template <int R>
__device__ __forceinline__ void butterfly(cf (&a)[R]);

template <>
__device__ __forceinline__ void butterfly<2>(cf (&a)[2]) {
  const cf t = a[0] - a[1];
  a[0] = a[0] + a[1];
  a[1] = t;
}
template <>
__device__ __forceinline__ void butterfly<4>(cf (&a)[4]) {
  const cf t0 = a[0] + a[2], t1 = a[0] - a[2], t2 = a[1] + a[3];
  const cf t3 = mul_neg_i(a[1] - a[3]);
  a[0] = t0 + t2;
  a[1] = t1 + t3;
  a[2] = t0 - t2;
  a[3] = t1 - t3;
}
template <>
__device__ __forceinline__ void butterfly<3>(cf (&a)[3]) {
  constexpr float c = 0.86602540378443864676f;  // sin(pi/3)
  const cf s = a[1] + a[2], d = a[1] - a[2];
  const cf m = a[0] - 0.5f * s;
  cf n;
  n.x = c * d.y;
  n.y = -c * d.x;
  a[0] = a[0] + s;
  a[1] = m + n;
  a[2] = m - n;
}
template <>
__device__ __forceinline__ void butterfly<5>(cf (&a)[5]) {
  constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
  constexpr float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
  const cf s14 = a[1] + a[4], d14 = a[1] - a[4];
  const cf s23 = a[2] + a[3], d23 = a[2] - a[3];
  const cf m1 = a[0] + c1 * s14 + c2 * s23, m2 = a[0] + c2 * s14 + c1 * s23;
  const cf n1 = mul_neg_i(s1 * d14 + s2 * d23);
  const cf n2 = mul_neg_i(s2 * d14 - s1 * d23);
  a[0] = a[0] + s14 + s23;
  a[1] = m1 + n1;
  a[4] = m1 - n1;
  a[2] = m2 + n2;
  a[3] = m2 - n2;
}

#ifndef WB2_FFT_ODD_FIRST
#define WB2_FFT_ODD_FIRST 0
#endif
#ifndef WB2_FFT_MIN_WAVES
#define WB2_FFT_MIN_WAVES 1
#endif
constexpr int pick_radix(int remaining) {
#if WB2_FFT_ODD_FIRST
  // odd radices first: their strided LDS writes (NS small) are conflict-free
  return remaining % 5 == 0 ? 5 : remaining % 3 == 0 ? 3
       : remaining % 4 == 0 ? 4 : remaining % 2 == 0 ? 2 : 0;
#else
  return remaining % 4 == 0 ? 4 : remaining % 2 == 0 ? 2
       : remaining % 3 == 0 ? 3 : remaining % 5 == 0 ? 5 : 0;
#endif
}
constexpr bool supported_half(int n2) {
  int r = n2;
  while (r > 1) {
    const int p = pick_radix(r);
    if (p == 0) return false;
    r /= p;
  }
  return n2 >= 4;
}

// One Stockham pass of radix R over the wave's slab; NS = product of the radices
// already applied.  twz[j] = exp(-2 pi i j / N2).
template <int N2, int R, int NS>
__device__ __forceinline__ void stockham_pass(cf* __restrict__ z,
                                              const cf* __restrict__ twz,
                                              int lane) {
  constexpr int T = N2 / R;                 // butterflies
  constexpr int ROUNDS = (T + kWave - 1) / kWave;
  constexpr int TWS = N2 / (NS * R);        // twiddle table stride
  cf v[ROUNDS][R];
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
    const int j = lane + rd * kWave;
    if (j < T) {
#pragma unroll
      for (int r = 0; r < R; ++r) v[rd][r] = z[j + r * T];
    }
  }
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
    const int j = lane + rd * kWave;
    if (j < T) {
      const int k = j % NS;
      if constexpr (NS > 1) {
#if WB2_FFT_TW_POWERS
        // one table read per butterfly; higher powers by multiplication
        const cf w1 = twz[k * TWS];
        cf w = w1;
#pragma unroll
        for (int r = 1; r < R; ++r) {
          v[rd][r] = cmul(v[rd][r], w);
          if (r + 1 < R) w = cmul(w, w1);
        }
#else
#pragma unroll
        for (int r = 1; r < R; ++r) v[rd][r] = cmul(v[rd][r], twz[k * r * TWS]);
#endif
      }
      butterfly<R>(v[rd]);
    }
  }
  // every read of this pass precedes every write (program order; DS operations
  // of one wave execute in order) -- the fence only restrains the compiler
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
    const int j = lane + rd * kWave;
    if (j < T) {
      const int k = j % NS;
      const int j0 = (j / NS) * NS * R + k;
#pragma unroll
      for (int t = 0; t < R; ++t) z[j0 + t * NS] = v[rd][t];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

template <int N2, int NS>
__device__ __forceinline__ void stockham_all(cf* z, const cf* twz, int lane) {
  if constexpr (NS < N2) {
    constexpr int R = pick_radix(N2 / NS);
    stockham_pass<N2, R, NS>(z, twz, lane);
    stockham_all<N2, NS * R>(z, twz, lane);
  }
}

struct FusedParams {
  const float* x;
  const cf* twz;   // [N2]      exp(-2 pi i j / N2)
  const cf* twn;   // [N2/2+1]  exp(-2 pi i k / N)
  const double* circ;
  double* out;
  long long n_rows;   // input rows
  long long n_time;   // TIME: input rows are [n_time][n_rows / n_time]
  int n_lat;
  int skipna;
};

// TIME: the mean over the leading time axis is fused (the time mean of
// scripts/compute_zonal_energy_spectrum.py:234): a wave owns one OUTPUT row,
// transforms its n_time input rows in time order with the bin powers summed in
// registers (fp64, NaN spectra skipped with skipna like xbeam.Mean) and stores
// the mean once -- 4 B read per grid point and almost nothing written.
template <int N2, bool TIME>
__global__ void __launch_bounds__(256, WB2_FFT_MIN_WAVES)
    fused_spectrum_kernel(const FusedParams p) {
  constexpr int N = 2 * N2, NB = N2 + 1, NWAVE = 4;
  constexpr int NH = N2 / 2 + 1;  // bin pairs (k, N2 - k), k = 0..N2/2
  constexpr int NIT = (NH + kWave - 1) / kWave;
  __shared__ cf s_twz[N2];
  __shared__ cf s_twn[NH];
  __shared__ __attribute__((aligned(16))) cf s_z[NWAVE][N2];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  for (int i = threadIdx.x; i < N2; i += blockDim.x) s_twz[i] = p.twz[i];
  for (int i = threadIdx.x; i < NH; i += blockDim.x) s_twn[i] = p.twn[i];
  __syncthreads();
  cf* z = s_z[wave];
  const float inv_n = 1.0f / (float)N;
  const long long stride = (long long)gridDim.x * NWAVE;
  const long long nt = TIME ? p.n_time : 1;
  const long long rows_out = p.n_rows / nt;
  for (long long orow_i = (long long)blockIdx.x * NWAVE + wave;
       orow_i < rows_out; orow_i += stride) {
   double sum1[NIT], sum2[NIT];
   int cnt1[NIT], cnt2[NIT];
#pragma unroll
   for (int i = 0; i < NIT; ++i) {
     sum1[i] = sum2[i] = 0.0;
     cnt1[i] = cnt2[i] = 0;
   }
   const double c = p.circ[(unsigned)(orow_i % p.n_lat)];
   double* orow = p.out + orow_i * NB;
   for (long long t = 0; t < nt; ++t) {
    const long long row = t * rows_out + orow_i;
#ifndef WB2_FFT_DIAG
#define WB2_FFT_DIAG 0  // 1: skip the FFT passes, 2: skip the epilogue stores
#endif
#if WB2_FFT_FIRST_FROM_GLOBAL && WB2_FFT_DIAG != 1 && !WB2_FFT_TW_GLOBAL
    // ---- first pass straight from HBM: every lane fetches the R inputs of its
    // butterflies itself (8-byte loads, consecutive lanes = consecutive complex
    // points), transforms them and writes the R outputs as ONE contiguous run:
    // no staging copy of the row in LDS, no strided (bank-conflicting) writes of
    // the NS = 1 pass.
    {
      constexpr int R = pick_radix(N2);
      constexpr int T = N2 / R;
      constexpr int ROUNDS = (T + kWave - 1) / kWave;
      const cf* src = reinterpret_cast<const cf*>(p.x + row * N);
      cf v[ROUNDS][R];
#pragma unroll
      for (int rd = 0; rd < ROUNDS; ++rd) {
        const int j = lane + rd * kWave;
        if (j < T) {
#pragma unroll
          for (int r = 0; r < R; ++r)
            v[rd][r] = __builtin_nontemporal_load(src + j + r * T);
        }
      }
#pragma unroll
      for (int rd = 0; rd < ROUNDS; ++rd) {
        const int j = lane + rd * kWave;
        if (j < T) {
          butterfly<R>(v[rd]);
          if constexpr (R % 2 == 0) {
            f4* dst = reinterpret_cast<f4*>(z + j * R);
#pragma unroll
            for (int h = 0; h < R / 2; ++h) {
              f4 w;
              w.x = v[rd][2 * h].x;
              w.y = v[rd][2 * h].y;
              w.z = v[rd][2 * h + 1].x;
              w.w = v[rd][2 * h + 1].y;
              dst[h] = w;
            }
          } else {
#pragma unroll
            for (int r = 0; r < R; ++r) z[j * R + r] = v[rd][r];
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      stockham_all<N2, R>(z, s_twz, lane);
    }
#else
    // ---- load: 16-byte nontemporal loads, two complex points per lane ----
    const f4* src = reinterpret_cast<const f4*>(p.x + row * N);
    f4* zq = reinterpret_cast<f4*>(z);
#pragma unroll
    for (int i = 0; i < (N2 / 2 + kWave - 1) / kWave; ++i) {
      const int q = lane + i * kWave;
      if (q < N2 / 2) zq[q] = __builtin_nontemporal_load(src + q);
    }
    if constexpr (N2 % 2 == 1) {  // odd N2: last complex point
      if (lane == 0) {
        cf last;
        last.x = p.x[row * N + N - 2];
        last.y = p.x[row * N + N - 1];
        z[N2 - 1] = last;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#if WB2_FFT_DIAG != 1
#if WB2_FFT_TW_GLOBAL
    stockham_all<N2, 1>(z, p.twz, lane);
#else
    stockham_all<N2, 1>(z, s_twz, lane);
#endif
#endif
#endif
    // ---- recombination + power for the bin pairs (k, N2 - k) ----
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int k = lane + i * kWave;
      if (k < NH) {
        const cf a = z[k];
        const cf b = z[k == 0 ? 0 : N2 - k];
        cf e, o;  // E = (a + conj b) / 2,  O = (a - conj b) / (2i)
        e.x = (a.x + b.x) * 0.5f;
        e.y = (a.y - b.y) * 0.5f;
        o.x = (a.y + b.y) * 0.5f;
        o.y = (b.x - a.x) * 0.5f;
        const cf wo = cmul(s_twn[k], o);
        const cf x1 = (e + wo) * inv_n;   // bin k        (norm='forward')
        const cf x2 = (e - wo) * inv_n;   // bin N2 - k   (conjugate: same power)
        const float p1 = x1.x * x1.x + x1.y * x1.y;
        const float p2 = x2.x * x2.x + x2.y * x2.y;
        // derived_variables.py:600: every bin but 0 is doubled (Nyquist too)
        const double v1 = ((double)p1 * (k == 0 ? 1.0 : 2.0)) * c;
        const double v2 = ((double)p2 * 2.0) * c;
        if constexpr (TIME) {
          const bool k1 = !(p.skipna && is_nan(v1));
          const bool k2 = !(p.skipna && is_nan(v2));
          sum1[i] += k1 ? v1 : 0.0;
          sum2[i] += k2 ? v2 : 0.0;
          cnt1[i] += k1 ? 1 : 0;
          cnt2[i] += k2 ? 1 : 0;
        } else {
#if WB2_FFT_DIAG == 2
          if (p1 + p2 == 1.2345f) orow[k] = p1;
#else
          __builtin_nontemporal_store(v1, orow + k);
          if (2 * k != N2) __builtin_nontemporal_store(v2, orow + N2 - k);
#endif
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
   }  // time
   if constexpr (TIME) {
#pragma unroll
     for (int i = 0; i < NIT; ++i) {
       const int k = lane + i * kWave;
       if (k < NH) {
         __builtin_nontemporal_store(sum1[i] / (double)cnt1[i], orow + k);
         if (2 * k != N2)
           __builtin_nontemporal_store(sum2[i] / (double)cnt2[i],
                                       orow + N2 - k);
       }
     }
   }
  }
}

__global__ void fused_twiddle_kernel(cf* twz, cf* twn, int n2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double sn, cs;
  if (i < n2) {
    sincospi(2.0 * (double)i / (double)n2, &sn, &cs);
    twz[i].x = (float)cs;
    twz[i].y = (float)(-sn);
  }
  if (i <= n2 / 2) {
    sincospi((double)i / (double)n2, &sn, &cs);
    twn[i].x = (float)cs;
    twn[i].y = (float)(-sn);
  }
}

template <int N2>
int launch(const FusedParams& p, hipStream_t s) {
  const long long rows_out = p.n_time > 0 ? p.n_rows / p.n_time : p.n_rows;
  long long blocks = (rows_out + 3) / 4;
  if (blocks > WB2_FFT_MAX_BLOCKS) blocks = WB2_FFT_MAX_BLOCKS;  // row-strided waves beyond that
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

int fused_spectrum_run(const float* x, long long n_rows, int n_lon,
                       const double* circ, int n_lat, long long n_time,
                       int skipna, double* out, void* tables, hipStream_t s) {
  using namespace fused;
  const int n2 = n_lon / 2;
  cf* twz = static_cast<cf*>(tables);  // filled once by fused_spectrum_tables
  cf* twn = twz + n2;
  FusedParams p{x, twz, twn, circ, out, n_rows, n_time, n_lat, skipna};
  switch (n2) {
#define WB2_CASE(N2) case N2: return launch<N2>(p, s);
    WB2_CASE(32) WB2_CASE(64) WB2_CASE(120) WB2_CASE(128) WB2_CASE(180)
    WB2_CASE(256) WB2_CASE(360) WB2_CASE(512) WB2_CASE(720)
#undef WB2_CASE
  }
  return fail("fused spectrum: n_lon=%d is not instantiated", n_lon);
}

}  // namespace wb2
