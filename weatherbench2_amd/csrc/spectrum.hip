// K4 of libwb2hip.so: zonal energy spectrum.
//
// Replaces ZonalEnergySpectrum.compute (reference =
// /root/reference/weatherbench2/derived_variables.py:592-626):
//   f_k = np.fft.rfft(f_x, axis=-1, norm='forward')            :597
//   power = real(f_k * conj(f_k)) * [1, 2, 2, ..., 2]          :598-602
//           (the LAST bin is doubled too, even for even N -- kept)
//   spectrum = power * circumference(latitude)                 :578-581, 626
// and, optionally, the time mean of scripts/compute_zonal_energy_spectrum.py:234.
//
// The batched FFT along longitude is rocFFT's (through hipFFT, as
// BASELINE.json's north_star prescribes).  For even n_lon the row is handed to
// rocFFT as n_lon/2 COMPLEX points (z[m] = x[2m] + i x[2m+1]) and the
// hand-written epilogue does the real-FFT recombination itself,
//   X[k] = E[k] + W^k O[k],  E = (Z[k] + conj Z[N/2-k]) / 2,
//                            O = (Z[k] - conj Z[N/2-k]) / 2i,  W = exp(-2 pi i / N),
// fused with the 1/N normalisation, |.|^2, the x2 of the non-zero wavenumbers,
// the circumference scale and (optionally) the deterministic time mean.  That
// removes rocFFT's separate r2c post-processing pass over the data (odd n_lon
// falls back to rocFFT's R2C).  fp64 appears exactly where numpy widens
// (float32 power * int64 -> float64).

#include "common.hpp"
#include "trace.hpp"
#include "wb2hip.h"

#include <hipfft/hipfft.h>

#include <mutex>

#include <cstdlib>
#include <cstring>

namespace wb2 {

// spectrum_fused.hip: single-kernel path (LDS FFT + fused epilogue)
bool fused_spectrum_supported(int dtype, int n_lon);
size_t fused_spectrum_table_bytes(int dtype, int n_lon);
int fused_spectrum_tables(void* tables, int dtype, int n_lon, hipStream_t s);
int fused_spectrum_run(const void* x, int dtype, long long n_rows, int n_lon,
                       const double* circ, int n_lat, long long n_time,
                       int skipna, double* out, void* tables, hipStream_t s);
int fused_spectrum_latmean_segments(int dtype, long long n_rows, int n_lon,
                                    int n_lat);
int fused_spectrum_latmean(const void* x, int dtype, long long n_rows,
                           int n_lon, const double* row_weight, int n_lat,
                           int n_seg, double scale, double* partial,
                           double* out, void* tables, hipStream_t s);

namespace {

struct SpectrumPlan {
  hipfftHandle fft = 0;
  int dtype = 0;
  int n_lon = 0;
  long long n_rows = 0;
  size_t complex_bytes = 0;  // [n_rows][n_lon/2 (+1)] complex
  size_t fft_work_bytes = 0;
  bool packed = false;       // even n_lon: C2C on n_lon/2 points + own recombination
  bool fused = false;        // spectrum_fused.hip handles (dtype, n_lon)
  void* tables = nullptr;    // fused path: twiddle tables, owned by the plan
  // A plan of a length the one-kernel transform handles makes its hipFFT plan
  // only when a call cannot take that kernel (x not 16-byte aligned, n_time >=
  // 65536): rocFFT compiles its kernels at plan creation -- seconds per (length,
  // batch) that almost no call needs.  Such a late plan owns its buffers
  // (`late_spec`, hipFFT's own work area) instead of using the caller's workspace.
  bool late = false;
  void* late_spec = nullptr;
  std::mutex mu;
};

// native 2-vectors (re, im): accepted by the nontemporal builtins
typedef float cf32 __attribute__((ext_vector_type(2)));
typedef double cf64 __attribute__((ext_vector_type(2)));

constexpr size_t kAlign = 256;
inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

// One thread per (output row, bin).  With n_time > 0 the thread walks the
// n_time input rows that map onto its output row in time order (deterministic
// mean); skipna drops NaN spectra like xbeam.Mean(skipna=True).
// PACKED: `spec` holds Z = FFT_{N/2}(x[2m] + i x[2m+1]) (n_half complex per row)
// and the real-FFT recombination happens here; otherwise `spec` already is the
// one-sided spectrum (n_bins complex per row).
// W^k = exp(-i pi k / n_half), k = 0..n_half, evaluated in fp64, rounded once.
template <typename C>
__global__ void twiddle_kernel(C* __restrict__ tw, int n_half) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > n_half) return;
  double sn, cs;
  sincospi((double)k / (double)n_half, &sn, &cs);
  tw[k].x = cs;
  tw[k].y = -sn;
}

// Power of one bin from the transform values (see file header).
template <typename T, typename C, bool PACKED>
__device__ __forceinline__ double bin_power(C a, C b, C w, T inv_n, double mult,
                                            double c) {
  T re, im;
  if constexpr (PACKED) {
    // E = (a + conj b) / 2,  O = (a - conj b) / (2i),  X = E + W^k O
    const T er = (a.x + b.x) * (T)0.5, ei = (a.y - b.y) * (T)0.5;
    const T orr = (a.y + b.y) * (T)0.5, oi = (b.x - a.x) * (T)0.5;
    re = (er + (w.x * orr - w.y * oi)) * inv_n;
    im = (ei + (w.x * oi + w.y * orr)) * inv_n;
  } else {
    // norm='forward': the transform is scaled by 1/N in the input dtype
    re = a.x * inv_n;
    im = a.y * inv_n;
  }
  const T p = re * re + im * im;  // real(f * conj(f)) in the input dtype
  return ((double)p * mult) * c;
}

template <typename T, typename C, bool PACKED>
__global__ void __launch_bounds__(256)
    power_kernel(const C* __restrict__ spec, const C* __restrict__ tw,
                 const double* __restrict__ circ, int n_lat, int n_bins,
                 long long rows_out, long long n_time, T inv_n, int skipna,
                 double* __restrict__ out) {
  // One workgroup per output row; a thread owns the bin PAIRS 2j, 2j+1 for
  // j = tid, tid + 256, ... (16-byte loads of the transform, 16-byte stores of
  // the powers); the row index, its circumference and all row pointers are
  // wave-uniform.
  typedef T C2 __attribute__((ext_vector_type(4)));        // two complex values
  typedef double D2 __attribute__((ext_vector_type(2)));
  const long long nt = n_time > 0 ? n_time : 1;
  const int n_half = n_bins - 1;            // PACKED: N/2 complex points per row
  const int row_len = PACKED ? n_half : n_bins;
  const long long r = (long long)blockIdx.y * gridDim.x + blockIdx.x;
  if (r >= rows_out) return;
  const double c = circ[(unsigned)(r % n_lat)];
  double* orow = out + r * n_bins;
  const bool vec_ok = (row_len % 2 == 0) &&
                      (reinterpret_cast<uintptr_t>(spec) % sizeof(C2) == 0);
  const int n_pair = vec_ok ? n_bins / 2 : 0;
  for (int j = threadIdx.x; j < n_pair; j += blockDim.x) {
    const int k = 2 * j;
    C w0 = {1, 0}, w1 = {1, 0};
    if constexpr (PACKED) {
      w0 = tw[k];
      w1 = tw[k + 1];
    }
    double s0 = 0.0, s1 = 0.0, n0 = 0.0, n1 = 0.0;
    for (long long t = 0; t < nt; ++t) {
      const C* row = spec + (t * rows_out + r) * row_len;
      const C2 a2 = __builtin_nontemporal_load(
          reinterpret_cast<const C2*>(row + k));
      const C a0 = {a2.x, a2.y}, a1 = {a2.z, a2.w};
      C b0 = a0, b1 = a1;
      if constexpr (PACKED) {
        b0 = row[k == 0 ? 0 : n_half - k];
        b1 = row[n_half - k - 1];
      }
      const double v0 = bin_power<T, C, PACKED>(a0, b0, w0, inv_n,
                                                k == 0 ? 1.0 : 2.0, c);
      const double v1 = bin_power<T, C, PACKED>(a1, b1, w1, inv_n, 2.0, c);
      const bool k0 = !(skipna && is_nan(v0)), k1 = !(skipna && is_nan(v1));
      s0 += k0 ? v0 : 0.0;
      n0 += k0 ? 1.0 : 0.0;
      s1 += k1 ? v1 : 0.0;
      n1 += k1 ? 1.0 : 0.0;
    }
    D2 o;
    o.x = n_time > 0 ? s0 / n0 : s0;
    o.y = n_time > 0 ? s1 / n1 : s1;
    // rows are 8-byte aligned only (n_bins is odd for even n_lon)
    __builtin_nontemporal_store(o.x, orow + k);
    __builtin_nontemporal_store(o.y, orow + k + 1);
  }
  // leftover bins (the last one when n_bins is odd; all of them if !vec_ok)
  for (int k = 2 * n_pair + threadIdx.x; k < n_bins; k += blockDim.x) {
    C w = {1, 0};
    if constexpr (PACKED) w = tw[k];
    double sum = 0.0, cnt = 0.0;
    for (long long t = 0; t < nt; ++t) {
      const C* row = spec + (t * rows_out + r) * row_len;
      const C a = row[PACKED && k == n_half ? 0 : k];
      C b = a;
      if constexpr (PACKED) b = row[k == 0 ? 0 : n_half - k];
      const double v = bin_power<T, C, PACKED>(a, b, w, inv_n,
                                               k == 0 ? 1.0 : 2.0, c);
      const bool keep = !(skipna && is_nan(v));
      sum += keep ? v : 0.0;
      cnt += keep ? 1.0 : 0.0;
    }
    orow[k] = n_time > 0 ? sum / cnt : sum;
  }
}

// The hipFFT plan of `p` (batched 1-D transform of its rows).  own_work_area:
// hipFFT allocates its work area itself (late plans); otherwise it is part of
// the caller's workspace and its size goes to p->fft_work_bytes.
int make_fft_plan(SpectrumPlan* p, bool own_work_area) {
  const int n_bins = p->n_lon / 2 + 1;
  int n[1] = {p->packed ? p->n_lon / 2 : p->n_lon};
  hipfftResult rc = hipfftCreate(&p->fft);
  if (rc == HIPFFT_SUCCESS)
    rc = hipfftSetAutoAllocation(p->fft, own_work_area ? 1 : 0);
  size_t work = 0;
  if (rc == HIPFFT_SUCCESS) {
    if (p->packed) {
      rc = hipfftMakePlanMany(p->fft, 1, n, nullptr, 1, n[0], nullptr, 1, n[0],
                              p->dtype == WB2_F32 ? HIPFFT_C2C : HIPFFT_Z2Z,
                              (int)p->n_rows, &work);
    } else {
      rc = hipfftMakePlanMany(p->fft, 1, n, nullptr, 1, p->n_lon, nullptr, 1,
                              n_bins,
                              p->dtype == WB2_F32 ? HIPFFT_R2C : HIPFFT_D2Z,
                              (int)p->n_rows, &work);
    }
  }
  if (rc != HIPFFT_SUCCESS) {
    if (p->fft) hipfftDestroy(p->fft);
    p->fft = 0;
    return fail("hipFFT plan creation failed (hipfftResult %d)", (int)rc);
  }
  p->fft_work_bytes = own_work_area ? 0 : work;
  return 0;
}

}  // namespace
}  // namespace wb2

extern "C" {

int wb2_spectrum_plan_create(int dtype, int32_t n_lon, int64_t n_rows,
                             void** plan_out) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(plan_out, "null plan_out");
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_REQUIRE(n_lon >= 2 && n_rows > 0 && n_rows < (1ll << 31), "bad sizes");
  auto* p = new SpectrumPlan();
  p->dtype = dtype;
  p->n_lon = n_lon;
  p->n_rows = n_rows;
  const int n_bins = n_lon / 2 + 1;
  p->packed = (n_lon % 2 == 0) && n_lon >= 4;
  const int row_len = p->packed ? n_lon / 2 : n_bins;
  p->complex_bytes =
      (size_t)n_rows * row_len * (dtype == WB2_F32 ? 8 : 16);
  // WB2HIP_SPECTRUM_BACKEND=rocfft forces the two-kernel path (tests, A/B runs)
  const char* backend = std::getenv("WB2HIP_SPECTRUM_BACKEND");
  p->fused = fused_spectrum_supported(dtype, n_lon) &&
             !(backend && std::strcmp(backend, "rocfft") == 0);
  p->late = p->fused;
  if (!p->late) {
    const int rc = make_fft_plan(p, /*own_work_area=*/false);
    if (rc != 0) {
      delete p;
      return rc;
    }
  }
  if (p->fused) {
    // the twiddle tables are generated once, here (device of the calling
    // thread), instead of by an extra kernel in front of every transform
    hipError_t e =
        hipMalloc(&p->tables, fused_spectrum_table_bytes(dtype, n_lon));
    if (e == hipSuccess &&
        fused_spectrum_tables(p->tables, dtype, n_lon, nullptr) != 0)
      e = hipErrorUnknown;
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
      if (p->tables) (void)hipFree(p->tables);
      if (p->fft) hipfftDestroy(p->fft);
      delete p;
      return fail("twiddle table setup failed: %s", hipGetErrorString(e));
    }
  }
  *plan_out = p;
  return 0;
}

int wb2_spectrum_plan_destroy(void* plan) {
  WB2_TRACE();
  auto* p = static_cast<wb2::SpectrumPlan*>(plan);
  if (!p) return 0;
  if (p->fft) hipfftDestroy(p->fft);
  if (p->tables) (void)hipFree(p->tables);
  if (p->late_spec) (void)hipFree(p->late_spec);
  delete p;
  return 0;
}

int64_t wb2_spectrum_plan_workspace(void* plan) {
  auto* p = static_cast<wb2::SpectrumPlan*>(plan);
  if (!p) return wb2::fail("null plan");
  size_t tw = (size_t)(p->n_lon / 2 + 1) * (p->dtype == WB2_F32 ? 8 : 16);
  if (p->fused && wb2::fused_spectrum_table_bytes(p->dtype, p->n_lon) > tw)
    tw = wb2::fused_spectrum_table_bytes(p->dtype, p->n_lon);
  // (a late hipFFT plan brings its own transform buffer and work area)
  const size_t spec = p->late ? 0 : p->complex_bytes;
  return (int64_t)(wb2::align_up(spec) + wb2::align_up(p->fft_work_bytes) +
                   wb2::align_up(tw));
}

int wb2_zonal_spectrum(void* plan, const void* x, const double* circumference,
                       int32_t n_lat, int64_t n_time, int skipna, double* out,
                       void* workspace, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  auto* p = static_cast<SpectrumPlan*>(plan);
  WB2_REQUIRE(p && x && circumference && out && workspace,
              "null pointer argument");
  WB2_REQUIRE(n_lat > 0 && p->n_rows % n_lat == 0,
              "n_rows=%lld is not a multiple of n_lat=%d", p->n_rows, n_lat);
  WB2_REQUIRE(n_time >= 0 && (n_time == 0 || p->n_rows % n_time == 0),
              "n_rows=%lld is not a multiple of n_time=%lld", p->n_rows,
              (long long)n_time);
  const long long rows_out = n_time > 0 ? p->n_rows / n_time : p->n_rows;
  WB2_REQUIRE(rows_out % n_lat == 0, "rows per time step not a multiple of n_lat");
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  // (the fused time mean keeps its per-bin sample counts in 16 bits)
  if (p->fused && reinterpret_cast<uintptr_t>(x) % 16 == 0 && n_time < 65536)
    return fused_spectrum_run(x, p->dtype, p->n_rows, p->n_lon, circumference,
                              n_lat, n_time, skipna, out, p->tables, s);
  void* spec = ws;
  if (p->late) {   // the first call that needs hipFFT after all makes its plan
    std::lock_guard<std::mutex> lock(p->mu);
    if (!p->fft) {
      const int made = make_fft_plan(p, /*own_work_area=*/true);
      if (made != 0) return made;
    }
    if (!p->late_spec)
      WB2_HIP_OK(hipMalloc(&p->late_spec, p->complex_bytes));
    spec = p->late_spec;
  }
  const size_t spec_in_ws = p->late ? 0 : p->complex_bytes;
  void* fft_work = ws + align_up(spec_in_ws);
  void* tw = ws + align_up(spec_in_ws) + align_up(p->fft_work_bytes);
  hipfftResult rc = hipfftSetStream(p->fft, s);
  if (rc == HIPFFT_SUCCESS && p->fft_work_bytes && !p->late)
    rc = hipfftSetWorkArea(p->fft, fft_work);
  if (rc == HIPFFT_SUCCESS) {
    if (p->packed) {
      rc = p->dtype == WB2_F32
               ? hipfftExecC2C(p->fft, (hipfftComplex*)x, (hipfftComplex*)spec,
                               HIPFFT_FORWARD)
               : hipfftExecZ2Z(p->fft, (hipfftDoubleComplex*)x,
                               (hipfftDoubleComplex*)spec, HIPFFT_FORWARD);
    } else {
      rc = p->dtype == WB2_F32
               ? hipfftExecR2C(p->fft, (hipfftReal*)x, (hipfftComplex*)spec)
               : hipfftExecD2Z(p->fft, (hipfftDoubleReal*)x,
                               (hipfftDoubleComplex*)spec);
    }
  }
  if (rc != HIPFFT_SUCCESS) return fail("hipFFT exec failed (%d)", (int)rc);
  const int n_bins = p->n_lon / 2 + 1;
  const unsigned gx = (unsigned)(rows_out < 65536 ? rows_out : 65536);
  const dim3 blocks(gx, (unsigned)((rows_out + gx - 1) / gx));
  if (p->packed) {  // 721 twiddles: regenerated per call, nothing is cached
    const unsigned tb = (unsigned)((n_bins + 255) / 256);
    if (p->dtype == WB2_F32)
      hipLaunchKernelGGL((twiddle_kernel<cf32>), dim3(tb), dim3(256), 0, s,
                         (cf32*)tw, n_bins - 1);
    else
      hipLaunchKernelGGL((twiddle_kernel<cf64>), dim3(tb), dim3(256), 0, s,
                         (cf64*)tw, n_bins - 1);
  }
#define WB2_POWER(T, C, PK)                                                   \
  hipLaunchKernelGGL((power_kernel<T, C, PK>), blocks, dim3(256), 0, s,       \
                     (const C*)spec, (const C*)tw, circumference, n_lat,      \
                     n_bins, rows_out, (long long)n_time, (T)1 / (T)p->n_lon, \
                     skipna, out)
  if (p->dtype == WB2_F32) {
    if (p->packed) WB2_POWER(float, cf32, true);
    else WB2_POWER(float, cf32, false);
  } else {
    if (p->packed) WB2_POWER(double, cf64, true);
    else WB2_POWER(double, cf64, false);
  }
#undef WB2_POWER
  WB2_HIP_OK(hipGetLastError());
  return 0;
}


int wb2_zonal_spectrum_latmean_segments(void* plan, int32_t n_lat) {
  using namespace wb2;
  auto* p = static_cast<SpectrumPlan*>(plan);
  WB2_REQUIRE(p && n_lat > 0 && p->n_rows % n_lat == 0, "bad plan / n_lat");
  if (!p->fused) return 0;  // 0: no fused latitude mean for this plan
  return fused_spectrum_latmean_segments(p->dtype, p->n_rows, p->n_lon, n_lat);
}

int wb2_zonal_spectrum_latmean(void* plan, const void* x,
                               const double* row_weight, int32_t n_lat,
                               int32_t n_seg, double scale, double* partial,
                               double* out, void* stream) {
  using namespace wb2;
  WB2_TRACE();
  auto* p = static_cast<SpectrumPlan*>(plan);
  WB2_REQUIRE(p && x && row_weight && partial && out, "null pointer argument");
  WB2_REQUIRE(n_lat > 0 && p->n_rows % n_lat == 0,
              "n_rows=%lld is not a multiple of n_lat=%d", p->n_rows, n_lat);
  WB2_REQUIRE(n_seg >= 1 && n_seg <= n_lat, "n_seg=%d outside [1, n_lat]", n_seg);
  WB2_REQUIRE(p->fused && reinterpret_cast<uintptr_t>(x) % 16 == 0,
              "the fused latitude mean needs rows of an instantiated "
              "length, 16-byte aligned (materialise with wb2_zonal_spectrum and "
              "reduce with wb2_axis_moments otherwise)");
  return fused_spectrum_latmean(x, p->dtype, p->n_rows, p->n_lon, row_weight,
                                n_lat, n_seg, scale, partial, out, p->tables,
                                static_cast<hipStream_t>(stream));
}

}  // extern "C"
