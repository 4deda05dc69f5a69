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
// The batched real-to-complex FFT along longitude is rocFFT's (through hipFFT,
// as BASELINE.json's north_star prescribes); the hand-written epilogue fuses
// the 1/N normalisation, |.|^2, the x2 of the non-zero wavenumbers, the
// circumference scale and (optionally) the deterministic time mean, and widens
// to fp64 exactly where numpy does (float32 power * int64 -> float64).

#include "common.hpp"
#include "wb2hip.h"

#include <hipfft/hipfft.h>

namespace wb2 {
namespace {

struct SpectrumPlan {
  hipfftHandle fft = 0;
  int dtype = 0;
  int n_lon = 0;
  long long n_rows = 0;
  size_t complex_bytes = 0;  // [n_rows][n_lon/2+1] complex
  size_t fft_work_bytes = 0;
};

constexpr size_t kAlign = 256;
inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

// One thread per (output row, bin).  With n_time > 0 the thread walks the
// n_time input rows that map onto its output row in time order (deterministic
// mean); skipna drops NaN spectra like xbeam.Mean(skipna=True).
template <typename T, typename C>
__global__ void __launch_bounds__(256)
    power_kernel(const C* __restrict__ spec, const double* __restrict__ circ,
                 int n_lat, int n_bins, long long rows_out, long long n_time,
                 T inv_n, int skipna, double* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows_out * n_bins) return;
  const long long r = idx / n_bins;
  const int k = (int)(idx - r * n_bins);
  const double c = circ[r % n_lat];
  const double mult = k == 0 ? 1.0 : 2.0;  // derived_variables.py:600
  const long long nt = n_time > 0 ? n_time : 1;
  double sum = 0.0, cnt = 0.0;
  for (long long t = 0; t < nt; ++t) {
    const C f = spec[(t * rows_out + r) * n_bins + k];
    // norm='forward': pocketfft scales the transform by 1/N in the input dtype
    const T re = f.x * inv_n, im = f.y * inv_n;
    const T p = re * re + im * im;  // real(f * conj(f)) in the input dtype
    const double v = ((double)p * mult) * c;
    if (skipna && is_nan(v)) continue;
    sum += v;
    cnt += 1.0;
  }
  out[idx] = n_time > 0 ? sum / cnt : sum;
}

}  // namespace
}  // namespace wb2

extern "C" {

int wb2_spectrum_plan_create(int dtype, int32_t n_lon, int64_t n_rows,
                             void** plan_out) {
  using namespace wb2;
  WB2_REQUIRE(plan_out, "null plan_out");
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_REQUIRE(n_lon >= 2 && n_rows > 0 && n_rows < (1ll << 31), "bad sizes");
  auto* p = new SpectrumPlan();
  p->dtype = dtype;
  p->n_lon = n_lon;
  p->n_rows = n_rows;
  const int n_bins = n_lon / 2 + 1;
  p->complex_bytes =
      (size_t)n_rows * n_bins * (dtype == WB2_F32 ? 8 : 16);
  int n[1] = {n_lon};
  hipfftResult rc = hipfftCreate(&p->fft);
  if (rc == HIPFFT_SUCCESS) rc = hipfftSetAutoAllocation(p->fft, 0);
  size_t work = 0;
  if (rc == HIPFFT_SUCCESS)
    rc = hipfftMakePlanMany(p->fft, 1, n, nullptr, 1, n_lon, nullptr, 1, n_bins,
                            dtype == WB2_F32 ? HIPFFT_R2C : HIPFFT_D2Z,
                            (int)n_rows, &work);
  if (rc != HIPFFT_SUCCESS) {
    if (p->fft) hipfftDestroy(p->fft);
    delete p;
    return fail("hipFFT plan creation failed (hipfftResult %d)", (int)rc);
  }
  p->fft_work_bytes = work;
  *plan_out = p;
  return 0;
}

int wb2_spectrum_plan_destroy(void* plan) {
  auto* p = static_cast<wb2::SpectrumPlan*>(plan);
  if (!p) return 0;
  hipfftDestroy(p->fft);
  delete p;
  return 0;
}

int64_t wb2_spectrum_plan_workspace(void* plan) {
  auto* p = static_cast<wb2::SpectrumPlan*>(plan);
  if (!p) return wb2::fail("null plan");
  return (int64_t)(wb2::align_up(p->complex_bytes) +
                   wb2::align_up(p->fft_work_bytes));
}

int wb2_zonal_spectrum(void* plan, const void* x, const double* circumference,
                       int32_t n_lat, int64_t n_time, int skipna, double* out,
                       void* workspace, void* stream) {
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
  void* spec = ws;
  void* fft_work = ws + align_up(p->complex_bytes);
  hipfftResult rc = hipfftSetStream(p->fft, s);
  if (rc == HIPFFT_SUCCESS && p->fft_work_bytes)
    rc = hipfftSetWorkArea(p->fft, fft_work);
  if (rc == HIPFFT_SUCCESS) {
    rc = p->dtype == WB2_F32
             ? hipfftExecR2C(p->fft, (hipfftReal*)x, (hipfftComplex*)spec)
             : hipfftExecD2Z(p->fft, (hipfftDoubleReal*)x,
                             (hipfftDoubleComplex*)spec);
  }
  if (rc != HIPFFT_SUCCESS) return fail("hipFFT exec failed (%d)", (int)rc);
  const int n_bins = p->n_lon / 2 + 1;
  const long long total = rows_out * n_bins;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (p->dtype == WB2_F32) {
    hipLaunchKernelGGL((power_kernel<float, float2>), dim3(blocks), dim3(256),
                       0, s, (const float2*)spec, circumference, n_lat, n_bins,
                       rows_out, (long long)n_time, 1.0f / (float)p->n_lon,
                       skipna, out);
  } else {
    hipLaunchKernelGGL((power_kernel<double, double2>), dim3(blocks),
                       dim3(256), 0, s, (const double2*)spec, circumference,
                       n_lat, n_bins, rows_out, (long long)n_time,
                       1.0 / (double)p->n_lon, skipna, out);
  }
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
