/*
 * A pure-C caller of libwb2hip.so: no Python, no torch -- only the HIP runtime
 * for device memory and the entry points declared in include/wb2hip.h.
 * It evaluates latitude-weighted MSE / RMSE / MAE / Bias / ACC of a tiny
 * (2 slabs of 7 lat x 12 lon) float32 case for two regions (global and
 * |lat| >= 20) and compares with a straightforward double-precision loop; then
 * the latitude weights, the running temporal mean and its RCCL all-reduce (a
 * one-rank communicator built here) -- the whole path without Python.
 *
 *   hipcc -x c -I include tests/c_abi/c_abi_example.c -L weatherbench2_amd -lwb2hip -o c_abi_example
 *   (run with LD_LIBRARY_PATH=weatherbench2_amd on a box with a GPU)
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "wb2hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_WB2(x) do { if ((x) != 0) { \
  fprintf(stderr, "%s: %s\n", #x, wb2_last_error()); return 3; } } while (0)

enum { N_OUTER = 2, N_LAT = 7, N_LON = 12, N_PT = N_LAT * N_LON, N_REGION = 2 };

static void* to_device(const void* host, size_t bytes) {
  void* d = NULL;
  if (hipMalloc(&d, bytes) != hipSuccess) return NULL;
  if (hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

int main(void) {
  /* inputs */
  static float f[N_OUTER * N_PT], t[N_OUTER * N_PT], c[N_OUTER * N_PT];
  unsigned s = 12345u;
  for (int i = 0; i < N_OUTER * N_PT; ++i) {
    s = s * 1664525u + 1013904223u; f[i] = (float)(s >> 8) / 8388608.0f - 1.0f;
    s = s * 1664525u + 1013904223u; t[i] = (float)(s >> 8) / 8388608.0f - 1.0f;
    s = s * 1664525u + 1013904223u; c[i] = (float)(s >> 8) / 16777216.0f;
  }
  /* latitude weights, metrics.py:40-60, for lat = -90, -60, ..., 90 */
  double lat[N_LAT], w[N_LAT], bounds[N_LAT + 1], mean = 0.0;
  const double pi = 3.14159265358979323846;
  for (int i = 0; i < N_LAT; ++i) lat[i] = (-90.0 + 30.0 * i) * pi / 180.0;
  bounds[0] = -pi / 2; bounds[N_LAT] = pi / 2;
  for (int i = 1; i < N_LAT; ++i) bounds[i] = 0.5 * (lat[i - 1] + lat[i]);
  for (int i = 0; i < N_LAT; ++i) { w[i] = sin(bounds[i + 1]) - sin(bounds[i]); mean += w[i] / N_LAT; }
  for (int i = 0; i < N_LAT; ++i) w[i] /= mean;

  /* plan for regions {global, |lat| >= 20}: rows -90,-60,-30 | 0 | 30,60,90
   * -> bands [0,3) [3,4) [4,7); one seg; one chunk per band, padded to 8 */
  const int32_t chunk_row0[8] = {0, 3, 4, 0, 0, 0, 0, 0};
  const int32_t chunk_nrow[8] = {3, 1, 3, 0, 0, 0, 0, 0};
  const int32_t band_chunk0[4] = {0, 1, 2, 3};
  const int32_t seg_col0[2] = {0, N_LON};
  const double coef_band[N_REGION * 3] = {1, 1, 1, /* global */ 1, 0, 1 /* extra-tropics */};
  const double coef_seg[N_REGION] = {1, 1};
  const int32_t region_wf[N_REGION] = {0, 0};
  double wsum[N_REGION] = {0, 0};
  for (int i = 0; i < N_LAT; ++i) {
    wsum[0] += w[i] * N_LON;
    if (fabs(-90.0 + 30.0 * i) >= 20.0) wsum[1] += w[i] * N_LON;
  }
  const int tile = wb2_tile_cols(WB2_F32, N_LON, 1);
  const int n_ctile = (N_LON + tile - 1) / tile;
  const int32_t seg_eoff[2] = {0, n_ctile};
  const int n_ts = n_ctile, n_chunk = 8;
  const int K = wb2_num_slots(WB2_MODE_DET_ACC, 0);

  void *df = to_device(f, sizeof f), *dt = to_device(t, sizeof t), *dc = to_device(c, sizeof c);
  double* dw = (double*)to_device(w, sizeof w);
  int32_t* d_row0 = (int32_t*)to_device(chunk_row0, sizeof chunk_row0);
  int32_t* d_nrow = (int32_t*)to_device(chunk_nrow, sizeof chunk_nrow);
  int32_t* d_seg = (int32_t*)to_device(seg_col0, sizeof seg_col0);
  int32_t* d_eoff = (int32_t*)to_device(seg_eoff, sizeof seg_eoff);
  int32_t* d_band = (int32_t*)to_device(band_chunk0, sizeof band_chunk0);
  double* d_cb = (double*)to_device(coef_band, sizeof coef_band);
  double* d_cs = (double*)to_device(coef_seg, sizeof coef_seg);
  int32_t* d_wf = (int32_t*)to_device(region_wf, sizeof region_wf);
  double* d_ws = (double*)to_device(wsum, sizeof wsum);
  if (!df || !dt || !dc || !dw || !d_row0 || !d_nrow || !d_seg || !d_eoff || !d_band ||
      !d_cb || !d_cs || !d_wf || !d_ws) { fprintf(stderr, "device alloc failed\n"); return 2; }
  double *d_part = NULL, *d_metrics = NULL;
  CHECK_HIP(hipMalloc((void**)&d_part, sizeof(double) * N_OUTER * n_chunk * n_ts * K));
  CHECK_HIP(hipMalloc((void**)&d_metrics, sizeof(double) * WB2_NMETRIC * N_REGION * N_OUTER));

  const void* in[3] = {df, dt, dc};
  const int64_t* slab[3] = {NULL, NULL, NULL};
  CHECK_WB2(wb2_stream_partials(WB2_MODE_DET_ACC, WB2_F32, 0, in, slab, N_OUTER, N_LAT, N_LON,
                                dw, NULL, NULL, d_row0, d_nrow, n_chunk, n_ctile, d_seg, d_eoff,
                                1, n_ts, d_part, NULL));
  CHECK_WB2(wb2_det_combine(WB2_MODE_DET_ACC, 0, d_part, N_OUTER, n_chunk, 1, 1, d_eoff, n_ts,
                            d_band, 3, d_cb, d_cs, d_wf, d_ws, N_REGION, NULL, d_metrics, NULL));
  double m[WB2_NMETRIC * N_REGION * N_OUTER];
  CHECK_HIP(hipMemcpy(m, d_metrics, sizeof m, hipMemcpyDeviceToHost));

  /* reference loop (float32 elementwise, double accumulation) */
  int bad = 0;
  for (int r = 0; r < N_REGION; ++r) for (int o = 0; o < N_OUTER; ++o) {
    double sd = 0, sa = 0, s2 = 0, sp = 0, sf = 0, st = 0, sw = 0;
    for (int i = 0; i < N_LAT; ++i) {
      if (r == 1 && fabs(-90.0 + 30.0 * i) < 20.0) continue;
      for (int j = 0; j < N_LON; ++j) {
        const int q = o * N_PT + i * N_LON + j;
        const float d = f[q] - t[q], fa = f[q] - c[q], ta = t[q] - c[q];
        const float d2 = d * d, p = fa * ta, fa2 = fa * fa, ta2 = ta * ta;
        sd += w[i] * d; sa += w[i] * fabsf(d); s2 += w[i] * d2;
        sp += w[i] * p; sf += w[i] * fa2; st += w[i] * ta2; sw += w[i];
      }
    }
    const double want[WB2_NMETRIC] = {s2 / sw, sqrt(s2 / sw), sa / sw, sd / sw,
                                      (sp / sw) / sqrt((sf / sw) * (st / sw))};
    for (int k = 0; k < WB2_NMETRIC; ++k) {
      const double got = m[(k * N_REGION + r) * N_OUTER + o];
      if (fabs(got - want[k]) > 1e-12 * (1.0 + fabs(want[k]))) {
        fprintf(stderr, "mismatch metric %d region %d slab %d: %.17g vs %.17g\n", k, r, o, got, want[k]);
        ++bad;
      }
    }
  }
  if (bad) return 1;

  /* the latitude weights through the C ABI (metrics.py:35-60) */
  double lat_deg[N_LAT], w_abi[N_LAT];
  for (int i = 0; i < N_LAT; ++i) lat_deg[i] = -90.0 + 30.0 * i;
  CHECK_WB2(wb2_lat_weights(WB2_F64, lat_deg, N_LAT, w_abi));
  for (int i = 0; i < N_LAT; ++i)
    if (fabs(w_abi[i] - w[i]) > 1e-14 * (1.0 + fabs(w[i]))) {
      fprintf(stderr, "wb2_lat_weights[%d]: %.17g vs %.17g\n", i, w_abi[i], w[i]);
      return 1;
    }

  /* the temporal mean and the path's one exchange step, without torch:
   * (sum, count) accumulated on the device (xbeam.Mean, evaluation.py:735-744),
   * all-reduced over an RCCL communicator this program builds itself (one rank
   * here; rank 0 of a real job would hand the 128-byte id to the others) */
  const long long n_acc = (long long)WB2_NMETRIC * N_REGION;
  double *d_sum = NULL, *d_cnt = NULL;
  CHECK_HIP(hipMalloc((void**)&d_sum, sizeof(double) * n_acc));
  CHECK_HIP(hipMalloc((void**)&d_cnt, sizeof(double) * n_acc));
  CHECK_HIP(hipMemset(d_sum, 0, sizeof(double) * n_acc));
  CHECK_HIP(hipMemset(d_cnt, 0, sizeof(double) * n_acc));
  /* metrics[metric][region][outer]: mean over the N_OUTER slabs ("times") */
  CHECK_WB2(wb2_time_accumulate(d_metrics, n_acc, N_OUTER, 1, 0, d_sum, d_cnt, NULL));
  unsigned char id[128];
  void* comm = NULL;
  CHECK_WB2(wb2_comm_unique_id(id));
  CHECK_WB2(wb2_comm_init_rank(id, 1, 0, &comm));
  CHECK_WB2(wb2_time_mean_allreduce(d_sum, d_cnt, n_acc, comm, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  double h_sum[WB2_NMETRIC * N_REGION], h_cnt[WB2_NMETRIC * N_REGION];
  CHECK_HIP(hipMemcpy(h_sum, d_sum, sizeof h_sum, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost));
  CHECK_WB2(wb2_comm_destroy(comm));
  for (int q = 0; q < WB2_NMETRIC * N_REGION; ++q) {
    double want_mean = 0.0;
    for (int o = 0; o < N_OUTER; ++o) want_mean += m[q * N_OUTER + o];
    want_mean /= N_OUTER;
    if (h_cnt[q] != (double)N_OUTER ||
        fabs(h_sum[q] / h_cnt[q] - want_mean) > 1e-13 * (1.0 + fabs(want_mean))) {
      fprintf(stderr, "time mean %d: %.17g / %.17g vs %.17g\n", q, h_sum[q], h_cnt[q], want_mean);
      return 1;
    }
  }
  printf("c_abi_example ok: %d metrics x %d regions x %d slabs match; lat weights and the RCCL "
         "time mean through the C ABI too\n", WB2_NMETRIC, N_REGION, N_OUTER);
  return 0;
}
