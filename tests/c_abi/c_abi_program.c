/*
 * A pure-C caller of the chunk-program entry points of libwb2hip.so: what a
 * non-Python host (the reference's Beam workers call _evaluate_chunk once per
 * chunk, /root/reference/weatherbench2/evaluation.py:583-599, and fold the
 * results with xbeam.Mean, :735-744) binds to evaluate chunk after chunk with
 * ONE call each.
 *
 * Three chunks of four slabs (two variables + one (u, v) wind pair; 9 lat x
 * 128 lon, float32 forecast / truth / climatology in allocations of their own)
 * are evaluated twice:
 *   A. call by call: wb2_det_wind_suite_step per chunk (by slab address), the
 *      running sum of every metric value kept on the host in chunk order;
 *   B. as a program: wb2_program_create / add_launch / add_sink / finalize,
 *      then ONE wb2_program_replay per chunk with the chunk's three base
 *      pointers.
 * B must give A's bits: the arena after the last replay == the last chunk's
 * metrics, the device accumulators == the host's running sums.
 *
 *   gcc -std=c11 -I include -I /opt/rocm/include tests/c_abi/c_abi_program.c \
 *       -L weatherbench2_amd -lwb2hip -L /opt/rocm/lib -lamdhip64 -lm
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "wb2hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_WB2(x) do { if ((x) != 0) { \
  fprintf(stderr, "%s: %s\n", #x, wb2_last_error()); return 3; } } while (0)

enum { N_CHUNK_IN = 3, N_OUTER = 4, N_PAIR = 1, N_LAT = 9, N_LON = 128,
       N_PT = N_LAT * N_LON, N_REGION = 2, N_IN = 3 };

static void* to_device(const void* host, size_t bytes) {
  void* d = NULL;
  if (hipMalloc(&d, bytes) != hipSuccess) return NULL;
  if (hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

static int same(double a, double b) {
  return memcmp(&a, &b, sizeof a) == 0 || (isnan(a) && isnan(b));
}

int main(void) {
  /* latitude weights (metrics.py:40-60) through the C ABI */
  double lat_deg[N_LAT], w[N_LAT];
  for (int i = 0; i < N_LAT; ++i) lat_deg[i] = -90.0 + 22.5 * i;
  CHECK_WB2(wb2_lat_weights(WB2_F64, lat_deg, N_LAT, w));

  /* plan for regions {global, |lat| >= 20}: rows [0,4) | [4,5) | [5,9) */
  const int32_t chunk_row0[8] = {0, 4, 5, 0, 0, 0, 0, 0};
  const int32_t chunk_nrow[8] = {4, 1, 4, 0, 0, 0, 0, 0};
  const int32_t band_chunk0[4] = {0, 1, 2, 3};
  const int32_t seg_col0[2] = {0, N_LON};
  const double coef_band[N_REGION * 3] = {1, 1, 1, 1, 0, 1};
  const double coef_seg[N_REGION] = {1, 1};
  const int32_t region_wf[N_REGION] = {0, 0};
  double wsum[N_REGION] = {0, 0};
  for (int i = 0; i < N_LAT; ++i) {
    wsum[0] += w[i] * N_LON;
    if (fabs(lat_deg[i]) >= 20.0) wsum[1] += w[i] * N_LON;
  }
  const int tile = wb2_tile_cols(WB2_F32, N_LON, 1);
  const int n_ctile = (N_LON + tile - 1) / tile;
  const int32_t seg_eoff[2] = {0, n_ctile};
  const int n_ts = n_ctile, n_chunk = 8;
  const int K = wb2_num_slots(WB2_MODE_DET_ACC, 0);
  const int KW = wb2_num_slots(WB2_MODE_WIND, 0);
  const int n_pair = wb2_pairs_supported(WB2_MODE_DET_ACC, WB2_F32, 0, 0, N_LON, 1)
                         ? N_PAIR : 0;

  wb2_plan_tables plan;
  memset(&plan, 0, sizeof plan);
  plan.n_row = N_LAT; plan.n_col = N_LON; plan.n_chunk = n_chunk;
  plan.n_ctile = n_ctile; plan.n_seg = 1; plan.n_ts = n_ts; plan.n_band = 3;
  plan.n_region = N_REGION; plan.wfield_dtype = WB2_F64;
  plan.w_row = (const double*)to_device(w, sizeof w);
  plan.chunk_row0 = (const int32_t*)to_device(chunk_row0, sizeof chunk_row0);
  plan.chunk_nrow = (const int32_t*)to_device(chunk_nrow, sizeof chunk_nrow);
  plan.seg_col0 = (const int32_t*)to_device(seg_col0, sizeof seg_col0);
  plan.seg_eoff = (const int32_t*)to_device(seg_eoff, sizeof seg_eoff);
  plan.band_chunk0 = (const int32_t*)to_device(band_chunk0, sizeof band_chunk0);
  plan.coef_band = (const double*)to_device(coef_band, sizeof coef_band);
  plan.coef_seg = (const double*)to_device(coef_seg, sizeof coef_seg);
  plan.region_wf = (const int32_t*)to_device(region_wf, sizeof region_wf);
  plan.region_wsum = (const double*)to_device(wsum, sizeof wsum);
  if (!plan.w_row || !plan.chunk_row0 || !plan.chunk_nrow || !plan.seg_col0 ||
      !plan.seg_eoff || !plan.band_chunk0 || !plan.coef_band || !plan.coef_seg ||
      !plan.region_wf || !plan.region_wsum) {
    fprintf(stderr, "device alloc failed\n");
    return 2;
  }

  /* the chunks: forecast, truth, climatology -- separate allocations each */
  static float host[N_OUTER * N_PT];
  void* chunk[N_CHUNK_IN][N_IN];
  unsigned s = 2024u;
  for (int k = 0; k < N_CHUNK_IN; ++k)
    for (int j = 0; j < N_IN; ++j) {
      for (int i = 0; i < N_OUTER * N_PT; ++i) {
        s = s * 1664525u + 1013904223u;
        host[i] = (float)(s >> 8) / 8388608.0f - 1.0f;
      }
      chunk[k][j] = to_device(host, sizeof host);
      if (!chunk[k][j]) { fprintf(stderr, "device alloc failed\n"); return 2; }
    }

  const long long n_det = (long long)WB2_NMETRIC * N_REGION * N_OUTER;
  const long long n_wind = (long long)WB2_NMETRIC * N_REGION * n_pair;
  const long long n_all = n_det + n_wind;
  const size_t part_bytes = sizeof(double) * N_OUTER * n_chunk * n_ts * K;
  const size_t wpart_bytes = sizeof(double) * (n_pair ? n_pair : 1) * n_chunk * n_ts * KW;
  double *part_a, *wpart_a, *metrics_a, *part_b, *wpart_b, *arena;
  CHECK_HIP(hipMalloc((void**)&part_a, part_bytes));
  CHECK_HIP(hipMalloc((void**)&wpart_a, wpart_bytes));
  CHECK_HIP(hipMalloc((void**)&metrics_a, sizeof(double) * n_all));
  CHECK_HIP(hipMalloc((void**)&part_b, part_bytes));
  CHECK_HIP(hipMalloc((void**)&wpart_b, wpart_bytes));
  CHECK_HIP(hipMalloc((void**)&arena, sizeof(double) * n_all));

  /* ---- A. call by call ---- */
  double* last_a = (double*)malloc(sizeof(double) * n_all);
  double* sum_a = (double*)calloc(n_all, sizeof(double));
  int64_t* addr = (int64_t*)malloc(sizeof(int64_t) * N_IN * N_OUTER);
  int64_t* d_addr = NULL;
  CHECK_HIP(hipMalloc((void**)&d_addr, sizeof(int64_t) * N_IN * N_OUTER));
  for (int k = 0; k < N_CHUNK_IN; ++k) {
    for (int j = 0; j < N_IN; ++j)
      for (int o = 0; o < N_OUTER; ++o)
        addr[j * N_OUTER + o] = (int64_t)(uintptr_t)chunk[k][j] +
                                (int64_t)o * N_PT * (int64_t)sizeof(float);
    CHECK_HIP(hipMemcpy(d_addr, addr, sizeof(int64_t) * N_IN * N_OUTER,
                        hipMemcpyHostToDevice));
    const int64_t* slab[N_IN] = {d_addr, d_addr + N_OUTER, d_addr + 2 * N_OUTER};
    if (n_pair) {
      CHECK_WB2(wb2_det_wind_suite_step(&plan, WB2_MODE_DET_ACC, WB2_F32, 0, NULL, slab,
                                        1, N_OUTER, n_pair, part_a, wpart_a, metrics_a,
                                        metrics_a + n_det, NULL));
    } else {
      CHECK_WB2(wb2_det_suite_step(&plan, WB2_MODE_DET_ACC, WB2_F32, 0, NULL, slab, 1,
                                   N_OUTER, part_a, metrics_a, 0, 0, 0, 0, NULL, NULL,
                                   NULL, NULL));
    }
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(last_a, metrics_a, sizeof(double) * n_all, hipMemcpyDeviceToHost));
    for (long long e = 0; e < n_all; ++e) sum_a[e] += last_a[e];
  }

  /* ---- B. the same chunks through one program ---- */
  void* program = NULL;
  CHECK_WB2(wb2_program_create(&program));
  int32_t slot[N_IN * N_OUTER];
  int64_t rel[N_IN * N_OUTER];
  for (int j = 0; j < N_IN; ++j)
    for (int o = 0; o < N_OUTER; ++o) {
      slot[j * N_OUTER + o] = j;   /* pointer j of a replay: forecast, truth, clim */
      rel[j * N_OUTER + o] = (int64_t)o * N_PT * (int64_t)sizeof(float);
    }
  CHECK_WB2(wb2_program_add_launch(program, &plan, WB2_MODE_DET_ACC, WB2_F32, 0, N_IN,
                                   N_OUTER, n_pair, slot, rel, part_b,
                                   n_pair ? wpart_b : NULL, 0, 0));
  /* the sink: every arena element into an accumulator of its own */
  int32_t* src = (int32_t*)malloc(sizeof(int32_t) * n_all);
  unsigned char* r32 = (unsigned char*)calloc(n_all, 1);
  int64_t* sum_addr = (int64_t*)malloc(sizeof(int64_t) * n_all);
  int64_t* cnt_addr = (int64_t*)malloc(sizeof(int64_t) * n_all);
  double *d_sum = NULL, *d_cnt = NULL;
  CHECK_HIP(hipMalloc((void**)&d_sum, sizeof(double) * n_all));
  CHECK_HIP(hipMalloc((void**)&d_cnt, sizeof(double) * n_all));
  CHECK_HIP(hipMemset(d_sum, 0, sizeof(double) * n_all));
  CHECK_HIP(hipMemset(d_cnt, 0, sizeof(double) * n_all));
  for (long long e = 0; e < n_all; ++e) {
    src[e] = (int32_t)e;
    sum_addr[e] = (int64_t)(uintptr_t)(d_sum + e);
    cnt_addr[e] = (int64_t)(uintptr_t)(d_cnt + e);
  }
  int32_t* d_src = (int32_t*)to_device(src, sizeof(int32_t) * n_all);
  unsigned char* d_r32 = (unsigned char*)to_device(r32, (size_t)n_all);
  int64_t* d_sum_addr = (int64_t*)to_device(sum_addr, sizeof(int64_t) * n_all);
  int64_t* d_cnt_addr = (int64_t*)to_device(cnt_addr, sizeof(int64_t) * n_all);
  if (!d_src || !d_r32 || !d_sum_addr || !d_cnt_addr) return 2;
  CHECK_WB2(wb2_program_add_sink(program, d_src, d_r32, n_all, 1, 0, NULL, NULL, 0));
  CHECK_WB2(wb2_program_finalize(program, arena, N_IN, 0));
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  for (int k = 0; k < N_CHUNK_IN; ++k) {
    const int64_t ptrs[N_IN] = {(int64_t)(uintptr_t)chunk[k][0],
                                (int64_t)(uintptr_t)chunk[k][1],
                                (int64_t)(uintptr_t)chunk[k][2]};
    const int64_t sink_args[3] = {(int64_t)(uintptr_t)d_sum_addr,
                                  (int64_t)(uintptr_t)d_cnt_addr, 0};
    CHECK_WB2(wb2_program_replay(program, ptrs, N_IN, NULL, 0, sink_args, NULL, 0,
                                 stream));   /* asynchronous: no wait between chunks */
  }
  CHECK_HIP(hipStreamSynchronize(stream));
  double seconds[5];
  int64_t replays = 0;
  CHECK_WB2(wb2_program_stats(program, seconds, &replays));
  double* last_b = (double*)malloc(sizeof(double) * n_all);
  double* sum_b = (double*)malloc(sizeof(double) * n_all);
  double* cnt_b = (double*)malloc(sizeof(double) * n_all);
  CHECK_HIP(hipMemcpy(last_b, arena, sizeof(double) * n_all, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(sum_b, d_sum, sizeof(double) * n_all, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(cnt_b, d_cnt, sizeof(double) * n_all, hipMemcpyDeviceToHost));
  CHECK_WB2(wb2_program_destroy(program));
  int bad = 0;
  for (long long e = 0; e < n_all; ++e) {
    if (!same(last_a[e], last_b[e]) || !same(sum_a[e], sum_b[e]) ||
        cnt_b[e] != (double)N_CHUNK_IN) {
      if (bad < 5)
        fprintf(stderr, "element %lld: last %.17g vs %.17g, sum %.17g vs %.17g, count %g\n",
                e, last_a[e], last_b[e], sum_a[e], sum_b[e], cnt_b[e]);
      ++bad;
    }
  }
  /* the values mean something: MSE of slab 0 over the globe against a loop */
  {
    static float hf[N_OUTER * N_PT], ht[N_OUTER * N_PT];
    CHECK_HIP(hipMemcpy(hf, chunk[N_CHUNK_IN - 1][0], sizeof hf, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(ht, chunk[N_CHUNK_IN - 1][1], sizeof ht, hipMemcpyDeviceToHost));
    double s2 = 0, sw = 0;
    for (int i = 0; i < N_LAT; ++i)
      for (int j = 0; j < N_LON; ++j) {
        const float d = hf[i * N_LON + j] - ht[i * N_LON + j];
        s2 += w[i] * (double)(d * d);
        sw += w[i];
      }
    const double want = s2 / sw, got = last_b[0];   /* [mse][global][slab 0] */
    if (fabs(got - want) > 1e-12 * (1.0 + fabs(want))) {
      fprintf(stderr, "MSE of slab 0: %.17g vs %.17g\n", got, want);
      ++bad;
    }
  }
  if (replays != N_CHUNK_IN) { fprintf(stderr, "replays %lld\n", (long long)replays); ++bad; }
  if (bad) return 1;
  printf("c_abi_program ok: %d chunks x %lld values (%d wind pair) replayed with one call "
         "each, bits of the call-by-call path\n", N_CHUNK_IN, n_all, n_pair);
  return 0;
}
