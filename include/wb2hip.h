/*
 * wb2hip.h -- C ABI of libwb2hip.so: MI355X (gfx950) kernels for the
 * WeatherBench2 per-chunk metric-evaluation hot path.
 *
 * The reference (google-research/weatherbench2) is pure Python; it has no FFI.
 * Its boundary for this path is the duck-typed operator protocol
 *   Metric.compute_chunk(forecast, truth, region, skipna)   weatherbench2/metrics.py:88-115
 *   Region.apply(dataset, weights)                          weatherbench2/regions.py:40-54
 *   DerivedVariable.compute(dataset)                        weatherbench2/derived_variables.py:54-56
 * called from evaluation._metric_and_region_loop            weatherbench2/evaluation.py:388-438.
 * The entry points below are what a ctypes binding of that protocol calls (see
 * INTEGRATION.md); each one names the reference arithmetic it replaces.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error; wb2_last_error() gives
 *    the (thread-local) message.  Nothing throws, nothing takes ownership of a
 *    caller buffer, nothing allocates device memory.
 *  - every pointer marked DEV is a device (HBM) pointer; `stream` is a
 *    hipStream_t passed as void* (NULL = the default stream).  All work is
 *    enqueued asynchronously on `stream`.
 *  - a call over zero units (n_outer, n_point, n_time ... == 0: an empty chunk)
 *    is a legal no-op and returns 0 whatever the data pointers are -- empty
 *    buffers have no address; a negative count is an error.
 *  - a "slab" is one 2-D (n_row, n_col) field, n_col contiguous.  For the
 *    (…, latitude, longitude) layout of 0.25-degree ERA5 rows are latitudes; for
 *    the (…, longitude, latitude) layout of the low-resolution/mocked datasets
 *    rows are longitudes.  `n_outer` counts the slabs = product of all
 *    non-spatial dims (time, lead, level, …).
 *  - regions are decomposed on the host into `bands` of rows and `segs` of
 *    columns with identical membership; the streaming kernel emits partial sums
 *    per (outer, row-chunk, col-tile, weight-field, seg) and the combine kernel
 *    folds them into every region at once.  See DESIGN.md.
 */
#ifndef WB2HIP_H_
#define WB2HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WB2_VERSION 1

/* element types of the data slabs */
#define WB2_F32 0
#define WB2_F64 1

/* streaming-reduction modes (which reference metrics one pass feeds) */
#define WB2_MODE_DET 0      /* forecast,truth        -> MSE/RMSE/MAE/Bias      metrics.py:236-359 */
#define WB2_MODE_DET_ACC 1  /* forecast,truth,clim   -> the above + ACC        metrics.py:377-414 */
#define WB2_MODE_WIND 2     /* fu,tu,fv,tv           -> WindVectorMSE/RMSE     metrics.py:175-233 */

#define WB2_MODE_ENS 3      /* members,truth         -> CRPS & ensemble moments  metrics.py:532-846,1161-1363
                             * (partials come from wb2_ens_partials)              */

#define WB2_MODE_GAUSS 4    /* mean,std,truth        -> GaussianCRPS, GaussianVariance  metrics.py:849-937 */
#define WB2_MODE_GAUSS_THR 5 /* mean,std,truth,thr   -> Gaussian Brier / ignorance / RPS part  metrics.py:975-1158 */
#define WB2_MODE_SEEPS 7     /* forecast,truth,wet-threshold + aux p1 field + scalar dry threshold
                             * -> SEEPS  metrics.py:417-524 (always NaN-skipping)             */
#define WB2_MODE_ENS_THR 6  /* members,truth,thr     -> ensemble Brier / debiased Brier / ignorance /
                             *                          RPS part  metrics.py:1524-1891
                             * (partials come from wb2_ens_threshold_partials)              */

/* number of metrics written by wb2_det_combine, in this order */
#define WB2_NMETRIC 5
#define WB2_METRIC_MSE 0
#define WB2_METRIC_RMSE 1
#define WB2_METRIC_MAE 2
#define WB2_METRIC_BIAS 3
#define WB2_METRIC_ACC 4
/* Modes >= WB2_MODE_GAUSS are "generic": their slots are [q_0..q_{KQ-1}] plus,
 * with skipna, the matching notnull weights [n_0..n_{KQ-1}], and
 * wb2_det_combine writes the KQ spatial means into metrics rows 0..KQ-1
 * (`metrics` must then hold KQ rows, not WB2_NMETRIC). */
#define WB2_GAUSS_CRPS 0
#define WB2_GAUSS_VARIANCE 1
#define WB2_GAUSS_THR_BRIER 0
#define WB2_GAUSS_THR_IGNORANCE 1
#define WB2_GAUSS_THR_RPS_PART 2
#define WB2_ENS_THR_BRIER 0
#define WB2_ENS_THR_DEBIASED_BRIER 1
#define WB2_ENS_THR_IGNORANCE 2
#define WB2_ENS_THR_RPS_PART 3

/* metrics written by wb2_ens_combine, in this order */
#define WB2_NMETRIC_ENS 8
#define WB2_ENS_CRPS 0          /* CRPS                                metrics.py:610-675  */
#define WB2_ENS_CRPS_SPREAD 1   /* CRPSSpread                          metrics.py:678-694  */
#define WB2_ENS_CRPS_SKILL 2    /* CRPSSkill                           metrics.py:697-715  */
#define WB2_ENS_MEAN_MSE 3      /* EnsembleMeanMSE                     metrics.py:1310-1333 */
#define WB2_ENS_MEAN_RMSE 4     /* EnsembleMeanRMSESqrtBeforeTimeAvg   metrics.py:1269-1307 */
#define WB2_ENS_VARIANCE 5      /* EnsembleVariance                    metrics.py:1213-1241 */
#define WB2_ENS_STDDEV 6        /* EnsembleStddevSqrtBeforeTimeAvg     metrics.py:1161-1210 */
#define WB2_ENS_DEBIASED_MSE 7  /* DebiasedEnsembleMeanMSE             metrics.py:1336-1363 */

int wb2_version(void);
const char* wb2_last_error(void);

/* Number of accumulated sums ("slots") per cell for a mode.
 *   DET      : S(w d) S(w|d|) S(w d^2)                                  [+ S(w notnull d)]
 *   DET_ACC  : the above + S(w fa ta) S(w fa^2) S(w ta^2)               [+ 3 more notnull sums]
 *   WIND     : S(w (du^2+dv^2))                                         [+ 1]
 *   GAUSS    : S(w crps) S(w std^2)                                     [+ 2]
 *   GAUSS_THR: S(w brier) S(w ignorance) S(w rps_part)                  [+ 3]
 *   ENS_THR  : S(w brier) S(w debiased) S(w ignorance) S(w rps_part)    [+ 4]
 *   SEEPS    : S(w seeps)                                               [+ 1]
 * The bracketed sums-of-weights exist only when skipna != 0 (xarray computes
 * them always, but without NaNs they are data independent: metrics.py:161-163). */
int wb2_num_slots(int mode, int skipna);

/* Columns one wavefront covers (64 lanes x 16-byte vectors whenever n_col holds
 * one vector: rows of any length and element-aligned base pointers take the wide
 * loads -- `aligned16` only matters with WB2HIP_UNALIGNED_VEC=0, the rule of
 * rounds 1-3: one column per lane unless everything is 16-byte aligned);
 * n_ctile = ceil(n_col / wb2_tile_cols(...)). */
int wb2_tile_cols(int dtype, int n_col, int aligned16);
/* The same for a given instantiation: an instantiation may cover fewer columns
 * per wavefront than the dtype's default (the build-time knob
 * WB2_F32_VEC_HEAVY cuts WB2_MODE_DET_ACC with skipna or a 2-D weight field to
 * 8-byte vectors; the shipped value keeps 16-byte vectors for all of them).
 * wb2_tile_cols(d, n, a) == wb2_tile_cols_ex(WB2_MODE_DET,
 * d, 0, 0, n, a).  Callers of wb2_stream_partials[_ex] size n_ctile / seg_eoff
 * with THIS function. */
int wb2_tile_cols_ex(int mode, int dtype, int skipna, int has_wfield, int n_col,
                     int aligned16);

/*
 * K1: fused weighted streaming reduction.  Replaces the elementwise temporaries
 * and the two einsums of _spatial_average (metrics.py:141-163) for every metric
 * of `mode` and every region, reading each input exactly once.
 *
 *  in[i]       DEV  input i (mode order above); slab o of input i starts at
 *                   element  (slab[i] ? slab[i][o] : o) * n_row * n_col
 *  slab[i]     DEV  int64[n_outer] or NULL (identity); lets truth / climatology
 *                   be gathered by valid time without a copy (metrics.py:398-404,
 *                   evaluation.py:474-475)
 *  w_row       DEV  double[n_row]  > 0   (latitude weights when rows = lat, else 1)
 *  w_col       DEV  double[n_col]  > 0   (latitude weights when cols = lat) or NULL
 *                   when every column weight is 1 (rows = lat)
 *  wfield      DEV  double[n_row*n_col] or NULL: a 2-D weight factor such as the
 *                   land-sea mask (regions.py:112-138); points with wfield == 0
 *                   are excluded (metrics.py:159-160).  PRECONDITION: every
 *                   value is finite and >= 0 -- the kernel multiplies the
 *                   weight through unconditionally and only clears the DATA of
 *                   excluded points, so a NaN cell would make every sum of its
 *                   tile NaN and a negative cell would be counted.  The Python
 *                   host checks this (plan.build_plan); other callers must.
 *  chunk_row0/chunk_nrow  DEV int32[n_chunk]: row range of each chunk (a chunk
 *                   never straddles a band boundary; nrow == 0 chunks are padding)
 *  seg_col0    DEV  int32[n_seg+1]: column range of each seg
 *  seg_eoff    DEV  int32[n_seg+1]: prefix sum of the number of column tiles each
 *                   seg intersects, tiles(s) = (seg_col0[s+1]-1)/T - seg_col0[s]/T + 1
 *                   with T = wb2_tile_cols(...).  A (seg, tile) pair is one
 *                   "entry" e = seg_eoff[s] + tile - seg_col0[s]/T; n_ts =
 *                   seg_eoff[n_seg] entries exist per (outer, chunk, weight field).
 *  partials    DEV  double[n_outer][n_chunk][nwf][n_ts][K] (out),
 *                   nwf = wfield ? 2 : 1, K = wb2_num_slots(mode, skipna).
 *                   Entries of padding chunks are not written.
 *  n_ctile          must equal ceil(n_col / wb2_tile_cols_ex(mode, dtype, skipna,
 *                   wfield != NULL, n_col, a16))
 *                   with a16 = all of in[] (and wfield) are 16-byte aligned; it
 *                   is passed explicitly so that the caller's allocation of
 *                   `partials` and the launch can never disagree.
 */
int wb2_stream_partials(int mode, int dtype, int skipna,
                        const void* const* in, const int64_t* const* slab,
                        int64_t n_outer, int32_t n_row, int32_t n_col,
                        const double* w_row, const double* w_col,
                        const double* wfield,
                        const int32_t* chunk_row0, const int32_t* chunk_nrow,
                        int32_t n_chunk, int32_t n_ctile,
                        const int32_t* seg_col0, const int32_t* seg_eoff,
                        int32_t n_seg, int32_t n_ts,
                        double* partials, void* stream);

/* As wb2_stream_partials, plus the mode-specific extras:
 *  aux     DEV double[n_row*n_col] or NULL  (WB2_MODE_SEEPS: the climatological
 *          dry fraction p1, NaN where it is masked out, metrics.py:504-506)
 *  scalar  (WB2_MODE_SEEPS: the dry threshold in data units, metrics.py:449)
 *  wfield_dtype  WB2_F64, or WB2_F32: the 2-D weight field stored as float32 --
 *          for a field whose values ARE float32 numbers (an ERA5 land-sea mask,
 *          a thresholded mask): the same results bit for bit, half the field
 *          bytes per point.  float32 inputs of the modes DET / DET_ACC / WIND
 *          only; anything else with WB2_F32 is an error. */
int wb2_stream_partials_ex(int mode, int dtype, int skipna,
                           const void* const* in, const int64_t* const* slab,
                           int64_t n_outer, int32_t n_row, int32_t n_col,
                           const double* w_row, const double* w_col,
                           const void* wfield, int wfield_dtype,
                           const double* aux,
                           double scalar, const int32_t* chunk_row0,
                           const int32_t* chunk_nrow, int32_t n_chunk,
                           int32_t n_ctile, const int32_t* seg_col0,
                           const int32_t* seg_eoff, int32_t n_seg,
                           int32_t n_ts, double* partials, void* stream);

/* As wb2_stream_partials_ex for slabs that do not share an allocation: input i
 * of outer slab o is the n_row x n_col slab at BYTE ADDRESS slab_addr[i][o].
 * This is how ONE launch serves every variable of a chunk (a Dataset's
 * variables are separate arrays; the reference walks them one after the other,
 * metrics.py:264, 284, 329, 358, 405-410 under xarray's Dataset arithmetic) and
 * several consecutive chunks of the Beam pipeline
 * (evaluation.py:583-599, 693-705: `input_chunks=init_time=1,lead_time=1`,
 * `split_vars=False`).  The fold is per slab, so results are bit-identical to
 * those of separate launches with the same chunk tables.
 *
 *  slab_addr[i] DEV int64[n_outer]: addresses of device memory, each a multiple
 *                   of the element size; required for every input of `mode`
 *  aligned16        nonzero iff every address (and wfield) is 16-byte aligned:
 *                   the caller built the tables and knows; selects the same
 *                   instantiation as aligned in[] would (n_ctile as above)
 */
int wb2_stream_partials_addr(int mode, int dtype, int skipna,
                             const int64_t* const* slab_addr, int aligned16,
                             int64_t n_outer, int32_t n_row, int32_t n_col,
                             const double* w_row, const double* w_col,
                             const void* wfield, int wfield_dtype,
                             const double* aux,
                             double scalar, const int32_t* chunk_row0,
                             const int32_t* chunk_nrow, int32_t n_chunk,
                             int32_t n_ctile, const int32_t* seg_col0,
                             const int32_t* seg_eoff, int32_t n_seg,
                             int32_t n_ts, double* partials, void* stream);

/* K1 with the wind-vector pairs of the launch answered from the SAME read as
 * their per-variable metrics.  The reference forms diff = forecast - truth once
 * and MSE.compute_chunk derives both the per-variable and the wind-vector
 * numbers from it (metrics.py:283-301 calling :194-201; scripts/evaluate.py:
 * 279-311, 420-425); a separate WB2_MODE_WIND launch reads u and v again.
 *
 *  mode        WB2_MODE_DET or WB2_MODE_DET_ACC
 *  in / slab   as wb2_stream_partials_ex; in == NULL: `slab` holds byte
 *              addresses as in wb2_stream_partials_addr (`aligned16` likewise;
 *              ignored otherwise)
 *  n_pair      the LAST 2 * n_pair of the n_outer slabs are the u slabs, then
 *              the v slabs, of n_pair pairs: pair k = slabs n_outer - 2 n_pair
 *              + k and n_outer - n_pair + k
 *  partials    as wb2_stream_partials for all n_outer slabs (the slabs outside
 *              the pairs go through the per-variable kernel)
 *  wind_partials DEV double[n_pair][n_chunk][nwf][n_ts][KW], KW =
 *              wb2_num_slots(WB2_MODE_WIND, skipna): what a WB2_MODE_WIND launch
 *              over (u, truth u, v, truth v) of the pairs writes, bit for bit
 *              (fold with wb2_det_combine(WB2_MODE_WIND, ...)).
 * Every slot -- per-variable and wind -- holds the bits of the separate
 * launches: same loads, same float32 du^2 + dv^2, same order of additions.
 * wb2_pairs_supported() says whether the launch geometry has a pair kernel
 * (rows too narrow for the wide loads have none: launch WB2_MODE_WIND then). */
int wb2_pairs_supported(int mode, int dtype, int skipna, int has_wfield,
                        int n_col, int aligned16);
int wb2_stream_partials_pairs(int mode, int dtype, int skipna,
                              const void* const* in,
                              const int64_t* const* slab, int aligned16,
                              int64_t n_outer, int64_t n_pair, int32_t n_row,
                              int32_t n_col, const double* w_row,
                              const double* w_col, const void* wfield,
                              int wfield_dtype, const int32_t* chunk_row0,
                              const int32_t* chunk_nrow, int32_t n_chunk,
                              int32_t n_ctile, const int32_t* seg_col0,
                              const int32_t* seg_eoff, int32_t n_seg,
                              int32_t n_ts, double* partials,
                              double* wind_partials, void* stream);

/*
 * K2: fold the partials into per-region sums and finalise the metrics.
 * Replaces the region loop + concat of evaluation.py:416-430 and the ratio /
 * sqrt epilogues of metrics.py:172, 233, 407-414.
 *
 *  band_chunk0 DEV int32[n_band+1]: chunks of band b are
 *                  [band_chunk0[b], band_chunk0[b+1])
 *  seg_eoff    DEV int32[n_seg+1]: as above (entries of seg s are
 *                  [seg_eoff[s], seg_eoff[s+1]))
 *  coef_band   DEV double[n_region][n_band]: multiplicity of the band's rows in
 *                  the region (0 = excluded; SliceRegion lists may repeat rows)
 *  coef_seg    DEV double[n_region][n_seg]:  same for columns
 *  region_wf   DEV int32[n_region]: 0 = uniform weights, 1 = `wfield`
 *  region_wsum DEV double[n_region]: sum of the region's weights (used as the
 *                  denominator when skipna == 0)
 *  sums        DEV double[n_outer][n_region][K]           (out, may be NULL)
 *  metrics     DEV double[WB2_NMETRIC][n_region][n_outer] (out, may be NULL).
 *                  For WB2_MODE_WIND only MSE and RMSE are meaningful.
 */
int wb2_det_combine(int mode, int skipna, const double* partials,
                    int64_t n_outer, int32_t n_chunk, int32_t nwf,
                    int32_t n_seg, const int32_t* seg_eoff, int32_t n_ts,
                    const int32_t* band_chunk0, int32_t n_band,
                    const double* coef_band, const double* coef_seg,
                    const int32_t* region_wf, const double* region_wsum,
                    int32_t n_region, double* sums, double* metrics,
                    void* stream);

/*
 * K3: fused ensemble pass.  Replaces the member mean / var(ddof=1) / abs-mean
 * sweeps (metrics.py:562-565, 824), the argsort ranks (:827-846) and the
 * rank-weighted sum (:804-813) with ONE read of the members: a lane keeps the M
 * values of its grid point in registers and sorts them with a sorting network.
 *
 *  ens         DEV  member m, slab s starts at element
 *                   m * member_stride + s * n_row * n_col  (member-major, e.g.
 *                   dims (realization, ..., lat, lon) as in schema.py:113-114)
 *  ens_slab / truth_slab  DEV int64[n_outer] or NULL (identity)
 *  n_member         1 <= M <= 128 (float32) / 64 (float64).  M == 1 yields
 *                   var = NaN; callers apply the reference's M == 1 special
 *                   cases (metrics.py:1196-1204) themselves.
 *  partials    DEV  double[n_outer][n_chunk][nwf][n_ts][K], K = wb2_ens_num_slots,
 *                   with the column tile width wb2_ens_tile_cols() (= 64)
 * Everything else as for wb2_stream_partials.
 */
int wb2_ens_num_slots(int skipna);
int wb2_ens_tile_cols(int32_t n_col);
int wb2_ens_partials(int dtype, int skipna, const void* ens,
                     const int64_t* ens_slab, const void* truth,
                     const int64_t* truth_slab, int32_t n_member,
                     int64_t member_stride, int64_t n_outer, int32_t n_row,
                     int32_t n_col, const double* w_row, const double* w_col,
                     const double* wfield, const int32_t* chunk_row0,
                     const int32_t* chunk_nrow, int32_t n_chunk,
                     int32_t n_ctile, const int32_t* seg_col0,
                     const int32_t* seg_eoff, int32_t n_seg, int32_t n_ts,
                     double* partials, void* stream);

/*
 * Ensemble threshold metrics: one read of the members (+ truth + threshold)
 * gives, per grid point, the exceedance counts behind EnsembleBrierScore,
 * DebiasedEnsembleBrierScore, EnsembleIgnoranceScore and the EnsembleRPS part
 * (metrics.py:1524-1560, 1720-1738, 1791-1802); any M >= 1, no sorting.
 * Slots/partials as for WB2_MODE_ENS_THR; fold with
 * wb2_det_combine(WB2_MODE_ENS_THR, ...).  `threshold` is [n_slab][n_row][n_col]
 * in the data dtype, resolved through thr_slab like truth.
 */
int wb2_ens_threshold_partials(
    int dtype, int skipna, const void* ens, const int64_t* ens_slab,
    const void* truth, const int64_t* truth_slab, const void* threshold,
    const int64_t* thr_slab, int32_t n_member, int64_t member_stride,
    int64_t n_outer, int32_t n_row, int32_t n_col, const double* w_row,
    const double* w_col, const double* wfield, const int32_t* chunk_row0,
    const int32_t* chunk_nrow, int32_t n_chunk, int32_t n_ctile,
    const int32_t* seg_col0, const int32_t* seg_eoff, int32_t n_seg,
    int32_t n_ts, double* partials, void* stream);

/*
 * Means / raw moments along one axis of a [n_lead][n_red][n_tail] view: the
 * compute core of scripts/compute_ensemble_mean.py:111-141 (xbeam.Mean over
 * realization), scripts/compute_averages.py:125-167 (v * lat_weights, mean over
 * averaging dims) and scripts/compute_statistical_moments.py:52-80 (means of
 * notnull(x), x, x**2).  For every (lead, tail):
 *   sum   = sum_r w_red[r] * x[l][r][t]          over the valid r
 *   sumsq = sum_r w_red[r] * (x * x)             (x * x in the input dtype;
 *                                                 NULL = not wanted)
 *   count = number of valid r                    (valid: !isnan(x), or every r
 *                                                 when skipna == 0)
 * in fp64.  w_red == NULL means all ones; otherwise w_red holds n_red /
 * w_repeat weights, each applying to w_repeat consecutive r (w_repeat = 1440
 * for latitude weights over a merged (latitude, longitude) axis, 1 for one
 * weight per element).  Deterministic: the reduced axis is cut into n_split
 * slices (wb2_axis_moments_splits() proposes a count) whose partials are
 * combined in slice order; `workspace` holds 3 * n_split * n_lead * n_tail
 * doubles.
 */
int wb2_axis_moments_splits(int64_t n_lead, int64_t n_red, int64_t n_tail,
                            int64_t w_repeat);
int wb2_axis_moments(int dtype, const void* x, int64_t n_lead, int64_t n_red,
                     int64_t n_tail, const double* w_red, int64_t w_repeat,
                     int skipna, int n_split, double* workspace, double* sum,
                     double* sumsq, double* count, void* stream);

/*
 * Spatial* threshold metrics (SpatialEnsembleBrierScore,
 * SpatialDebiasedEnsembleBrierScore, SpatialEnsembleIgnoranceScore,
 * SpatialEnsembleRPS; metrics.py:1615-1638, 1697-1719, 1780-1802, 1870-1891):
 * the four per-point scores of wb2_ens_threshold_partials, unreduced, to
 * maps[4][n_outer][n_point] (fp64; Brier, debiased Brier, ignorance, RPS
 * part).  Slab addressing as in wb2_rank_histogram (n_point = n_row * n_col).
 */
int wb2_ens_threshold_maps(int dtype, int skipna, const void* ens,
                           const int64_t* ens_slab, const void* truth,
                           const int64_t* truth_slab, const void* threshold,
                           const int64_t* thr_slab, int32_t n_member,
                           int64_t member_stride, int64_t n_outer,
                           int64_t n_point, double* maps, void* stream);

/*
 * SpatialSEEPS (metrics.py:418-509): the per-point SEEPS score (NaN where the
 * forecast, truth or masked dry fraction is NaN) to out[n_outer][n_point]
 * (fp64).  in = {forecast, truth, wet threshold} in `dtype`, each resolved
 * through its slab table (NULL = identity); aux = p1[n_point] with NaN where
 * p1 is outside (min_p1, max_p1); scalar = dry threshold in data units.
 */
int wb2_seeps_map(int dtype, const void* const* in, const int64_t* const* slab,
                  int64_t n_outer, int64_t n_point, const double* aux,
                  double scalar, double* out, void* stream);

/* wb2_seeps_map for slabs given by ADDRESS: addr[i][o] (DEV int64[n_outer], i =
 * forecast, truth, wet threshold) is the byte address of input i's slab of
 * outer index o -- the (init_time=1, lead_time=1) chunks of a window are
 * allocations of their own; one launch scores the k chunks that carry one lead
 * label (map_suite.py) and wb2_time_accumulate_runs adds them in chunk order. */
int wb2_seeps_map_addr(int dtype, const int64_t* const* addr, int64_t n_outer,
                       int64_t n_point, const double* aux, double scalar,
                       double* out, void* stream);

/*
 * RankHistogram (metrics.py:1894-2042): truth's rank among the n_member
 * members of each sample, binned by (n_member + 1) / n_bins, as float64.
 * Sample (o, pt): members at ens[(m * member_stride) + ens_slab[o] * n_point + pt]
 * (member_stride in ELEMENTS), truth at truth[truth_slab[o] * n_point + pt];
 * NULL tables mean identity.  rank = #{x_m < t}; ties (x_m == t) are broken
 * uniformly at random from a counter-based hash of (seed, o, pt) when
 * break_ties != 0 (the reference perturbs with NumPy's RNG, :1955-1980 --
 * same distribution, different stream), else truth goes first.  NaN members
 * rank highest.
 *   acc_row == NULL: out[n_outer][n_point][n_bins] is fully written (one-hot).
 *   acc_row != NULL: out[acc_row[o]][n_point][n_bins] += 1 per sample (the
 *                    caller zero-fills `out`; the temporal mean of Metric.compute
 *                    :117-138 is then out / n_time).
 * Fails when (n_member + 1) % n_bins != 0 (:1933-1938).
 */
int wb2_rank_histogram(int dtype, const void* ens, const int64_t* ens_slab,
                       const void* truth, const int64_t* truth_slab,
                       int32_t n_member, int64_t member_stride, int64_t n_outer,
                       int64_t n_point, int32_t n_bins, int break_ties,
                       uint64_t seed, const int64_t* acc_row, double* out,
                       void* stream);

/* As wb2_ens_partials, additionally storing the six pointwise values
 * (skill, spread, (t-mean)^2, var, std^2, debiased; NaN where undefined) to
 * maps[6][n_outer][n_row*n_col] (fp64, may be NULL): the Spatial* ensemble
 * metrics, metrics.py:718-772, 1244-1266, 1366-1399. */
int wb2_ens_partials_maps(int dtype, int skipna, const void* ens,
                          const int64_t* ens_slab, const void* truth,
                          const int64_t* truth_slab, int32_t n_member,
                          int64_t member_stride, int64_t n_outer,
                          int32_t n_row, int32_t n_col, const double* w_row,
                          const double* w_col, const double* wfield,
                          const int32_t* chunk_row0, const int32_t* chunk_nrow,
                          int32_t n_chunk, int32_t n_ctile,
                          const int32_t* seg_col0, const int32_t* seg_eoff,
                          int32_t n_seg, int32_t n_ts, double* partials,
                          double* maps, void* stream);

/* As wb2_ens_partials for slabs given by ADDRESS: ens_addr[o] is the byte
 * address of member 0's slab of outer index o (member m follows at
 * m * member_stride elements), truth_addr[o] that of the truth slab (both DEV
 * int64[n_outer]).  The variables of one `init_time=1,lead_time=1` ensemble
 * chunk (docs/source/official-evaluation.md:765-860) are separate allocations:
 * one launch takes all that share a member stride.  Same kernels (and bits) as
 * wb2_ens_partials for every member count. */
int wb2_ens_partials_addr(int dtype, int skipna, const int64_t* ens_addr,
                          const int64_t* truth_addr, int32_t n_member,
                          int64_t member_stride, int64_t n_outer, int32_t n_row,
                          int32_t n_col, const double* w_row,
                          const double* w_col, const double* wfield,
                          const int32_t* chunk_row0, const int32_t* chunk_nrow,
                          int32_t n_chunk, int32_t n_ctile,
                          const int32_t* seg_col0, const int32_t* seg_eoff,
                          int32_t n_seg, int32_t n_ts, double* partials,
                          void* stream);

/* As wb2_ens_partials_maps for a GATHERED ensemble: member m of outer index o
 * is the [n_row][n_col] slab at device address member_ptr[o * n_member + m]
 * (DEV int64[n_outer][n_member]) -- no stride, no common base: the forecast of
 * a probabilistic-climatology baseline (evaluation.py:458-470, 714-726: one
 * member per climatological year, gathered by (dayofyear, hour) of the valid
 * time) is read in place from the resident observations; a hole of the gather
 * (29 February in a common year) points at a slab of NaNs.  Register-sort
 * kernels only: n_member <= 128 (float32) / 64 (float64). */
int wb2_ens_partials_gather(int dtype, int skipna, const int64_t* member_ptr,
                            const void* truth, const int64_t* truth_slab,
                            int32_t n_member, int64_t n_outer, int32_t n_row,
                            int32_t n_col, const double* w_row,
                            const double* w_col, const double* wfield,
                            const int32_t* chunk_row0,
                            const int32_t* chunk_nrow, int32_t n_chunk,
                            int32_t n_ctile, const int32_t* seg_col0,
                            const int32_t* seg_eoff, int32_t n_seg,
                            int32_t n_ts, double* partials, double* maps,
                            void* stream);

/* Region fold + finalisation for the ensemble pass (same tables as
 * wb2_det_combine); metrics is double[WB2_NMETRIC_ENS][n_region][n_outer]. */
int wb2_ens_combine(int skipna, const double* partials, int64_t n_outer,
                    int32_t n_chunk, int32_t nwf, int32_t n_seg,
                    const int32_t* seg_eoff, int32_t n_ts,
                    const int32_t* band_chunk0, int32_t n_band,
                    const double* coef_band, const double* coef_seg,
                    const int32_t* region_wf, const double* region_wsum,
                    int32_t n_region, double* sums, double* metrics,
                    void* stream);

/*
 * Running temporal mean (Metric.compute's .mean(init_time) metrics.py:125-138
 * and the (sum, count) combiner of xbeam.Mean, evaluation.py:740-744).
 *   values DEV double[n_lead][n_time][n_tail] viewed as (lead.., time, tail..)
 *   sum, count DEV double[n_lead][n_tail]  (in/out, caller zero-initialises)
 * skipna != 0: NaN values add nothing to sum nor count.
 * The sum continues FROM the accumulator, value by value in time order
 * (sum = ((sum + v_0) + v_1) + ...): feeding the time steps one call at a time
 * or several per call gives the same bits.
 */
int wb2_time_accumulate(const double* values, int64_t n_lead, int64_t n_time,
                        int64_t n_tail, int skipna, double* sum, double* count,
                        void* stream);

/* The same for float32 or float64 values (dtype WB2_F32 / WB2_F64; float32
 * widens exactly: metric results in the reference's float32 result dtype need
 * no conversion pass) with a destination table: result element idx of
 * [n_lead][n_tail] goes to accumulator element dst[idx] (DEV
 * int64[n_lead * n_tail], entries distinct).  Chunks that split the lead dim as well
 * (`input_chunks=init_time=1,lead_time=1`, docs/source/official-evaluation.md:
 * 537-549) accumulate into the rows of their lead labels: xbeam.Mean combines
 * per chunk key (evaluation.py:740-744).  dst == NULL: identity. */
int wb2_time_accumulate_scatter(int dtype, const void* values, int64_t n_lead,
                                int64_t n_time, int64_t n_tail, int skipna,
                                const int64_t* dst, double* sum, double* count,
                                void* stream);
/* The same with one table entry per RUN of `run` consecutive result elements
 * that go to consecutive accumulator elements: element idx goes to
 * dst[idx / run] + idx % run (dst: DEV int64[n_lead * n_tail / run]; run must
 * divide n_lead * n_tail; the destination ranges must not overlap).  Map-valued
 * results (the Spatial* metrics, metrics.py:304-374) split by lead time need
 * one entry per slab instead of one per grid point.  count may be NULL when
 * skipna == 0 (every element gains n_time: a caller may keep that number on
 * the host). */
int wb2_time_accumulate_runs(int dtype, const void* values, int64_t n_lead,
                             int64_t n_time, int64_t n_tail, int skipna,
                             const int64_t* dst, int64_t run, double* sum,
                             double* count, void* stream);

/*
 * One chunk of the deterministic suite in ONE call: K1 (wb2_stream_partials_ex /
 * _addr) -> K2 (wb2_det_combine) -> the running temporal mean
 * (wb2_time_accumulate_scatter), enqueued back to back on `stream`.  This is
 * what evaluation._evaluate_chunk + TemporalMean do for one chunk
 * (evaluation.py:583-599, 735-744) and what a caller that evaluates many
 * chunks on one grid should call: the host work per chunk is one function call
 * (no per-kernel argument marshalling), and the kernels and their bits are
 * exactly those of the three separate entry points.
 *
 * `plan` collects the tables that do not change between chunks of one grid and
 * region set (every pointer DEV, meanings as in wb2_stream_partials_ex /
 * wb2_det_combine); the struct itself lives in HOST memory and is only read
 * during the call.
 *
 *  in / slab     as wb2_stream_partials_ex; in == NULL selects the by-address
 *                form (slab[i][o] = byte address of input i's slab o,
 *                `aligned16` as in wb2_stream_partials_addr; ignored otherwise)
 *  partials      DEV scratch, double[n_outer][n_chunk][nwf][n_ts][K]
 *  metrics       DEV double[NM][n_region][n_outer] (out; NM = WB2_NMETRIC for
 *                DET / DET_ACC / WIND, KQ for the generic modes)
 *  sum / count   DEV accumulators or NULL (no temporal accumulation).  `metrics`
 *                is then read as [acc_lead][acc_time][acc_tail] (their product
 *                must be NM * n_region * n_outer) and summed over acc_time into
 *                sum / count [acc_lead][acc_tail], through `dst` if not NULL,
 *                NaNs skipped when acc_skipna != 0 -- wb2_time_accumulate_scatter.
 */
typedef struct wb2_plan_tables {
  int32_t n_row, n_col;
  int32_t n_chunk, n_ctile;
  int32_t n_seg, n_ts;
  int32_t n_band, n_region;
  const double* w_row;
  const double* w_col;
  const void* wfield;          /* NULL: no 2-D weight field (nwf = 1) */
  int32_t wfield_dtype;        /* WB2_F64 / WB2_F32 */
  int32_t reserved;            /* 0 */
  const double* aux;           /* WB2_MODE_SEEPS: p1; else NULL */
  double scalar;               /* WB2_MODE_SEEPS: dry threshold */
  const int32_t* chunk_row0;
  const int32_t* chunk_nrow;
  const int32_t* seg_col0;
  const int32_t* seg_eoff;
  const int32_t* band_chunk0;
  const double* coef_band;
  const double* coef_seg;
  const int32_t* region_wf;
  const double* region_wsum;
} wb2_plan_tables;

int wb2_det_suite_step(const wb2_plan_tables* plan, int mode, int dtype,
                       int skipna, const void* const* in,
                       const int64_t* const* slab, int aligned16,
                       int64_t n_outer, double* partials, double* metrics,
                       int64_t acc_lead, int64_t acc_time, int64_t acc_tail,
                       int acc_skipna, const int64_t* dst, double* sum,
                       double* count, void* stream);

/* wb2_det_suite_step for a launch with wind-vector pairs
 * (wb2_stream_partials_pairs): K1 over the slabs outside the pairs, the pair
 * kernel, K2 over all n_outer slabs -> metrics[WB2_NMETRIC][n_region][n_outer],
 * K2 in mode WIND over the pairs -> wind_metrics[WB2_NMETRIC][n_region][n_pair]
 * (MSE and RMSE rows meaningful).  No accumulation step: the caller adds the
 * chunk's values where it wants them (wb2_gather_accumulate). */
int wb2_det_wind_suite_step(const wb2_plan_tables* plan, int mode, int dtype,
                            int skipna, const void* const* in,
                            const int64_t* const* slab, int aligned16,
                            int64_t n_outer, int64_t n_pair, double* partials,
                            double* wind_partials, double* metrics,
                            double* wind_metrics, void* stream);

/* The running temporal mean of a whole chunk result in ONE launch (what
 * TemporalMean / xbeam.Mean does with the Dataset _evaluate_chunk returns,
 * evaluation.py:583-599, 735-744, for every variable of it at once): output
 * element e sums, over the chunk's n_time time steps in order,
 *   v = src[e * n_time + t] < 0 ? NaN : arena[src[e * n_time + t]]
 * (rounded to float32 first where round32[e] != 0: the reference's float32
 * result dtype) into the accumulators at the DEVICE ADDRESSES sum_addr[e] /
 * count_addr[e] (distinct per e), NaNs skipped when skipna != 0.  `arena` holds
 * the `metrics` outputs of the chunk's wb2_det_suite_step calls side by side;
 * the caller derives `src` once per chunk structure (which result element
 * reads which metrics element).  All pointers DEV. */
int wb2_gather_accumulate(const double* arena, const int32_t* src,
                          const uint8_t* round32, int64_t n_out, int64_t n_time,
                          int skipna, const int64_t* sum_addr,
                          const int64_t* count_addr, void* stream);

/* wb2_gather_accumulate for a sink that KEEPS the time steps
 * (`temporal_mean=False`, evaluation.py:735: the reference skips TemporalMean
 * and the output keeps init_time): entry e is added rows[sel[e]] * rsz8[e]
 * BYTES behind sum_addr[e] / count_addr[e].  `rows` (DEV int64) are the storage
 * rows of the chunk's (time, lead) label pairs -- all a chunk brings --, `sel`
 * (DEV int32[n_out]) which of them entry e belongs to and `rsz8` (DEV
 * int64[n_out]) the bytes of one row of e's storage: structural. */
int wb2_gather_accumulate_rows(const double* arena, const int32_t* src,
                               const uint8_t* round32, int64_t n_out,
                               int64_t n_time, int skipna,
                               const int64_t* sum_addr,
                               const int64_t* count_addr, const int64_t* rows,
                               const int32_t* sel, const int64_t* rsz8,
                               void* stream);

/*
 * Chunk programs: every launch of one chunk STRUCTURE recorded once, replayed
 * per chunk in ONE call -- what _evaluate_chunk + TemporalMean do per chunk
 * (evaluation.py:583-599, 735-744), 2 920 x 40 times in the official 0.25
 * degree run, with nothing changing between chunks but the addresses of their
 * arrays and their valid times.
 *
 *  wb2_program_add_launch   one K1 + K2 step (wb2_det_suite_step, or
 *      wb2_det_wind_suite_step when n_pair > 0) over n_outer slabs given by
 *      address: input j of slab o is read at ptrs[slot[j * n_outer + o]] +
 *      rel[j * n_outer + o] (slot / rel: HOST, copied).  `partials` (and
 *      `wind_partials`) are the launch's DEV scratch, sized as for the suite
 *      step; its metrics are written at arena + arena_offset (per-variable
 *      block, then the wind block).  side_stream != 0: the launch runs on the
 *      program's second stream beside the others (small, latency-bound
 *      passes) and is joined before the sinks.
 *  wb2_program_add_ens_launch   one ensemble pass (wb2_ens_partials_addr +
 *      wb2_ens_combine: the `probabilistic` config, scripts/evaluate.py:496-520)
 *      over n_outer slabs: slot / rel [2][n_outer] = (member 0's slab, truth
 *      slab) of every outer index; `plan` carries the ENSEMBLE tile geometry
 *      (n_ctile, seg_eoff, n_ts for wb2_ens_tile_cols(n_col); wfield float64 or
 *      NULL); `partials` sized as for wb2_ens_partials; the WB2_NMETRIC_ENS x
 *      n_region x n_outer values are written at arena + arena_offset.
 *  wb2_program_add_gather   entries [first, first + count) of input `input` of
 *      the launch added LAST are read from a resident array by valid time
 *      (climatology.sel(dayofyear, hour), metrics.py:398-404) instead:
 *      address = ptrs[source] + (values[value_offset + cell[k]] + base[k]) *
 *      step_bytes -- values = the slab number of every (time, lead) cell of
 *      the chunk (per chunk), cell / base structural (HOST, copied).
 *  wb2_program_add_sink     one wb2_gather_accumulate over the arena per eval
 *      config (src / round32: DEV, kept by the caller); sel / rsz8 != NULL:
 *      wb2_gather_accumulate_rows with at most max_rows rows per chunk.
 *  wb2_program_finalize     arena = DEV double[...], the launches' outputs side
 *      by side; n_ptrs / n_values = lengths every replay must bring.
 *  wb2_program_replay       ptrs / values: HOST.  sink_args: HOST int64[3] per
 *      sink = {sum table address, count table address (DEV int64 tables as in
 *      wb2_gather_accumulate), number of rows of this sink in `rows`}; rows:
 *      HOST int64[n_rows], the kept sinks' rows one sink after the other.
 *      Everything is enqueued on `stream` (use ONE stream per program); the
 *      host only waits when it is more than four replays ahead of the copies.
 * Results are those of the separate calls, bit for bit (same kernels, same
 * order of additions). */
int wb2_program_create(void** program);
int wb2_program_destroy(void* program);
int wb2_program_add_launch(void* program, const wb2_plan_tables* plan, int mode,
                           int dtype, int skipna, int32_t n_in, int64_t n_outer,
                           int64_t n_pair, const int32_t* slot,
                           const int64_t* rel, double* partials,
                           double* wind_partials, int64_t arena_offset,
                           int side_stream);
int wb2_program_add_ens_launch(void* program, const wb2_plan_tables* plan,
                               int dtype, int skipna, int32_t n_member,
                               int64_t member_stride, int64_t n_outer,
                               const int32_t* slot, const int64_t* rel,
                               double* partials, int64_t arena_offset);
int wb2_program_add_gather(void* program, int32_t input, int64_t first,
                           int64_t count, int32_t source, int64_t step_bytes,
                           int32_t value_offset, const int32_t* cell,
                           const int64_t* base);
int wb2_program_add_sink(void* program, const int32_t* src,
                         const uint8_t* round32, int64_t n_out, int64_t n_time,
                         int skipna, const int32_t* sel, const int64_t* rsz8,
                         int64_t max_rows);
int wb2_program_finalize(void* program, double* arena, int32_t n_ptrs,
                         int32_t n_values);
int wb2_program_replay(void* program, const int64_t* ptrs, int32_t n_ptrs,
                       const int64_t* values, int32_t n_values,
                       const int64_t* sink_args, const int64_t* rows,
                       int32_t n_rows, void* stream);
/* Host seconds the replays of `program` have spent, by phase: seconds[0]
 * waiting for a free table slot (the GPU is more than four replays behind:
 * the run is GPU-bound), [1] filling the address table, [2] its copy, [3] the
 * launches, [4] the sinks; *replays = number of calls. */
int wb2_program_stats(void* program, double* seconds, int64_t* replays);

/*
 * The energy score in ONE read of the ensemble (metrics.py:1403-1517;
 * scripts/evaluate.py:541-565 evaluates score, spread and skill per chunk):
 *   skill  = mean_m     sqrt(_spatial_average((x_m - y)^2))
 *   spread = mean_{m<M-1} sqrt(_spatial_average((x_m - x_{m+1})^2))   (0 for M = 1)
 *   score  = skill - 0.5 spread
 * for every region of `plan` at once: out[3][n_region][n_outer] = (score,
 * spread, skill), fp64.  The 2 M - 1 weighted sums per region are accumulated
 * by blocks of `block` consecutive members (one wave per block, the blocks of a
 * row chunk side by side in one workgroup: HBM traffic (M + 1) elements per
 * point), folded by the combine kernel and finished by a small third kernel.
 * skipna as in _spatial_average / .mean(ensemble_dim, skipna): NaNs drop out of
 * the spatial means and NaN members out of the member means.
 *
 *  plan      tables as for wb2_det_suite_step, with n_ctile / n_ts / seg_eoff
 *            for the ENSEMBLE tile width (wb2_ens_tile_cols), wfield float64
 *  ens, ens_slab, truth, truth_slab, n_member, member_stride: as wb2_ens_partials
 *  partials  DEV scratch double[n_outer * n_block][n_chunk][nwf][n_ts][n_slot]
 *  means     DEV scratch double[2 * block][n_region][n_outer * n_block]
 *            (block, n_block, n_slot from wb2_energy_layout)
 */
int wb2_energy_layout(int32_t n_member, int skipna, int has_wfield,
                      int32_t* block, int32_t* n_block, int32_t* n_slot);
int wb2_energy_score(int dtype, int skipna, const void* ens,
                     const int64_t* ens_slab, const void* truth,
                     const int64_t* truth_slab, int32_t n_member,
                     int64_t member_stride, int64_t n_outer,
                     const wb2_plan_tables* plan, double* partials,
                     double* means, double* out, void* stream);

/*
 * K5: the Spatial* metrics (no spatial reduction): SpatialBias / SpatialMSE /
 * SpatialMAE (weatherbench2/metrics.py:304-374).
 *
 * wb2_spatial_maps: per-time maps, written in the INPUT dtype like the
 *   reference's elementwise arrays; slab o of `forecast`/`truth` is resolved
 *   through the optional int64[n_outer] tables; any of bias/mse/mae may be NULL.
 *   Outputs are [n_outer][n_point].
 * wb2_spatial_accumulate: the temporal mean of those maps (Metric.compute
 *   :117-138; xbeam.Mean evaluation.py:740-744) without materialising them:
 *   adds, for every (rest, point), the sum over the chunk's n_time steps of
 *   d, d^2, |d| (d = forecast - truth in the input dtype) to
 *   sum[3][n_rest][n_point] (order bias, mse, mae) and, when skipna, the number
 *   of non-NaN terms to count[3][n_rest][n_point].  Slab of (time i, rest j) =
 *   table[i * n_rest + j] (identity when NULL).
 * wb2_spatial_accumulate_addr: the same for EVERY variable of a chunk (or of a
 *   window of chunks) in one launch, into accumulators that are separate
 *   allocations -- what `deterministic_spatial` (scripts/evaluate.py:471-478)
 *   does per `init_time=1,lead_time=1` chunk before xbeam.Mean.  Destination j
 *   (a (variable, lead, level) slab of the running mean) receives the chunk's
 *   n_time steps in order; the sums CONTINUE from the accumulators value by
 *   value (sum = ((sum + v_0) + v_1) ...: the bits do not depend on how many
 *   steps a call brings).
 *     f_addr, t_addr  DEV int64[n_time][n_dst]: byte address of the slab
 *                     (n_point elements of `dtype`) of step i for destination j
 *     sum_addr        DEV int64[3][n_dst]: byte address of double[n_point] for
 *                     bias / mse / mae of destination j; 0 = not wanted
 *     count_addr      the same for the counts of non-NaN terms (skipna != 0
 *                     only; NULL otherwise: the caller adds n_time itself)
 *     aligned16       != 0: every slab address is 16-byte aligned and n_point
 *                     is a multiple of 16 / sizeof(dtype)
 */
int wb2_spatial_accumulate_addr(int dtype, int skipna, int aligned16,
                                const int64_t* f_addr, const int64_t* t_addr,
                                int64_t n_time, int64_t n_dst, int64_t n_point,
                                const int64_t* sum_addr,
                                const int64_t* count_addr, void* stream);
int wb2_spatial_maps(int dtype, const void* forecast, const int64_t* f_slab,
                     const void* truth, const int64_t* t_slab, int64_t n_outer,
                     int64_t n_point, void* bias, void* mse, void* mae,
                     void* stream);
int wb2_spatial_accumulate(int dtype, int skipna, const void* forecast,
                           const int64_t* f_slab, const void* truth,
                           const int64_t* t_slab, int64_t n_time,
                           int64_t n_rest, int64_t n_point, double* sum,
                           double* count, void* stream);

/*
 * K4: zonal energy spectrum, ZonalEnergySpectrum.compute
 * (weatherbench2/derived_variables.py:592-626): batched real-to-complex FFT
 * along longitude (rocFFT through hipFFT), then
 *   S[row, k] = |F[k] / n_lon|^2 * (k == 0 ? 1 : 2) * circumference[row % n_lat]
 * in fp64 (numpy widens float32 power * int64 to float64 at the same point).
 *
 *  plan        host handle for (dtype, n_lon, n_rows) -- create once, reuse.
 *              For the 22 row lengths with a one-kernel LDS transform
 *              (spectrum_fused.hip) creation makes twiddle tables only; the
 *              hipFFT plan behind the other lengths takes seconds (rocFFT
 *              compiles its kernels at run time) and is made at creation --
 *              or, for a one-kernel length, by the first call that cannot take
 *              that kernel (x not 16-byte aligned, n_time >= 65536).
 *  x           DEV [n_rows][n_lon], longitude contiguous; rows ordered
 *              (time, ..., latitude) with latitude the fastest row index
 *  circumference DEV double[n_lat] = cos(lat * pi / 180) * 2 pi R   (:578-581)
 *  n_time      0: out is double[n_rows][n_lon/2+1] (what compute() returns);
 *              > 0: rows are [n_time][n_rows / n_time] and out is the mean over
 *              time, double[n_rows / n_time][n_lon/2+1]
 *              (scripts/compute_zonal_energy_spectrum.py:234); skipna as in
 *              xbeam.Mean.
 *  workspace   DEV, wb2_spectrum_plan_workspace(plan) bytes, 256-byte aligned
 */
int wb2_spectrum_plan_create(int dtype, int32_t n_lon, int64_t n_rows,
                             void** plan);
int wb2_spectrum_plan_destroy(void* plan);
int64_t wb2_spectrum_plan_workspace(void* plan);
int wb2_zonal_spectrum(void* plan, const void* x, const double* circumference,
                       int32_t n_lat, int64_t n_time, int skipna, double* out,
                       void* workspace, void* stream);

/* wb2_rank_histogram with the reference's SEEDED tie breaking reproduced
 * (metrics.py:1955-1980: np.random.default_rng(seed).uniform over the
 * concatenated [truth, members] array): pcg_state_inc[4] (HOST) = the 128-bit
 * state and increment of np.random.PCG64(seed) as (state hi, state lo, inc hi,
 * inc lo); element (outer o, row r, col c, j) of the reference's concatenated
 * array (j = 0 truth, j >= 1 member j - 1; n_point = n_row * n_col points per
 * slab) has the C-order index
 *   ref_outer_off[o] (DEV int64[n_outer]) + r ref_strides[0] + c ref_strides[1]
 *   + j ref_strides[2]      (ref_strides: HOST int64[3]),
 * which is where in NumPy's stream its perturbation comes from.  Everything else
 * as wb2_rank_histogram with break_ties = 1. */
int wb2_rank_histogram_seeded(
    int dtype, const void* ens, const int64_t* ens_slab, const void* truth,
    const int64_t* truth_slab, int32_t n_member, int64_t member_stride,
    int64_t n_outer, int64_t n_point, int32_t n_col, int32_t n_bins,
    const uint64_t* pcg_state_inc, const int64_t* ref_outer_off,
    const int64_t* ref_strides, const int64_t* acc_row, double* out,
    void* stream);

/* The histogram summed (mean == 0) or averaged (mean != 0: counts / n_time,
 * the temporal mean of the one-hots, Metric.compute metrics.py:117-138) over
 * the MIDDLE axis of the outer index o = (l * n_time + t) * n_tail + j, without
 * the per-sample one-hots and without atomics (a wave owns 64 points of a
 * result row, counts in LDS): out[n_lead * n_tail][n_point][n_bins] is WRITTEN,
 * every element once.  Samples, slab tables, ranks, ties as above;
 * pcg_state_inc != NULL selects the seeded (NumPy PCG64) ties of
 * wb2_rank_histogram_seeded and needs ref_outer_off[n_lead * n_time * n_tail],
 * ref_strides and n_col.  n_bins <= 256. */
int wb2_rank_histogram_mean(
    int dtype, const void* ens, const int64_t* ens_slab, const void* truth,
    const int64_t* truth_slab, int32_t n_member, int64_t member_stride,
    int64_t n_lead, int64_t n_time, int64_t n_tail, int64_t n_point,
    int32_t n_col, int32_t n_bins, int break_ties, uint64_t seed,
    const uint64_t* pcg_state_inc, const int64_t* ref_outer_off,
    const int64_t* ref_strides, int mean, double* out, void* stream);

/* BASELINE configs[3] in one pass: the latitude-weighted mean of the zonal energy
 * spectrum WITHOUT materialising the per-latitude spectra,
 *   out[field][k] = scale * sum_lat row_weight[lat] * S[field][lat][k],
 * S = ZonalEnergySpectrum.compute (derived_variables.py:592-626) of the plan's
 * rows viewed as [field][n_lat]; row_weight[n_lat] (DEV, float64) = latitude
 * weight x circumference (the x circumference of :626 folded in), scale =
 * 1 / sum(latitude weights) for the area-weighted mean.  Every (field, latitude
 * segment) is reduced by one wave in registers into partial[field][n_seg][bins]
 * (DEV scratch, float64), the segments are then added in order (deterministic).
 * n_seg from wb2_zonal_spectrum_latmean_segments (0 = this plan has no fused
 * path: use wb2_zonal_spectrum + wb2_axis_moments); out[field][n_lon/2+1] DEV. */
int wb2_zonal_spectrum_latmean_segments(void* plan, int32_t n_lat);
int wb2_zonal_spectrum_latmean(void* plan, const void* x,
                               const double* row_weight, int32_t n_lat,
                               int32_t n_seg, double scale, double* partial,
                               double* out, void* stream);

/* ---------------------------------------------------------------------------
 * Host-side helpers and the path's one exchange step (csrc/comm.cpp)
 * ------------------------------------------------------------------------- */

/* Latitude / area weights, normalised to mean 1: replaces get_lat_weights
 * (weatherbench2/metrics.py:35-60).  `latitude` (HOST, increasing, degrees) and
 * `out` (HOST, n values) have element type `dtype`; the arithmetic runs in that
 * type like NumPy's does (metrics.py:41, 57: the weights inherit the coordinate
 * dtype).  float64: the C library's sin -- bit-identical to NumPy where NumPy
 * calls libm, else within 1-2 ulp; float32: NumPy's SIMD sinf differs from libm's
 * in the last ulp, which the cell-area difference sin(upper) - sin(lower)
 * amplifies next to the poles (<= 5e-5 relative there).  The Python host keeps
 * using NumPy itself (plan.get_lat_weights). */
int wb2_lat_weights(int dtype, const void* latitude, int64_t n, void* out);

/* Staging of PAGEABLE host chunks: the Beam pipeline hands _evaluate_chunk
 * NumPy-backed datasets (weatherbench2/evaluation.py:583-599, 693-705), i.e.
 * malloc'ed pages a DMA engine cannot read.  An uploader owns a ring of
 * `n_slots` page-locked slots of `slot_bytes` each and a pool of `n_threads`
 * copy threads: wb2_uploader_upload cuts the source into slot-sized slices,
 * the pool copies slice k + 1 into a free slot while the DMA of slice k
 * (hipMemcpyAsync on `stream`) is in flight.  The call returns once the last
 * slice has been STAGED -- `src` may be reused or freed; `dst` (DEV) is
 * complete in `stream` order.  One uploader serves one calling thread at a
 * time.  (HOST pointers: uploader_out, src.) */
int wb2_uploader_create(int32_t n_threads, int64_t slot_bytes, int32_t n_slots,
                        void** uploader_out);
/* The pool's copy on its own: dst[0:nbytes] = src[0:nbytes] (HOST, HOST) by
 * n_threads threads over 4 KiB-aligned pieces; for callers that keep their own
 * pinned buffers (feeder.ChunkFeeder). */
int wb2_host_copy(void* dst, const void* src, int64_t nbytes,
                  int32_t n_threads);
int wb2_uploader_destroy(void* uploader);
int wb2_uploader_upload(void* uploader, void* dst, const void* src,
                        int64_t nbytes, void* stream);
/* The same for the n buffers of one chunk (its variables) in ONE call:
 * dst[i] (DEV) <- src[i] (HOST), nbytes[i] each; the arrays themselves are
 * HOST.  A Python caller holds no interpreter lock for the whole chunk. */
int wb2_uploader_upload_many(void* uploader, int32_t n, void* const* dst,
                             const void* const* src, const int64_t* nbytes,
                             void* stream);

/* The way back: results that leave for the host (the float64 time-mean maps
 * of the Spatial* metrics are gigabytes per variable; the reference writes
 * them out after xbeam.Mean, weatherbench2/evaluation.py:735-752).  dst (HOST,
 * pageable) <- src (DEV), nbytes, through the same ring: the DMAs of the next
 * n_slots - 1 slices run on `stream` (behind whatever produced src there) while
 * the pool copies the slice that has arrived out of its slot.  Returns when
 * dst is complete. */
int wb2_uploader_download(void* uploader, void* dst, const void* src,
                          int64_t nbytes, void* stream);

/* RCCL communicator for callers that do not use torch.distributed: rank 0 makes
 * a 128-byte id (wb2_comm_unique_id), distributes it by any means (file, MPI,
 * socket), every rank calls wb2_comm_init_rank with the device it computes on
 * current (hipSetDevice).  `*comm_out` is an ncclComm_t. */
int wb2_comm_unique_id(void* id128);
int wb2_comm_init_rank(const void* id128, int32_t n_ranks, int32_t rank,
                       void** comm_out);
int wb2_comm_destroy(void* comm);

/* The all-reduce of the temporal mean over init-time shards: replaces the
 * combiner of xbeam.Mean (weatherbench2/evaluation.py:735-744) across ranks.
 * sum[n], count[n] (DEV, float64, e.g. the accumulators of wb2_time_accumulate)
 * are summed IN PLACE over all ranks of `comm` (an ncclComm_t, from
 * wb2_comm_init_rank or any other RCCL initialisation) as one grouped RCCL
 * operation on `stream`; the caller then divides sum / count.  Every rank must
 * call it with the same n. */
int wb2_time_mean_allreduce(double* sum, double* count, int64_t n, void* comm,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif  /* WB2HIP_H_ */
