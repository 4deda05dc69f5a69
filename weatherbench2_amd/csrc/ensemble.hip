// K3 of libwb2hip.so: fused ensemble metrics.  One read of the M members (+ the
// truth) feeds CRPS skill / spread, ensemble-mean MSE, ensemble variance,
// stddev^2 and the debiased MSE, for every region at once.
//
// Replaces (reference = /root/reference/weatherbench2/metrics.py):
//   :532-565   _debiased_ensemble_mean_mse      (mean, var(ddof=1) over members)
//   :781-813   _pointwise_crps_spread           (argsort ranks, sum (2r-M-1) x)
//   :816-824   _pointwise_crps_skill            (mean |t - x|)
//   :827-846   _rank_ds / _rankdata             (the argsort + put_along_axis)
//   :1161-1363 EnsembleStddev / Variance / MeanRMSE / MeanMSE / DebiasedMeanMSE
// followed by the same _spatial_average (:141-163) as the deterministic path.
//
// One lane owns one grid point (column) of the current row and keeps its M
// member values in VGPRs; ranks never materialise: the members are sorted with
// a straight-line Batcher network (sort_networks.inc) and the rank-weighted sum
// is taken in sorted order.  Member statistics are computed in the input dtype
// in member order (numpy reduces the leading axis sequentially); the
// rank-weighted sum is fp64 (numpy: int64 * float32 -> float64); all spatial
// sums are fp64.
//
// Slots: 0 skill, 1 spread, 2 (t-mean)^2, 3 var, 4 std^2, 5 debiased
//        [skipna: 6 n(skill,mse), 7 n(spread), 8 n(var,std^2), 9 n(debiased)]

#include "ensemble_kernels.hpp"

namespace wb2 {
namespace {

// ---------------------------------------------------------------------------
// Ensemble threshold metrics (metrics.py:1524-1891): exceedance counts only.
// Slots: 0 Brier, 1 debiased Brier, 2 ignorance, 3 RPS part [+ 4 notnull].
// ---------------------------------------------------------------------------
struct EnsThrParams {
  EnsParams e;
  const void* thr;
  const long long* thr_slab;
};

// One grid point of the ensemble threshold metrics: v = (Brier, debiased
// Brier, ignorance, RPS part).  `xb` points at member 0 of this point.
template <typename T, bool SKIPNA>
__device__ __forceinline__ void ens_thr_point(const T* xb,
                                              long long member_stride, int M,
                                              T t, T thr, double (&v)[4]) {
  const double nan = __builtin_nan("");
  int gt = 0, lt = 0, nn = 0;  // members above / below / not NaN
  int m = 0;
  for (; m + 4 <= M; m += 4) {  // four loads in flight
    T x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      x[u] = __builtin_nontemporal_load(xb + (m + u) * member_stride);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      gt += x[u] > thr ? 1 : 0;
      lt += x[u] < thr ? 1 : 0;
      nn += is_nan(x[u]) ? 0 : 1;
    }
  }
  for (; m < M; ++m) {
    const T x = __builtin_nontemporal_load(xb + m * member_stride);
    gt += x > thr ? 1 : 0;
    lt += x < thr ? 1 : 0;
    nn += is_nan(x) ? 0 : 1;
  }
  // metrics.py:1535-1560: probabilities are NaN where the input is NaN
  // (a NaN threshold makes the comparison false, like xr.where does)
  const double tp_b = is_nan(t) ? nan : (t > thr ? 1.0 : 0.0);
  const int n = SKIPNA ? nn : M;
  double pm = (double)gt / (double)n;           // n == 0 -> NaN
  if (!SKIPNA && nn != M) pm = nan;
  const double eb = pm - tp_b;
  const double brier = eb * eb;
  // var(ddof=1) of the 0/1 member probabilities (two-pass form)
  double var = ((double)gt * (1.0 - pm) * (1.0 - pm) +
                (double)(n - gt) * pm * pm) / (double)(n - 1);
  if (n <= 1) var = nan;
  const double debiased = brier - var / (double)M;
  // metrics.py:1728-1738 / 1799-1802: plain comparisons (NaN -> 0)
  const double pe = (double)gt / (double)M, pl = (double)lt / (double)M;
  const double ign = -((t > thr) ? log(pe) : log(1.0 - pe));
  const double dr = pl - ((t < thr) ? 1.0 : 0.0);
  v[0] = brier;
  v[1] = debiased;
  v[2] = ign;
  v[3] = dr * dr;
}

template <typename T, bool SKIPNA, bool WF>
__global__ void __launch_bounds__(256)
    ens_threshold_kernel(const EnsThrParams q) {
  const EnsParams& p = q.e;
  constexpr int K = SKIPNA ? 8 : 4, NWF = WF ? 2 : 1;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int nwave = blockDim.x / kWave;
  const unsigned bx = blockIdx.x;
  const unsigned tblk = bx / (unsigned)p.n_chunk;
  const long long o = (long long)blockIdx.z * gridDim.y + blockIdx.y;
  // chunks rotated by the slab number: a fixed chunk -> XCD map (workgroup x
  // lands on XCD x % 8) would give some XCDs the short chunks of every slab
  const int chunk = (int)(((long long)(bx - tblk * (unsigned)p.n_chunk) +
                           (WB2_ROTATE_CHUNKS ? o : 0)) % p.n_chunk);
  const int tile = (int)tblk * nwave + wave;
  const int row0 = p.chunk_row0[chunk];
  const int nrow = p.chunk_nrow[chunk];
  const int col0 = tile * kWave + lane;
  const bool active = tile < p.n_ctile && col0 < p.n_col;
  if (nrow <= 0 || tile >= p.n_ctile || o >= p.n_outer) return;
  const long long es = p.ens_slab ? p.ens_slab[o] : o;
  const long long ts = p.truth_slab ? p.truth_slab[o] : o;
  const long long hs = q.thr_slab ? q.thr_slab[o] : o;
  const int M = p.n_member;

  double acc[NWF][1][K];
#pragma unroll
  for (int w = 0; w < NWF; ++w)
#pragma unroll
    for (int k = 0; k < K; ++k) acc[w][0][k] = 0.0;

  if (active) {
    const long long slab_elems = (long long)p.n_row * p.n_col;
    const T* xb = static_cast<const T*>(p.ens) + es * slab_elems +
                  (long long)row0 * p.n_col + col0;
    const T* tb = static_cast<const T*>(p.truth) + ts * slab_elems +
                  (long long)row0 * p.n_col + col0;
    const T* hb = static_cast<const T*>(q.thr) + hs * slab_elems +
                  (long long)row0 * p.n_col + col0;
    const double* wfp = WF ? p.wfield + (long long)row0 * p.n_col + col0
                           : nullptr;
    for (int r = 0; r < nrow; ++r) {
      const long long off = (long long)r * p.n_col;
      const T t = __builtin_nontemporal_load(tb + off);
      const T thr = __builtin_nontemporal_load(hb + off);
      double v[4];
      ens_thr_point<T, SKIPNA>(xb + off, p.member_stride, M, t, thr, v);
      const double wr = p.w_row[row0 + r];
      double wf = 1.0;
      if constexpr (WF) wf = wfp[off];
      const bool inside = !WF || wf > 0.0;
      const double w2 = inside ? wr * wf : 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double xk = v[k], ck = 1.0;
        if constexpr (SKIPNA) {
          const bool ok = !is_nan(v[k]);
          xk = ok ? v[k] : 0.0;
          ck = ok ? 1.0 : 0.0;
        }
        acc[0][0][k] = __builtin_fma(wr, xk, acc[0][0][k]);
        if constexpr (SKIPNA)
          acc[0][0][4 + k] = __builtin_fma(wr, ck, acc[0][0][4 + k]);
        if constexpr (WF) {
          acc[1][0][k] = __builtin_fma(w2, inside ? xk : 0.0, acc[1][0][k]);
          if constexpr (SKIPNA)
            acc[1][0][4 + k] =
                __builtin_fma(w2, inside ? ck : 0.0, acc[1][0][4 + k]);
        }
      }
    }
    if (p.w_col) {
      const double wc = p.w_col[col0];
#pragma unroll
      for (int w = 0; w < NWF; ++w)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[w][0][k] *= wc;
    }
  }
  fold_tile_to_segs<NWF, 1, K>(
      acc, lane, tile, col0, col0, p.n_col, p.seg_col0, p.seg_eoff, p.n_seg,
      p.n_ts,
      p.partials + (o * p.n_chunk + chunk) * (long long)(NWF * p.n_ts * K));
}

// Pointwise maps of the same four scores (Spatial* threshold metrics,
// metrics.py:1615-1638, 1697-1719, 1780-1802, 1870-1891): one lane per grid
// point, no weights / regions (the reference ignores `region` here).
struct EnsThrMapParams {
  const void* ens;
  const void* truth;
  const void* thr;
  const long long* ens_slab;
  const long long* truth_slab;
  const long long* thr_slab;
  double* maps;  // [4][n_outer][n_point]
  long long member_stride, n_outer, n_point;
  int n_member;
};

template <typename T, bool SKIPNA>
__global__ void __launch_bounds__(256)
    ens_threshold_maps_kernel(const EnsThrMapParams p) {
  const long long pt = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long o = blockIdx.y + (long long)blockIdx.z * gridDim.y;
  if (pt >= p.n_point || o >= p.n_outer) return;
  const long long es = p.ens_slab ? p.ens_slab[o] : o;
  const long long ts = p.truth_slab ? p.truth_slab[o] : o;
  const long long hs = p.thr_slab ? p.thr_slab[o] : o;
  const T t = __builtin_nontemporal_load(static_cast<const T*>(p.truth) +
                                         ts * p.n_point + pt);
  const T thr = __builtin_nontemporal_load(static_cast<const T*>(p.thr) +
                                           hs * p.n_point + pt);
  double v[4];
  ens_thr_point<T, SKIPNA>(static_cast<const T*>(p.ens) + es * p.n_point + pt,
                           p.member_stride, p.n_member, t, thr, v);
  const long long plane = p.n_outer * p.n_point;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    __builtin_nontemporal_store(v[k], p.maps + k * plane + o * p.n_point + pt);
}

template <typename T>
int launch_ens_threshold(const EnsThrParams& q, bool skipna, bool wf,
                         hipStream_t stream) {
  const EnsParams& p = q.e;
  int nwave = p.n_ctile < WB2_ENS_WG_WAVES ? p.n_ctile : WB2_ENS_WG_WAVES;
  const int n_tblk = (p.n_ctile + nwave - 1) / nwave;
  const long long gy = p.n_outer < 32768 ? p.n_outer : 32768;
  const long long gz = (p.n_outer + gy - 1) / gy;
  const dim3 grid((unsigned)(p.n_chunk * n_tblk), (unsigned)gy, (unsigned)gz);
  const dim3 block(nwave * kWave);
#define WB2_L(S, W) \
  hipLaunchKernelGGL((ens_threshold_kernel<T, S, W>), grid, block, 0, stream, q)
  if (skipna) {
    if (wf) WB2_L(true, true); else WB2_L(true, false);
  } else {
    if (wf) WB2_L(false, true); else WB2_L(false, false);
  }
#undef WB2_L
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

template <typename T>
int launch_ens_npad(const EnsParams& p, bool skipna, bool wf,
                    hipStream_t stream) {
  const int m = p.n_member;
  // exact-size kernel for the operational ensemble size (IFS ENS: 50 members).
  // Its float32 / no-skipna instantiation is the ONE place where K3 is not
  // operation for operation the reference's arithmetic (rank-weighted sum over
  // (hi, lo) pairs in float32, ens_point): WB2HIP_ENS_REFERENCE_SPREAD=1 takes
  // the padded 64-network with the reference's fp64 chain (int64 x float32 ->
  // float64, metrics.py:806-812) instead -- slower, selectable, tested beside
  // the default (tests/test_ens_gpu.py).
  static const bool reference_spread = [] {
    const char* v = getenv("WB2HIP_ENS_REFERENCE_SPREAD");
    return v && v[0] && v[0] != '0';
  }();
  if (m == 50 && !reference_spread && !p.member_ptr)  // gathers: runtime-M forms
    return launch_ens<T, 64, 50>(p, skipna, wf, stream);
  if constexpr (sizeof(T) == 4) {
    // the other member counts with kernels of their own (float32, members at
    // a constant stride): compile-time M, no selects, a 2-/3-sorter program
    if (!p.member_ptr) {
      switch (m) {
#define WB2_ENS_CASE(M, NPAD) \
  case M:                     \
    return launch_ens_exact_f32_##M(p, skipna, wf, stream);
        WB2_ENS_EXACT_SIZES(WB2_ENS_CASE)
#undef WB2_ENS_CASE
        default: break;
      }
    }
  }
  if constexpr (sizeof(T) == 4) {
    // any other float32 member count up to the largest program (and gathered
    // members of any count): hosted by the smallest exact program that holds
    // it -- the dead slots at +inf -- instead of a padded power-of-two network
    // (WB2HIP_ENS_HOSTED=0: the padded networks, for A/B runs)
    static const bool hosted = [] {
      const char* v = getenv("WB2HIP_ENS_HOSTED");
      return !(v && v[0] == '0');
    }();
    if (hosted && m >= 2) {
#define WB2_ENS_HOST_CASE(M, NPAD) \
  if (m <= M) return launch_ens_hosted_f32_##M(p, skipna, wf, stream);
      WB2_ENS_EXACT_SIZES(WB2_ENS_HOST_CASE)
#undef WB2_ENS_HOST_CASE
    }
  }
  if (m <= 4) return launch_ens<T, 4, 0>(p, skipna, wf, stream);
  if (m <= 16) return launch_ens<T, 16, 0>(p, skipna, wf, stream);
  if (m <= 32) return launch_ens<T, 32, 0>(p, skipna, wf, stream);
  if (m <= 64) return launch_ens<T, 64, 0>(p, skipna, wf, stream);
  if constexpr (sizeof(T) == 4) {
    if (m <= 128) return launch_ens<T, 128, 0>(p, skipna, wf, stream);
  }
  return launch_ens<T, 0, 0>(p, skipna, wf, stream);  // any size, no sort
}

}  // namespace
}  // namespace wb2

extern "C" {

int wb2_ens_num_slots(int skipna) { return skipna ? 10 : 6; }

int wb2_ens_tile_cols(int32_t n_col) {
  (void)n_col;
  return wb2::kWave;
}

int wb2_ens_partials(int dtype, int skipna, const void* ens,
                     const int64_t* ens_slab, const void* truth,
                     const int64_t* truth_slab, int32_t n_member,
                     int64_t member_stride, int64_t n_outer, int32_t n_row,
                     int32_t n_col, const double* w_row, const double* w_col,
                     const double* wfield, const int32_t* chunk_row0,
                     const int32_t* chunk_nrow, int32_t n_chunk,
                     int32_t n_ctile, const int32_t* seg_col0,
                     const int32_t* seg_eoff, int32_t n_seg, int32_t n_ts,
                     double* partials, void* stream) {
  WB2_TRACE();
  return wb2_ens_partials_maps(dtype, skipna, ens, ens_slab, truth, truth_slab,
                               n_member, member_stride, n_outer, n_row, n_col,
                               w_row, w_col, wfield, chunk_row0, chunk_nrow,
                               n_chunk, n_ctile, seg_col0, seg_eoff, n_seg,
                               n_ts, partials, nullptr, stream);
}

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
                          double* maps, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_EMPTY_OK(n_outer);
  WB2_REQUIRE(ens && truth && w_row && chunk_row0 && chunk_nrow && seg_col0 &&
                  seg_eoff && partials,
              "null pointer argument");
  WB2_REQUIRE(n_member >= 1, "n_member=%d", n_member);
  WB2_REQUIRE(n_outer >= 0 && n_row > 0 && n_col > 0 && n_chunk > 0 &&
                  n_seg > 0 && n_ts >= n_seg,
              "bad sizes");
  WB2_REQUIRE(n_chunk % 8 == 0, "n_chunk=%d must be a multiple of 8", n_chunk);
  WB2_REQUIRE(n_outer < (1ll << 31), "n_outer=%lld too large",
              (long long)n_outer);
  WB2_REQUIRE(n_ctile == (n_col + kWave - 1) / kWave,
              "n_ctile=%d does not match ceil(n_col / 64)", n_ctile);
  if (n_outer == 0) return 0;
  EnsParams p{};
  p.ens = ens;
  p.truth = truth;
  p.ens_slab = reinterpret_cast<const long long*>(ens_slab);
  p.truth_slab = reinterpret_cast<const long long*>(truth_slab);
  p.w_row = w_row;
  p.w_col = w_col;
  p.wfield = wfield;
  p.chunk_row0 = chunk_row0;
  p.chunk_nrow = chunk_nrow;
  p.seg_col0 = seg_col0;
  p.seg_eoff = seg_eoff;
  p.partials = partials;
  p.maps = maps;
  p.member_stride = member_stride;
  p.ens_scale = p.truth_scale =
      (long long)n_row * n_col * (dtype == WB2_F32 ? 4 : 8);
  p.n_outer = n_outer;
  p.n_member = n_member;
  p.n_row = n_row;
  p.n_col = n_col;
  p.n_chunk = n_chunk;
  p.n_ctile = n_ctile;
  p.n_seg = n_seg;
  p.n_ts = n_ts;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == WB2_F32)
    return launch_ens_npad<float>(p, skipna != 0, wfield != nullptr, s);
  return launch_ens_npad<double>(p, skipna != 0, wfield != nullptr, s);
}

int wb2_ens_partials_addr(int dtype, int skipna, const int64_t* ens_addr,
                          const int64_t* truth_addr, int32_t n_member,
                          int64_t member_stride, int64_t n_outer, int32_t n_row,
                          int32_t n_col, const double* w_row,
                          const double* w_col, const double* wfield,
                          const int32_t* chunk_row0, const int32_t* chunk_nrow,
                          int32_t n_chunk, int32_t n_ctile,
                          const int32_t* seg_col0, const int32_t* seg_eoff,
                          int32_t n_seg, int32_t n_ts, double* partials,
                          void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_EMPTY_OK(n_outer);
  WB2_REQUIRE(ens_addr && truth_addr && w_row && chunk_row0 && chunk_nrow &&
                  seg_col0 && seg_eoff && partials,
              "null pointer argument");
  WB2_REQUIRE(n_member >= 1 && n_outer >= 0 && n_row > 0 && n_col > 0 &&
                  n_chunk > 0 && n_seg > 0 && n_ts >= n_seg,
              "bad sizes");
  WB2_REQUIRE(n_chunk % 8 == 0, "n_chunk=%d must be a multiple of 8", n_chunk);
  WB2_REQUIRE(n_outer < (1ll << 31), "n_outer=%lld too large",
              (long long)n_outer);
  WB2_REQUIRE(n_ctile == (n_col + kWave - 1) / kWave,
              "n_ctile=%d does not match ceil(n_col / 64)", n_ctile);
  if (n_outer == 0) return 0;
  EnsParams p{};
  p.ens = nullptr;    // the tables hold byte addresses
  p.truth = nullptr;
  p.ens_scale = p.truth_scale = 1;
  p.ens_slab = reinterpret_cast<const long long*>(ens_addr);
  p.truth_slab = reinterpret_cast<const long long*>(truth_addr);
  p.w_row = w_row;
  p.w_col = w_col;
  p.wfield = wfield;
  p.chunk_row0 = chunk_row0;
  p.chunk_nrow = chunk_nrow;
  p.seg_col0 = seg_col0;
  p.seg_eoff = seg_eoff;
  p.partials = partials;
  p.member_stride = member_stride;
  p.n_outer = n_outer;
  p.n_member = n_member;
  p.n_row = n_row;
  p.n_col = n_col;
  p.n_chunk = n_chunk;
  p.n_ctile = n_ctile;
  p.n_seg = n_seg;
  p.n_ts = n_ts;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == WB2_F32)
    return launch_ens_npad<float>(p, skipna != 0, wfield != nullptr, s);
  return launch_ens_npad<double>(p, skipna != 0, wfield != nullptr, s);
}

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
                            void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_EMPTY_OK(n_outer);
  WB2_REQUIRE(member_ptr && truth && w_row && chunk_row0 && chunk_nrow &&
                  seg_col0 && seg_eoff && partials,
              "null pointer argument");
  const int max_member = dtype == WB2_F32 ? 128 : 64;
  WB2_REQUIRE(n_member >= 1 && n_member <= max_member,
              "n_member=%d: gathered ensembles go through the register sort "
              "(<= %d members of this dtype)", n_member, max_member);
  WB2_REQUIRE(n_outer >= 0 && n_row > 0 && n_col > 0 && n_chunk > 0 &&
                  n_seg > 0 && n_ts >= n_seg,
              "bad sizes");
  WB2_REQUIRE(n_chunk % 8 == 0, "n_chunk=%d must be a multiple of 8", n_chunk);
  WB2_REQUIRE(n_outer < (1ll << 31), "n_outer=%lld too large",
              (long long)n_outer);
  WB2_REQUIRE(n_ctile == (n_col + kWave - 1) / kWave,
              "n_ctile=%d does not match ceil(n_col / 64)", n_ctile);
  if (n_outer == 0) return 0;
  EnsParams p{};
  p.ens = truth;  // unused with member_ptr; any valid address
  p.truth = truth;
  p.ens_scale = p.truth_scale =
      (long long)n_row * n_col * (dtype == WB2_F32 ? 4 : 8);
  p.truth_slab = reinterpret_cast<const long long*>(truth_slab);
  p.member_ptr = reinterpret_cast<const long long*>(member_ptr);
  p.w_row = w_row;
  p.w_col = w_col;
  p.wfield = wfield;
  p.chunk_row0 = chunk_row0;
  p.chunk_nrow = chunk_nrow;
  p.seg_col0 = seg_col0;
  p.seg_eoff = seg_eoff;
  p.partials = partials;
  p.maps = maps;
  p.n_outer = n_outer;
  p.n_member = n_member;
  p.n_row = n_row;
  p.n_col = n_col;
  p.n_chunk = n_chunk;
  p.n_ctile = n_ctile;
  p.n_seg = n_seg;
  p.n_ts = n_ts;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == WB2_F32)
    return launch_ens_npad<float>(p, skipna != 0, wfield != nullptr, s);
  return launch_ens_npad<double>(p, skipna != 0, wfield != nullptr, s);
}

int wb2_ens_threshold_partials(
    int dtype, int skipna, const void* ens, const int64_t* ens_slab,
    const void* truth, const int64_t* truth_slab, const void* threshold,
    const int64_t* thr_slab, int32_t n_member, int64_t member_stride,
    int64_t n_outer, int32_t n_row, int32_t n_col, const double* w_row,
    const double* w_col, const double* wfield, const int32_t* chunk_row0,
    const int32_t* chunk_nrow, int32_t n_chunk, int32_t n_ctile,
    const int32_t* seg_col0, const int32_t* seg_eoff, int32_t n_seg,
    int32_t n_ts, double* partials, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_EMPTY_OK(n_outer);
  WB2_REQUIRE(ens && truth && threshold && w_row && chunk_row0 && chunk_nrow &&
                  seg_col0 && seg_eoff && partials,
              "null pointer argument");
  WB2_REQUIRE(n_member >= 1, "n_member=%d", n_member);
  WB2_REQUIRE(n_outer >= 0 && n_outer < (1ll << 31) && n_row > 0 && n_col > 0 &&
                  n_chunk > 0 && n_seg > 0 && n_ts >= n_seg,
              "bad sizes");
  WB2_REQUIRE(n_chunk % 8 == 0, "n_chunk=%d must be a multiple of 8", n_chunk);
  WB2_REQUIRE(n_ctile == (n_col + kWave - 1) / kWave,
              "n_ctile=%d does not match ceil(n_col / 64)", n_ctile);
  if (n_outer == 0) return 0;
  EnsThrParams q{};
  EnsParams& p = q.e;
  p.ens = ens;
  p.truth = truth;
  p.ens_slab = reinterpret_cast<const long long*>(ens_slab);
  p.truth_slab = reinterpret_cast<const long long*>(truth_slab);
  p.w_row = w_row;
  p.w_col = w_col;
  p.wfield = wfield;
  p.chunk_row0 = chunk_row0;
  p.chunk_nrow = chunk_nrow;
  p.seg_col0 = seg_col0;
  p.seg_eoff = seg_eoff;
  p.partials = partials;
  p.member_stride = member_stride;
  p.n_outer = n_outer;
  p.n_member = n_member;
  p.n_row = n_row;
  p.n_col = n_col;
  p.n_chunk = n_chunk;
  p.n_ctile = n_ctile;
  p.n_seg = n_seg;
  p.n_ts = n_ts;
  q.thr = threshold;
  q.thr_slab = reinterpret_cast<const long long*>(thr_slab);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == WB2_F32)
    return launch_ens_threshold<float>(q, skipna != 0, wfield != nullptr, s);
  return launch_ens_threshold<double>(q, skipna != 0, wfield != nullptr, s);
}

int wb2_ens_threshold_maps(int dtype, int skipna, const void* ens,
                           const int64_t* ens_slab, const void* truth,
                           const int64_t* truth_slab, const void* threshold,
                           const int64_t* thr_slab, int32_t n_member,
                           int64_t member_stride, int64_t n_outer,
                           int64_t n_point, double* maps, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_EMPTY_OK(n_outer);
  WB2_EMPTY_OK(n_point);
  WB2_REQUIRE(ens && truth && threshold && maps, "null pointer argument");
  WB2_REQUIRE(n_member >= 1, "n_member=%d", n_member);
  WB2_REQUIRE(n_outer >= 0 && n_point >= 0, "bad sizes");
  if (n_outer == 0 || n_point == 0) return 0;
  EnsThrMapParams p{};
  p.ens = ens;
  p.truth = truth;
  p.thr = threshold;
  p.ens_slab = reinterpret_cast<const long long*>(ens_slab);
  p.truth_slab = reinterpret_cast<const long long*>(truth_slab);
  p.thr_slab = reinterpret_cast<const long long*>(thr_slab);
  p.maps = maps;
  p.member_stride = member_stride;
  p.n_outer = n_outer;
  p.n_point = n_point;
  p.n_member = n_member;
  const long long gy = n_outer < 32768 ? n_outer : 32768;
  const long long gz = (n_outer + gy - 1) / gy;
  WB2_REQUIRE(gz <= 65535, "n_outer=%lld too large", (long long)n_outer);
  const dim3 grid((unsigned)((n_point + 255) / 256), (unsigned)gy, (unsigned)gz);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define WB2_L(T, S) \
  hipLaunchKernelGGL((ens_threshold_maps_kernel<T, S>), grid, dim3(256), 0, s, p)
  if (dtype == WB2_F32) {
    if (skipna) WB2_L(float, true); else WB2_L(float, false);
  } else {
    if (skipna) WB2_L(double, true); else WB2_L(double, false);
  }
#undef WB2_L
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
