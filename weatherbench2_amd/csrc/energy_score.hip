// The energy score in ONE read of the ensemble.
//
// Replaces (reference = /root/reference/weatherbench2/metrics.py):
//   :1403-1465  EnergyScore        = skill - 0.5 spread
//   :1468-1498  EnergyScoreSpread  = mean_m sqrt(_spatial_average((x_m - x_{m+1})^2)),
//                                    the M - 1 adjacent member differences
//   :1501-1517  EnergyScoreSkill   = mean_m sqrt(_spatial_average((x_m - y)^2))
// (scripts/evaluate.py:541-565 evaluates the three of them per chunk).  Per
// member these are area-weighted spatial means of float32 squares, i.e. 2 M - 1
// weighted sums per region -- 99 fp64 accumulators per grid point column for 50
// members, more than a lane can hold.  Members are therefore cut into BLOCKS of
// B consecutive members; the waves of a workgroup take one block each over the
// SAME rows and columns (grid x = chunk fastest: the blocks of a (chunk, tile)
// run on one XCD side by side), so that HBM sees every member and the truth
// once -- the block's neighbour member and the truth are re-read through L2 --
// and a lane keeps 2 B sums: B skill sums, B adjacent-difference sums.
//
// Block b of outer slab o is VIRTUAL slab o * n_block + b of the partials
// [n_outer * n_block][n_chunk][nwf][n_ts][K], K = 2 B (x 2 with skipna: the
// matching sums of weights), folded by the generic branch of the combine kernel
// (stream_reduce.hip) into means[K][n_region][n_outer * n_block]; a last small
// kernel takes the square roots and the member means.
#include "ensemble_kernels.hpp"

namespace wb2 {
namespace {

constexpr int kEnergyWaves = 8;  // member blocks per workgroup
#ifndef WB2_ENERGY_B
#define WB2_ENERGY_B 8   // members per block (wave): 2 B sums per lane
#endif
#ifndef WB2_ENERGY_ROWS
#define WB2_ENERGY_ROWS 4  // rows in flight per wave: their loads before the sums
#endif

template <typename T, int B, bool SKIPNA, bool WF>
__global__ void __launch_bounds__(kEnergyWaves* kWave)
    energy_partials_kernel(const EnsParams p, const int n_block) {
  constexpr int KQ = 2 * B, K = SKIPNA ? 2 * KQ : KQ, NWF = WF ? 2 : 1;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const unsigned bx = blockIdx.x;
  const long long o = (long long)blockIdx.z * gridDim.y + blockIdx.y;
  // x = chunk (fastest) + n_chunk * (column tile + n_ctile * block group)
  const unsigned rest = bx / (unsigned)p.n_chunk;
  const int chunk = (int)(((long long)(bx - rest * (unsigned)p.n_chunk) +
                           (WB2_ROTATE_CHUNKS ? o : 0)) % p.n_chunk);
  const int tile = (int)(rest % (unsigned)p.n_ctile);
  const int block = (int)(rest / (unsigned)p.n_ctile) * kEnergyWaves + wave;
  const int row0 = p.chunk_row0[chunk];
  const int nrow = p.chunk_nrow[chunk];
  if (nrow <= 0 || block >= n_block || o >= p.n_outer) return;
  const int M = p.n_member;
  const int m0 = block * B;
  const int nm = min(B, M - m0);      // members of this block: skill sums
  const int np = min(B, M - 1 - m0);  // pairs (m, m + 1) that START in it
  const int nload = min(B + 1, M - m0);
  const long long es = p.ens_slab ? p.ens_slab[o] : o;
  const long long ts = p.truth_slab ? p.truth_slab[o] : o;
  const int col0 = tile * kWave + lane;
  const bool active = col0 < p.n_col;
  const int colc = active ? col0 : p.n_col - 1;

  double acc[NWF][1][K];
#pragma unroll
  for (int w = 0; w < NWF; ++w)
#pragma unroll
    for (int k = 0; k < K; ++k) acc[w][0][k] = 0.0;

  const long long slab_elems = (long long)p.n_row * p.n_col;
  const T* xrow0 = static_cast<const T*>(p.ens) + es * slab_elems +
                   (long long)m0 * p.member_stride + (long long)row0 * p.n_col;
  const int lane_bytes = colc * (int)sizeof(T);
  const T* tb = static_cast<const T*>(p.truth) + ts * slab_elems +
                (long long)row0 * p.n_col + colc;
  const double* wfp =
      WF ? p.wfield + (long long)row0 * p.n_col + colc : nullptr;
  // U rows per iteration: their (B + 1) + 1 loads each are issued before the
  // first sum (a wave has only ~10 loads per row to hide the latency with)
  constexpr int U = WB2_ENERGY_ROWS;
  auto rows = [&](const int r, auto count_tag) {
    constexpr int N = decltype(count_tag)::value;
    T x[N][B + 1], t[N];
    double wr[N], wf[N];
#pragma unroll
    for (int u = 0; u < N; ++u) {
      const long long off = (long long)(r + u) * p.n_col;
      const T* xrow = xrow0 + off;
#pragma unroll
      for (int j = 0; j <= B; ++j)  // a load past the block's members is off
        x[u][j] = member_load<T, true>(xrow + j * p.member_stride, lane_bytes,
                                       j < nload ? 0x7fffffff : 0);
      t[u] = __builtin_nontemporal_load(tb + off);
      wr[u] = p.w_row[row0 + r + u];
      wf[u] = 1.0;
      if constexpr (WF) wf[u] = wfp[off];
    }
#pragma unroll
    for (int u = 0; u < N; ++u) {
      const bool inside = !WF || wf[u] > 0.0;
      const double w2 = (WF && inside) ? wr[u] * wf[u] : 0.0;
      // (forecast - truth) ** 2 and (x_m - x_{m+1}) ** 2 in the input dtype
      // like NumPy (metrics.py:1489-1493, 1512); the weighted sums in fp64
      // (:161-163)
      auto add = [&](int k, T q, bool live) {
        if (!live) return;  // wave-uniform
        double v = (double)q, c = 1.0;
        if constexpr (SKIPNA) {
          const bool ok = !is_nan(q);
          v = ok ? v : 0.0;
          c = ok ? 1.0 : 0.0;
        }
        acc[0][0][k] = __builtin_fma(wr[u], v, acc[0][0][k]);
        if constexpr (SKIPNA)
          acc[0][0][KQ + k] = __builtin_fma(wr[u], c, acc[0][0][KQ + k]);
        if constexpr (WF) {
          acc[1][0][k] = __builtin_fma(w2, inside ? v : 0.0, acc[1][0][k]);
          if constexpr (SKIPNA)
            acc[1][0][KQ + k] =
                __builtin_fma(w2, inside ? c : 0.0, acc[1][0][KQ + k]);
        }
      };
#pragma unroll
      for (int j = 0; j < B; ++j) {
        const T d = x[u][j] - t[u];
        add(j, d * d, j < nm);
        const T e = x[u][j] - x[u][j + 1];
        add(B + j, e * e, j < np);
      }
    }
  };
  int r = 0;
#pragma clang loop unroll(disable)
  for (; r + U <= nrow; r += U) rows(r, std::integral_constant<int, U>{});
#pragma clang loop unroll(disable)
  for (; r < nrow; ++r) rows(r, std::integral_constant<int, 1>{});
  if (p.w_col) {
    const double wc = p.w_col[colc];
#pragma unroll
    for (int w = 0; w < NWF; ++w)
#pragma unroll
      for (int k = 0; k < K; ++k) acc[w][0][k] *= wc;
  }
  if (!active) {
#pragma unroll
    for (int w = 0; w < NWF; ++w)
#pragma unroll
      for (int k = 0; k < K; ++k) acc[w][0][k] = 0.0;
  }
  fold_tile_to_segs<NWF, 1, K>(
      acc, lane, tile, col0, col0, p.n_col, p.seg_col0, p.seg_eoff, p.n_seg,
      p.n_ts,
      p.partials + ((o * n_block + block) * p.n_chunk + chunk) *
                       (long long)(NWF * p.n_ts * K));
}

// means[KQ][n_region][n_outer * n_block] -> out[3][n_region][n_outer]:
// (score, spread, skill); one wave per (region, outer slab): the lanes fetch
// the 2 M - 1 spatial means and take the square roots side by side, lane 0 adds
// them in member order in fp64 (NumPy's reduction over the leading axis) --
// one thread walking them alone took 26 us on dependent loads.
constexpr int kEnergyMaxMembers = 1024;
__global__ void __launch_bounds__(kWave)
    energy_finalize_kernel(const double* __restrict__ means, int block,
                           int n_block, int n_member, int skipna, int n_region,
                           long long n_outer, double* __restrict__ out) {
  __shared__ double root[2][kEnergyMaxMembers];
  const long long idx = blockIdx.x;
  const int lane = threadIdx.x;
  const int r = (int)(idx / n_outer);
  const long long o = idx - (long long)r * n_outer;
  const long long n_virtual = n_outer * n_block;
  const long long row = (long long)n_region * n_virtual;  // one slot's plane
  const double* base = means + (long long)r * n_virtual + o * n_block;
  for (int m = lane; m < n_member; m += kWave) {
    root[0][m] = sqrt(base[(long long)(m % block) * row + m / block]);
    if (m + 1 < n_member)
      root[1][m] = sqrt(base[(long long)(block + m % block) * row + m / block]);
  }
  __syncthreads();
  if (lane != 0) return;
  auto member_mean = [&](const double* v, int count) {
    double s = 0.0, n = 0.0;
    for (int m = 0; m < count; ++m) {
      const bool keep = !(skipna && is_nan(v[m]));
      s += keep ? v[m] : 0.0;
      n += keep ? 1.0 : 0.0;
    }
    return s / n;  // 0 / 0: an all-NaN mean is NaN (xarray, skipna)
  };
  const double skill = member_mean(root[0], n_member);
  // metrics.py:1479-1488: one member has no spread -- zeros, whatever the data
  const double spread = n_member == 1 ? 0.0 : member_mean(root[1], n_member - 1);
  const long long plane = (long long)n_region * n_outer;
  out[idx] = skill - 0.5 * spread;
  out[plane + idx] = spread;
  out[2 * plane + idx] = skill;
}

template <typename T, int B>
int launch_energy(const EnsParams& p, int n_block, bool skipna, bool wf,
                  hipStream_t stream) {
  const int n_bgrp = (n_block + kEnergyWaves - 1) / kEnergyWaves;
  const int waves = n_block < kEnergyWaves ? n_block : kEnergyWaves;
  const long long gy = p.n_outer < 32768 ? p.n_outer : 32768;
  const long long gz = (p.n_outer + gy - 1) / gy;
  const dim3 grid((unsigned)(p.n_chunk * p.n_ctile * n_bgrp), (unsigned)gy,
                  (unsigned)gz);
  const dim3 blockdim(waves * kWave);
#define WB2_L(S, W)                                                       \
  hipLaunchKernelGGL((energy_partials_kernel<T, B, S, W>), grid, blockdim, 0, \
                     stream, p, n_block)
  if (skipna) {
    if (wf) WB2_L(true, true); else WB2_L(true, false);
  } else {
    if (wf) WB2_L(false, true); else WB2_L(false, false);
  }
#undef WB2_L
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

// Members per block: 8 (16 sums per lane; 32 with skipna), 4 when a 2-D weight
// field AND NaN skipping double the sums twice over.
int energy_block(bool skipna, bool wf) {
  return (skipna && wf) ? 4 : WB2_ENERGY_B;
}

}  // namespace
}  // namespace wb2

extern "C" {

int wb2_energy_layout(int32_t n_member, int skipna, int has_wfield,
                      int32_t* block, int32_t* n_block, int32_t* n_slot) {
  using namespace wb2;
  WB2_REQUIRE(n_member >= 1, "n_member=%d", n_member);
  WB2_REQUIRE(block && n_block && n_slot, "null pointer argument");
  const int b = energy_block(skipna != 0, has_wfield != 0);
  *block = b;
  *n_block = (n_member + b - 1) / b;
  *n_slot = (skipna ? 4 : 2) * b;
  return 0;
}

int wb2_energy_score(int dtype, int skipna, const void* ens,
                     const int64_t* ens_slab, const void* truth,
                     const int64_t* truth_slab, int32_t n_member,
                     int64_t member_stride, int64_t n_outer,
                     const wb2_plan_tables* plan, double* partials,
                     double* means, double* out, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(dtype == WB2_F32 || dtype == WB2_F64, "unknown dtype %d", dtype);
  WB2_EMPTY_OK(n_outer);
  WB2_REQUIRE(plan != nullptr, "null plan");
  WB2_REQUIRE(ens && truth && partials && means && out, "null pointer argument");
  WB2_REQUIRE(n_member >= 1 && n_member <= kEnergyMaxMembers,
              "n_member=%d (1 ... %d)", n_member, kEnergyMaxMembers);
  const wb2_plan_tables& t = *plan;
  WB2_REQUIRE(t.w_row && t.chunk_row0 && t.chunk_nrow && t.seg_col0 &&
                  t.seg_eoff && t.band_chunk0 && t.coef_band && t.coef_seg &&
                  t.region_wf && t.region_wsum,
              "null table in the plan");
  WB2_REQUIRE(t.n_row > 0 && t.n_col > 0 && t.n_chunk > 0 && t.n_seg > 0 &&
                  t.n_ts >= t.n_seg && t.n_band > 0 && t.n_region > 0,
              "bad sizes in the plan");
  WB2_REQUIRE(t.n_chunk % 8 == 0, "n_chunk=%d must be a multiple of 8",
              t.n_chunk);
  WB2_REQUIRE(t.n_ctile == (t.n_col + kWave - 1) / kWave,
              "n_ctile=%d does not match ceil(n_col / 64): the energy-score "
              "pass uses the ensemble tile width (wb2_ens_tile_cols)",
              t.n_ctile);
  WB2_REQUIRE(!t.wfield || t.wfield_dtype == WB2_F64,
              "the energy-score pass reads a float64 weight field");
  int32_t block = 0, n_block = 0, k = 0;
  if (wb2_energy_layout(n_member, skipna, t.wfield != nullptr, &block, &n_block,
                        &k) != 0)
    return -1;
  WB2_REQUIRE(n_outer * n_block < (1ll << 31), "n_outer=%lld too large",
              (long long)n_outer);
  EnsParams p{};
  p.ens = ens;
  p.truth = truth;
  p.ens_slab = reinterpret_cast<const long long*>(ens_slab);
  p.truth_slab = reinterpret_cast<const long long*>(truth_slab);
  p.w_row = t.w_row;
  p.w_col = t.w_col;
  p.wfield = static_cast<const double*>(t.wfield);
  p.chunk_row0 = t.chunk_row0;
  p.chunk_nrow = t.chunk_nrow;
  p.seg_col0 = t.seg_col0;
  p.seg_eoff = t.seg_eoff;
  p.partials = partials;
  p.member_stride = member_stride;
  p.n_outer = n_outer;
  p.n_member = n_member;
  p.n_row = t.n_row;
  p.n_col = t.n_col;
  p.n_chunk = t.n_chunk;
  p.n_ctile = t.n_ctile;
  p.n_seg = t.n_seg;
  p.n_ts = t.n_ts;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool sk = skipna != 0, wf = t.wfield != nullptr;
  int rc;
  if (dtype == WB2_F32)
    rc = block != 4 ? launch_energy<float, WB2_ENERGY_B>(p, n_block, sk, wf, s)
                    : launch_energy<float, 4>(p, n_block, sk, wf, s);
  else
    rc = block != 4 ? launch_energy<double, WB2_ENERGY_B>(p, n_block, sk, wf, s)
                    : launch_energy<double, 4>(p, n_block, sk, wf, s);
  if (rc != 0) return rc;
  // spatial means of every (virtual slab, slot): the generic combine
  rc = combine_slots(WB2_MODE_GAUSS, skipna, k, partials, n_outer * n_block,
                     t.n_chunk, wf ? 2 : 1, t.n_seg, t.seg_eoff, t.n_ts,
                     t.band_chunk0, t.n_band, t.coef_band, t.coef_seg,
                     t.region_wf, t.region_wsum, t.n_region, nullptr, means,
                     stream);
  if (rc != 0) return rc;
  const long long n = (long long)t.n_region * n_outer;
  WB2_REQUIRE(n < (1ll << 31), "n_region * n_outer too large");
  hipLaunchKernelGGL(energy_finalize_kernel, dim3((unsigned)n), dim3(kWave), 0,
                     s, means, block, n_block, n_member, skipna, t.n_region,
                     (long long)n_outer, out);
  WB2_HIP_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
