// The normal cdf / pdf behind the Gaussian scores (the reference calls
// scipy.stats.norm.cdf / pdf in float64: metrics.py:895-905, 975-1121).
//
//   cdf(z) = 1 - q or q,  q = 0.5 erfc(|z| / sqrt 2) = 0.5 e erfcx(|z| / sqrt 2)
//   pdf(z) = e / sqrt(2 pi),                            e = exp(-z^2 / 2)
//
// ONE exp serves both: erfcx(x) = exp(x^2) erfc(x) is smooth and comes from a
// table of degree-9 Taylor polynomials on 192 intervals of [0, 12) kept in LDS
// (five 16-byte reads + 9 FMAs, straight-line), the asymptotic series beyond
// (tools/gen_gauss_tables.py: <= 2.5e-16 relative against 50-digit values).
// The device libm's erfc -- two exp calls and a division inside, 135 VALU
// instructions, every branch taken by some lane of a wave -- was two thirds of
// the Gaussian kernels' arithmetic.
#pragma once

#include "common.hpp"
#include "gauss_tables.inc"

namespace wb2 {

constexpr int kErfcxDoubles = WB2_ERFCX_ROWS * WB2_ERFCX_COEFS;

__device__ const double kErfcxTable[kErfcxDoubles] = {WB2_ERFCX_TABLE};

// the workgroup copies the table (15 KB) into LDS; ends with a barrier
__device__ __forceinline__ void load_erfcx_table(double* lds) {
  for (int i = threadIdx.x; i < kErfcxDoubles; i += blockDim.x)
    lds[i] = kErfcxTable[i];
  __syncthreads();
}

// erfcx(x) for 0 <= x < 12 from the table (x >= 12 reads the last row: garbage,
// replaced by the caller; NaN: row 0, d = NaN, NaN out).  Straight-line code:
// the points of a lane's load interleave in one basic block.
__device__ __forceinline__ double erfcx_table(double x, const double* lds) {
  int k = (int)(x * (double)WB2_ERFCX_STEP_INV);
  k = k < WB2_ERFCX_ROWS - 1 ? k : WB2_ERFCX_ROWS - 1;
  const double d = __builtin_fma(-((double)k + 0.5), 1.0 / WB2_ERFCX_STEP_INV,
                                 x);
  typedef double D2 __attribute__((ext_vector_type(2)));
  const D2* row = reinterpret_cast<const D2*>(lds + k * WB2_ERFCX_COEFS);
  static_assert(WB2_ERFCX_COEFS == 10, "five 16-byte reads per row");
  const D2 c01 = row[0], c23 = row[1], c45 = row[2], c67 = row[3],
           c89 = row[4];
  double acc = c89[1];
  acc = __builtin_fma(acc, d, c89[0]);
  acc = __builtin_fma(acc, d, c67[1]);
  acc = __builtin_fma(acc, d, c67[0]);
  acc = __builtin_fma(acc, d, c45[1]);
  acc = __builtin_fma(acc, d, c45[0]);
  acc = __builtin_fma(acc, d, c23[1]);
  acc = __builtin_fma(acc, d, c23[0]);
  acc = __builtin_fma(acc, d, c01[1]);
  return __builtin_fma(acc, d, c01[0]);
}

// erfcx(x) for x >= 12 (|z| >= 17): the asymptotic series
//   1 / (x sqrt pi) * sum_n (-1)^n (2n - 1)!! / (2 x^2)^n;  +inf -> 0
__device__ __forceinline__ double erfcx_series(double x) {
  constexpr double a[WB2_ERFCX_ASYMPTOTIC_TERMS] = {WB2_ERFCX_ASYMPTOTIC};
  const double r = 1.0 / (2.0 * x * x);
  double s = a[WB2_ERFCX_ASYMPTOTIC_TERMS - 1];
#pragma unroll
  for (int n = WB2_ERFCX_ASYMPTOTIC_TERMS - 2; n >= 0; --n)
    s = __builtin_fma(s, r, a[n]);
  return s / x * 0.56418958354775628695;
}

// log(p) for p in [0, 1] (0 -> -inf, NaN -> NaN): the classic reduction
// p = 2^e m, m in [sqrt 1/2, sqrt 2), s = (m - 1) / (m + 1), an even polynomial
// in s of degree 14 (the minimax coefficients of the freely distributable
// fdlibm e_log.c, < 1 ulp); ~40 instructions against the 94 of the libm call.
__device__ __forceinline__ double log_unit(double p) {
  int e = __builtin_amdgcn_frexp_exp(p);
  double m = __builtin_amdgcn_frexp_mant(p);  // [0.5, 1)
  const bool low = m < 0.70710678118654752440;
  m = low ? m + m : m;
  e = low ? e - 1 : e;
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * __builtin_fma(
      w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01),
      3.999999999940941908e-01);
  const double t2 = z * __builtin_fma(
      w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01,
                                        1.818357216161805012e-01),
                       2.857142874366239149e-01),
      6.666666666666735130e-01);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)e;
  const double lo = __builtin_fma(dk, 1.90821492927058770002e-10,
                                  s * (hfsq + R));
  const double v = __builtin_fma(dk, 6.93147180369123816490e-01,
                                 -((hfsq - lo) - f));
  return p == 0.0 ? -__builtin_huge_val() : v;
}

// cdf and pdf of the standard normal at z (float64); NaN in, NaN out.
// FAR = false: table only -- returns true when |z| is beyond the table (the
// results are then garbage and the caller repeats the point with FAR = true,
// behind a wave-uniform branch: almost no wave ever does).
template <bool FAR>
__device__ __forceinline__ bool normal_cdf_pdf(double z, const double* lds,
                                               double& cdf, double& pdf) {
  const double e = exp(-0.5 * z * z);
  const double x = __builtin_fabs(z) * 0.70710678118654752440;
  const bool far = x >= WB2_ERFCX_X_MAX;
  double y = erfcx_table(x, lds);
  if constexpr (FAR) y = far ? erfcx_series(x) : y;
  const double q = 0.5 * e * y;  // the tail beyond |z|
  cdf = z > 0.0 ? 1.0 - q : q;
  pdf = e * 0.39894228040143267794;
  return far;
}

}  // namespace wb2
