// Per-lane building blocks of the LDS real FFT of K4f (spectrum_fused.hip).
//
// Everything here is __host__ __device__ and free of wave intrinsics, so the
// exact code the kernel runs per lane can also be executed on the CPU by a
// lane-by-lane emulation (tools/fft_host_check.hip, tests/test_fft_core_cpu.py):
// index maps, twiddle tables, composite butterflies and the real-FFT
// recombination are checked against numpy without a GPU.
//
// Arithmetic contract (ZonalEnergySpectrum.compute,
// /root/reference/weatherbench2/derived_variables.py:592-626): np.fft.rfft(x,
// norm='forward') of a float32 row is a complex64 transform; here the row is
// transformed as N/2 complex points z[m] = x[2m] + i x[2m+1] with a Stockham
// FFT in float32 and recombined, see recombine_pair().
#pragma once

#include <hip/hip_runtime.h>

namespace wb2 {
namespace fftcore {

#define WB2_HD __host__ __device__ __forceinline__

// How a pass stores one complex point into the slab; spectrum_fused.hip
// overrides it for device code (single ds_write_b64 instead of the compiler's
// ds_write2_b64 pairs), the host check keeps the plain assignment.
#ifndef WB2_FFT_SLAB_STORE
#define WB2_FFT_SLAB_STORE(ptr, value) (*(ptr) = (value))
#endif

// complex (re, im) of float (the float32 transform: complex64 like NumPy's
// rfft of float32 data) or double (float64 rows: complex128)
template <typename S>
using cx = S __attribute__((ext_vector_type(2)));
template <typename S>
using cx2 = S __attribute__((ext_vector_type(4)));  // two adjacent points
typedef cx<float> cf;
typedef cx<double> cd;
template <typename C>
struct ScalarOf;
template <>
struct ScalarOf<cf> {
  typedef float type;
};
template <>
struct ScalarOf<cd> {
  typedef double type;
};
template <typename C>
using scalar_t = typename ScalarOf<C>::type;
// a literal in the transform's precision: the float32 code keeps its float
// literals (no double rounding), the float64 code gets the double ones
template <typename S>
constexpr S lit(float f, double d) {
  return sizeof(S) == 4 ? (S)f : (S)d;
}

constexpr int kLanes = 64;  // a wavefront on gfx950

#ifndef WB2_FFT_PAIRED_PAD0
#define WB2_FFT_PAIRED_PAD0 2    // slab padding of PairedPlan<720>, see Plan
#endif
#ifndef WB2_FFT_PAIRED_PAD1
#define WB2_FFT_PAIRED_PAD1 8
#endif

// ---- compile-time cos / sin of 2 pi m / r (Taylor series on (-pi, pi]) ------
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double taylor_sin(double x) {
  double term = x, sum = x;
  for (int n = 1; n < 20; ++n) {
    term *= -x * x / (double)((2 * n) * (2 * n + 1));
    sum += term;
  }
  return sum;
}
constexpr double taylor_cos(double x) {
  double term = 1.0, sum = 1.0;
  for (int n = 1; n < 20; ++n) {
    term *= -x * x / (double)((2 * n - 1) * (2 * n));
    sum += term;
  }
  return sum;
}
constexpr int reduce_turn(int m, int r) {
  int mm = ((m % r) + r) % r;
  if (2 * mm > r) mm -= r;
  return mm;
}
constexpr double unit_cos(int m, int r) {
  return taylor_cos(2.0 * kPi * (double)reduce_turn(m, r) / (double)r);
}
constexpr double unit_sin(int m, int r) {
  return taylor_sin(2.0 * kPi * (double)reduce_turn(m, r) / (double)r);
}

// ---- compile-time loop -------------------------------------------------------
template <int I>
struct Idx {
  static constexpr int value = I;
};
template <int I, int N, typename F>
WB2_HD void static_for(F&& f) {
  if constexpr (I < N) {
    f(Idx<I>{});
    static_for<I + 1, N>(f);
  }
}

// ---- complex helpers ---------------------------------------------------------
template <typename C>
WB2_HD C dup_x(C a) { return __builtin_shufflevector(a, a, 0, 0); }
template <typename C>
WB2_HD C dup_y(C a) { return __builtin_shufflevector(a, a, 1, 1); }
template <typename C>
WB2_HD C vfma(C a, C b, C c) { return __builtin_elementwise_fma(a, b, c); }
template <typename C>
WB2_HD C splat(scalar_t<C> s) { return C{s, s}; }
// Multiplications by -i / +i as ONE packed multiply of the swapped operand with a
// literal sign pair: a half-negated vector (a.y, -a.x) costs hipcc a v_xor and
// a v_mov, while swaps (op_sel) and literal operands are free.
template <typename C>
WB2_HD C swap_xy(C a) { return __builtin_shufflevector(a, a, 1, 0); }
template <typename C>
WB2_HD C mul_neg_i(C a) { return swap_xy(a) * C{1, -1}; }  // a * (-i)
template <typename C>
WB2_HD C mul_pos_i(C a) { return swap_xy(a) * C{-1, 1}; }  // a * (+i)
// m + s * (-i) * d  for a real scalar s: one packed FMA
template <typename C>
WB2_HD C add_neg_i(C m, C d, scalar_t<C> s) {
  return vfma(swap_xy(d), C{s, -s}, m);
}

// a * w: one packed multiply + one packed FMA (w and its rotation i*w are
// separate operands; for compile-time w both are literals)
template <typename C>
WB2_HD C cmul(C a, C w) {
  const C wr = {-w.y, w.x};
  return vfma(dup_y(a), wr, dup_x(a) * w);
}

// a * exp(-2 pi i M / R) for compile-time M, R
template <int M, int R, typename C>
WB2_HD C mul_w(C a) {
  typedef scalar_t<C> S;
  constexpr int m = ((M % R) + R) % R;
  if constexpr (m == 0) {
    return a;
  } else if constexpr (4 * m == R) {
    return mul_neg_i(a);
  } else if constexpr (2 * m == R) {
    return -a;
  } else if constexpr (4 * m == 3 * R) {
    return mul_pos_i(a);
  } else {
    constexpr S c = (S)unit_cos(m, R), s = (S)(-unit_sin(m, R));
    const C w = {c, s}, wr = {-s, c};
    return vfma(dup_y(a), wr, dup_x(a) * w);
  }
}

// ---- butterflies: in-place DFT of R points, X[k] = sum_n a[n] exp(-2 pi i nk/R)
template <int R>
struct Radix;

template <>
struct Radix<2> {
  template <typename C>
  static WB2_HD void run(C (&a)[2]) {
    const C t = a[0] - a[1];
    a[0] = a[0] + a[1];
    a[1] = t;
  }
};
template <>
struct Radix<3> {
  template <typename C>
  static WB2_HD void run(C (&a)[3]) {
    typedef scalar_t<C> S;
    constexpr S c = lit<S>(0.86602540378443864676f,
                           0.86602540378443864676);  // sin(pi/3)
    const C s = a[1] + a[2], d = a[1] - a[2];
    const C m = vfma(splat<C>((S)-0.5), s, a[0]);
    a[0] = a[0] + s;
    a[1] = add_neg_i(m, d, c);
    a[2] = add_neg_i(m, d, -c);
  }
};
template <>
struct Radix<4> {
  template <typename C>
  static WB2_HD void run(C (&a)[4]) {
    typedef scalar_t<C> S;
    const C t0 = a[0] + a[2], t1 = a[0] - a[2], t2 = a[1] + a[3];
    const C d = a[1] - a[3];
    a[0] = t0 + t2;
    a[1] = add_neg_i(t1, d, (S)1);
    a[2] = t0 - t2;
    a[3] = add_neg_i(t1, d, (S)-1);
  }
};
template <>
struct Radix<5> {
  template <typename C>
  static WB2_HD void run(C (&a)[5]) {
    typedef scalar_t<C> S;
    constexpr S c1 = lit<S>(0.30901699437494742410f, 0.30901699437494742410);
    constexpr S c2 = lit<S>(-0.80901699437494742410f, -0.80901699437494742410);
    constexpr S s1 = lit<S>(0.95105651629515357212f, 0.95105651629515357212);
    constexpr S s2 = lit<S>(0.58778525229247312917f, 0.58778525229247312917);
    const C s14 = a[1] + a[4], d14 = a[1] - a[4];
    const C s23 = a[2] + a[3], d23 = a[2] - a[3];
    const C m1 = vfma(splat<C>(c2), s23, vfma(splat<C>(c1), s14, a[0]));
    const C m2 = vfma(splat<C>(c1), s23, vfma(splat<C>(c2), s14, a[0]));
    const C q1 = vfma(splat<C>(s2), d23, splat<C>(s1) * d14);
    const C q2 = vfma(splat<C>(-s1), d23, splat<C>(s2) * d14);
    a[0] = a[0] + s14 + s23;
    a[1] = add_neg_i(m1, q1, (S)1);
    a[4] = add_neg_i(m1, q1, (S)-1);
    a[2] = add_neg_i(m2, q2, (S)1);
    a[3] = add_neg_i(m2, q2, (S)-1);
  }
};

// R = A * B (Cooley-Tukey inside the registers of one lane): input index
// n = B n1 + n2, output index k = k1 + A k2,
//   X[k1 + A k2] = sum_n2 W_B^(n2 k2) W_R^(n2 k1) sum_n1 a[B n1 + n2] W_A^(n1 k1)
template <int A, int B>
struct Composite {
  template <typename C>
  static WB2_HD void run(C (&a)[A * B]) {
    C y[B][A];
    static_for<0, B>([&](auto n2c) {
      constexpr int n2 = decltype(n2c)::value;
      C t[A];
#pragma unroll
      for (int n1 = 0; n1 < A; ++n1) t[n1] = a[B * n1 + n2];
      Radix<A>::run(t);
      static_for<0, A>([&](auto k1c) {
        constexpr int k1 = decltype(k1c)::value;
        y[n2][k1] = mul_w<n2 * k1, A * B>(t[k1]);
      });
    });
#pragma unroll
    for (int k1 = 0; k1 < A; ++k1) {
      C u[B];
#pragma unroll
      for (int n2 = 0; n2 < B; ++n2) u[n2] = y[n2][k1];
      Radix<B>::run(u);
#pragma unroll
      for (int k2 = 0; k2 < B; ++k2) a[k1 + A * k2] = u[k2];
    }
  }
};
template <> struct Radix<6> : Composite<2, 3> {};
template <> struct Radix<9> : Composite<3, 3> {};
template <> struct Radix<15> : Composite<3, 5> {};
template <> struct Radix<8> : Composite<2, 4> {};
template <> struct Radix<10> : Composite<2, 5> {};
template <> struct Radix<12> : Composite<4, 3> {};
template <> struct Radix<16> : Composite<4, 4> {};
template <> struct Radix<20> : Composite<4, 5> {};

// ---- pass plans: N2 = product of up to three radices --------------------------
// Chosen so that every pass has at most a few rounds of <= 64 butterflies and
// (for the sizes that matter: 0.25 / 0.5 degree grids) nearly full lanes:
// 720 = 12 x 12 x 5 -> 60, 60 and 3 x 48 lanes.
template <int N2>
struct Plan;
// PAD0 / PAD1: complex slots of padding after every R0 outputs of pass 0 / every
// R0 R1 outputs of pass 1 in the LDS slab, chosen so that the strided 16-byte
// and 8-byte writes of those passes hit distinct banks (720: rows of 12 -> 14
// complex = 28 dwords, blocks of 144 -> 156; measured 18 % of the LDS cycles
// were bank conflicts without them).  The last pass always writes compactly.
#define WB2_FFT_PLAN(N2_, A_, B_, C_, PAD0_, PAD1_)          \
  template <>                                                \
  struct Plan<N2_> {                                         \
    static constexpr int R0 = A_, R1 = B_, R2 = C_;          \
    static constexpr int PAD0 = PAD0_, PAD1 = (C_ > 1) ? PAD1_ : 0; \
    static_assert(A_ * B_ * C_ == N2_, "plan");              \
    static_assert(PAD0_ % 2 == 0, "16-byte aligned runs");   \
  };
WB2_FFT_PLAN(32, 4, 8, 1, 0, 0)
WB2_FFT_PLAN(64, 8, 8, 1, 0, 0)
WB2_FFT_PLAN(120, 4, 5, 6, 0, 0)
WB2_FFT_PLAN(128, 4, 4, 8, 0, 0)
WB2_FFT_PLAN(180, 5, 6, 6, 0, 0)
WB2_FFT_PLAN(256, 4, 8, 8, 0, 0)
WB2_FFT_PLAN(360, 6, 6, 10, 0, 0)
WB2_FFT_PLAN(512, 8, 8, 8, 0, 0)
WB2_FFT_PLAN(720, 12, 12, 5, 2, 12)
// Grids beyond the WeatherBench 2 datasets' own (N = 64, 128, 240, 256, 360,
// 512, 720, 1024, 1440 above): every other even row length whose half is a
// multiple of 4 (the adjacent-bin epilogue) and a product of up to three
// radices out of 2 ... 20, in common use -- 3.75, 1.25, 1.125, 0.9375, 0.75,
// 0.5625, 0.46875, 0.28125, 0.2, 0.17578125, 0.140625, 0.125 and 0.1 degree
// grids.  Unpadded slabs (not tuned); any other
// length keeps the hipFFT path (spectrum.hip).
WB2_FFT_PLAN(48, 6, 8, 1, 0, 0)        // N = 96
WB2_FFT_PLAN(144, 12, 12, 1, 0, 0)     // 288
WB2_FFT_PLAN(160, 10, 16, 1, 0, 0)     // 320
WB2_FFT_PLAN(192, 12, 16, 1, 0, 0)     // 384
WB2_FFT_PLAN(240, 12, 20, 1, 0, 0)     // 480
WB2_FFT_PLAN(320, 16, 20, 1, 0, 0)     // 640
WB2_FFT_PLAN(384, 6, 8, 8, 0, 0)       // 768
WB2_FFT_PLAN(640, 8, 8, 10, 0, 0)      // 1280
WB2_FFT_PLAN(900, 10, 10, 9, 0, 0)     // 1800
WB2_FFT_PLAN(1024, 8, 8, 16, 0, 0)     // 2048
WB2_FFT_PLAN(1280, 8, 10, 16, 0, 0)    // 2560
WB2_FFT_PLAN(1440, 12, 12, 10, 0, 0)   // 2880
WB2_FFT_PLAN(1800, 10, 12, 15, 0, 0)   // 3600
#undef WB2_FFT_PLAN

// ---- one Stockham pass of radix R; NS = product of the radices already done ---
// Butterfly j (0 <= j < T = N2 / R), k = j mod NS:
//   inputs   v[r] = src[j + r T] * exp(-2 pi i k r / (NS R))      r = 0..R-1
//   outputs  dst[(j / NS) NS R + k + t NS] = DFT_R(v)[t]          t = 0..R-1
// A lane owns the butterflies j = lane + 64 rd.  twz[m] = exp(-2 pi i m / N2).
// Slab layout: logical index L of the pass's INPUT lives at
//   (L / IN_BLOCK) (IN_BLOCK + IN_PAD) + L mod IN_BLOCK
// and its OUTPUT blocks of NS R points are OUT_PAD slots apart (0 = compact).
template <int N2, int R, int NS, int IN_BLOCK = 1, int IN_PAD = 0,
          int OUT_PAD = 0>
struct Pass {
  static constexpr int T = N2 / R;
  static_assert(IN_PAD == 0 || T % IN_BLOCK == 0, "padded input layout");
  static constexpr int IN_STEP =  // slots between the inputs r and r + 1
      IN_PAD == 0 ? T : (T / IN_BLOCK) * (IN_BLOCK + IN_PAD);
  // slab slots this pass's output needs
  static constexpr int OUT_SLOTS = (N2 / (NS * R)) * (NS * R + OUT_PAD);
  static constexpr int ROUNDS = (T + kLanes - 1) / kLanes;
  static constexpr int TWS = N2 / (NS * R);
  static constexpr int NTW = R > 1 ? R - 1 : 1;
  // distinct twiddle rows of this pass (k = j mod NS takes min(NS, T) values)
  static constexpr int KP = NS < T ? NS : T;

  static WB2_HD bool live(int lane, int rd) {
    return (rd + 1) * kLanes <= T || lane + rd * kLanes < T;
  }

  // the R - 1 non-trivial twiddles of each of this lane's butterflies
  template <typename C>
  static WB2_HD void load_twiddles(const C* __restrict__ twz, int lane,
                                   C (&tw)[ROUNDS][NTW]) {
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
      const int j = lane + rd * kLanes;
      const int k = (j < T ? j : 0) % NS;
#pragma unroll
      for (int r = 1; r < R; ++r) tw[rd][r - 1] = twz[k * r * TWS];
    }
  }

  // Compact table of this pass: tbl[(r - 1) KP + k] = exp(-2 pi i k r / (NS R)),
  // filled cooperatively (`tid` of `nthread`), read back per lane and round.
  template <typename C>
  static WB2_HD void fill_table(const C* __restrict__ twz, C* __restrict__ tbl,
                                int tid, int nthread) {
    for (int i = tid; i < (R - 1) * KP; i += nthread) {
      const int r = i / KP + 1, k = i % KP;
      tbl[i] = twz[k * r * TWS];
    }
  }
  static WB2_HD int table_row(int lane, int rd) {
    const int j = lane + rd * kLanes;
    return (j < T ? j : 0) % NS;
  }
  template <typename C>
  static WB2_HD void load_twiddles_table(const C* __restrict__ tbl,
                                         const int (&row)[ROUNDS],
                                         C (&tw)[ROUNDS][NTW]) {
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd)
#pragma unroll
      for (int r = 1; r < R; ++r) tw[rd][r - 1] = tbl[(r - 1) * KP + row[rd]];
  }

  // Lanes without a butterfly in a round re-read the last one's inputs (in
  // range, never stored): unconditional loads keep the wave free of exec
  // branches and of the zero-fills hipcc adds for half-defined registers.
  template <typename Load, typename C>  // Load: index -> C
  static WB2_HD void load(const Load& src, int lane, C (&v)[ROUNDS][R]) {
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
      const int j0 = lane + rd * kLanes;
      const int j = (rd + 1) * kLanes <= T ? j0 : (j0 < T ? j0 : T - 1);
      const int base =
          IN_PAD == 0 ? j : (j / IN_BLOCK) * (IN_BLOCK + IN_PAD) + j % IN_BLOCK;
#pragma unroll
      for (int r = 0; r < R; ++r) v[rd][r] = src(base + r * IN_STEP);
    }
  }

  template <typename C>
  static WB2_HD void twiddle(C (&v)[ROUNDS][R], const C (&tw)[ROUNDS][NTW]) {
    if constexpr (NS > 1) {
#pragma unroll
      for (int rd = 0; rd < ROUNDS; ++rd)
#pragma unroll
        for (int r = 1; r < R; ++r) v[rd][r] = cmul(v[rd][r], tw[rd][r - 1]);
    }
  }

  template <typename C>
  static WB2_HD void butterflies(C (&v)[ROUNDS][R]) {
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) Radix<R>::run(v[rd]);
  }

  template <typename C>
  static WB2_HD void store(C* __restrict__ z, int lane,
                           const C (&v)[ROUNDS][R]) {
    typedef cx2<scalar_t<C>> C2;
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
      const int j = lane + rd * kLanes;
      if (live(lane, rd)) {
        if constexpr (NS == 1 && R % 2 == 0) {
          // the R outputs of a first-pass butterfly are one contiguous run
          C2* dst = reinterpret_cast<C2*>(z + j * (R + OUT_PAD));
#pragma unroll
          for (int h = 0; h < R / 2; ++h) {
            C2 w;
            w.x = v[rd][2 * h].x;
            w.y = v[rd][2 * h].y;
            w.z = v[rd][2 * h + 1].x;
            w.w = v[rd][2 * h + 1].y;
            dst[h] = w;
          }
        } else {
          const int k = j % NS;
          const int j0 = (j / NS) * (NS * R + OUT_PAD) + k;
#pragma unroll
          for (int t = 0; t < R; ++t) WB2_FFT_SLAB_STORE(z + j0 + t * NS, v[rd][t]);
        }
      }
    }
  }
};

// Slab slots a wave needs for a row of N2 complex points (largest layout).
template <int N2, typename PL = Plan<N2>>
constexpr int slab_slots() {
  constexpr int a = (N2 / PL::R0) * (PL::R0 + PL::PAD0);
  constexpr int b = (N2 / (PL::R0 * PL::R1)) * (PL::R0 * PL::R1 + PL::PAD1);
  return a > b ? (a > N2 ? a : N2) : (b > N2 ? b : N2);
}

// ---- real-FFT recombination ----------------------------------------------------
// Z = FFT_{N2}(x[2m] + i x[2m+1]).  With a = Z[k], b = Z[N2 - k] (Z[N2] = Z[0]):
//   E = (a + conj b) / 2,  O = (a - conj b) / (2i),  X[k] = E + W^k O,
//   X[N2 - k] = conj(E - W^k O),  W = exp(-2 pi i / N).
// `wq` = W^k * (-i) * (0.5 / N) and `half_inv_n` = 0.5 / N carry the factor 1/2
// and the 1/N of norm='forward'; returns |X[k]|^2 and |X[N2-k]|^2 (float32,
// like real(f_k * conj(f_k)) of a complex64 transform).
template <typename C>
WB2_HD void recombine_pair(C a, C b, C wq, scalar_t<C> half_inv_n,
                           scalar_t<C>& p1, scalar_t<C>& p2) {
  const C u = vfma(b, C{1, -1}, a);  // a + conj b
  const C v = vfma(b, C{-1, 1}, a);  // a - conj b
  const C wv = cmul(v, wq);
  const C x1 = vfma(splat<C>(half_inv_n), u, wv);
  const C x2 = vfma(splat<C>(half_inv_n), u, -wv);
  p1 = x1.x * x1.x + x1.y * x1.y;
  p2 = x2.x * x2.x + x2.y * x2.y;
}

// ---- the last pass with its butterflies PAIRED in the lane ---------------------
// The real-FFT recombination needs Z[k] and Z[N2 - k].  In the last pass (radix
// R, NS = T = N2 / R) butterfly j produces the bins j + t T, t = 0..R-1, and the
// partner of bin j + t T is N2 - j - t T = (T - j) + (R - 1 - t) T: an output of
// butterfly T - j.  A lane that runs the butterflies j AND T - j therefore holds
// every pair it needs in registers: no store of the last pass, no load of the
// recombination -- two thirds of the LDS traffic of a row and one dependent LDS
// round trip less.  T / 2 + 1 lanes have work (j = 0 .. T / 2): 61 of 64 for
// 720 = 20 x 6 x 6.  Lane 0 (j = 0) and lane T / 2 pair inside ONE butterfly.
//
// Pairs of a lane, s = 0 / 1, u = 0 .. R/2 - 1 (A / B = outputs of j / T - j):
//   s = 0:  a = A[u], b = B[R-1-u]   bins k = j + u T          and N2 - k
//   s = 1:  a = B[u], b = A[R-1-u]   bins k = (T - j) + u T    and N2 - k
// (k <= N2 / 2 in both: `a` is the low bin recombine_pair() wants first).
template <int N2, int R, int IN_PAD>
struct PairedLast {
  static constexpr int T = N2 / R;
  static constexpr int H = R / 2;
  static constexpr int NP = T / 2 + 1;       // lanes with work
  static constexpr int IN_STEP = T + IN_PAD;  // slots between inputs r, r + 1
  static_assert(R % 2 == 0 && T % 2 == 0 && N2 % R == 0, "paired last pass");
  static_assert(NP <= kLanes, "one round of pairs");

  static WB2_HD int ja(int lane) { return lane < NP ? lane : NP - 1; }
  static WB2_HD bool self_paired(int lane) {
    const int j = ja(lane);
    return j == 0 || 2 * j == T;
  }
  static WB2_HD int jb(int lane) {
    return self_paired(lane) ? ja(lane) : T - ja(lane);
  }
  // bin of pair (s, u): the low one (p1 of recombine_pair); the high one is
  // N2 - low
  static WB2_HD int low_bin(int lane, int s, int u) {
    const int j = ja(lane);
    return (s == 0 ? j : T - j) + u * T;
  }
  // which results of pair (s, u) are this lane's to keep: p1 / p2
  static WB2_HD bool keeps(int lane, int s, int u, int which) {
    if (lane >= NP) return false;
    if (s == 0) return true;
    const int j = ja(lane);
    if (2 * j == T) return false;                 // B is A: s = 1 repeats s = 0
    if (j == 0) return u == H - 1 && which == 0;  // bin N2 / 2, once
    return true;
  }
  template <typename C>
  static WB2_HD void load_twiddles(const C* __restrict__ twz, int lane,
                                   C (&tw)[2][R - 1]) {
#pragma unroll
    for (int r = 1; r < R; ++r) {
      tw[0][r - 1] = twz[ja(lane) * r];
      tw[1][r - 1] = twz[jb(lane) * r];
    }
  }
  // twq[k] of the low bins of the lane's pairs (twq has N2 / 2 + 1 entries)
  template <typename C>
  static WB2_HD void load_recombination(const C* __restrict__ twq, int lane,
                                        C (&wq)[2][H]) {
#pragma unroll
    for (int u = 0; u < H; ++u) {
      wq[0][u] = twq[low_bin(lane, 0, u)];
      const int k = low_bin(lane, 1, u);
      wq[1][u] = twq[k <= N2 / 2 ? k : N2 / 2];
    }
  }
  template <typename Load, typename C>
  static WB2_HD void load(const Load& src, int lane, C (&v)[2][R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      v[0][r] = src(ja(lane) + r * IN_STEP);
      v[1][r] = src(jb(lane) + r * IN_STEP);
    }
  }
  // after twiddles and butterflies: lane 0's partner outputs are its own,
  // Z[T + t T] = A[(t + 1) mod R]
  template <typename C>
  static WB2_HD void fix_lane0(int lane, C (&v)[2][R]) {
    const bool first = lane == 0;
#pragma unroll
    for (int t = 0; t < R; ++t) {
      const C own = v[0][(t + 1) % R];
      v[1][t].x = first ? own.x : v[1][t].x;
      v[1][t].y = first ? own.y : v[1][t].y;
    }
  }
  template <typename C>
  static WB2_HD void recombine(const C (&v)[2][R], const C (&wq)[2][H],
                               scalar_t<C> half_inv_n,
                               scalar_t<C> (&p)[2][H][2]) {
#pragma unroll
    for (int u = 0; u < H; ++u) {
      recombine_pair(v[0][u], v[1][R - 1 - u], wq[0][u], half_inv_n, p[0][u][0],
                     p[0][u][1]);
      recombine_pair(v[1][u], v[0][R - 1 - u], wq[1][u], half_inv_n, p[1][u][0],
                     p[1][u][1]);
    }
  }
};

// Plans whose last pass is paired (selected by the reducing modes of K4f where
// measured faster): 720 = 20 x 6 x 6 -> 36 lanes x 20 points from HBM, two
// rounds of radix 6, 61 lanes x two radix-6 butterflies.
template <int N2>
struct PairedPlan;
template <>
struct PairedPlan<720> {
  static constexpr int R0 = 20, R1 = 6, R2 = 6;
  static constexpr int PAD0 = WB2_FFT_PAIRED_PAD0, PAD1 = WB2_FFT_PAIRED_PAD1;
};

// The two tables a plan keeps (evaluated in fp64, rounded once):
//   twz[j] = exp(-2 pi i j / N2)                    j = 0 .. N2-1
//   twq[k] = exp(-2 pi i k / N) * (-i) * (0.5 / N)  k = 0 .. N2/2
template <typename C>
WB2_HD void table_entry_z(int j, int n2, double cs, double sn, C& out) {
  (void)j;
  (void)n2;
  out.x = (scalar_t<C>)cs;
  out.y = (scalar_t<C>)(-sn);
}
template <typename C>
WB2_HD void table_entry_q(int n2, double cs, double sn, C& out) {
  // (cs - i sn) * (-i) = -sn - i cs
  const double s = 0.5 / (2.0 * (double)n2);
  out.x = (scalar_t<C>)(-sn * s);
  out.y = (scalar_t<C>)(-cs * s);
}

}  // namespace fftcore
}  // namespace wb2
