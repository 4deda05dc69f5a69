// CPU emulation of the K4f wave program: runs the per-lane code of
// weatherbench2_amd/csrc/fft_core.hpp lane by lane (every read phase of a pass
// before its write phase, exactly the order the DS queue of one wave enforces)
// and compares the one-sided power spectrum with a direct O(N^2) DFT in double.
//
//   hipcc --cuda-host-only -O2 -std=c++17 -I weatherbench2_amd/csrc \
//       tools/fft_host_check.hip -o /tmp/fft_host_check && /tmp/fft_host_check
//
// Prints one line per instantiated size: "N <n_lon> err <max |got-ref| / sum ref>".
#include "fft_core.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace wb2::fftcore;

template <typename P, int R, bool FIRST>
void run_pass(std::vector<cf>& z, const std::vector<cf>& twz, const float* x) {
  static cf v[kLanes][P::ROUNDS][R];
  for (int lane = 0; lane < kLanes; ++lane) {
    if constexpr (FIRST) {
      P::load([&](int i) { return cf{x[2 * i], x[2 * i + 1]}; }, lane, v[lane]);
    } else {
      cf tw[P::ROUNDS][P::NTW];
      P::load_twiddles(twz.data(), lane, tw);
      P::load([&](int i) { return z[i]; }, lane, v[lane]);
      P::twiddle(v[lane], tw);
    }
    P::butterflies(v[lane]);
  }
  for (int lane = 0; lane < kLanes; ++lane) P::store(z.data(), lane, v[lane]);
}

template <int N2>
double check(unsigned seed) {
  using PL = Plan<N2>;
  constexpr int N = 2 * N2, NH = N2 / 2 + 1;
  std::vector<cf> twz(N2), twq(NH), z(N2);
  for (int j = 0; j < N2; ++j) {
    const double a = 2.0 * kPi * j / N2;
    table_entry_z(j, N2, std::cos(a), std::sin(a), twz[j]);
  }
  for (int k = 0; k < NH; ++k) {
    const double a = kPi * k / N2;
    table_entry_q(N2, std::cos(a), std::sin(a), twq[k]);
  }
  std::vector<float> x(N);
  srand(seed);
  for (int i = 0; i < N; ++i)
    x[i] = (float)(rand() / (double)RAND_MAX - 0.5) * 4.0f +
           (float)std::cos(2.0 * kPi * 7 * i / N);
  using P0 = Pass<N2, PL::R0, 1, 1, 0, PL::PAD0>;
  using P1 = Pass<N2, PL::R1, PL::R0, PL::R0, PL::PAD0, PL::PAD1>;
  using P2 = Pass<N2, PL::R2, PL::R0 * PL::R1, PL::R0 * PL::R1, PL::PAD1, 0>;
  z.assign(slab_slots<N2>(), cf{0.0f, 0.0f});
  run_pass<P0, PL::R0, true>(z, twz, x.data());
  run_pass<P1, PL::R1, false>(z, twz, x.data());
  if constexpr (PL::R2 > 1) run_pass<P2, PL::R2, false>(z, twz, x.data());
  std::vector<double> got(N2 + 1, -1.0), ref(N2 + 1);
  const float half_inv_n = 0.5f / (float)N;
  constexpr int NIT = (NH + kLanes - 1) / kLanes;
  for (int lane = 0; lane < kLanes; ++lane)
    for (int i = 0; i < NIT; ++i) {
      const int k = lane + i * kLanes;
      if (k < NH) {
        float p1, p2;
        recombine_pair(z[k], z[k == 0 ? 0 : N2 - k], twq[k], half_inv_n, p1, p2);
        got[k] = (double)p1 * (k == 0 ? 1.0 : 2.0);
        if (2 * k != N2) got[N2 - k] = (double)p2 * 2.0;
      }
    }
  double tot = 0.0;
  for (int k = 0; k <= N2; ++k) {
    double re = 0.0, im = 0.0;
    for (int n = 0; n < N; ++n) {
      const double a = 2.0 * kPi * (double)((long long)n * k % N) / N;
      re += x[n] * std::cos(a);
      im -= x[n] * std::sin(a);
    }
    re /= N;
    im /= N;
    ref[k] = (re * re + im * im) * (k == 0 ? 1.0 : 2.0);
    tot += ref[k];
  }
  double err = 0.0;
  for (int k = 0; k <= N2; ++k) err = std::fmax(err, std::fabs(got[k] - ref[k]));
  return err / tot;
}

// The paired last pass (fft_core.hpp: PairedLast / PairedPlan): passes 0 and 1
// as above under the paired plan, then every lane's two butterflies, lane 0's
// fix-up and the in-register recombination; every bin must be produced exactly
// once.
template <int N2>
double check_paired(unsigned seed) {
  using PL = PairedPlan<N2>;
  constexpr int N = 2 * N2, NH = N2 / 2 + 1;
  std::vector<cf> twz(N2), twq(NH), z;
  for (int j = 0; j < N2; ++j) {
    const double a = 2.0 * kPi * j / N2;
    table_entry_z(j, N2, std::cos(a), std::sin(a), twz[j]);
  }
  for (int k = 0; k < NH; ++k) {
    const double a = kPi * k / N2;
    table_entry_q(N2, std::cos(a), std::sin(a), twq[k]);
  }
  std::vector<float> x(N);
  srand(seed);
  for (int i = 0; i < N; ++i)
    x[i] = (float)(rand() / (double)RAND_MAX - 0.5) * 4.0f +
           (float)std::cos(2.0 * kPi * 7 * i / N);
  using P0 = Pass<N2, PL::R0, 1, 1, 0, PL::PAD0>;
  using P1 = Pass<N2, PL::R1, PL::R0, PL::R0, PL::PAD0, PL::PAD1>;
  using PP = PairedLast<N2, PL::R2, PL::PAD1>;
  constexpr int R = PL::R2, H = PP::H;
  z.assign(slab_slots<N2, PL>(), cf{0.0f, 0.0f});
  run_pass<P0, PL::R0, true>(z, twz, x.data());
  run_pass<P1, PL::R1, false>(z, twz, x.data());
  std::vector<double> got(N2 + 1, 0.0), ref(N2 + 1);
  std::vector<int> times(N2 + 1, 0);
  const float half_inv_n = 0.5f / (float)N;
  for (int lane = 0; lane < kLanes; ++lane) {
    cf v[2][R], tw[2][R - 1], wq[2][H];
    PP::load_twiddles(twz.data(), lane, tw);
    PP::load_recombination(twq.data(), lane, wq);
    PP::load([&](int i) { return z[i]; }, lane, v);
    for (int s = 0; s < 2; ++s)
      for (int r = 1; r < R; ++r) v[s][r] = cmul(v[s][r], tw[s][r - 1]);
    Radix<R>::run(v[0]);
    Radix<R>::run(v[1]);
    PP::fix_lane0(lane, v);
    float p[2][H][2];
    PP::recombine(v, wq, half_inv_n, p);
    for (int s = 0; s < 2; ++s)
      for (int u = 0; u < H; ++u)
        for (int which = 0; which < 2; ++which)
          if (PP::keeps(lane, s, u, which)) {
            const int low = PP::low_bin(lane, s, u);
            const int k = which == 0 ? low : N2 - low;
            got[k] = (double)p[s][u][which] * (k == 0 ? 1.0 : 2.0);
            ++times[k];
          }
  }
  for (int k = 0; k <= N2; ++k)
    if (times[k] != 1) {
      std::printf("paired N %d: bin %d produced %d times\n", N, k, times[k]);
      return 1.0;
    }
  double tot = 0.0;
  for (int k = 0; k <= N2; ++k) {
    double re = 0.0, im = 0.0;
    for (int n = 0; n < N; ++n) {
      const double a = 2.0 * kPi * (double)((long long)n * k % N) / N;
      re += x[n] * std::cos(a);
      im -= x[n] * std::sin(a);
    }
    re /= N;
    im /= N;
    ref[k] = (re * re + im * im) * (k == 0 ? 1.0 : 2.0);
    tot += ref[k];
  }
  double err = 0.0;
  for (int k = 0; k <= N2; ++k) err = std::fmax(err, std::fabs(got[k] - ref[k]));
  return err / tot;
}

int main() {
  double worst = 0.0;
  {
    const double e = check_paired<720>(4321u);
    std::printf("P %d err %.3e\n", 1440, e);
    worst = std::fmax(worst, e);
  }
#define WB2_CHECK(N2)                                   \
  {                                                     \
    const double e = check<N2>(1234u + N2);             \
    std::printf("N %d err %.3e\n", 2 * N2, e);          \
    worst = std::fmax(worst, e);                        \
  }
  WB2_CHECK(32) WB2_CHECK(64) WB2_CHECK(120) WB2_CHECK(128) WB2_CHECK(180)
  WB2_CHECK(256) WB2_CHECK(360) WB2_CHECK(512) WB2_CHECK(720)
  WB2_CHECK(48) WB2_CHECK(144) WB2_CHECK(160) WB2_CHECK(192)
  WB2_CHECK(240) WB2_CHECK(320) WB2_CHECK(384) WB2_CHECK(640)
  WB2_CHECK(900) WB2_CHECK(1024) WB2_CHECK(1280) WB2_CHECK(1440) WB2_CHECK(1800)
  std::printf("worst %.3e\n", worst);
  return worst < 2e-6 ? 0 : 1;
}
