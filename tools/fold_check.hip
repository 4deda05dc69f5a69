// Checks reduce_common.hpp's halving-tree sums against a plain xor-butterfly
// per value, bit for bit, on the GPU (and each exchange primitive by itself).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iweatherbench2_amd/csrc -Iinclude -o build/fold_check tools/fold_check.hip
#include "reduce_common.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace wb2;

template <int N>
__global__ void check(const double* in, double* tree, int* slots, double* ref) {
  const int lane = threadIdx.x;
  double v[N], r[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    v[j] = in[j * 64 + lane];
    r[j] = wave_allsum(v[j]);
  }
  int slot;
  bool writes;
  wave_sum_many<N>(v, lane, slot, writes);
  tree[lane] = v[0];
  slots[lane] = writes ? slot : -1 - slot;
#pragma unroll
  for (int j = 0; j < N; ++j)
    if (lane == 0) ref[j] = r[j];
}

__global__ void prims(const double* in, double* out) {
  const int lane = threadIdx.x;
  double a = in[lane], b = in[64 + lane];
  double a32 = a, b32 = b;
  swap_halves32(a32, b32);
  double a16 = a, b16 = b;
  swap_halves16(a16, b16);
  out[lane] = a32;
  out[64 + lane] = b32;
  out[128 + lane] = a16;
  out[192 + lane] = b16;
  out[256 + lane] = dpp_move<0x128>(a);
  out[320 + lane] = dpp_move<0x4E>(a);
  out[384 + lane] = dpp_move<0xB1>(a);
  const unsigned long long u = __builtin_bit_cast(unsigned long long, a);
  const unsigned lo = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)u, 0x101F);
  const unsigned hi = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)(u >> 32), 0x101F);
  out[448 + lane] = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// the real epilogue: K = 3 and K = 6 slots over the SAME per-column sums for
// slots 0..2 must store the same bits (MSE of a DET pass == MSE of a DET_ACC pass)
template <int VEC, int K>
__global__ void fold_kernel(const double* in, const int* seg_col0,
                            const int* seg_eoff, int n_seg, int n_ts, int n_col,
                            double* out) {
  const int lane = threadIdx.x;
  double acc[1][VEC][K];
#pragma unroll
  for (int e = 0; e < VEC; ++e)
#pragma unroll
    for (int k = 0; k < K; ++k) acc[0][e][k] = in[(k * 8 + e) * 64 + lane];
  fold_tile_to_segs<1, VEC, K>(acc, lane, 0, lane * VEC, lane * VEC, n_col,
                               seg_col0, seg_eoff, n_seg, n_ts, out);
}

template <int VEC>
int run_fold(const std::vector<double>& host) {
  const int n_col = 64 * VEC - 5;
  std::vector<int> col0 = {0, 7, 7 + 64, n_col - 30, n_col};  // 4 segs
  const int n_seg = 4;
  std::vector<int> eoff = {0, 1, 2, 3, 4};
  double *in, *o3, *o6;
  int *c, *e;
  (void)hipMalloc(&in, host.size() * 8);
  (void)hipMalloc(&o3, 4 * 3 * 8);
  (void)hipMalloc(&o6, 4 * 6 * 8);
  (void)hipMalloc(&c, 5 * 4);
  (void)hipMalloc(&e, 5 * 4);
  (void)hipMemcpy(in, host.data(), host.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(c, col0.data(), 5 * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(e, eoff.data(), 5 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((fold_kernel<VEC, 3>), dim3(1), dim3(64), 0, 0, in, c, e, n_seg, 4, n_col, o3);
  hipLaunchKernelGGL((fold_kernel<VEC, 6>), dim3(1), dim3(64), 0, 0, in, c, e, n_seg, 4, n_col, o6);
  std::vector<double> h3(12), h6(24);
  (void)hipMemcpy(h3.data(), o3, 12 * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(h6.data(), o6, 24 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int s = 0; s < 4; ++s)
    for (int k = 0; k < 3; ++k)
      if (h3[s * 3 + k] != h6[s * 6 + k]) {
        ++bad;
        printf("  VEC=%d seg %d slot %d: K=3 %.17g  K=6 %.17g\n", VEC, s, k, h3[s * 3 + k], h6[s * 6 + k]);
      }
  printf("fold VEC=%d: K=3 vs K=6 mismatches=%d\n", VEC, bad);
  return bad;
}

template <int N>
int run(const std::vector<double>& host) {
  double *in, *tree, *ref;
  int* slots;
  (void)hipMalloc(&in, N * 64 * 8);
  (void)hipMalloc(&tree, 64 * 8);
  (void)hipMalloc(&ref, N * 8);
  (void)hipMalloc(&slots, 64 * 4);
  (void)hipMemcpy(in, host.data(), N * 64 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check<N>, dim3(1), dim3(64), 0, 0, in, tree, slots, ref);
  std::vector<double> t(64), r(N);
  std::vector<int> s(64);
  (void)hipMemcpy(t.data(), tree, 64 * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(r.data(), ref, N * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(s.data(), slots, 64 * 4, hipMemcpyDeviceToHost);
  int bad = 0, seen = 0;
  std::vector<int> have(N, 0);
  for (int l = 0; l < 64; ++l) {
    if (s[l] < 0) continue;
    ++seen;
    have[s[l]]++;
    if (t[l] != r[s[l]]) {
      ++bad;
      printf("  N=%d lane %d slot %d tree %.17g butterfly %.17g\n", N, l, s[l], t[l], r[s[l]]);
    }
  }
  int missing = 0;
  for (int j = 0; j < N; ++j) missing += have[j] != 1;
  printf("N=%d writers=%d mismatches=%d slots_not_written_once=%d\n", N, seen, bad, missing);
  return bad + missing;
}

int main() {
  std::vector<double> host(64 * 64);
  srand(1);
  for (auto& x : host) x = (rand() / (double)RAND_MAX - 0.5) * (1 << (rand() % 20));
  // primitives on lane ids
  std::vector<double> ids(128);
  for (int l = 0; l < 64; ++l) { ids[l] = l; ids[64 + l] = 100 + l; }
  double *pin, *pout;
  (void)hipMalloc(&pin, 128 * 8);
  (void)hipMalloc(&pout, 512 * 8);
  (void)hipMemcpy(pin, ids.data(), 128 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(prims, dim3(1), dim3(64), 0, 0, pin, pout);
  std::vector<double> po(512);
  (void)hipMemcpy(po.data(), pout, 512 * 8, hipMemcpyDeviceToHost);
  const char* names[8] = {"swap32 a", "swap32 b", "swap16 a", "swap16 b", "row_ror:8", "quad 0x4E", "quad 0xB1", "swizzle xor4"};
  for (int k = 0; k < 8; ++k) {
    printf("%-13s", names[k]);
    for (int l = 0; l < 64; ++l) printf(" %g", po[k * 64 + l]);
    printf("\n");
  }
  int bad = 0;
  bad += run<1>(host);
  bad += run<3>(host);
  bad += run<6>(host);
  bad += run<10>(host);
  bad += run<12>(host);
  bad += run<20>(host);
  bad += run_fold<4>(host);
  bad += run_fold<2>(host);
  bad += run_fold<1>(host);
  printf(bad ? "FAILED\n" : "OK\n");
  return bad != 0;
}
