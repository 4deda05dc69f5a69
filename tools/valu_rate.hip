// Issue-rate microbenchmark for the instruction forms K4f / K3 are built from
// (round 3: is the VALU really only 60 % busy in K4f, or do the packed-fp32
// FMAs take more than the 4 cycles SQ_ACTIVE_INST_VALU books for them?).
//
//   hipcc --offload-arch=gfx950 -O3 -o build/valu_rate tools/valu_rate.hip
//   ./build/valu_rate            -> one line per instruction form:
//       cycles per wave-instruction per SIMD, at 1 / 2 / 3 waves per SIMD
//
// Every kernel runs REPS x 16 independent instructions of one form per wave
// (16 accumulator chains, so dependent-issue latency does not enter), one
// workgroup per CU; the clock comes from s_memrealtime-free wall time against
// the plain v_fma_f32 loop, which is known to issue in 4 cycles.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REPEAT16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum Form {
  FMA_F32 = 0, PK_FMA, PK_FMA_OPSEL, PK_MUL, PK_ADD, PK_FMA_SHARED, FMA_F64,
  ADD_F64, MUL_F64, CVT_F64_F32, MED3_F32, MIN_F32, PK_FMA_LITERAL,
  DS_READ_B64, DS_READ2_B64, DS_READ_B128, DS_WRITE_B64, DS_WRITE2_B64,
  DS_WRITE_B128, N_FORMS
};
static const char* kNames[N_FORMS] = {
    "v_fma_f32", "v_pk_fma_f32 (3 distinct 64-bit sources)",
    "v_pk_fma_f32 op_sel/neg (the cmul form)", "v_pk_mul_f32", "v_pk_add_f32",
    "v_pk_fma_f32 (a, a, acc: 2 distinct sources)", "v_fma_f64", "v_add_f64",
    "v_mul_f64", "v_cvt_f64_f32", "v_med3_f32", "v_min_f32",
    "v_pk_fma_f32 with an SGPR-pair constant", "ds_read_b64", "ds_read2_b64",
    "ds_read_b128", "ds_write_b64", "ds_write2_b64", "ds_write_b128"};

template <int FORM>
__global__ void __launch_bounds__(768) rate_kernel(float* out, int reps,
                                                   float seed) {
  __shared__ __attribute__((aligned(16))) float lds[768 * 8 + 64];
  const int t = threadIdx.x;
  f2 a[16], b = {seed, seed * 0.5f}, c = {0.25f, seed};
  double d[16], e = (double)seed;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a[i] = f2{seed + i, seed - i};
    d[i] = (double)(seed + i);
  }
  // conflict-free addresses: consecutive lanes own consecutive 8 / 16 bytes
  constexpr bool kWide = FORM == DS_READ2_B64 || FORM == DS_READ_B128 ||
                         FORM == DS_WRITE2_B64 || FORM == DS_WRITE_B128;
  float* mine = lds + t * (kWide ? 4 : 2);
  for (int i = t; i < 768 * 8; i += blockDim.x) lds[i] = seed + i;
  __syncthreads();
  for (int r = 0; r < reps; ++r) {
    if constexpr (FORM == FMA_F32) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
      REPEAT16(X)
#undef X
    } else if constexpr (FORM == PK_FMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      REPEAT16(X)
#undef X
    } else if constexpr (FORM == PK_FMA_OPSEL) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(a[i]) : "v"(b), "v"(c));
      REPEAT16(X)
#undef X
    } else if constexpr (FORM == PK_MUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REPEAT16(X)
#undef X
    } else if constexpr (FORM == PK_ADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REPEAT16(X)
#undef X
    } else if constexpr (FORM == PK_FMA_SHARED) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b));
      REPEAT16(X)
#undef X
    } else if constexpr (FORM == PK_FMA_LITERAL) {
      f2 k = {0.5f, -0.5f};
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "s"(k));
      REPEAT16(X)
#undef X
    } else if constexpr (FORM == FMA_F64) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d[i]) : "v"(e));
      REPEAT16(X)
#undef X
    } else if constexpr (FORM == ADD_F64) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e));
      REPEAT16(X)
#undef X
    } else if constexpr (FORM == MUL_F64) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e));
      REPEAT16(X)
#undef X
    } else if constexpr (FORM == CVT_F64_F32) {
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i].x));
      REPEAT16(X)
#undef X
    } else if constexpr (FORM == MED3_F32) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
      REPEAT16(X)
#undef X
    } else if constexpr (FORM == MIN_F32) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
      REPEAT16(X)
#undef X
    } else if constexpr (FORM == DS_READ_B64) {
      const unsigned addr = (unsigned)(size_t)mine;
#define X(i) asm volatile("ds_read_b64 %0, %1" : "=v"(a[i]) : "v"(addr));
      REPEAT16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else if constexpr (FORM == DS_READ2_B64) {
      const unsigned addr = (unsigned)(size_t)mine;
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 q[8];
#define X(i) if (i < 8) asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(q[i & 7]) : "v"(addr));
      REPEAT16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i].x += q[i].x + q[i].w;
    } else if constexpr (FORM == DS_READ_B128) {
      const unsigned addr = (unsigned)(size_t)mine;
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 q[8];
#define X(i) if (i < 8) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 7]) : "v"(addr));
      REPEAT16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i].x += q[i].x + q[i].w;
    } else if constexpr (FORM == DS_WRITE_B64) {
      const unsigned addr = (unsigned)(size_t)mine;
#define X(i) asm volatile("ds_write_b64 %0, %1" : : "v"(addr), "v"(a[i]) : "memory");
      REPEAT16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else if constexpr (FORM == DS_WRITE2_B64) {
      const unsigned addr = (unsigned)(size_t)mine;
#define X(i) if (i < 8) asm volatile("ds_write2_b64 %0, %1, %2 offset1:1" : : "v"(addr), "v"(a[i]), "v"(a[i + 8]) : "memory");
      REPEAT16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else if constexpr (FORM == DS_WRITE_B128) {
      const unsigned addr = (unsigned)(size_t)mine;
      typedef float f4 __attribute__((ext_vector_type(4)));
#define X(i) if (i < 8) { f4 q = {a[i].x, a[i].y, a[i + 8].x, a[i + 8].y}; asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(q) : "memory"); }
      REPEAT16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)");
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y + (float)d[i];
  if (s == 1.2345f) out[t] = s + mine[0];
}

template <int FORM>
float time_form(int threads, int reps, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  for (int w = 0; w < 2; ++w)
    hipLaunchKernelGGL(rate_kernel<FORM>, dim3(cus), dim3(threads), 0, 0, out,
                       reps, 1.0f);
  hipEventRecord(e0, 0);
  for (int w = 0; w < 5; ++w)
    hipLaunchKernelGGL(rate_kernel<FORM>, dim3(cus), dim3(threads), 0, 0, out,
                       reps, 1.0f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5.f;
}

template <int FORM>
void report(float* out, const float* base_ms) {
  const int reps = 4096;
  // wave-instructions per wave: reads2 / writes2 / b128 forms issue 8 per rep
  const bool wide = FORM == DS_READ2_B64 || FORM == DS_READ_B128 ||
                    FORM == DS_WRITE2_B64 || FORM == DS_WRITE_B128;
  const double per_rep = wide ? 8.0 : 16.0;
  printf("%-48s", kNames[FORM]);
  for (int wps = 1; wps <= 3; ++wps) {
    const float ms = time_form<FORM>(256 * wps, reps, out);
    // the v_fma_f32 loop at the same occupancy issues one instruction per 4
    // cycles and SIMD: cycles = 4 * (ms / base_ms) * (16 / per_rep)
    const double cyc = 4.0 * ms / base_ms[wps - 1] * (16.0 / per_rep);
    printf("  %6.2f", cyc);
  }
  const bool lds = FORM >= DS_READ_B64;
  printf("   cycles per wave-instruction per %s (1 / 2 / 3 waves per SIMD)\n",
         lds ? "SIMD [x4 SIMDs share one LDS]" : "SIMD");
}

int main() {
  float* out = nullptr;
  hipMalloc(&out, 4096 * sizeof(float));
  // reference: the plain v_fma_f32 loop at the same occupancy issues one
  // instruction per 4 cycles and SIMD (VALU-bound: its time grows with the
  // number of waves), so for any form
  //   cycles per wave-instruction per SIMD = 4 * ms_form(w) / ms_fma(w)
  float ref[3];
  for (int wps = 1; wps <= 3; ++wps)
    ref[wps - 1] = time_form<FMA_F32>(256 * wps, 4096, out);
  printf("v_fma_f32 loop: %.4f / %.4f / %.4f ms at 1 / 2 / 3 waves per SIMD "
         "(proportional = VALU-bound)\n", ref[0], ref[1], ref[2]);
  report<FMA_F32>(out, ref);
  report<PK_FMA>(out, ref);
  report<PK_FMA_OPSEL>(out, ref);
  report<PK_FMA_SHARED>(out, ref);
  report<PK_FMA_LITERAL>(out, ref);
  report<PK_MUL>(out, ref);
  report<PK_ADD>(out, ref);
  report<FMA_F64>(out, ref);
  report<ADD_F64>(out, ref);
  report<MUL_F64>(out, ref);
  report<CVT_F64_F32>(out, ref);
  report<MED3_F32>(out, ref);
  report<MIN_F32>(out, ref);
  report<DS_READ_B64>(out, ref);
  report<DS_READ2_B64>(out, ref);
  report<DS_READ_B128>(out, ref);
  report<DS_WRITE_B64>(out, ref);
  report<DS_WRITE2_B64>(out, ref);
  report<DS_WRITE_B128>(out, ref);
  hipFree(out);
  return 0;
}
