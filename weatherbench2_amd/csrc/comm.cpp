// Host-side pieces of the C ABI that are not kernels:
//   wb2_lat_weights          latitude / area weights        metrics.py:35-60
//   wb2_comm_* / wb2_time_mean_allreduce
//                            the path's ONE exchange step: the all-reduce of the
//                            (sum, count) accumulators of the temporal mean over
//                            init-time shards (xbeam.Mean, evaluation.py:735-744)
//                            on RCCL, for callers that do not go through
//                            torch.distributed
//   roctx ranges             trace.hpp
// RCCL and the marker library are opened with dlopen on first use (a process
// that already loaded them -- PyTorch ships its own copies -- gets those).
#include "common.hpp"
#include "trace.hpp"
#include "wb2hip.h"

#include <dlfcn.h>

#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace wb2 {
namespace {

void* open_first(const char* const* names) {
  for (; *names; ++names)
    if (void* h = dlopen(*names, RTLD_NOW | RTLD_GLOBAL)) return h;
  return nullptr;
}

// ---- roctx ---------------------------------------------------------------------
typedef int (*range_push_fn)(const char*);
typedef int (*range_pop_fn)();
struct Markers {
  range_push_fn push = nullptr;
  range_pop_fn pop = nullptr;
  Markers() {
    static const char* const libs[] = {"librocprofiler-sdk-roctx.so",
                                       "librocprofiler-sdk-roctx.so.0",
                                       "libroctx64.so", "libroctx64.so.4",
                                       nullptr};
    if (void* h = open_first(libs)) {
      push = reinterpret_cast<range_push_fn>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<range_pop_fn>(dlsym(h, "roctxRangePop"));
      if (!push || !pop) push = nullptr, pop = nullptr;
    }
  }
};
const Markers& markers() {
  static const Markers m;
  return m;
}

// ---- RCCL ----------------------------------------------------------------------
// The handful of entry points used, with RCCL's (= NCCL's) C signatures.
struct NcclUniqueId {
  char internal[128];
};
typedef int (*get_unique_id_fn)(NcclUniqueId*);
typedef int (*comm_init_rank_fn)(void**, int, NcclUniqueId, int);
typedef int (*comm_destroy_fn)(void*);
typedef int (*all_reduce_fn)(const void*, void*, size_t, int, int, void*,
                             hipStream_t);
typedef int (*group_fn)();
typedef const char* (*error_string_fn)(int);
constexpr int kNcclFloat64 = 8, kNcclSum = 0;  // ncclDataType_t / ncclRedOp_t

struct Rccl {
  get_unique_id_fn get_unique_id = nullptr;
  comm_init_rank_fn comm_init_rank = nullptr;
  comm_destroy_fn comm_destroy = nullptr;
  all_reduce_fn all_reduce = nullptr;
  group_fn group_start = nullptr, group_end = nullptr;
  error_string_fn error_string = nullptr;
  bool ok = false;
  // why `ok` is false, captured ONCE here: dlerror() returns NULL when the
  // library opened but a symbol is missing, and after any earlier dlerror()
  std::string why;
  Rccl() {
    static const char* const libs[] = {"librccl.so.1", "librccl.so", nullptr};
    void* h = open_first(libs);
    if (!h) {
      const char* msg = dlerror();
      why = msg ? msg : "librccl.so.1 / librccl.so not found";
      return;
    }
    get_unique_id =
        reinterpret_cast<get_unique_id_fn>(dlsym(h, "ncclGetUniqueId"));
    comm_init_rank =
        reinterpret_cast<comm_init_rank_fn>(dlsym(h, "ncclCommInitRank"));
    comm_destroy = reinterpret_cast<comm_destroy_fn>(dlsym(h, "ncclCommDestroy"));
    all_reduce = reinterpret_cast<all_reduce_fn>(dlsym(h, "ncclAllReduce"));
    group_start = reinterpret_cast<group_fn>(dlsym(h, "ncclGroupStart"));
    group_end = reinterpret_cast<group_fn>(dlsym(h, "ncclGroupEnd"));
    error_string =
        reinterpret_cast<error_string_fn>(dlsym(h, "ncclGetErrorString"));
    ok = get_unique_id && comm_init_rank && comm_destroy && all_reduce &&
         group_start && group_end;
    if (!ok) why = "librccl opened but an ncclXxx entry point is missing";
  }
};
const Rccl& rccl() {
  static const Rccl r;
  return r;
}
int rccl_fail(const char* what, int rc) {
  const Rccl& r = rccl();
  return fail("%s failed: %s (ncclResult %d)", what,
              r.error_string ? r.error_string(rc) : "?", rc);
}

// ---- latitude weights -----------------------------------------------------------
// metrics.py:35-60, operation by operation in the coordinate's dtype T:
//   bounds = [-pi/2, midpoints of deg2rad(lat), +pi/2]        (must increase)
//   w = sin(upper) - sin(lower);  w /= mean(w)
// np.mean is a pairwise sum; sin is the C library's (NumPy's SIMD sin may differ
// in the last ulp -- the Python host keeps using NumPy itself, plan.py).
template <typename T>
T pairwise_sum(const T* a, long long n) {
  if (n <= 8) {
    T s = 0;
    for (long long i = 0; i < n; ++i) s += a[i];
    return s;
  }
  const long long h = n / 2;
  return pairwise_sum(a, h) + pairwise_sum(a + h, n - h);
}

template <typename T>
int lat_weights(const T* lat, long long n, T* out) {
  const T pi_over_2 = (T)1.5707963267948966192313216916397514L;
  const T deg = (T)0.017453292519943295769236907684886127L;  // pi / 180
  std::vector<T> bounds((size_t)n + 1);
  bounds[0] = -pi_over_2;
  for (long long i = 0; i + 1 < n; ++i)
    bounds[(size_t)i + 1] = (lat[i] * deg + lat[i + 1] * deg) / (T)2;
  bounds[(size_t)n] = pi_over_2;
  for (long long i = 0; i < n; ++i)
    if (!(bounds[(size_t)i + 1] > bounds[(size_t)i]))
      return fail("latitude cell bounds are not increasing at index %lld "
                  "(flip a decreasing latitude axis first, evaluation.py:41-47)",
                  i);
  for (long long i = 0; i < n; ++i)
    out[i] = std::sin(bounds[(size_t)i + 1]) - std::sin(bounds[(size_t)i]);
  const T mean = pairwise_sum(out, n) / (T)n;
  for (long long i = 0; i < n; ++i) out[i] /= mean;
  return 0;
}

}  // namespace

void trace_push(const char* name) {
  const Markers& m = markers();
  if (m.push) m.push(name);
}
void trace_pop() {
  const Markers& m = markers();
  if (m.pop) m.pop();
}

}  // namespace wb2

extern "C" {

int wb2_lat_weights(int dtype, const void* latitude, int64_t n, void* out) {
  using namespace wb2;
  WB2_TRACE();
  WB2_REQUIRE(latitude && out && n >= 1, "null pointer or empty latitude");
  if (dtype == WB2_F64)
    return lat_weights(static_cast<const double*>(latitude), n,
                       static_cast<double*>(out));
  if (dtype == WB2_F32)
    return lat_weights(static_cast<const float*>(latitude), n,
                       static_cast<float*>(out));
  return fail("unknown dtype %d", dtype);
}

int wb2_comm_unique_id(void* id128) {
  using namespace wb2;
  WB2_TRACE();
  WB2_REQUIRE(id128, "null pointer argument");
  const Rccl& r = rccl();
  WB2_REQUIRE(r.ok, "RCCL is unavailable: %s", r.why.c_str());
  NcclUniqueId id;
  const int rc = r.get_unique_id(&id);
  if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
  std::memcpy(id128, id.internal, sizeof(id.internal));
  return 0;
}

int wb2_comm_init_rank(const void* id128, int32_t n_ranks, int32_t rank,
                       void** comm_out) {
  using namespace wb2;
  WB2_TRACE();
  WB2_REQUIRE(id128 && comm_out, "null pointer argument");
  WB2_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "rank %d of %d", rank,
              n_ranks);
  const Rccl& r = rccl();
  WB2_REQUIRE(r.ok, "RCCL is unavailable: %s", r.why.c_str());
  NcclUniqueId id;
  std::memcpy(id.internal, id128, sizeof(id.internal));
  void* comm = nullptr;
  const int rc = r.comm_init_rank(&comm, n_ranks, id, rank);
  if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
  *comm_out = comm;
  return 0;
}

int wb2_comm_destroy(void* comm) {
  using namespace wb2;
  WB2_TRACE();
  if (!comm) return 0;
  const Rccl& r = rccl();
  WB2_REQUIRE(r.ok, "RCCL is unavailable: %s", r.why.c_str());
  const int rc = r.comm_destroy(comm);
  if (rc != 0) return rccl_fail("ncclCommDestroy", rc);
  return 0;
}

int wb2_time_mean_allreduce(double* sum, double* count, int64_t n, void* comm,
                            void* stream) {
  using namespace wb2;
  WB2_TRACE();
  WB2_REQUIRE(sum && count && comm, "null pointer argument");
  WB2_REQUIRE(n >= 0, "n=%lld", (long long)n);
  if (n == 0) return 0;
  const Rccl& r = rccl();
  WB2_REQUIRE(r.ok, "RCCL is unavailable: %s", r.why.c_str());
  hipStream_t s = static_cast<hipStream_t>(stream);
  // one fused exchange: both buffers inside a group, in place
  int rc = r.group_start();
  if (rc != 0) return rccl_fail("ncclGroupStart", rc);
  const int rc1 = r.all_reduce(sum, sum, (size_t)n, kNcclFloat64, kNcclSum, comm, s);
  const int rc2 =
      r.all_reduce(count, count, (size_t)n, kNcclFloat64, kNcclSum, comm, s);
  rc = r.group_end();
  if (rc1 != 0) return rccl_fail("ncclAllReduce(sum)", rc1);
  if (rc2 != 0) return rccl_fail("ncclAllReduce(count)", rc2);
  if (rc != 0) return rccl_fail("ncclGroupEnd", rc);
  return 0;
}

}  // extern "C"
