// Chunk programs: the launches of one chunk STRUCTURE, recorded once by the
// Python host (weatherbench2_amd/program.py) and replayed per chunk in ONE
// C-ABI call.
//
// The reference evaluates one chunk per call of _evaluate_chunk
// (/root/reference/weatherbench2/evaluation.py:583-599) and folds the chunk
// results with xbeam.Mean (:735-744): 2 920 x 40 calls in the official 0.25
// degree run, each walking the same Python.  The chunks of one evaluation share
// everything but the addresses of their arrays and their valid times, so a
// replay needs, per chunk:
//   ptrs[]    the base address of every array the launches read (the variables
//             of the forecast / truth chunk; resident arrays stay the same),
//   values[]  per gathered input (a climatology read by valid time) the slab
//             number of every (time, lead) cell of the chunk,
//   sinks     where the chunk's values are accumulated (device tables),
// and does, on the caller's stream:
//   1. the slab address of every input of every launch -> one pinned table,
//      one asynchronous copy on a copy stream of the program's (a ring of
//      tables: the table of chunk k travels while the kernels of chunk k - 1
//      run; the host waits only when it is kRing chunks ahead of the GPU);
//   2. every recorded launch -- wb2_det_suite_step / wb2_det_wind_suite_step,
//      or wb2_ens_partials_addr + wb2_ens_combine for an ensemble pass --
//      i.e. the kernels of the generic path, bit for bit; launches marked
//      `side` (the one-slab SEEPS passes: latency-bound) run on a second
//      stream beside the big one and join before step 3;
//   3. one wb2_gather_accumulate[_rows] per sink (eval config).
#include "common.hpp"
#include "suite_streams.hpp"
#include "trace.hpp"
#include "wb2hip.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

namespace wb2 {
namespace {

constexpr int kRing = 4;
constexpr int kSide = 2;

struct Gather {
  int input;
  long long first, count;
  int source;
  long long step_bytes;
  int value_offset;
  std::vector<int> cell;
  std::vector<long long> base;
};

struct Launch {
  wb2_plan_tables plan;
  int mode, dtype, skipna, n_in, side;
  // an ensemble pass (K3 + its fold): inputs = (member 0's slab, truth slab)
  bool ensemble = false;
  int n_member = 0;
  long long member_stride = 0;
  long long n_outer, n_pair;
  std::vector<int> slot;        // [n_in][n_outer]
  std::vector<long long> rel;   // [n_in][n_outer]
  std::vector<Gather> gathers;
  double* partials;
  double* wind_partials;
  long long arena_offset;
  long long table_offset;       // int64 entries into the chunk table
};

struct Sink {
  const int32_t* src;
  const uint8_t* round32;
  long long n_out, n_time;
  int skipna;
  const int32_t* sel;           // kept sinks: row of the chunk per entry
  const int64_t* rsz8;          //             bytes per row of its storage
};

struct Program {
  std::vector<Launch> launches;
  std::vector<Sink> sinks;
  double* arena = nullptr;
  long long table_len = 0;      // addresses of all launches
  long long max_rows = 0;       // rows of kept sinks ride behind them
  int n_ptrs = 0, n_values = 0;
  bool finalized = false;
  void* pinned[kRing] = {};
  void* device[kRing] = {};
  hipEvent_t copied[kRing] = {};   // table r is on the device (copy stream)
  hipEvent_t done[kRing] = {};     // the replay that used table r has finished
  bool busy[kRing] = {};
  int next = 0;
  // the table of replay k travels while the kernels of replay k - 1 run
  hipStream_t copy_stream = nullptr;
  // small latency-bound launches run beside the big ones: two high-priority
  // streams (their workgroups take the slots the streaming kernel frees)
  hipStream_t side[kSide] = {};
  hipEvent_t fork = nullptr, join[kSide] = {};
  bool any_side = false;
  // the pair kernel of a launch with wind-vector pairs beside its
  // per-variable kernel (WB2HIP_PAIR_STREAM=1; default: behind it)
  hipStream_t pair_stream = nullptr;
  hipEvent_t pair_join = nullptr;
  bool any_pairs = false;
  // host seconds spent in replay, by phase (wb2_program_stats): waiting for a
  // ring slot, filling the table, the copy, the launches, the sinks
  double spent[5] = {};
  long long replays = 0;
};

inline double now_s() {
  return std::chrono::duration<double>(
             std::chrono::steady_clock::now().time_since_epoch())
      .count();
}

int release(Program* p) {
  for (int i = 0; i < kRing; ++i) {
    if (p->busy[i]) (void)hipEventSynchronize(p->done[i]);
    if (p->copied[i]) (void)hipEventDestroy(p->copied[i]);
    if (p->done[i]) (void)hipEventDestroy(p->done[i]);
    if (p->pinned[i]) (void)hipHostFree(p->pinned[i]);
    if (p->device[i]) (void)hipFree(p->device[i]);
  }
  for (int i = 0; i < kSide; ++i) {
    if (p->side[i]) {
      (void)hipStreamSynchronize(p->side[i]);
      (void)hipStreamDestroy(p->side[i]);
    }
    if (p->join[i]) (void)hipEventDestroy(p->join[i]);
  }
  if (p->pair_stream) {
    (void)hipStreamSynchronize(p->pair_stream);
    (void)hipStreamDestroy(p->pair_stream);
  }
  if (p->pair_join) (void)hipEventDestroy(p->pair_join);
  if (p->fork) (void)hipEventDestroy(p->fork);
  if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
  delete p;
  return 0;
}

}  // namespace
}  // namespace wb2

extern "C" {

int wb2_program_create(void** program) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(program != nullptr, "null pointer argument");
  *program = new Program();
  return 0;
}

int wb2_program_destroy(void* program) {
  WB2_TRACE();
  if (!program) return 0;
  return wb2::release(static_cast<wb2::Program*>(program));
}

int wb2_program_add_launch(void* program, const wb2_plan_tables* plan, int mode,
                           int dtype, int skipna, int32_t n_in, int64_t n_outer,
                           int64_t n_pair, const int32_t* slot,
                           const int64_t* rel, double* partials,
                           double* wind_partials, int64_t arena_offset,
                           int side_stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(program && plan && slot && rel && partials,
              "null pointer argument");
  auto* p = static_cast<Program*>(program);
  WB2_REQUIRE(!p->finalized, "the program is finalized");
  WB2_REQUIRE(n_in >= 1 && n_in <= 4 && n_outer > 0 && n_pair >= 0 &&
                  2 * n_pair <= n_outer && arena_offset >= 0,
              "bad sizes: n_in=%d n_outer=%lld n_pair=%lld", n_in,
              (long long)n_outer, (long long)n_pair);
  WB2_REQUIRE(n_pair == 0 || wind_partials != nullptr,
              "a launch with pairs needs wind_partials");
  Launch la{};
  la.plan = *plan;
  la.mode = mode;
  la.dtype = dtype;
  la.skipna = skipna;
  la.n_in = n_in;
  la.side = side_stream != 0;
  la.n_outer = n_outer;
  la.n_pair = n_pair;
  const size_t n = (size_t)n_in * (size_t)n_outer;
  la.slot.assign(slot, slot + n);
  la.rel.assign(reinterpret_cast<const long long*>(rel),
                reinterpret_cast<const long long*>(rel) + n);
  for (size_t i = 0; i < n; ++i)
    WB2_REQUIRE(la.slot[i] >= 0, "negative pointer slot");
  la.partials = partials;
  la.wind_partials = wind_partials;
  la.arena_offset = arena_offset;
  la.table_offset = p->table_len;
  p->table_len += (long long)n;
  p->any_side = p->any_side || la.side;
  p->any_pairs = p->any_pairs || (la.n_pair > 0 && !la.side);
  p->launches.push_back(std::move(la));
  return 0;
}

int wb2_program_add_ens_launch(void* program, const wb2_plan_tables* plan,
                               int dtype, int skipna, int32_t n_member,
                               int64_t member_stride, int64_t n_outer,
                               const int32_t* slot, const int64_t* rel,
                               double* partials, int64_t arena_offset) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(program && plan && slot && rel && partials,
              "null pointer argument");
  auto* p = static_cast<Program*>(program);
  WB2_REQUIRE(!p->finalized, "the program is finalized");
  WB2_REQUIRE(n_member >= 1 && member_stride >= 0 && n_outer > 0 &&
                  arena_offset >= 0,
              "bad sizes: n_member=%d member_stride=%lld n_outer=%lld",
              n_member, (long long)member_stride, (long long)n_outer);
  WB2_REQUIRE(plan->wfield == nullptr || plan->wfield_dtype == WB2_F64,
              "the ensemble kernels read a float64 weight field");
  Launch la{};
  la.plan = *plan;
  la.mode = WB2_MODE_ENS;
  la.dtype = dtype;
  la.skipna = skipna;
  la.n_in = 2;
  la.ensemble = true;
  la.n_member = n_member;
  la.member_stride = member_stride;
  la.n_outer = n_outer;
  const size_t n = 2 * (size_t)n_outer;
  la.slot.assign(slot, slot + n);
  la.rel.assign(reinterpret_cast<const long long*>(rel),
                reinterpret_cast<const long long*>(rel) + n);
  for (size_t i = 0; i < n; ++i)
    WB2_REQUIRE(la.slot[i] >= 0, "negative pointer slot");
  la.partials = partials;
  la.arena_offset = arena_offset;
  la.table_offset = p->table_len;
  p->table_len += (long long)n;
  p->launches.push_back(std::move(la));
  return 0;
}

int wb2_program_add_gather(void* program, int32_t input, int64_t first,
                           int64_t count, int32_t source, int64_t step_bytes,
                           int32_t value_offset, const int32_t* cell,
                           const int64_t* base) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(program && cell && base, "null pointer argument");
  auto* p = static_cast<Program*>(program);
  WB2_REQUIRE(!p->finalized && !p->launches.empty(),
              "add a launch first (and before finalizing)");
  Launch& la = p->launches.back();
  WB2_REQUIRE(input >= 0 && input < la.n_in && first >= 0 && count > 0 &&
                  first + count <= la.n_outer && source >= 0 &&
                  value_offset >= 0,
              "bad gather: input=%d first=%lld count=%lld", input,
              (long long)first, (long long)count);
  Gather g{};
  g.input = input;
  g.first = first;
  g.count = count;
  g.source = source;
  g.step_bytes = step_bytes;
  g.value_offset = value_offset;
  g.cell.assign(cell, cell + count);
  g.base.assign(reinterpret_cast<const long long*>(base),
                reinterpret_cast<const long long*>(base) + count);
  for (long long k = 0; k < count; ++k)
    WB2_REQUIRE(g.cell[k] >= 0, "negative cell");
  la.gathers.push_back(std::move(g));
  return 0;
}

int wb2_program_add_sink(void* program, const int32_t* src,
                         const uint8_t* round32, int64_t n_out, int64_t n_time,
                         int skipna, const int32_t* sel, const int64_t* rsz8,
                         int64_t max_rows) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(program && src && round32, "null pointer argument");
  auto* p = static_cast<Program*>(program);
  WB2_REQUIRE(!p->finalized, "the program is finalized");
  WB2_REQUIRE(n_out > 0 && n_time > 0 && (sel == nullptr) == (rsz8 == nullptr) &&
                  max_rows >= 0,
              "bad sink");
  Sink s{};
  s.src = src;
  s.round32 = round32;
  s.n_out = n_out;
  s.n_time = n_time;
  s.skipna = skipna;
  s.sel = sel;
  s.rsz8 = rsz8;
  if (sel) p->max_rows += max_rows;
  p->sinks.push_back(s);
  return 0;
}

int wb2_program_finalize(void* program, double* arena, int32_t n_ptrs,
                         int32_t n_values) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(program && arena, "null pointer argument");
  auto* p = static_cast<Program*>(program);
  WB2_REQUIRE(!p->finalized && !p->launches.empty(), "nothing to finalize");
  WB2_REQUIRE(n_ptrs > 0 && n_values >= 0, "bad sizes");
  for (const Launch& la : p->launches) {
    for (int s : la.slot)
      WB2_REQUIRE(s < n_ptrs, "pointer slot %d >= n_ptrs=%d", s, n_ptrs);
    for (const Gather& g : la.gathers) {
      WB2_REQUIRE(g.source < n_ptrs, "gather source out of range");
      for (int c : g.cell)
        WB2_REQUIRE(g.value_offset + c < n_values,
                    "gather cell %d beyond n_values=%d", g.value_offset + c,
                    n_values);
    }
  }
  p->arena = arena;
  p->n_ptrs = n_ptrs;
  p->n_values = n_values;
  const size_t bytes = (size_t)(p->table_len + p->max_rows) * sizeof(int64_t);
  for (int i = 0; i < kRing; ++i) {
    WB2_HIP_OK(hipHostMalloc(&p->pinned[i], bytes, hipHostMallocDefault));
    WB2_HIP_OK(hipMalloc(&p->device[i], bytes));
    WB2_HIP_OK(hipEventCreateWithFlags(&p->copied[i], hipEventDisableTiming));
    WB2_HIP_OK(hipEventCreateWithFlags(&p->done[i], hipEventDisableTiming));
  }
  WB2_HIP_OK(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
  if (p->any_side) {
    int least = 0, greatest = 0;
    WB2_HIP_OK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    for (int i = 0; i < kSide; ++i) {
      WB2_HIP_OK(hipStreamCreateWithPriority(&p->side[i], hipStreamNonBlocking,
                                             greatest));
      WB2_HIP_OK(hipEventCreateWithFlags(&p->join[i], hipEventDisableTiming));
    }
    WB2_HIP_OK(hipEventCreateWithFlags(&p->fork, hipEventDisableTiming));
  }
  // measured (profiles/r06_round_log.md): beside each other the two kernels
  // gain 2 % in 24-chunk windows and lose 5 % chunk by chunk -- off unless
  // WB2HIP_PAIR_STREAM=1
  static const bool pair_stream_on = [] {
    const char* e = getenv("WB2HIP_PAIR_STREAM");
    return e && e[0] == '1';
  }();
  if (p->any_pairs && pair_stream_on) {
    WB2_HIP_OK(hipStreamCreateWithFlags(&p->pair_stream, hipStreamNonBlocking));
    WB2_HIP_OK(hipEventCreateWithFlags(&p->pair_join, hipEventDisableTiming));
    if (!p->fork)
      WB2_HIP_OK(hipEventCreateWithFlags(&p->fork, hipEventDisableTiming));
  }
  p->finalized = true;
  return 0;
}

int wb2_program_replay(void* program, const int64_t* ptrs, int32_t n_ptrs,
                       const int64_t* values, int32_t n_values,
                       const int64_t* sink_args, const int64_t* rows,
                       int32_t n_rows, void* stream) {
  WB2_TRACE();
  using namespace wb2;
  WB2_REQUIRE(program && ptrs && sink_args, "null pointer argument");
  auto* p = static_cast<Program*>(program);
  WB2_REQUIRE(p->finalized, "finalize the program first");
  WB2_REQUIRE(n_ptrs == p->n_ptrs && n_values == p->n_values &&
                  (n_values == 0 || values) && n_rows >= 0 &&
                  n_rows <= p->max_rows && (n_rows == 0 || rows),
              "argument counts differ from the program's (ptrs %d/%d, values "
              "%d/%d, rows %d/%lld)",
              n_ptrs, p->n_ptrs, n_values, p->n_values, n_rows, p->max_rows);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int r = p->next;
  p->next = (r + 1) % kRing;
  double t0 = now_s();
  // (table r -- pinned and device -- is free once the replay that used it has
  // finished; this is also what keeps the host at most kRing replays ahead)
  if (p->busy[r]) WB2_HIP_OK(hipEventSynchronize(p->done[r]));
  double t1 = now_s();
  p->spent[0] += t1 - t0;
  t0 = t1;
  // ---- 1. the chunk table ----
  long long* table = static_cast<long long*>(p->pinned[r]);
  std::vector<int> aligned(p->launches.size());
  for (size_t li = 0; li < p->launches.size(); ++li) {
    const Launch& la = p->launches[li];
    long long* out = table + la.table_offset;
    const size_t n = (size_t)la.n_in * (size_t)la.n_outer;
    long long low = 0;
    for (size_t i = 0; i < n; ++i) out[i] = ptrs[la.slot[i]] + la.rel[i];
    for (const Gather& g : la.gathers) {
      long long* o = out + (size_t)g.input * la.n_outer + g.first;
      const long long base_ptr = ptrs[g.source];
      const int64_t* v = values + g.value_offset;
      for (long long k = 0; k < g.count; ++k)
        o[k] = base_ptr + (v[g.cell[k]] + g.base[k]) * g.step_bytes;
    }
    for (size_t i = 0; i < n; ++i) low |= out[i];
    aligned[li] = (low & 15) == 0;
  }
  if (n_rows)
    std::memcpy(table + p->table_len, rows, (size_t)n_rows * sizeof(int64_t));
  const size_t bytes = (size_t)(p->table_len + n_rows) * sizeof(int64_t);
  t1 = now_s();
  p->spent[1] += t1 - t0;
  t0 = t1;
  WB2_HIP_OK(hipMemcpyAsync(p->device[r], table, bytes, hipMemcpyHostToDevice,
                            p->copy_stream));
  WB2_HIP_OK(hipEventRecord(p->copied[r], p->copy_stream));
  WB2_HIP_OK(hipStreamWaitEvent(s, p->copied[r], 0));
  p->busy[r] = true;
  t1 = now_s();
  p->spent[2] += t1 - t0;
  t0 = t1;
  // ---- 2. the launches: the side ones first (their packets are in their
  // queues before the streaming kernels fill the device) ----
  const int64_t* dev = static_cast<const int64_t*>(p->device[r]);
  auto run = [&](const Launch& la, int aligned16, hipStream_t ls) -> int {
    const int64_t* slabs[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int j = 0; j < la.n_in; ++j)
      slabs[j] = dev + la.table_offset + (long long)j * la.n_outer;
    double* metrics = p->arena + la.arena_offset;
    if (la.ensemble) {
      const wb2_plan_tables& t = la.plan;
      int rc = wb2_ens_partials_addr(
          la.dtype, la.skipna, slabs[0], slabs[1], la.n_member,
          la.member_stride, la.n_outer, t.n_row, t.n_col, t.w_row, t.w_col,
          static_cast<const double*>(t.wfield), t.chunk_row0, t.chunk_nrow,
          t.n_chunk, t.n_ctile, t.seg_col0, t.seg_eoff, t.n_seg, t.n_ts,
          la.partials, ls);
      if (rc != 0) return rc;
      return wb2_ens_combine(la.skipna, la.partials, la.n_outer, t.n_chunk,
                             t.wfield ? 2 : 1, t.n_seg, t.seg_eoff, t.n_ts,
                             t.band_chunk0, t.n_band, t.coef_band, t.coef_seg,
                             t.region_wf, t.region_wsum, t.n_region, nullptr,
                             metrics, ls);
    }
    if (la.n_pair > 0) {
      const long long n_det =
          (long long)WB2_NMETRIC * la.plan.n_region * la.n_outer;
      hipStream_t ps = ls == s ? p->pair_stream : nullptr;
      return det_wind_suite_step_streams(
          &la.plan, la.mode, la.dtype, la.skipna, nullptr, slabs, aligned16,
          la.n_outer, la.n_pair, la.partials, la.wind_partials, metrics,
          metrics + n_det, ls, ps, ps ? p->pair_join : nullptr);
    }
    return wb2_det_suite_step(&la.plan, la.mode, la.dtype, la.skipna, nullptr,
                              slabs, aligned16, la.n_outer, la.partials,
                              metrics, 0, 0, 0, 0, nullptr, nullptr, nullptr,
                              ls);
  };
  int n_side = 0;
  if (p->any_side || p->pair_stream) WB2_HIP_OK(hipEventRecord(p->fork, s));
  if (p->pair_stream) WB2_HIP_OK(hipStreamWaitEvent(p->pair_stream, p->fork, 0));
  if (p->any_side) {
    for (size_t li = 0; li < p->launches.size(); ++li) {
      const Launch& la = p->launches[li];
      if (!la.side) continue;
      hipStream_t ls = p->side[n_side % kSide];
      if (n_side < kSide) WB2_HIP_OK(hipStreamWaitEvent(ls, p->fork, 0));
      ++n_side;
      const int rc = run(la, aligned[li], ls);
      if (rc != 0) return rc;
    }
  }
  for (size_t li = 0; li < p->launches.size(); ++li) {
    const Launch& la = p->launches[li];
    if (la.side) continue;
    const int rc = run(la, aligned[li], s);
    if (rc != 0) return rc;
  }
  for (int i = 0; i < kSide && i < n_side; ++i) {
    WB2_HIP_OK(hipEventRecord(p->join[i], p->side[i]));
    WB2_HIP_OK(hipStreamWaitEvent(s, p->join[i], 0));
  }
  t1 = now_s();
  p->spent[3] += t1 - t0;
  t0 = t1;
  // ---- 3. the sinks ----
  long long row0 = 0;
  for (size_t si = 0; si < p->sinks.size(); ++si) {
    const Sink& k = p->sinks[si];
    const int64_t* a = sink_args + 3 * si;
    const int64_t* sum_addr = reinterpret_cast<const int64_t*>(a[0]);
    const int64_t* count_addr = reinterpret_cast<const int64_t*>(a[1]);
    int rc;
    if (k.sel) {
      const long long mine = a[2];
      WB2_REQUIRE(mine >= 0 && row0 + mine <= n_rows,
                  "sink %zu: rows out of range", si);
      rc = wb2_gather_accumulate_rows(p->arena, k.src, k.round32, k.n_out,
                                      k.n_time, k.skipna, sum_addr, count_addr,
                                      dev + p->table_len + row0, k.sel, k.rsz8,
                                      s);
      row0 += mine;
    } else {
      rc = wb2_gather_accumulate(p->arena, k.src, k.round32, k.n_out, k.n_time,
                                 k.skipna, sum_addr, count_addr, s);
    }
    if (rc != 0) return rc;
  }
  WB2_HIP_OK(hipEventRecord(p->done[r], s));
  p->spent[4] += now_s() - t0;
  ++p->replays;
  return 0;
}

int wb2_program_stats(void* program, double* seconds, int64_t* replays) {
  using namespace wb2;
  WB2_REQUIRE(program && seconds && replays, "null pointer argument");
  auto* p = static_cast<Program*>(program);
  for (int i = 0; i < 5; ++i) seconds[i] = p->spent[i];
  *replays = p->replays;
  return 0;
}

}  // extern "C"
