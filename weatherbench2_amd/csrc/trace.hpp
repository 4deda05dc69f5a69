// roctx ranges around every C-ABI call (tracing): with
// `rocprofv3 --marker-trace` a timeline shows which wb2_* call enqueued which
// kernels.  The marker library is looked up at run time (rocprofiler-sdk's
// roctx, then the legacy libroctx64); without one the ranges are no-ops, so
// libwb2hip.so has no link-time dependency on a profiler.
#pragma once

namespace wb2 {

void trace_push(const char* name);
void trace_pop();

struct TraceRange {
  explicit TraceRange(const char* name) { trace_push(name); }
  ~TraceRange() { trace_pop(); }
  TraceRange(const TraceRange&) = delete;
  TraceRange& operator=(const TraceRange&) = delete;
};

}  // namespace wb2

#define WB2_TRACE() ::wb2::TraceRange wb2_trace_range_(__func__)
