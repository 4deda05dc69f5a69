// wb2_det_wind_suite_step with the pair kernel on a stream of its own beside
// the per-variable kernel (stream_reduce.hip; used by program.cpp): the two
// kernels read disjoint slabs, their ramps and tails overlap.  `pair_stream`
// is already ordered behind whatever produced the inputs; `stream` waits for
// `join_event` (recorded on pair_stream) before the folds.  pair_stream ==
// NULL: everything on `stream`, i.e. wb2_det_wind_suite_step.
#pragma once

#include "wb2hip.h"

namespace wb2 {

int det_wind_suite_step_streams(const wb2_plan_tables* plan, int mode,
                                int dtype, int skipna, const void* const* in,
                                const int64_t* const* slab, int aligned16,
                                int64_t n_outer, int64_t n_pair,
                                double* partials, double* wind_partials,
                                double* metrics, double* wind_metrics,
                                void* stream, void* pair_stream,
                                void* join_event);

}  // namespace wb2
