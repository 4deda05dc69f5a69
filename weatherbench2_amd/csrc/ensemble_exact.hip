// K3 with a compile-time member count: compiled once per entry of
// WB2_SORT3_SIZES (build.py passes -DWB2_ENS_M=<members> -DWB2_ENS_NPAD=<padded
// register count>), one object each -- float32 members at a constant stride,
// the 2- / 3-sorter program of sort3_networks.inc, no per-member selects.  One
// translation unit per size keeps the build parallel.
#include "ensemble_kernels.hpp"

#if !defined(WB2_ENS_M) || !defined(WB2_ENS_NPAD)
#error "compile with -DWB2_ENS_M=<members> -DWB2_ENS_NPAD=<padded count>"
#endif

#define WB2_CAT2(a, b) a##b
#define WB2_CAT(a, b) WB2_CAT2(a, b)

namespace wb2 {

int WB2_CAT(launch_ens_exact_f32_, WB2_ENS_M)(const EnsParams& p, bool skipna,
                                              bool wf, hipStream_t stream) {
  return launch_ens<float, WB2_ENS_NPAD, WB2_ENS_M>(p, skipna, wf, stream);
}

// ... and as the host of every smaller runtime member count (ens_point_hosted)
int WB2_CAT(launch_ens_hosted_f32_, WB2_ENS_M)(const EnsParams& p, bool skipna,
                                               bool wf, hipStream_t stream) {
  return launch_ens_hosted<WB2_ENS_NPAD, WB2_ENS_M>(p, skipna, wf, stream);
}

}  // namespace wb2
