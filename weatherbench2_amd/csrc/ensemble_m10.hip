// K3 with a compile-time member count of 10 float32 members (padded to 16
// registers): the sorting program of sort3_networks.inc, no per-member
// selects.  One translation unit per size keeps the build parallel.
#include "ensemble_kernels.hpp"

namespace wb2 {

int launch_ens_exact_f32_10(const EnsParams& p, bool skipna, bool wf,
                            hipStream_t stream) {
  return launch_ens<float, 16, 10>(p, skipna, wf, stream);
}

}  // namespace wb2
