// K3 with a compile-time member count of 30 float32 members (padded to 32
// registers): the sorting program of sort3_networks.inc, no per-member
// selects.  One translation unit per size keeps the build parallel.
#include "ensemble_kernels.hpp"

namespace wb2 {

int launch_ens_exact_f32_30(const EnsParams& p, bool skipna, bool wf,
                            hipStream_t stream) {
  return launch_ens<float, 32, 30>(p, skipna, wf, stream);
}

}  // namespace wb2
