// ltr_k_plan.hip -- the plan kernel (ltr_dp_plan.hpp): every one-wave class and packed strip width of a plan in one
// persistent launch, pairs whose certificate fails scored in line (ltr_dp_redo.hpp).  Symmetric indel models only (the
// LongTR defaults and every model with ins->match == del->match, match->ins == match->del); others keep a launch per class.
#include <hip/hip_runtime.h>

#include "ltr_kernels.h"

namespace {
#include "ltr_dp_kernel.hpp"
#include "ltr_dp_pack.hpp"
#include "ltr_dp_chain.hpp"
#include "ltr_dp_plan.hpp"
}  // namespace

namespace ltrk {
hipError_t occ_plan(int* per_cu) { return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ltr_dp_plan_kernel<true>, 64 * kBlockWaves, 0); }
void launch_plan(dim3 grid, hipStream_t st, const KernelArgs& A) {
  hipLaunchKernelGGL((ltr_dp_plan_kernel<true>), grid, dim3(64 * kBlockWaves), 0, st, A);
}
}  // namespace ltrk
