// ltr_k_plan.hip -- the plan kernel (ltr_dp_plan.hpp): every one-wave class and packed strip width of a plan in one
// persistent launch, pairs whose certificate fails scored in line (ltr_dp_redo.hpp).  Two instances: symmetric indel models (the
// LongTR defaults and every model with ins->match == del->match, match->ins == match->del: the 11-operation cell) and the
// general model (any seven negative transitions, HapAligner.h:111-119: 13 operations, failed certificates by the generic body).
#include <hip/hip_runtime.h>

#include "ltr_kernels.h"

namespace {
#include "ltr_dp_kernel.hpp"
#include "ltr_dp_pack.hpp"
#include "ltr_dp_chain.hpp"
#include "ltr_dp_plan.hpp"
}  // namespace

namespace ltrk {
hipError_t occ_plan(bool sym, int* per_cu) {
  if (sym) return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ltr_dp_plan_kernel<true>, 64 * kBlockWaves, 0);
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ltr_dp_plan_kernel<false>, 64 * kBlockWaves, 0);
}
void launch_plan(bool sym, dim3 grid, hipStream_t st, const KernelArgs& A) {
  if (sym) hipLaunchKernelGGL((ltr_dp_plan_kernel<true>), grid, dim3(64 * kBlockWaves), 0, st, A);
  else hipLaunchKernelGGL((ltr_dp_plan_kernel<false>), grid, dim3(64 * kBlockWaves), 0, st, A);
}
}  // namespace ltrk
