// ltr_k_exact.hip -- the exact (redo) kernels: the reference's cell-by-cell band-penalised row maximum
// (HapAligner.cpp:297-306).  One wavefront per pair (ltr_dp_kernel.hpp, EXACT = true) for reads of up to 1281
// bases and beyond 10241, a 4- / 8-wave workgroup per pair (ltr_dp_wg.hpp, ltr_dp_wgx_kernel) in between.
#include <hip/hip_runtime.h>

#include "ltr_kernels.h"

namespace {
#include "ltr_dp_kernel.hpp"
#include "ltr_dp_wg.hpp"
}  // namespace

namespace ltrk {
int exact_block_threads(int which) { return which == kXWg4 ? 64 * 4 : (which == kXWg8 ? 64 * 8 : 64 * kBlockWaves); }

hipError_t occ_exact(int which, int* per_cu) {
  switch (which) {
    // the general (non-SYM) body is the larger one: its occupancy is valid for both
    case kXGeneric: return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ltr_dp_kernel<kExactW, true, false, false>, 64 * kBlockWaves, 0);
    case kXShort: return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ltr_dp_kernel<kXShortW, true, true, true>, 64 * kBlockWaves, 0);
    case kXMid: return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ltr_dp_kernel<kXMidW, true, true, true>, 64 * kBlockWaves, 0);
    case kXLong: return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ltr_dp_kernel<kXLongW, true, true, true>, 64 * kBlockWaves, 0);
    case kXWg4: return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ltr_dp_wgx_kernel<4, 6, 10, 14>, 64 * 4, 0);
    case kXWg8: return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ltr_dp_wgx_kernel<8, 12, 16, 20>, 64 * 8, 0);
    case kXWideLaunch: return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ltr_dp_kernel<kXWideW, true, true, true>, 64 * kBlockWaves, 0);
    default: return hipErrorInvalidValue;
  }
}

void launch_exact(int which, bool sym, dim3 g, hipStream_t st, const KernelArgs& A) {
  const dim3 blk(64 * kBlockWaves);
  switch (which) {
    case kXGeneric:
      if (sym) hipLaunchKernelGGL((ltr_dp_kernel<kExactW, true, true, false>), g, blk, 0, st, A);
      else hipLaunchKernelGGL((ltr_dp_kernel<kExactW, true, false, false>), g, blk, 0, st, A);
      break;
    case kXShort: hipLaunchKernelGGL((ltr_dp_kernel<kXShortW, true, true, true>), g, blk, 0, st, A); break;
    case kXMid: hipLaunchKernelGGL((ltr_dp_kernel<kXMidW, true, true, true>), g, blk, 0, st, A); break;
    case kXLong: hipLaunchKernelGGL((ltr_dp_kernel<kXLongW, true, true, true>), g, blk, 0, st, A); break;
    case kXWideLaunch: hipLaunchKernelGGL((ltr_dp_kernel<kXWideW, true, true, true>), g, blk, 0, st, A); break;
    case kXWg4: hipLaunchKernelGGL((ltr_dp_wgx_kernel<4, 6, 10, 14>), g, dim3(64 * 4), 0, st, A); break;
    case kXWg8: hipLaunchKernelGGL((ltr_dp_wgx_kernel<8, 12, 16, 20>), g, dim3(64 * 8), 0, st, A); break;
    default: break;
  }
}
}  // namespace ltrk
