// ltr_k_wgt.hip -- the workgroup-per-pair kernels with the exact threshold test (ltr_dp_wg.hpp, FULL = true) as the FIRST pass of a
// workgroup class: 4 waves W 6..20, 8 waves W 8..20, even strip widths (an odd class is launched with the next even width).
// Symmetric indel models only.  Replaces HapAligner::align_seq_to_hap (HapAligner.cpp:236-343) including its row abort
// (:283, :297-306) in one pass.
#include <hip/hip_runtime.h>

#include "ltr_kernels.h"

namespace {
#include "ltr_dp_kernel.hpp"
#include "ltr_dp_wg.hpp"

template <int NW, int WT, int WMIN, bool END = (WT < WMIN)>
struct WgtKernels {
  static hipError_t occupancy(int w, int* per_cu) {
    if (w != WT) return WgtKernels<NW, WT - 2, WMIN>::occupancy(w, per_cu);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ltr_dp_wg_kernel<WT, NW, true, true>, 64 * NW, 0);
  }
  static void launch(int w, dim3 grid, hipStream_t st, const KernelArgs& A) {
    if (w != WT) { WgtKernels<NW, WT - 2, WMIN>::launch(w, grid, st, A); return; }
    hipLaunchKernelGGL((ltr_dp_wg_kernel<WT, NW, true, true>), grid, dim3(64 * NW), 0, st, A);
  }
};
template <int NW, int WT, int WMIN>
struct WgtKernels<NW, WT, WMIN, true> {
  static hipError_t occupancy(int, int*) { return hipErrorInvalidValue; }
  static void launch(int, dim3, hipStream_t, const KernelArgs&) {}
};
static_assert(kWgWMax % 2 == 0 && kWg4MaxW % 2 == 0, "the recursion walks the even strip widths down from the widest");
}  // namespace

namespace ltrk {
int wgt_width(int W) { return W + (W & 1); }
hipError_t occ_wgt(int NW, int W, int* per_cu) {
  if (NW == 4) return WgtKernels<4, kWg4MaxW, 6>::occupancy(wgt_width(W), per_cu);
  return WgtKernels<8, kWgWMax, 8>::occupancy(wgt_width(W), per_cu);
}
void launch_wgt(int NW, int W, dim3 grid, hipStream_t st, const KernelArgs& A) {
  if (NW == 4) WgtKernels<4, kWg4MaxW, 6>::launch(wgt_width(W), grid, st, A);
  else WgtKernels<8, kWgWMax, 8>::launch(wgt_width(W), grid, st, A);
}
}  // namespace ltrk
