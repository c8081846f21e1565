// ltr_k_one.hip -- the one-pair-per-wavefront certificate kernels (ltr_dp_kernel.hpp), strip widths 1..kWMax.
#include <hip/hip_runtime.h>

#include "ltr_kernels.h"

namespace {
#include "ltr_dp_kernel.hpp"

template <int WT>
struct FastKernels {
  static hipError_t occupancy(int w, int* per_cu) {
    if (w != WT) return FastKernels<WT - 1>::occupancy(w, per_cu);
    // the general (non-SYM) body is the larger one: its occupancy is valid for both
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ltr_dp_kernel<WT, false, false, true>, 64 * kBlockWaves, 0);
  }
  static void launch(int w, bool sym, dim3 grid, hipStream_t st, const KernelArgs& A) {
    if (w != WT) { FastKernels<WT - 1>::launch(w, sym, grid, st, A); return; }
    if (sym) hipLaunchKernelGGL((ltr_dp_kernel<WT, false, true, true>), grid, dim3(64 * kBlockWaves), 0, st, A);
    else hipLaunchKernelGGL((ltr_dp_kernel<WT, false, false, true>), grid, dim3(64 * kBlockWaves), 0, st, A);
  }
};
template <>
struct FastKernels<0> {
  static hipError_t occupancy(int, int*) { return hipErrorInvalidValue; }
  static void launch(int, bool, dim3, hipStream_t, const KernelArgs&) {}
};
}  // namespace

namespace ltrk {
hipError_t occ_multi(int* per_cu) { return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ltr_dp_multi_kernel<false>, 64 * kBlockWaves, 0); }
void launch_multi(bool sym, dim3 grid, hipStream_t st, const KernelArgs& A) {
  if (sym) hipLaunchKernelGGL((ltr_dp_multi_kernel<true>), grid, dim3(64 * kBlockWaves), 0, st, A);
  else hipLaunchKernelGGL((ltr_dp_multi_kernel<false>), grid, dim3(64 * kBlockWaves), 0, st, A);
}
hipError_t occ_onewave(int W, int* per_cu) { return FastKernels<kWMax>::occupancy(W, per_cu); }
void launch_onewave(int W, bool sym, dim3 grid, hipStream_t st, const KernelArgs& A) { FastKernels<kWMax>::launch(W, sym, grid, st, A); }
}  // namespace ltrk
