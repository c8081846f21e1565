// ltr_k_wg.hip -- the workgroup-per-pair certificate kernels (ltr_dp_wg.hpp): 4 waves W 5..14, 8 waves W 8..20,
// 1 wave W 1..16 (the latency variant).  Symmetric indel models only.
#include <hip/hip_runtime.h>

#include "ltr_kernels.h"

namespace {
#include "ltr_dp_kernel.hpp"
#include "ltr_dp_wg.hpp"

template <int NW, int WT, int WMIN, bool END = (WT < WMIN)>
struct WgKernels {
  static hipError_t occupancy(int w, int* per_cu) {
    if (w != WT) return WgKernels<NW, WT - 1, WMIN>::occupancy(w, per_cu);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ltr_dp_wg_kernel<WT, NW, true>, 64 * NW, 0);
  }
  static void launch(int w, dim3 grid, hipStream_t st, const KernelArgs& A) {
    if (w != WT) { WgKernels<NW, WT - 1, WMIN>::launch(w, grid, st, A); return; }
    hipLaunchKernelGGL((ltr_dp_wg_kernel<WT, NW, true>), grid, dim3(64 * NW), 0, st, A);
  }
};
template <int NW, int WT, int WMIN>
struct WgKernels<NW, WT, WMIN, true> {
  static hipError_t occupancy(int, int*) { return hipErrorInvalidValue; }
  static void launch(int, dim3, hipStream_t, const KernelArgs&) {}
};
}  // namespace

namespace ltrk {
hipError_t occ_wg(int NW, int W, int* per_cu) {
  if (NW == 4) return WgKernels<4, kWg4MaxW, kWg4MinW>::occupancy(W, per_cu);
  if (NW == 8) return WgKernels<8, kWgWMax, kWg8MinW>::occupancy(W, per_cu);
  return WgKernels<1, kWg1MaxW, 1>::occupancy(W, per_cu);
}
void launch_wg(int NW, int W, dim3 grid, hipStream_t st, const KernelArgs& A) {
  if (NW == 4) WgKernels<4, kWg4MaxW, kWg4MinW>::launch(W, grid, st, A);
  else if (NW == 8) WgKernels<8, kWgWMax, kWg8MinW>::launch(W, grid, st, A);
  else WgKernels<1, kWg1MaxW, 1>::launch(W, grid, st, A);
}
}  // namespace ltrk
