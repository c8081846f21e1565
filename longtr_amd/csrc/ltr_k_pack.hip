// ltr_k_pack.hip -- the several-pairs-per-wavefront certificate kernels (ltr_dp_pack.hpp), strip widths 1..kPackWMax.
#include <hip/hip_runtime.h>

#include "ltr_kernels.h"

namespace {
#include "ltr_dp_kernel.hpp"
#include "ltr_dp_pack.hpp"

template <int WT>
struct PackKernels {
  static hipError_t occupancy(int w, int* per_cu) {
    if (w != WT) return PackKernels<WT - 1>::occupancy(w, per_cu);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ltr_dp_pack_kernel<WT, false>, 64 * kBlockWaves, 0);
  }
  static void launch(int w, bool sym, dim3 grid, hipStream_t st, const KernelArgs& A) {
    if (w != WT) { PackKernels<WT - 1>::launch(w, sym, grid, st, A); return; }
    if (sym) hipLaunchKernelGGL((ltr_dp_pack_kernel<WT, true>), grid, dim3(64 * kBlockWaves), 0, st, A);
    else hipLaunchKernelGGL((ltr_dp_pack_kernel<WT, false>), grid, dim3(64 * kBlockWaves), 0, st, A);
  }
};
template <>
struct PackKernels<0> {
  static hipError_t occupancy(int, int*) { return hipErrorInvalidValue; }
  static void launch(int, bool, dim3, hipStream_t, const KernelArgs&) {}
};
}  // namespace

namespace ltrk {
hipError_t occ_pack_multi(int* per_cu) { return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, ltr_dp_pack_multi_kernel<false>, 64 * kBlockWaves, 0); }
void launch_pack_multi(bool sym, dim3 grid, hipStream_t st, const KernelArgs& A) {
  if (sym) hipLaunchKernelGGL((ltr_dp_pack_multi_kernel<true>), grid, dim3(64 * kBlockWaves), 0, st, A);
  else hipLaunchKernelGGL((ltr_dp_pack_multi_kernel<false>), grid, dim3(64 * kBlockWaves), 0, st, A);
}
hipError_t occ_pack(int W, int* per_cu) { return PackKernels<kPackWMax>::occupancy(W, per_cu); }
void launch_pack(int W, bool sym, dim3 grid, hipStream_t st, const KernelArgs& A) { PackKernels<kPackWMax>::launch(W, sym, grid, st, A); }
}  // namespace ltrk
