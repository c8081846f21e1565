// ltr_kernels.h -- host-callable entry points of the kernel translation units (ltr_k_*.hip).
// Every family of DP kernels is compiled in a TU of its own (the template instances of one family take
// tens of seconds of hipcc each; side by side they build in the time of the slowest) and is reached from
// ltr_gpu.hip through these launchers.  `occ_*` = resident workgroups per CU
// (hipOccupancyMaxActiveBlocksPerMultiprocessor) of the family's kernel with strip width W.
#ifndef LTR_KERNELS_H_
#define LTR_KERNELS_H_

#include <hip/hip_runtime.h>

#include "ltr_dp_types.h"

namespace ltrk {

// one pair per wavefront, certificate kernels (ltr_dp_kernel.hpp), W = 1..kWMax
hipError_t occ_onewave(int W, int* per_cu);
void launch_onewave(int W, bool sym, dim3 grid, hipStream_t st, const KernelArgs& A);

// ... strip widths kMultiMinW .. kWMax as one persistent launch over several classes (KernelArgs::mk_*)
hipError_t occ_multi(int* per_cu);
void launch_multi(bool sym, dim3 grid, hipStream_t st, const KernelArgs& A);

// 64 / LP pairs per wavefront (ltr_dp_pack.hpp), W = 1..kPackWMax; LP = 1 << A.lp_shift
hipError_t occ_pack(int W, int* per_cu);
void launch_pack(int W, bool sym, dim3 grid, hipStream_t st, const KernelArgs& A);

// ... strip widths kPackMultiMinW .. kPackWMax as one persistent launch (KernelArgs::pk_tabs)
hipError_t occ_pack_multi(int* per_cu);
void launch_pack_multi(bool sym, dim3 grid, hipStream_t st, const KernelArgs& A);

// the plan kernel (ltr_dp_plan.hpp): every one-wave class and packed strip width of a plan in one persistent launch, failed
// certificates scored in line (KernelArgs::pl_*, pk_tabs); `sym`: the symmetric-model instance (11-operation cell) or the general one
hipError_t occ_plan(bool sym, int* per_cu);
void launch_plan(bool sym, dim3 grid, hipStream_t st, const KernelArgs& A);

// one pair per workgroup of NW = 1 / 4 / 8 wavefronts (ltr_dp_wg.hpp; symmetric models only)
hipError_t occ_wg(int NW, int W, int* per_cu);
void launch_wg(int NW, int W, dim3 grid, hipStream_t st, const KernelArgs& A);

// ... with the exact threshold test in the same pass (FULL; NW = 4 / 8): class (NW, W) is scored by the kernel of strip width
// wgt_width(W) = W rounded up to even
int wgt_width(int W);
hipError_t occ_wgt(int NW, int W, int* per_cu);
void launch_wgt(int NW, int W, dim3 grid, hipStream_t st, const KernelArgs& A);

// exact (redo) kernels: which = kXGeneric .. kXWg8, or kXWideLaunch (the W = 20 one-wave kernel that shares the
// four-wave list)
constexpr int kXWideLaunch = kNumExact;
// (the eight-wave list's reads of up to kXWg8NarrowMaxC columns are scored by launch_wgt(8, 10, ..) with the list as its index: four
// waves per SIMD, two workgroups a CU; the list's own launch takes the longer ones on strips of 12 / 16 / 20 columns)
constexpr int kXWg8NarrowMaxC = 8 * 64 * 10;
hipError_t occ_exact(int which, int* per_cu);
void launch_exact(int which, bool sym, dim3 grid, hipStream_t st, const KernelArgs& A);
int exact_block_threads(int which);

}  // namespace ltrk

#endif
