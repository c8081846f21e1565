// VALU issue-rate probe: cycles per wave-instruction for the instruction kinds the DP kernel is made of.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 2; } } while (0)

template <int KIND>
__global__ void k_rate(double* out, unsigned long long* cyc, int iters, double c0, double c1) {
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {  // 8 independent v_add_f64
      asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                   "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c0));
    } else if (KIND == 1) {  // 8 independent v_max_f64
      asm volatile("v_max_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n"
                   "v_max_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_max_f64 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c0));
    } else if (KIND == 2) {  // 8 independent v_cndmask_b32 (vcc)
      asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                   "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(it) : "vcc");
    } else if (KIND == 3) {  // 8 independent v_add_u32
      asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                   "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(it));
    } else if (KIND == 4) {  // the cell mix: 4 add_f64, 2 max_f64, 2 cndmask  (per 8)
      asm volatile("v_add_f64 %0, %0, %10\n v_max_f64 %1, %1, %10\n v_add_f64 %2, %2, %10\n v_cndmask_b32 %8, %8, %11, vcc\n"
                   "v_add_f64 %4, %4, %10\n v_max_f64 %5, %5, %10\n v_add_f64 %6, %6, %10\n v_cndmask_b32 %9, %9, %11, vcc\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(i0), "+v"(i1) : "v"(c0), "v"(it) : "vcc");
    } else if (KIND == 5) {  // 8 v_mov_b32 dpp wave_shr
      asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
    } else if (KIND == 6) {  // 8 v_cmp_eq_u32 (writes vcc)
      asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_cmp_eq_u32 vcc, %1, %2\n v_cmp_eq_u32 vcc, %2, %3\n v_cmp_eq_u32 vcc, %3, %4\n"
                   "v_cmp_eq_u32 vcc, %4, %5\n v_cmp_eq_u32 vcc, %5, %6\n v_cmp_eq_u32 vcc, %6, %7\n v_cmp_eq_u32 vcc, %7, %0\n"
                   :: "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(i4), "v"(i5), "v"(i6), "v"(i7) : "vcc");
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7 + c1;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND> int run(const char* name, int waves_per_simd) {
  const int iters = 20000;
  const int blocks = 256 * 4 * waves_per_simd;   // 64-thread blocks: one wave each
  double* out; unsigned long long* cyc; CK(hipMalloc(&out, (size_t)blocks * 64 * 8)); CK(hipMalloc(&cyc, (size_t)blocks * 8));
  hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, -1.0, 0.5); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, -1.0, 0.5); hipEventRecord(e1);
  CK(hipDeviceSynchronize()); float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long* h = (unsigned long long*)malloc((size_t)blocks * 8); CK(hipMemcpy(h, cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost));
  double avg = 0; for (int i = 0; i < blocks; i++) avg += (double)h[i]; avg /= blocks;
  // s_memtime ticks at 100 MHz (constant clock) on gfx9: report wall-derived issue rate instead
  const double insts = (double)iters * 8 * blocks;             // wave-instructions
  const double per_simd_per_s = insts / 1024.0 / (ms * 1e-3);
  printf("%-14s waves/SIMD %d: %.3f ms  -> %.3f Ginst/s/SIMD = %.2f cycles/inst at 2.4 GHz (memtime avg %.0f ticks)\n", name, waves_per_simd, ms,
         per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, avg);
  hipFree(out); hipFree(cyc); free(h); return 0;
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_add_f64", w); run<1>("v_max_f64", w); run<2>("v_cndmask_b32", w); run<3>("v_add_u32", w);
    run<4>("cell mix", w); run<5>("v_mov_dpp", w); run<6>("v_cmp_eq_u32", w);
  }
  return 0;
}
