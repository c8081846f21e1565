// Hardware probes used while bringing the kernel up (manual; not part of pytest).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 2; } } while (0)

__global__ void k_dpp(int* out) {
  int lane = threadIdx.x;
  int v = lane * 10;
  int r = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);
  out[lane] = r;
}
__global__ void k_queue(unsigned* q, int n, int* out) {
  int lane = threadIdx.x;
  int cnt = 0;
  for (;;) {
    int x = 0;
    if (lane == 0) x = (int)atomicAdd(q, 1u);
    x = __builtin_amdgcn_readfirstlane(x);
    if (x >= n) break;
    cnt++;
  }
  if (lane == 0) atomicAdd(out, cnt);
}
__global__ void k_atomic_load(const double* p, double* out) {
  out[threadIdx.x] = __hip_atomic_load(p + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_f64(double* x) {
  int i = threadIdx.x;
  double a = x[i];
  for (int k = 0; k < 100; ++k) a = fmax(a + -1.0, a + -0.458675);
  x[i] = a;
}
int main(int argc, char** argv) {
  const char* which = argc > 1 ? argv[1] : "dpp";
  if (!strcmp(which, "dpp")) {
    int* d; int h[64]; CK(hipMalloc(&d, 256));
    hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, d); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d, 256, hipMemcpyDeviceToHost));
    int ok = h[0] == -1; for (int i = 1; i < 64; i++) ok &= (h[i] == (i - 1) * 10);
    printf("dpp wave_shr:1 %s  h[0]=%d h[1]=%d h[16]=%d h[32]=%d h[63]=%d\n", ok ? "OK" : "BAD", h[0], h[1], h[16], h[32], h[63]);
  } else if (!strcmp(which, "queue")) {
    unsigned* q; int* o; int ho = 0; CK(hipMalloc(&q, 4)); CK(hipMalloc(&o, 4)); CK(hipMemset(q, 0, 4)); CK(hipMemset(o, 0, 4));
    hipLaunchKernelGGL(k_queue, dim3(8), dim3(64), 0, 0, q, 1000, o); CK(hipDeviceSynchronize());
    CK(hipMemcpy(&ho, o, 4, hipMemcpyDeviceToHost)); printf("queue processed %d (want 1000)\n", ho);
  } else if (!strcmp(which, "aload")) {
    double* p; double* o; double h[64]; CK(hipMalloc(&p, 512)); CK(hipMalloc(&o, 512));
    for (int i = 0; i < 64; i++) h[i] = i * 1.5; CK(hipMemcpy(p, h, 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_atomic_load, dim3(1), dim3(64), 0, 0, p, o); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, o, 512, hipMemcpyDeviceToHost)); printf("atomic load h[5]=%g (want 7.5)\n", h[5]);
  } else if (!strcmp(which, "f64")) {
    double* p; double h[64]; for (int i = 0; i < 64; i++) h[i] = -i; CK(hipMalloc(&p, 512)); CK(hipMemcpy(p, h, 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_f64, dim3(1), dim3(64), 0, 0, p); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, p, 512, hipMemcpyDeviceToHost)); printf("f64 h[1]=%.6f\n", h[1]);
  }
  return 0;
}
