// Host->device upload paths for a plan's inputs (~100 MB of read bytes per chunk): pageable hipMemcpy,
// pinned hipMemcpyAsync, and pageable -> pinned staging on N host threads followed by the async copy.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d h2d.hip -lpthread && /tmp/h2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  const size_t sizes[] = {(size_t)4 << 20, (size_t)32 << 20, (size_t)128 << 20};
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (size_t n : sizes) {
    std::vector<uint8_t> pageable(n, 1);
    void* d = nullptr; hipMalloc(&d, n);
    void* pin = nullptr;
    double t0 = now_ms(); hipHostMalloc(&pin, n, hipHostMallocDefault); const double t_pin_alloc = now_ms() - t0;
    std::memset(pin, 2, n);
    hipMemcpy(d, pageable.data(), n, hipMemcpyHostToDevice);                       // warm
    t0 = now_ms(); for (int i = 0; i < 5; ++i) hipMemcpy(d, pageable.data(), n, hipMemcpyHostToDevice);
    const double t_page = (now_ms() - t0) / 5;
    t0 = now_ms(); for (int i = 0; i < 5; ++i) { hipMemcpyAsync(d, pin, n, hipMemcpyHostToDevice, st); } const double t_issue = (now_ms() - t0) / 5;
    hipStreamSynchronize(st);
    t0 = now_ms(); for (int i = 0; i < 5; ++i) { hipMemcpyAsync(d, pin, n, hipMemcpyHostToDevice, st); } hipStreamSynchronize(st);
    const double t_pinned = (now_ms() - t0) / 5;
    for (int nt : {1, 4, 8, 16}) {
      t0 = now_ms();
      for (int i = 0; i < 5; ++i) {
        std::vector<std::thread> th;
        for (int k = 0; k < nt; ++k) th.emplace_back([&, k]() { const size_t a = n * k / nt, b = n * (k + 1) / nt; std::memcpy((uint8_t*)pin + a, pageable.data() + a, b - a); });
        for (auto& t : th) t.join();
      }
      std::printf("  %zu MB: staging memcpy on %2d threads %.2f ms (%.1f GB/s)\n", n >> 20, nt, (now_ms() - t0) / 5, n / ((now_ms() - t0) / 5) / 1e6);
    }
    t0 = now_ms(); hipHostRegister(pageable.data(), n, hipHostRegisterDefault); const double t_reg = now_ms() - t0;
    t0 = now_ms(); hipHostUnregister(pageable.data()); const double t_unreg = now_ms() - t0;
    std::printf("%zu MB: pageable hipMemcpy %.2f ms (%.1f GB/s); pinned async %.2f ms (%.1f GB/s), issue %.3f ms; hipHostMalloc %.2f ms; register %.2f / unregister %.2f ms\n",
                n >> 20, t_page, n / t_page / 1e6, t_pinned, n / t_pinned / 1e6, t_issue, t_pin_alloc, t_reg, t_unreg);
    hipFree(d); hipHostFree(pin);
  }
  return 0;
}
