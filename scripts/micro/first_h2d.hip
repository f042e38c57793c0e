// What does the first host-to-device copy of a process cost, by kind?  (cold-start cost of a solver handle)
// usage: first_h2d <mode>   mode 0: pageable first; 1: pinned first, then pageable; 2: kernel reading a host-mapped buffer first
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void copyk(const float *src, float *dst, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) dst[i] = src[i]; }
int main(int argc, char **argv) {
  int mode = argc > 1 ? atoi(argv[1]) : 0;
  const int n = 10000;
  double t0 = now();
  hipSetDevice(0);
  hipStream_t s; hipStreamCreate(&s);
  float *d; hipMalloc(&d, 4 << 20);
  hipLaunchKernelGGL(copyk, dim3(1), dim3(256), 0, s, d, d, 0); hipStreamSynchronize(s);
  printf("mode %d: init+stream+malloc+first kernel %.3f ms\n", mode, (now() - t0) * 1e3);
  std::vector<float> h(1 << 20, 1.f);
  float *p = nullptr;
  auto step = [&](const char *name, auto fn) { double t = now(); fn(); hipStreamSynchronize(s); printf("  %-38s %.3f ms\n", name, (now() - t) * 1e3); };
  if (mode == 1 || mode == 2) step("hipHostMalloc 4 MB (mapped)", [&] { hipHostMalloc(&p, 4 << 20, hipHostMallocMapped); });
  if (mode == 1) { memcpy(p, h.data(), n * 4); step("H2D 40 KB from pinned", [&] { hipMemcpyAsync(d, p, n * 4, hipMemcpyHostToDevice, s); }); 
                   step("H2D 40 KB from pinned again", [&] { hipMemcpyAsync(d, p, n * 4, hipMemcpyHostToDevice, s); }); }
  if (mode == 2) { memcpy(p, h.data(), n * 4); step("kernel copy 40 KB from mapped host", [&] { hipLaunchKernelGGL(copyk, dim3((n + 255) / 256), dim3(256), 0, s, p, d, n); });
                   step("kernel copy 2.4 MB from mapped host", [&] { hipLaunchKernelGGL(copyk, dim3((600000 + 255) / 256), dim3(256), 0, s, p, d, 600000); }); }
  step("H2D 40 KB pageable (first)", [&] { hipMemcpyAsync(d, h.data(), n * 4, hipMemcpyHostToDevice, s); });
  step("H2D 40 KB pageable (second)", [&] { hipMemcpyAsync(d, h.data(), n * 4, hipMemcpyHostToDevice, s); });
  step("H2D 2.4 MB pageable", [&] { hipMemcpyAsync(d, h.data(), 2400000, hipMemcpyHostToDevice, s); });
  step("D2H 40 KB pageable (first)", [&] { hipMemcpyAsync(h.data(), d, n * 4, hipMemcpyDeviceToHost, s); });
  step("D2H 40 KB pageable (second)", [&] { hipMemcpyAsync(h.data(), d, n * 4, hipMemcpyDeviceToHost, s); });
  return 0;
}
