// Development micro-benchmark: cost of hipMalloc / hipFree / hipMemset by size on this box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(float *p) { p[threadIdx.x] = 1; }
int main() {
  hipFree(0);
  float *w; hipMalloc(&w, 1 << 20);
  double t0 = now(); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, w); hipDeviceSynchronize();
  printf("first launch %.3f ms\n", (now() - t0) * 1e3);
  for (size_t mb : {1ul, 4ul, 40ul, 80ul, 400ul, 1600ul, 4000ul, 6400ul}) {
    void *p;
    t0 = now(); hipMalloc(&p, mb << 20); double t1 = now();
    hipMemset(p, 0, mb << 20); hipDeviceSynchronize(); double t2 = now();
    hipFree(p); double t3 = now();
    printf("%6zu MB: malloc %.3f ms  memset %.3f ms  free %.3f ms\n", mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
  }
  // many small allocations
  std::vector<void *> v(40);
  t0 = now(); for (auto &p : v) hipMalloc(&p, 400000); double t1 = now();
  for (auto &p : v) hipFree(p);
  printf("40 x 400 KB malloc %.3f ms, free %.3f ms\n", (t1 - t0) * 1e3, (now() - t1) * 1e3);
  // async pool
  hipStream_t s; hipStreamCreate(&s);
  for (int rep = 0; rep < 2; ++rep) {
    void *p;
    t0 = now(); hipMallocAsync(&p, 1600ul << 20, s); t1 = now(); hipFreeAsync(p, s); double t2 = now(); hipStreamSynchronize(s);
    printf("async 1600 MB: malloc %.3f ms free %.3f ms sync %.3f\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (now() - t2) * 1e3);
  }
  return 0;
}
