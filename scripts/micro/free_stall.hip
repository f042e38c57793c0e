// Does a hipFree of a medium buffer stall later kernel submissions?  (development aid)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void touch(float *p, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] += 1.f;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  size_t mb = argc > 1 ? atoi(argv[1]) : 20;
  int keep = argc > 2 ? atoi(argv[2]) : 0;
  hipStream_t s;
  hipStreamCreate(&s);
  float *kept = nullptr;
  for (int call = 0; call < 8; ++call) {
    double t0 = now();
    float *p = kept;
    size_t n = mb << 18;
    if (!p) hipMalloc(&p, n * 4);
    double t1 = now();
    double worst = 0;
    for (int it = 0; it < 400; ++it) {
      double a = now();
      touch<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(p, n);
      if (it % 4 == 3) hipStreamSynchronize(s);
      double b = now() - a;
      if (b > worst) worst = b;
    }
    hipStreamSynchronize(s);
    double t2 = now();
    if (keep) kept = p; else hipFree(p);
    double t3 = now();
    printf("call %d: malloc %.2f ms, loop %.2f ms (worst step %.2f), free %.2f\n", call, t1 - t0, t2 - t1, worst, t3 - t2);
  }
  return 0;
}
