// Development micro-benchmark: cost of the first kernel launch on a fresh stream.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(float *p) { p[threadIdx.x] = 1; }
int main() {
  float *w; hipMalloc(&w, 1 << 20);
  double t0 = now(); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, w); hipDeviceSynchronize();
  printf("first launch (null stream) %.3f ms\n", (now() - t0) * 1e3);
  for (int i = 0; i < 6; ++i) {
    hipStream_t s;
    t0 = now(); hipStreamCreateWithFlags(&s, hipStreamNonBlocking); double t1 = now();
    hipMemsetAsync(w, 0, 1024, s); hipStreamSynchronize(s); double t2 = now();
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, w); double t3 = now(); hipStreamSynchronize(s); double t4 = now();
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, w); hipStreamSynchronize(s); double t5 = now();
    hipStreamDestroy(s); double t6 = now();
    printf("stream %d: create %.3f  memset+sync %.3f  first launch call %.3f  sync %.3f  second launch+sync %.3f  destroy %.3f ms\n",
           i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3);
  }
  return 0;
}
