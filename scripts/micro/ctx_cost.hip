// Cost of the calls in Ctx::init (development aid).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(float *p) { p[0] = 1.f; }
int main() {
  float *w; (void)hipMalloc(&w, 4096); k<<<1, 1>>>(w); (void)hipDeviceSynchronize();   // runtime up
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now();
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    double t1 = now();
    int cu = 0; (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, 0);
    double t2 = now();
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double t3 = now();
    k<<<1, 1, 0, s>>>(w); (void)hipStreamSynchronize(s);
    double t4 = now();
    double *h; (void)hipHostMalloc(reinterpret_cast<void **>(&h), 4096, hipHostMallocMapped | hipHostMallocCoherent);
    double t5 = now();
    double *d; (void)hipMalloc(&d, 4096);
    double t6 = now();
    printf("rep %d: getDeviceProperties %.3f ms, getAttribute %.3f, streamCreate %.3f, first launch+sync on it %.3f, hostMalloc mapped %.3f, malloc %.3f\n",
           rep, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5);
  }
  return 0;
}
