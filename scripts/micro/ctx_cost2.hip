// Is there a cheaper stream than hipStreamCreate?  (development aid)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(float *p) { p[0] = 1.f; }
int main() {
  float *w; (void)hipMalloc(&w, 4096); k<<<1, 1>>>(w); (void)hipDeviceSynchronize();   // runtime up, null stream used
  double t0 = now();
  k<<<1, 1, 0, hipStreamPerThread>>>(w); (void)hipStreamSynchronize(hipStreamPerThread);
  double t1 = now();
  k<<<1, 1, 0, hipStreamPerThread>>>(w); (void)hipStreamSynchronize(hipStreamPerThread);
  double t2 = now();
  hipStream_t s1; (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  double t3 = now();
  hipStream_t s2; (void)hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, 0);
  double t4 = now();
  (void)hipStreamDestroy(s1);
  double t5 = now();
  hipStream_t s3; (void)hipStreamCreateWithFlags(&s3, hipStreamNonBlocking);
  double t6 = now();
  printf("first use of hipStreamPerThread %.3f ms, second %.3f; create %.3f, create with priority %.3f, destroy %.3f, create after destroy %.3f\n",
         t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5);
  // creation on a helper thread while the main thread works on the null stream
  double t7 = now();
  hipStream_t s4 = nullptr;
  std::thread th([&] { (void)hipStreamCreateWithFlags(&s4, hipStreamNonBlocking); });
  for (int i = 0; i < 200; ++i) k<<<1, 1>>>(w);
  (void)hipDeviceSynchronize();
  double t8 = now();
  th.join();
  double t9 = now();
  printf("200 null-stream launches + sync while a helper creates a stream: %.3f ms, join after %.3f ms\n", t8 - t7, t9 - t8);
  return 0;
}
