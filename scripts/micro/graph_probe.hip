// Development aid (round 6): would a captured hipGraph shorten an ADMM iteration?  The iteration's shape -- the host polls a word the
// last launch publishes, then enqueues five dependent launches (10, 14, 7, 600, 6 us) -- with kernels that only spin:
//   (a) five hipLaunchKernelGGL calls;  (b) one hipGraphLaunch of the same five nodes;  (c) as (b), every node's parameters rewritten
//   first (the real iteration's arguments change every time).
// Prints the time per iteration of each and the GPU's idle share.   hipcc --offload-arch=gfx950 -O3 graph_probe.hip -o graph_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(unsigned long long ticks, int *sink) {   // wall_clock64: 100 MHz
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && threadIdx.x == 12345) *sink = 1;
}
__global__ void publish_kernel(unsigned long long ticks, volatile unsigned long long *host_word, unsigned long long seq) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0) {
    __atomic_store_n(const_cast<unsigned long long *>(host_word), seq, __ATOMIC_RELEASE);
  }
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const double us[5] = {10, 14, 7, 600, 6};
  hipStream_t s;
  CK(hipStreamCreate(&s));
  unsigned long long *word, *word_dev;
  CK(hipHostMalloc(&word, sizeof(*word), hipHostMallocMapped));
  CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&word_dev), word, 0));
  *word = 0;
  unsigned long long seq = 0;
  auto wait = [&](unsigned long long want) { while (__atomic_load_n(word, __ATOMIC_ACQUIRE) != want) __builtin_ia32_pause(); };
  auto ticks = [&](int k) { return static_cast<unsigned long long>(us[k] * 100.0); };
  double busy = 0;
  for (double u : us) busy += u;

  auto run = [&](const char *name, auto &&enqueue) {
    for (int w = 0; w < 50; ++w) { enqueue(++seq); wait(seq); }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) { enqueue(++seq); wait(seq); }
    const double per = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
    printf("%-64s %8.2f us per iteration  (kernels %.0f us: %.2f us not computing, %.2f %%)\n", name, per, busy, per - busy, 100.0 * (per - busy) / per);
  };

  run("(a) five launches", [&](unsigned long long q) {
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, ticks(k), static_cast<int *>(nullptr));
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, s, ticks(4), word_dev, q);
  });

  // the same five nodes as a graph (explicit nodes: every iteration's publish carries a new sequence number, so that node's
  // parameters are rewritten in (b) too -- the minimum any real use needs)
  hipGraph_t g;
  CK(hipGraphCreate(&g, 0));
  std::vector<hipGraphNode_t> nodes(5);
  unsigned long long tk[5];
  int *nullp = nullptr;
  unsigned long long q0 = 0;
  void *args[5][3];
  hipKernelNodeParams np[5];
  for (int k = 0; k < 5; ++k) {
    tk[k] = ticks(k);
    np[k] = hipKernelNodeParams{};
    np[k].blockDim = k < 4 ? dim3(256) : dim3(64);
    np[k].gridDim = k < 4 ? dim3(256) : dim3(1);
    np[k].sharedMemBytes = 0;
    if (k < 4) {
      np[k].func = reinterpret_cast<void *>(spin_kernel);
      args[k][0] = &tk[k]; args[k][1] = &nullp;
    } else {
      np[k].func = reinterpret_cast<void *>(publish_kernel);
      args[k][0] = &tk[k]; args[k][1] = &word_dev; args[k][2] = &q0;
    }
    np[k].kernelParams = args[k];
    np[k].extra = nullptr;
    CK(hipGraphAddKernelNode(&nodes[k], g, k ? &nodes[k - 1] : nullptr, k ? 1 : 0, &np[k]));
  }
  hipGraphExec_t ge;
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  run("(b) one graph launch (the publish node's sequence number rewritten)", [&](unsigned long long q) {
    q0 = q;
    CK(hipGraphExecKernelNodeSetParams(ge, nodes[4], &np[4]));
    CK(hipGraphLaunch(ge, s));
  });
  run("(c) one graph launch, all five nodes' parameters rewritten", [&](unsigned long long q) {
    q0 = q;
    for (int k = 0; k < 5; ++k) CK(hipGraphExecKernelNodeSetParams(ge, nodes[k], &np[k]));
    CK(hipGraphLaunch(ge, s));
  });
  // (d) the look-ahead shape: the three short launches of the NEXT iteration are already in the queue when the host polls; after the
  //     poll it enqueues only the long one and the publish (and then the front after next) -- the upper bound of what speculating on
  //     the decision could give
  for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, ticks(k), static_cast<int *>(nullptr));
  run("(d) the three short launches enqueued before the poll", [&](unsigned long long q) {
    hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, ticks(3), static_cast<int *>(nullptr));
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, s, ticks(4), word_dev, q);
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, ticks(k), static_cast<int *>(nullptr));
  });
  CK(hipStreamSynchronize(s));
  // (e) as (d), but only the FIRST short launch ahead
  hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, ticks(0), static_cast<int *>(nullptr));
  run("(e) only the first short launch enqueued before the poll", [&](unsigned long long q) {
    for (int k = 1; k < 4; ++k) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, ticks(k), static_cast<int *>(nullptr));
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, s, ticks(4), word_dev, q);
    hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, ticks(0), static_cast<int *>(nullptr));
  });
  CK(hipStreamSynchronize(s));
  run("(a) five launches, again", [&](unsigned long long q) {
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, ticks(k), static_cast<int *>(nullptr));
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, s, ticks(4), word_dev, q);
  });
  return 0;
}
