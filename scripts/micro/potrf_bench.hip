// Development micro-benchmark: potrf_inv_kernel on one 128x128 block, timing + residual check.
#include "../../pogs_amd/csrc/gemm.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace pogs_amd;
int main() {
  const int NB = 128, n = 128;
  std::vector<float> A(n * n), G(n * n);
  unsigned s = 1;
  for (auto &v : A) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
    double acc = (i == j) ? 1.0 : 0.0;
    for (int k = 0; k < n; ++k) acc += (double)A[i * n + k] * A[j * n + k];
    G[i * n + j] = (float)acc;
  }
  float *dG, *dW, *dG0;
  hipMalloc(&dG, n * n * 4); hipMalloc(&dW, n * n * 4); hipMalloc(&dG0, n * n * 4);
  hipMemcpy(dG0, G.data(), n * n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(dG, dG0, n * n * 4, hipMemcpyDeviceToDevice);
    hipMemset(dW, 0, n * n * 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((potrf_inv_kernel<float, NB>), dim3(1), dim3(256), 0, 0, dG, (size_t)n, n, dW, (size_t)n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("potrf_inv 128: %.1f us\n", ms * 1e3);
  }
  std::vector<float> L(n * n), W(n * n);
  hipMemcpy(L.data(), dG, n * n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(W.data(), dW, n * n * 4, hipMemcpyDeviceToHost);
  double e_llt = 0, e_inv = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
    double acc = 0; for (int k = 0; k <= j; ++k) acc += (double)L[i * n + k] * L[j * n + k];
    e_llt = fmax(e_llt, fabs(acc - G[i * n + j]));
    double a2 = 0; for (int k = j; k <= i; ++k) a2 += (double)L[i * n + k] * W[k * n + j];
    e_inv = fmax(e_inv, fabs(a2 - (i == j ? 1.0 : 0.0)));
  }
  printf("max |LL^T - G| = %.3e   max |L W - I| = %.3e\n", e_llt, e_inv);
  return 0;
}
