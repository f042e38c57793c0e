// How fast can this part READ a 4 GB array?  The ceiling of the one-pass iteration kernel
// (csrc/stream.h) -- plain grid-stride reads with 16-byte loads, temporal and non-temporal, several
// workgroup shapes and unroll depths.   hipcc --offload-arch=gfx950 -O3 read_bw.hip -o bin/read_bw
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float v4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void __launch_bounds__(256) read_kernel(const v4 *__restrict__ a, size_t nvec, float *out) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  v4 acc = {0, 0, 0, 0};
  for (; i + (U - 1) * stride < nvec; i += U * stride) {
    v4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  for (; i < nvec; i += stride) acc += a[i];
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

// row-block form: a workgroup reads whole rows of `rowvec` vectors (contiguous 40 KB at C2), NV per thread
template <int NV, int R, bool NT>
__global__ void __launch_bounds__(256) rows_kernel(const v4 *__restrict__ a, int rows, int rowvec, float *out) {
  v4 acc = {0, 0, 0, 0};
  for (int r0 = blockIdx.x * R; r0 < rows; r0 += gridDim.x * R) {
    v4 v[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int c = k * 256 + threadIdx.x;
        const v4 *p = a + static_cast<size_t>(min(r0 + r, rows - 1)) * rowvec + min(c, rowvec - 1);
        v[r][k] = NT ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int k = 0; k < NV; ++k) acc += v[r][k];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

template <typename F>
double time_ms(F &&launch, int reps = 10) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const int rows = 100000, n = 10000, rowvec = n / 4;
  const size_t nvec = static_cast<size_t>(rows) * rowvec;
  v4 *a;
  float *out;
  hipMalloc(&a, nvec * sizeof(v4));
  hipMalloc(&out, 64);
  hipMemset(a, 0, nvec * sizeof(v4));
  const double gb = nvec * 16.0 / 1e9;
  for (int grid : {512, 1024, 2048, 4096, 8192}) {
    double t;
    t = time_ms([&] { hipLaunchKernelGGL((read_kernel<4, false>), dim3(grid), dim3(256), 0, 0, a, nvec, out); });
    printf("grid-stride U=4  temporal      grid %5d: %.3f ms  %.0f GB/s\n", grid, t, gb / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((read_kernel<4, true>), dim3(grid), dim3(256), 0, 0, a, nvec, out); });
    printf("grid-stride U=4  non-temporal  grid %5d: %.3f ms  %.0f GB/s\n", grid, t, gb / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((read_kernel<8, true>), dim3(grid), dim3(256), 0, 0, a, nvec, out); });
    printf("grid-stride U=8  non-temporal  grid %5d: %.3f ms  %.0f GB/s\n", grid, t, gb / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((read_kernel<16, true>), dim3(grid), dim3(256), 0, 0, a, nvec, out); });
    printf("grid-stride U=16 non-temporal  grid %5d: %.3f ms  %.0f GB/s\n", grid, t, gb / t * 1e3);
  }
  for (int grid : {512, 1024, 2048}) {
    double t;
    t = time_ms([&] { hipLaunchKernelGGL((rows_kernel<10, 1, true>), dim3(grid), dim3(256), 0, 0, a, rows, rowvec, out); });
    printf("row blocks NV=10 R=1 non-temporal grid %5d: %.3f ms  %.0f GB/s\n", grid, t, gb / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((rows_kernel<10, 2, true>), dim3(grid), dim3(256), 0, 0, a, rows, rowvec, out); });
    printf("row blocks NV=10 R=2 non-temporal grid %5d: %.3f ms  %.0f GB/s\n", grid, t, gb / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((rows_kernel<10, 4, true>), dim3(grid), dim3(256), 0, 0, a, rows, rowvec, out); });
    printf("row blocks NV=10 R=4 non-temporal grid %5d: %.3f ms  %.0f GB/s\n", grid, t, gb / t * 1e3);
  }
  return 0;
}
