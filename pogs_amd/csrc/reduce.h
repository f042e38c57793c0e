// Wavefront (64-lane) and workgroup reductions, fixed order => bit-reproducible.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace pogs_amd {
namespace dev {

// Butterfly sum over the 64 lanes of a wavefront; every lane gets the total.
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum NS doubles held by every thread of the workgroup; result valid in thread 0.
// `smem` must hold NS * (TPB/64) doubles.  Ends with a barrier-free state: the
// caller must not reuse smem before a __syncthreads().
template <int NS, int TPB>
__device__ __forceinline__ void block_sum(double (&v)[NS], double *smem) {
  constexpr int NW = TPB / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    double s = wave_sum(v[k]);
    if (NW > 1) {
      if (lane == 0) smem[k * NW + wave] = s;
    } else {
      v[k] = s;
    }
  }
  if (NW > 1) {
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        double s = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += smem[k * NW + w];
        v[k] = s;
      }
    }
  }
}

}  // namespace dev
}  // namespace pogs_amd
