// Wavefront (64-lane) and workgroup reductions, fixed order => bit-reproducible.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace pogs_amd {
namespace dev {

// Butterfly sum over the 64 lanes of a wavefront; every lane gets the total.  The tree is the one of
//   for (off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
// (partners at distance 32 first, then 16, 8, 4, 2, 1).  __shfl_xor is a ds_bpermute_b32, an LDS round trip per
// level; gfx950 can do every level in the vector ALU:
//   32, 16  v_permlane32_swap / v_permlane16_swap of two copies of v: one result holds the value of the lower
//           half (row pair), the other the upper one's, in every lane; their sum is v + partner (a + b = b + a
//           bit for bit, so which of the two is "mine" does not matter);
//   8, 4    DPP row_ror:8 is lane ^ 8 inside a row of 16; after it the values repeat every 8 lanes, so row_ror:4
//           delivers the value lane ^ 4 holds;
//   2, 1    DPP quad_perm [2,3,0,1] and [1,0,3,2].
// Same tree, same bits (tests/test_gpu_dense.py: test_wavefront_sum_..., PogsAmdWaveSumCheck).
//
// Used for DOUBLES (two halves each): the scalar sums of every kernel, the device-resident CG loop's records, the fp64
// row dots -- C2 in fp64 +1.3 %, C4 +1.1 %, every fixture bit for bit as before (round 5, profiles/NOTES_r05.md).
// fp32 sums stay on the LDS crossbar.  The fp32 reduction in the ALU is just as exact, but it was measured (two
// boxes, alternating runs) at +0.5 % on C2 and -1.5 % on C3, and it changed the code AROUND it: -ffp-contract=fast
// fuses the products of the row dots (a.x * b.x + a.y * b.y + ...) as instruction selection sees fit, and with the
// sums in the ALU the 64-thread kernels came out with unfused v_pk_mul_f32 where there had been v_pk_fma_f32 -- a
// 33 x 40001 fp32 problem moved by 1e-4 against the oracle.  scripts/fp_op_diff.py compares the multiplies / FMAs /
// adds of every kernel between two builds; with doubles only, no kernel's products change.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xf, 0xf, true));
}
struct WavePair32 { unsigned a, b; };
template <int WIDTH>   // 32 or 16: {value of the partner group's lower member, upper member}
__device__ __forceinline__ WavePair32 swap_u32(unsigned v) {
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  const u2 r = WIDTH == 32 ? __builtin_amdgcn_permlane32_swap(v, v, false, false)
                           : __builtin_amdgcn_permlane16_swap(v, v, false, false);
  return WavePair32{r.x, r.y};
}
__device__ __forceinline__ float wave_sum(float v) {   // (fp32 sums stay on the LDS crossbar: see above)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// (the fp32 tree in the vector ALU, bit for bit wave_sum(float)'s: for the one kernel whose step time is a chain of
//  LDS round trips -- stream_rows2_pf_kernel, stream.h)
__device__ __forceinline__ float wave_sum_valu(float v) {
  auto bits = [](float f) { return __builtin_bit_cast(unsigned, f); };
  auto flt = [](unsigned u) { return __builtin_bit_cast(float, u); };
  WavePair32 p = swap_u32<32>(bits(v));
  v = flt(p.a) + flt(p.b);
  p = swap_u32<16>(bits(v));
  v = flt(p.a) + flt(p.b);
  v += flt(dpp_u32<0x128>(bits(v)));
  v += flt(dpp_u32<0x124>(bits(v)));
  v += flt(dpp_u32<0x4E>(bits(v)));
  v += flt(dpp_u32<0xB1>(bits(v)));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
  auto halves = [](double d, unsigned &lo, unsigned &hi) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, d);
    lo = static_cast<unsigned>(u);
    hi = static_cast<unsigned>(u >> 32);
  };
  auto whole = [](unsigned lo, unsigned hi) {
    return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
  };
  unsigned lo, hi;
  halves(v, lo, hi);
  WavePair32 pl = swap_u32<32>(lo), ph = swap_u32<32>(hi);
  v = whole(pl.a, ph.a) + whole(pl.b, ph.b);
  halves(v, lo, hi);
  pl = swap_u32<16>(lo);
  ph = swap_u32<16>(hi);
  v = whole(pl.a, ph.a) + whole(pl.b, ph.b);
  halves(v, lo, hi); v += whole(dpp_u32<0x128>(lo), dpp_u32<0x128>(hi));
  halves(v, lo, hi); v += whole(dpp_u32<0x124>(lo), dpp_u32<0x124>(hi));
  halves(v, lo, hi); v += whole(dpp_u32<0x4E>(lo), dpp_u32<0x4E>(hi));
  halves(v, lo, hi); v += whole(dpp_u32<0xB1>(lo), dpp_u32<0xB1>(hi));
  return v;
}
// (the same tree through the LDS crossbar: what the two above are tested against, tests/test_gpu_kernels)
__device__ __forceinline__ double wave_sum_valu(double v) { return wave_sum(v); }
template <typename T>
__device__ __forceinline__ T wave_sum_shfl(T v) {
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
