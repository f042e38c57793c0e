// Kernels that were built, measured and NOT shipped in round 6 (profiles/NOTES_r06.md section 4): kept as the subjects of
// scripts/micro/c3_bisect.hip, outside the library.  Same contracts as stream_rows2_kernel (csrc/stream.h).
#pragma once
#include "stream.h"

namespace pogs_amd {

constexpr int kFwThreads = 64;   // the functor wavefront

namespace dev {
__device__ __forceinline__ float wave_sum_alu(float v) {
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
__device__ __forceinline__ double wave_sum_alu(double v) { return wave_sum(v); }
}  // namespace dev

// ---------------------------------------------------------------------------
// stream_rows2_db_kernel: the one-pass iteration kernel with ONE row per step and the NEXT row in flight.
// At 256 x 5 (rows of <= 1280 vectors: C3) stream_rows2_kernel takes two rows per step and holds them through
// dot -> barrier -> row functor (one lane per row) -> barrier -> column sums before it asks for the next two:
// every workgroup's loads stop for the length of that chain, and at this row length nothing else on the CU
// covers it (the pass ran at 0.77 of the data-sheet peak where Sinkhorn-Knopp's pass over the same matrix,
// the same skeleton with a three-instruction functor, reaches 0.86).  Here the register tile is the same size
// -- two rows -- but they are consecutive STEPS: row k + 1 (and its functor operands) is requested before row
// k is reduced, so the chain of row k runs under the load of row k + 1.  Same arithmetic per row, same
// round-robin dealing of rows to workgroups (row = blockIdx.x + k gridDim.x): the column partials of a workgroup
// add the same rows in the same order as a two-row step's did only when the grid is the same, so results are
// compared through the usual tolerances, not bit for bit (the second stage's partial count changes with R).
// ---------------------------------------------------------------------------
// R rows per step (the tile in flight is R rows too); BPC: workgroups per CU the register budget is held to
template <typename Op, bool DEFER> struct DeferPost { struct type {}; };
template <typename Op> struct DeferPost<Op, true> { using type = typename Op::Post; };
template <typename T, int TPB, int NV, int R, int ND, int NA, int BPC, typename Op, bool X1REG = false, bool ALUSUM = false, bool DEFER = false, int LATE = 0>
__global__ void __launch_bounds__(TPB, (BPC * (TPB / 64) + 3) / 4) stream_rows2_db_kernel(StreamArgs2<T> a, Op op) {
  using V = typename Vec16<T>::type;
  using Pre = typename Op::Pre;
  constexpr int VEC = Vec16<T>::N;
  constexpr int NW = TPB / 64;
  constexpr int NS = Op::NS > 0 ? Op::NS : 1;
  static_assert(ND > 0, "the column-sum-only form has no chain to hide");
  __shared__ T s_part[2 * R * ND * NW];
  __shared__ T s_u[2 * R * NA];
  __shared__ double s_red[NS * NW];
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  T *s_x1 = reinterpret_cast<T *>(s_dyn);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

  V xv[NV], x1v[(ND > 1 && X1REG) ? NV : 1];
  V acc[NA][NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * TPB + t) * VEC;
    xv[v] = (col < a.n_pad) ? *reinterpret_cast<const V *>(a.xin0 + col) : dev::vzero<V>();
    if constexpr (ND > 1 && X1REG) {
      x1v[v] = (col < a.n_pad) ? *reinterpret_cast<const V *>(a.xin1 + col) : dev::vzero<V>();
    } else if (ND > 1 && col < a.n_pad) {
      *reinterpret_cast<V *>(s_x1 + col) = *reinterpret_cast<const V *>(a.xin1 + col);
    }
  }
  if (ND > 1 && !X1REG) __syncthreads();
#pragma unroll
  for (int q = 0; q < NA; ++q)
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[q][v] = dev::vzero<V>();
  double sacc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) sacc[k] = 0.0;

  const int nblk = (a.m + R - 1) / R;
  int blk = blockIdx.x;
  V cur[R][NV];
  Pre pre_cur;
  if (t < R && blk < nblk && blk * R + t < a.m) pre_cur = op.prefetch(blk * R + t);
  // Every tile load is UNCONDITIONAL (row and column clamped into the matrix; the dot vectors are zero in the
  // lanes past n_pad, the functor returns u = 0 for rows past m and the column partials of those lanes are
  // never stored).  A guarded load is a branch, and behind a branch the compiler cannot count how many loads
  // are younger than the tile it is about to use: it waits for vmcnt(0), i.e. for the NEXT tile as well, and
  // the prefetch is gone (the first form of this kernel did exactly that).
  int colc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * TPB + t) * VEC;
    colc[v] = col < a.n_pad ? col : a.n_pad - VEC;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = blk * R + r;
    const T *rp = a.A + static_cast<size_t>(row < a.m ? row : a.m - 1) * a.lda;
#pragma unroll
    for (int v = 0; v < NV; ++v) cur[r][v] = stream_load<V>(rp + colc[v]);
  }
  int slot = 0;
  typename DeferPost<Op, DEFER>::type post_prev{};
  int prev_row = 0;
  bool have_prev = false;
  for (; blk < nblk; blk += gridDim.x, slot ^= 1) {
    const int row0 = blk * R;
    // the next tile of this workgroup and its functor's operands: requested first (the functor's operands before
    // the tile, so that waiting for them next step does not wait for anything younger)
    const int nblk_ = blk + gridDim.x, nrow0 = nblk_ * R;
    Pre pre_nxt;
    if (t < R && nblk_ < nblk && nrow0 + t < a.m) pre_nxt = op.prefetch(nrow0 + t);
    V nxt[R][NV];
    // LATE = 0: the whole next tile requested here; 1: after the dots' barrier (in flight under the functor and the column
    // sums only); 2: its first row here, the others after the barrier
    auto issue_next = [&](int r0, int r1) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (r < r0 || r >= r1) continue;
        const int row = nrow0 + r;
        const T *rp = a.A + static_cast<size_t>(row < a.m ? row : a.m - 1) * a.lda;
#pragma unroll
        for (int v = 0; v < NV; ++v) nxt[r][v] = stream_load<V>(rp + colc[v]);
      }
    };
    if constexpr (LATE == 0) issue_next(0, R);
    if constexpr (LATE == 2) issue_next(0, 1);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      T s0 = 0, s1 = 0;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        s0 += dev::vdot(cur[r][v], xv[v]);
        if constexpr (ND > 1) {
          if constexpr (X1REG) {
            s1 += dev::vdot(cur[r][v], x1v[v]);
          } else {
            const int col = (v * TPB + t) * VEC;
            if (col < a.n_pad) s1 += dev::vdot(cur[r][v], *reinterpret_cast<const V *>(s_x1 + col));
          }
        }
      }
      s0 = ALUSUM ? dev::wave_sum_alu(s0) : dev::wave_sum(s0);
      if (ND > 1) s1 = ALUSUM ? dev::wave_sum_alu(s1) : dev::wave_sum(s1);
      if (lane == 0) {
        s_part[((slot * R + r) * ND + 0) * NW + wave] = s0;
        if (ND > 1) s_part[((slot * R + r) * ND + 1) * NW + wave] = s1;
      }
    }
    __syncthreads();
    if constexpr (LATE == 1) issue_next(0, R);
    if constexpr (LATE == 2) issue_next(1, R);
    if (t < R) {
      const int row = row0 + t;
      T uu[NA];
#pragma unroll
      for (int q = 0; q < NA; ++q) uu[q] = 0;
      if (row < a.m) {
        T dots[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
          T s = 0;
#pragma unroll
          for (int w = 0; w < NW; ++w) s += s_part[((slot * R + t) * ND + d) * NW + w];
          dots[d] = s;
        }
        if constexpr (DEFER) {
          if (have_prev) op.commit(prev_row, post_prev);
          post_prev = op.compute(pre_cur, dots, sacc, uu);
          prev_row = row;
          have_prev = true;
        } else {
          op.row(row, pre_cur, dots, sacc, uu);
        }
      }
#pragma unroll
      for (int q = 0; q < NA; ++q) s_u[(slot * R + t) * NA + q] = uu[q];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      T uu[NA];
#pragma unroll
      for (int q = 0; q < NA; ++q) uu[q] = s_u[(slot * R + r) * NA + q];
#pragma unroll
      for (int q = 0; q < NA; ++q)
#pragma unroll
        for (int v = 0; v < NV; ++v) dev::vfma(acc[q][v], uu[q], cur[r][v]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int v = 0; v < NV; ++v) cur[r][v] = nxt[r][v];
    pre_cur = pre_nxt;
  }
  if constexpr (DEFER) {
    if (have_prev) op.commit(prev_row, post_prev);
  }
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    T *out = (q == 0 ? a.col_partials0 : a.col_partials1) + static_cast<size_t>(blockIdx.x) * a.n_pad;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * TPB + t) * VEC;
      if (col < a.n_pad) *reinterpret_cast<V *>(out + col) = acc[q][v];
    }
  }
  if (Op::NS > 0) {
    __syncthreads();
    dev::block_sum<NS, TPB>(sacc, s_red);
    if (t == 0) {
#pragma unroll
      for (int k = 0; k < NS; ++k) a.scalar_partials[static_cast<size_t>(blockIdx.x) * NS + k] = sacc[k];
    }
  }
}

// ---------------------------------------------------------------------------
// stream_rows2_fw_kernel: the one-pass iteration kernel with a FUNCTOR WAVEFRONT.
// What round 6's bisection found (profiles/NOTES_r06.md section 4): two workgroups per CU stream 8 % faster than three,
// and the third is only there to cover the row functor -- a serial chain on one lane per row (logistic: ~1 us per step)
// that sits between the row dots and the column sums of the SAME tile in the same wavefronts.  Prefetching the next
// tile does not help, because the chain is compute.  Here the chain leaves the streaming wavefronts: a workgroup is
// 4 streaming wavefronts + 1 functor wavefront (320 threads), and the column sums of tile k are taken one step late:
//   streaming wavefronts, step k:  dots(k) -> partials | barrier | column sums of tile k - 1 (u from the functor
//                                  wavefront, the tile from the thread's own LDS slots) | tile k -> LDS | load tile k + 1
//   functor wavefront, step k:     | barrier | functor of the rows of tile k (partials of step k) -> u(k); operands of k + 1
// One barrier per step; partials and u are double-buffered by step parity; the LDS copy of a tile is thread-private (every
// thread re-reads exactly what it wrote), so it needs no barrier.  The functor has a whole step (~3 us at two workgroups
// per CU) for its ~1 us.  Same arithmetic per row and per column partial as stream_rows2_kernel with the same R and grid.
// ---------------------------------------------------------------------------
template <typename T, int TPB, int NV, int R, int ND, int NA, int BPC, typename Op>
__global__ void __launch_bounds__(TPB + kFwThreads, (BPC * ((TPB + kFwThreads) / 64) + 3) / 4)
    stream_rows2_fw_kernel(StreamArgs2<T> a, Op op) {
  using V = typename Vec16<T>::type;
  using Pre = typename Op::Pre;
  constexpr int VEC = Vec16<T>::N;
  constexpr int NW = TPB / 64;
  constexpr int NS = Op::NS > 0 ? Op::NS : 1;
  static_assert(ND > 0 && R <= kFwThreads, "one functor lane per row of the step");
  __shared__ T s_part[2 * R * ND * NW];
  __shared__ T s_u[2 * R * NA];
  // dynamic LDS: [n_pad] second dot vector (ND > 1), then the tile copy: R * NV vectors per streaming thread
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  T *s_x1 = reinterpret_cast<T *>(s_dyn);
  V *s_tile = reinterpret_cast<V *>(s_dyn + (ND > 1 ? (static_cast<size_t>(a.n_pad) * sizeof(T) + 15) / 16 * 16 : 0));
  const int t = threadIdx.x;
  const int nblk = (a.m + R - 1) / R;

  if (t >= TPB) {
    // ---------------- functor wavefront
    const int fl = t - TPB;
    double sacc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) sacc[k] = 0.0;
    Pre pre;
    int blk = blockIdx.x;
    if (fl < R && blk < nblk && blk * R + fl < a.m) pre = op.prefetch(blk * R + fl);
    __syncthreads();   // (the streaming side's start barrier)
    int slot = 0;
    for (; blk < nblk; blk += gridDim.x, slot ^= 1) {
      __syncthreads();   // A(k): the partials of this step are there
      const int row = blk * R + fl;
      const int nb = blk + gridDim.x;
      Pre pre_nxt;
      if (fl < R && nb < nblk && nb * R + fl < a.m) pre_nxt = op.prefetch(nb * R + fl);
      if (fl < R) {
        T uu[NA];
#pragma unroll
        for (int q = 0; q < NA; ++q) uu[q] = 0;
        if (row < a.m) {
          T dots[ND];
#pragma unroll
          for (int d = 0; d < ND; ++d) {
            T sum = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += s_part[((slot * R + fl) * ND + d) * NW + w];
            dots[d] = sum;
          }
          op.row(row, pre, dots, sacc, uu);
        }
#pragma unroll
        for (int q = 0; q < NA; ++q) s_u[(slot * R + fl) * NA + q] = uu[q];
      }
      pre = pre_nxt;
    }
    __syncthreads();   // the streaming side's closing barrier: u of the last step is there
    if (Op::NS > 0) {
#pragma unroll
      for (int k = 0; k < NS; ++k) sacc[k] = dev::wave_sum(sacc[k]);
      if (fl == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) a.scalar_partials[static_cast<size_t>(blockIdx.x) * NS + k] = sacc[k];
      }
    }
    return;
  }

  // ---------------- streaming wavefronts
  const int lane = t & 63, wave = t >> 6;
  V xv[NV];
  V acc[NA][NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * TPB + t) * VEC;
    xv[v] = (col < a.n_pad) ? *reinterpret_cast<const V *>(a.xin0 + col) : dev::vzero<V>();
    if (ND > 1 && col < a.n_pad) *reinterpret_cast<V *>(s_x1 + col) = *reinterpret_cast<const V *>(a.xin1 + col);
  }
#pragma unroll
  for (int q = 0; q < NA; ++q)
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[q][v] = dev::vzero<V>();
  __syncthreads();   // start: the second dot vector is in LDS
  V av[R][NV];
  auto load_tile = [&](int blk) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = blk * R + r;
      const T *rp = a.A + static_cast<size_t>(row) * a.lda;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int col = (v * TPB + t) * VEC;
        av[r][v] = (col < a.n_pad && blk < nblk && row < a.m) ? stream_load<V>(rp + col) : dev::vzero<V>();
      }
    }
  };
  auto column_sums = [&](int slot) {   // of the tile in the LDS copy, with the u of step parity `slot`
#pragma unroll
    for (int r = 0; r < R; ++r) {
      T uu[NA];
#pragma unroll
      for (int q = 0; q < NA; ++q) uu[q] = s_u[(slot * R + r) * NA + q];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const V tv = s_tile[(r * NV + v) * TPB + t];
#pragma unroll
        for (int q = 0; q < NA; ++q) dev::vfma(acc[q][v], uu[q], tv);
      }
    }
  };
  int blk = blockIdx.x;
  load_tile(blk);
  int slot = 0;
  bool first = true;
  for (; blk < nblk; blk += gridDim.x, slot ^= 1) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      T s0 = 0, s1 = 0;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        s0 += dev::vdot(av[r][v], xv[v]);
        if (ND > 1) {
          const int col = (v * TPB + t) * VEC;
          if (col < a.n_pad) s1 += dev::vdot(av[r][v], *reinterpret_cast<const V *>(s_x1 + col));
        }
      }
      s0 = dev::wave_sum(s0);
      if (ND > 1) s1 = dev::wave_sum(s1);
      if (lane == 0) {
        s_part[((slot * R + r) * ND + 0) * NW + wave] = s0;
        if (ND > 1) s_part[((slot * R + r) * ND + 1) * NW + wave] = s1;
      }
    }
    __syncthreads();   // A(k)
    if (!first) column_sums(slot ^ 1);   // tile k - 1: its u was written during the previous step
    first = false;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int v = 0; v < NV; ++v) s_tile[(r * NV + v) * TPB + t] = av[r][v];
    load_tile(blk + gridDim.x);
  }
  __syncthreads();   // closing barrier
  if (!first) column_sums(slot ^ 1);
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    T *out = (q == 0 ? a.col_partials0 : a.col_partials1) + static_cast<size_t>(blockIdx.x) * a.n_pad;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * TPB + t) * VEC;
      if (col < a.n_pad) *reinterpret_cast<V *>(out + col) = acc[q][v];
    }
  }
}
template <typename T>
inline size_t stream2_fw_lds(int n_pad, int tpb, int nv, int rows, int nd) {
  return (nd > 1 ? (static_cast<size_t>(n_pad) * sizeof(T) + 15) / 16 * 16 : 0) + static_cast<size_t>(rows) * nv * tpb * 16;
}


// ---------------------------------------------------------------------------
// stream_rows2_timed_kernel: stream_rows2_kernel (ND > 0) with cycle stamps around the phases of a workgroup step, taken by
// thread 0 (wavefront 0): wait for the tile | dots + wavefront sums | first barrier | row functor | second barrier | column
// sums + the next tile's load issue.  Adds an explicit wait for the tile before the dots (the compiler's own would sit
// inside them).  stamps[blockIdx.x][0..5] = summed cycles per phase (s_memtime), [6] = steps.
// ---------------------------------------------------------------------------
template <typename T, int TPB, int NV, int R, int ND, int NA, typename Op>
__global__ void __launch_bounds__(TPB, stream2_waves_per_simd(TPB, NV, ND, NA)) stream_rows2_timed_kernel(StreamArgs2<T> a, Op op,
                                                                                                           unsigned long long *stamps) {
  using V = typename Vec16<T>::type;
  constexpr int VEC = Vec16<T>::N;
  constexpr int NW = TPB / 64;
  constexpr int NS = Op::NS > 0 ? Op::NS : 1;
  static_assert(ND > 0, "timed form of the dot + column-sum pass");
  __shared__ T s_part[2 * R * ND * NW];
  __shared__ T s_u[2 * R * NA];
  __shared__ double s_red[NS * NW];
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  T *s_x1 = reinterpret_cast<T *>(s_dyn);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, nsteps = 0;
  V xv[NV];
  V acc[NA][NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * TPB + t) * VEC;
    xv[v] = (col < a.n_pad) ? *reinterpret_cast<const V *>(a.xin0 + col) : dev::vzero<V>();
    if (ND > 1 && col < a.n_pad) *reinterpret_cast<V *>(s_x1 + col) = *reinterpret_cast<const V *>(a.xin1 + col);
  }
  if (ND > 1) __syncthreads();
#pragma unroll
  for (int q = 0; q < NA; ++q)
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[q][v] = dev::vzero<V>();
  double sacc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) sacc[k] = 0.0;
  const int nblk = (a.m + R - 1) / R;
  int slot = 0;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x, slot ^= 1) {
    const int row0 = blk * R;
    V av[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = row0 + r;
      const T *rp = a.A + static_cast<size_t>(row) * a.lda;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int col = (v * TPB + t) * VEC;
        V val = dev::vzero<V>();
        if (col < a.n_pad && row < a.m) val = stream_load<V>(rp + col);
        av[r][v] = val;
      }
    }
    typename Op::Pre pre;
    if (t < R && row0 + t < a.m) pre = op.prefetch(row0 + t);
    const unsigned long long c0 = __builtin_readcyclecounter();
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the tile (and the functor's operands) have arrived
    const unsigned long long c1 = __builtin_readcyclecounter();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      T s0 = 0, s1 = 0;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        s0 += dev::vdot(av[r][v], xv[v]);
        if (ND > 1) {
          const int col = (v * TPB + t) * VEC;
          if (col < a.n_pad) s1 += dev::vdot(av[r][v], *reinterpret_cast<const V *>(s_x1 + col));
        }
      }
      s0 = dev::wave_sum(s0);
      if (ND > 1) s1 = dev::wave_sum(s1);
      if (lane == 0) {
        s_part[((slot * R + r) * ND + 0) * NW + wave] = s0;
        if (ND > 1) s_part[((slot * R + r) * ND + 1) * NW + wave] = s1;
      }
    }
    const unsigned long long c2 = __builtin_readcyclecounter();
    __syncthreads();
    const unsigned long long c3 = __builtin_readcyclecounter();
    if (t < R) {
      const int row = row0 + t;
      T uu[NA];
#pragma unroll
      for (int q = 0; q < NA; ++q) uu[q] = 0;
      if (row < a.m) {
        T dots[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
          T sum = 0;
#pragma unroll
          for (int w = 0; w < NW; ++w) sum += s_part[((slot * R + t) * ND + d) * NW + w];
          dots[d] = sum;
        }
        op.row(row, pre, dots, sacc, uu);
      }
#pragma unroll
      for (int q = 0; q < NA; ++q) s_u[(slot * R + t) * NA + q] = uu[q];
    }
    const unsigned long long c4 = __builtin_readcyclecounter();
    __syncthreads();
    const unsigned long long c5 = __builtin_readcyclecounter();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      T uu[NA];
#pragma unroll
      for (int q = 0; q < NA; ++q) uu[q] = s_u[(slot * R + r) * NA + q];
#pragma unroll
      for (int q = 0; q < NA; ++q)
#pragma unroll
        for (int v = 0; v < NV; ++v) dev::vfma(acc[q][v], uu[q], av[r][v]);
    }
    const unsigned long long c6 = __builtin_readcyclecounter();
    ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += c3 - c2; ph[3] += c4 - c3; ph[4] += c5 - c4; ph[5] += c6 - c5;
    ++nsteps;
  }
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    T *out = (q == 0 ? a.col_partials0 : a.col_partials1) + static_cast<size_t>(blockIdx.x) * a.n_pad;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * TPB + t) * VEC;
      if (col < a.n_pad) *reinterpret_cast<V *>(out + col) = acc[q][v];
    }
  }
  if (Op::NS > 0) {
    __syncthreads();
    dev::block_sum<NS, TPB>(sacc, s_red);
    if (t == 0) {
#pragma unroll
      for (int k = 0; k < NS; ++k) a.scalar_partials[static_cast<size_t>(blockIdx.x) * NS + k] = sacc[k];
    }
  }
  if (t == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) stamps[static_cast<size_t>(blockIdx.x) * 8 + k] = ph[k];
    stamps[static_cast<size_t>(blockIdx.x) * 8 + 6] = nsteps;
  }
}


// ---------------------------------------------------------------------------
// stream_rows2_fast_kernel: what table 6 pointed at.  A workgroup step of stream_rows2_kernel spends ~45 % of its time in the
// DOTS phase -- not in the arithmetic (80 FMAs) but in LDS latency chains: ten serialised ds_read_b128 of the second dot vector
// (five per row, re-read for every row of the step) and four wavefront sums of six dependent ds_bpermute round trips each.
// Here (X1REG) the second dot vector lives in registers like the first (two workgroups per CU leave 256 VGPRs a wavefront),
// and (ALUSUM) the fp32 wavefront sums run in the vector ALU (v_permlane32/16_swap + DPP: the same tree, the same bits).
// ---------------------------------------------------------------------------
template <typename T, int TPB, int NV, int R, int ND, int NA, int BPC, bool X1REG, bool ALUSUM, typename Op>
__global__ void __launch_bounds__(TPB, (BPC * (TPB / 64) + 3) / 4) stream_rows2_fast_kernel(StreamArgs2<T> a, Op op) {
  using V = typename Vec16<T>::type;
  constexpr int VEC = Vec16<T>::N;
  constexpr int NW = TPB / 64;
  constexpr int NS = Op::NS > 0 ? Op::NS : 1;
  static_assert(ND > 0, "dot + column-sum pass");
  __shared__ T s_part[2 * R * ND * NW];
  __shared__ T s_u[2 * R * NA];
  __shared__ double s_red[NS * NW];
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  T *s_x1 = reinterpret_cast<T *>(s_dyn);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  V xv[NV], x1v[(ND > 1 && X1REG) ? NV : 1];
  V acc[NA][NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * TPB + t) * VEC;
    xv[v] = (col < a.n_pad) ? *reinterpret_cast<const V *>(a.xin0 + col) : dev::vzero<V>();
    if constexpr (ND > 1 && X1REG) {
      x1v[v] = (col < a.n_pad) ? *reinterpret_cast<const V *>(a.xin1 + col) : dev::vzero<V>();
    } else if (ND > 1 && col < a.n_pad) {
      *reinterpret_cast<V *>(s_x1 + col) = *reinterpret_cast<const V *>(a.xin1 + col);
    }
  }
  if (ND > 1 && !X1REG) __syncthreads();
#pragma unroll
  for (int q = 0; q < NA; ++q)
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[q][v] = dev::vzero<V>();
  double sacc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) sacc[k] = 0.0;
  const int nblk = (a.m + R - 1) / R;
  int slot = 0;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x, slot ^= 1) {
    const int row0 = blk * R;
    V av[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = row0 + r;
      const T *rp = a.A + static_cast<size_t>(row) * a.lda;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int col = (v * TPB + t) * VEC;
        V val = dev::vzero<V>();
        if (col < a.n_pad && row < a.m) val = stream_load<V>(rp + col);
        av[r][v] = val;
      }
    }
    typename Op::Pre pre;
    if (t < R && row0 + t < a.m) pre = op.prefetch(row0 + t);
    T s0[R], s1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s0[r] = 0; s1[r] = 0;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        s0[r] += dev::vdot(av[r][v], xv[v]);
        if constexpr (ND > 1) {
          if constexpr (X1REG) {
            s1[r] += dev::vdot(av[r][v], x1v[v]);
          } else {
            const int col = (v * TPB + t) * VEC;
            if (col < a.n_pad) s1[r] += dev::vdot(av[r][v], *reinterpret_cast<const V *>(s_x1 + col));
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s0[r] = ALUSUM ? dev::wave_sum_alu(s0[r]) : dev::wave_sum(s0[r]);
      if (ND > 1) s1[r] = ALUSUM ? dev::wave_sum_alu(s1[r]) : dev::wave_sum(s1[r]);
    }
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        s_part[((slot * R + r) * ND + 0) * NW + wave] = s0[r];
        if (ND > 1) s_part[((slot * R + r) * ND + 1) * NW + wave] = s1[r];
      }
    }
    __syncthreads();
    if (t < R) {
      const int row = row0 + t;
      T uu[NA];
#pragma unroll
      for (int q = 0; q < NA; ++q) uu[q] = 0;
      if (row < a.m) {
        T dots[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
          T sum = 0;
#pragma unroll
          for (int w = 0; w < NW; ++w) sum += s_part[((slot * R + t) * ND + d) * NW + w];
          dots[d] = sum;
        }
        op.row(row, pre, dots, sacc, uu);
      }
#pragma unroll
      for (int q = 0; q < NA; ++q) s_u[(slot * R + t) * NA + q] = uu[q];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      T uu[NA];
#pragma unroll
      for (int q = 0; q < NA; ++q) uu[q] = s_u[(slot * R + r) * NA + q];
#pragma unroll
      for (int q = 0; q < NA; ++q)
#pragma unroll
        for (int v = 0; v < NV; ++v) dev::vfma(acc[q][v], uu[q], av[r][v]);
    }
  }
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    T *out = (q == 0 ? a.col_partials0 : a.col_partials1) + static_cast<size_t>(blockIdx.x) * a.n_pad;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * TPB + t) * VEC;
      if (col < a.n_pad) *reinterpret_cast<V *>(out + col) = acc[q][v];
    }
  }
  if (Op::NS > 0) {
    __syncthreads();
    dev::block_sum<NS, TPB>(sacc, s_red);
    if (t == 0) {
#pragma unroll
      for (int k = 0; k < NS; ++k) a.scalar_partials[static_cast<size_t>(blockIdx.x) * NS + k] = sacc[k];
    }
  }
}

}  // namespace pogs_amd
