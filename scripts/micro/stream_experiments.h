// Kernels that were built, measured and NOT shipped in round 6 (profiles/NOTES_r06.md section 4): kept as the subjects of
// scripts/micro/c3_bisect.hip, outside the library.  Same contracts as stream_rows2_kernel (csrc/stream.h).
#pragma once
#include "stream.h"

namespace pogs_amd {

constexpr int kFwThreads = 64;   // the functor wavefront

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
template <typename T, int TPB, int NV, int R, int ND, int NA, int BPC, typename Op>
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
  int blk = blockIdx.x;
  V cur[R][NV];
  Pre pre_cur;
  if (t < R && blk < nblk && blk * R + t < a.m) pre_cur = op.prefetch(blk * R + t);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = blk * R + r;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * TPB + t) * VEC;
      cur[r][v] = (col < a.n_pad && blk < nblk && row < a.m) ? stream_load<V>(a.A + static_cast<size_t>(row) * a.lda + col)
                                                              : dev::vzero<V>();
    }
  }
  int slot = 0;
  for (; blk < nblk; blk += gridDim.x, slot ^= 1) {
    const int row0 = blk * R;
    // the next tile of this workgroup and its functor's operands: requested first (the functor's operands before
    // the tile, so that waiting for them next step does not wait for anything younger)
    const int nblk_ = blk + gridDim.x, nrow0 = nblk_ * R;
    Pre pre_nxt;
    if (t < R && nblk_ < nblk && nrow0 + t < a.m) pre_nxt = op.prefetch(nrow0 + t);
    V nxt[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = nrow0 + r;
      const T *rp = a.A + static_cast<size_t>(row) * a.lda;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int col = (v * TPB + t) * VEC;
        nxt[r][v] = (col < a.n_pad && nblk_ < nblk && row < a.m) ? stream_load<V>(rp + col) : dev::vzero<V>();
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      T s0 = 0, s1 = 0;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        s0 += dev::vdot(cur[r][v], xv[v]);
        if (ND > 1) {
          const int col = (v * TPB + t) * VEC;
          if (col < a.n_pad) s1 += dev::vdot(cur[r][v], *reinterpret_cast<const V *>(s_x1 + col));
        }
      }
      s0 = dev::wave_sum(s0);
      if (ND > 1) s1 = dev::wave_sum(s1);
      if (lane == 0) {
        s_part[((slot * R + r) * ND + 0) * NW + wave] = s0;
        if (ND > 1) s_part[((slot * R + r) * ND + 1) * NW + wave] = s1;
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
          T s = 0;
#pragma unroll
          for (int w = 0; w < NW; ++w) s += s_part[((slot * R + t) * ND + d) * NW + w];
          dots[d] = s;
        }
        op.row(row, pre_cur, dots, sacc, uu);
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

}  // namespace pogs_amd
