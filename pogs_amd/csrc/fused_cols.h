// Column-side kernels of the one-pass dense iteration (dense_solver.h: iteration_fused).
//
// pre_cols_kernel: ONE launch for everything that happens per column between two passes over A
//   -- the second stage of both column-sum sets the pass left behind (A^T yhat, and the exact
//   dual residual's A^T (y12 + c yt - yprev)), the x half of the prox step (pogs.cpp:257-278) and
//   the exact dual residual itself (pogs.cpp:366-373).  It replaces three launches
//   (admm_pre's x half + two reduce_cols), ~18 us + two kernel boundaries at C2.
// pack_cols_kernel (row shards): the same second stage, written as fp64 into the pack buffer that
//   ONE ncclAllReduce then sums over the ranks, plus -- in an extra workgroup -- the y-side scalar
//   sums that ride in the tail of that buffer (one collective per iteration).
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"
#include "prox.h"
#include "reduce.h"
#include "stream.h"
#include "vec_kernels.h"

namespace pogs_amd {

// total[j] = sum_b partials[b][j] exactly as reduce_cols_kernel forms it (8 groups of every 8th
// partial, then the groups in order), so that both second stages give the same bits.
template <typename T>
__device__ __forceinline__ typename Vec16<T>::type colsum_group(const T *partials, int nparts, int n_pad, int col, int g) {
  using V = typename Vec16<T>::type;
  V sum = dev::vzero<V>();
  int b = g;
  for (; b + 56 < nparts; b += 64) {
    V v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const V *>(partials + static_cast<size_t>(b + 8 * q) * n_pad + col);
#pragma unroll
    for (int q = 0; q < 8; ++q) dev::vadd(sum, v[q]);
  }
  for (; b < nparts; b += 8) {
    const V v = *reinterpret_cast<const V *>(partials + static_cast<size_t>(b) * n_pad + col);
    dev::vadd(sum, v);
  }
  return sum;
}

template <typename T>
struct PreColsArgs {
  const T *part0, *part1;   // SRC64 = false: [nparts][n_pad] partial column sums of the two sets; part1 == nullptr:
                            // the second set was not formed (fp64 lean passes): its totals and sum read as zero
  int nparts;
  const double *tot64;      // SRC64 = true: [2][n_pad] totals (summed over the ranks)
  int n, n_pad;
  FnView<T> g;              // scaled g
  const T *x_cur, *xt;
  T zt_scale, rho, alpha;
  T *x12, *xtemp;           // out: prox point, over-relaxed point
  T *rhs;                   // out: A^T yhat (padding columns zero)
  double *partials;         // out: [grid][4] = sum w h, |w|^2, |h|^2, |exact dual residual|^2
};

// Workgroup = 256 threads = 16 column vectors (64 fp32 / 32 fp64 columns) x 16 groups of partials;
// then one thread per column finishes the sums and runs the column's x-half.  (Narrow column
// blocks: the launch has ~n / 64 workgroups, enough to spread the 40 MB of partials over the CUs.)
constexpr int kPreColsVecs = 16, kPreColsGroups = 16;
inline int pre_cols_grid(int n_pad, int vec) { return (n_pad / vec + kPreColsVecs - 1) / kPreColsVecs; }

template <typename T, bool SRC64>
__global__ void __launch_bounds__(256) pre_cols_kernel(PreColsArgs<T> a) {
  using V = typename Vec16<T>::type;
  constexpr int VEC = Vec16<T>::N;
  constexpr int NC = kPreColsVecs * VEC;   // columns per workgroup
  __shared__ T s_v[2][kPreColsGroups][NC];
  __shared__ double s_red[4 * 4];
  const int cx = threadIdx.x % kPreColsVecs, g = threadIdx.x / kPreColsVecs;
  const int col0 = blockIdx.x * NC;
  if (!SRC64) {
    const int col = col0 + cx * VEC;
    V s0 = dev::vzero<V>(), s1 = dev::vzero<V>();
    if (col < a.n_pad) {
      // two independent chains per set keep more loads in flight (the partials sit in L2 / Infinity Cache)
      V t0 = dev::vzero<V>(), t1 = dev::vzero<V>();
      int b = g;
      if (a.part1) {   // (uniform)
        for (; b + kPreColsGroups < a.nparts; b += 2 * kPreColsGroups) {
          const V u0 = *reinterpret_cast<const V *>(a.part0 + static_cast<size_t>(b) * a.n_pad + col);
          const V u1 = *reinterpret_cast<const V *>(a.part1 + static_cast<size_t>(b) * a.n_pad + col);
          const V w0 = *reinterpret_cast<const V *>(a.part0 + static_cast<size_t>(b + kPreColsGroups) * a.n_pad + col);
          const V w1 = *reinterpret_cast<const V *>(a.part1 + static_cast<size_t>(b + kPreColsGroups) * a.n_pad + col);
          dev::vadd(s0, u0);
          dev::vadd(s1, u1);
          dev::vadd(t0, w0);
          dev::vadd(t1, w1);
        }
        if (b < a.nparts) {
          dev::vadd(s0, *reinterpret_cast<const V *>(a.part0 + static_cast<size_t>(b) * a.n_pad + col));
          dev::vadd(s1, *reinterpret_cast<const V *>(a.part1 + static_cast<size_t>(b) * a.n_pad + col));
        }
      } else {   // the first set alone, in the same order
        for (; b + kPreColsGroups < a.nparts; b += 2 * kPreColsGroups) {
          const V u0 = *reinterpret_cast<const V *>(a.part0 + static_cast<size_t>(b) * a.n_pad + col);
          const V w0 = *reinterpret_cast<const V *>(a.part0 + static_cast<size_t>(b + kPreColsGroups) * a.n_pad + col);
          dev::vadd(s0, u0);
          dev::vadd(t0, w0);
        }
        if (b < a.nparts)
          dev::vadd(s0, *reinterpret_cast<const V *>(a.part0 + static_cast<size_t>(b) * a.n_pad + col));
      }
      dev::vadd(s0, t0);
      dev::vadd(s1, t1);
    }
    *reinterpret_cast<V *>(&s_v[0][g][cx * VEC]) = s0;
    *reinterpret_cast<V *>(&s_v[1][g][cx * VEC]) = s1;
    __syncthreads();
  }
  double sacc[4] = {0.0, 0.0, 0.0, 0.0};
  const int j = col0 + static_cast<int>(threadIdx.x);
  if (threadIdx.x < NC && j < a.n_pad) {
    T t0, t1;
    if (SRC64) {
      t0 = static_cast<T>(a.tot64[j]);
      t1 = static_cast<T>(a.tot64[a.n_pad + j]);
    } else {
      t0 = s_v[0][0][threadIdx.x];
      t1 = s_v[1][0][threadIdx.x];
#pragma unroll
      for (int q = 1; q < kPreColsGroups; ++q) {   // groups in order: a fixed summation order
        t0 += s_v[0][q][threadIdx.x];
        t1 += s_v[1][q][threadIdx.x];
      }
    }
    if (j < a.n) {
      const T prev = a.x_cur[j];
      const T xtj = a.xt[j];
      const T ztv = a.zt_scale * xtj;
      const T v = prev - ztv;                                                          // pogs.cpp:257
      const T h = dev::ProxEval(a.g.h[j], a.g.a[j], a.g.b[j], a.g.c[j], a.g.d[j], a.g.e[j], v, a.rho);   // :263
      const T w = v - h;                                                               // :267
      a.x12[j] = h;
      a.xtemp[j] = ztv + a.alpha * h + (static_cast<T>(1) - a.alpha) * prev;           // :276-278
      sacc[0] = static_cast<double>(w) * h;                                            // :268
      sacc[1] = static_cast<double>(w) * w;
      sacc[2] = static_cast<double>(h) * h;
      a.rhs[j] = t0;
      const T sd = t1 + h + a.zt_scale * xtj - prev;                                   // :366-373
      sacc[3] = (SRC64 || a.part1) ? static_cast<double>(sd) * sd : 0.0;
    } else {
      a.rhs[j] = static_cast<T>(0);
    }
  }
  __syncthreads();
  dev::block_sum<4, 256>(sacc, s_red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) a.partials[static_cast<size_t>(blockIdx.x) * 4 + k] = sacc[k];
  }
}

template <typename T>
void launch_pre_cols(const PreColsArgs<T> &a, bool src64, hipStream_t s) {
  const int grid = pre_cols_grid(a.n_pad, Vec16<T>::N);
  if (src64) hipLaunchKernelGGL((pre_cols_kernel<T, true>), dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((pre_cols_kernel<T, false>), dim3(grid), dim3(256), 0, s, a);
}

// Workgroups [0, ncb): pack[j] = total0[j], pack[n_pad + j] = total1[j] as doubles.
// Workgroup ncb: the scalar sums of up to two jobs into pack[2 n_pad ...] (job order, k order).
struct PackJobs {
  SumJob j[2];
  int njobs;
};
template <typename T>
__global__ void __launch_bounds__(256) pack_cols_kernel(const T *part0, const T *part1, int nparts, int n_pad,
                                                        double *pack, PackJobs jobs) {
  using V = typename Vec16<T>::type;
  constexpr int VEC = Vec16<T>::N;
  __shared__ V s_v[2][8][32];
  __shared__ double s_w[4];
  const int ncb = (n_pad / VEC + 31) / 32;   // = reduce_cols_grid
  if (static_cast<int>(blockIdx.x) == ncb) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double *out = pack + 2 * static_cast<size_t>(n_pad);
    for (int q = 0; q < jobs.njobs; ++q) {
      const SumJob job = jobs.j[q];
      const size_t stride = job.stride > 0 ? job.stride : job.ns;
      for (int k = 0; k < job.ns; ++k) {
        const double *p = job.partials + job.offset + k;
        double sum = 0;
        for (int b = t; b < job.nparts; b += 256) sum += p[static_cast<size_t>(b) * stride];
        sum = dev::wave_sum(sum);
        if (lane == 0) s_w[wave] = sum;
        __syncthreads();
        if (t == 0) *out = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        ++out;
        __syncthreads();
      }
    }
    return;
  }
  const int cx = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int col = (blockIdx.x * 32 + cx) * VEC;
  V s0 = dev::vzero<V>(), s1 = dev::vzero<V>();
  if (col < n_pad) {
    s0 = colsum_group<T>(part0, nparts, n_pad, col, g);
    s1 = colsum_group<T>(part1, nparts, n_pad, col, g);
  }
  s_v[0][g][cx] = s0;
  s_v[1][g][cx] = s1;
  __syncthreads();
  if (g == 0 && col < n_pad) {
    V tot0 = s_v[0][0][cx], tot1 = s_v[1][0][cx];
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      dev::vadd(tot0, s_v[0][q][cx]);
      dev::vadd(tot1, s_v[1][q][cx]);
    }
    T t0[VEC], t1[VEC];
    __builtin_memcpy(t0, &tot0, sizeof(V));
    __builtin_memcpy(t1, &tot1, sizeof(V));
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      pack[col + i] = static_cast<double>(t0[i]);
      pack[n_pad + col + i] = static_cast<double>(t1[i]);
    }
  }
}

template <typename T>
void launch_pack_cols(const T *part0, const T *part1, int nparts, int n_pad, double *pack, const PackJobs &jobs,
                      hipStream_t s) {
  const int grid = reduce_cols_grid(n_pad, Vec16<T>::N) + 1;
  hipLaunchKernelGGL((pack_cols_kernel<T>), dim3(grid), dim3(256), 0, s, part0, part1, nparts, n_pad, pack, jobs);
}

}  // namespace pogs_amd
