// Device-resident CGLS loop of the sparse projector (single GPU, tiled lane-stream storage).
//
// Reference: ProjectorCgls::Project (src/cpu/projector/projector_cgls.cpp:52-88) -> cgls::Solve
// (src/cpu/include/cgls.h:200-323).  The reference's loop reads three scalars per step (alpha,
// beta, the stopping test); round 2 formed alpha and beta on the device but still polled the host
// once per step for the stopping test and spent ~9 launches per step.  Here a CG step is six
// launches that never involve the host:
//
//   L1  q~ = A p           spmv_sell_kernel (partial sums per column group)
//   R1  q = sum of groups, partial |q|^2                                   cgf_reduce_kernel
//   U1  alpha (every block adds up R1's and U2's records itself, in one fixed order);
//       x += alpha p, r -= alpha q, y_new += alpha q, partial |x|^2        cgf_step_a_kernel
//   L2  s~ = A^T r         spmv_sell_kernel
//   R2  s = sum of groups - x, partial |s|^2                               cgf_reduce_kernel
//   U2  beta, gamma, the stopping test of cgls.h:301-305 -> S[kFcDone];
//       p = s + beta p, partial |p|^2                                      cgf_step_b_kernel
//
// "Every block adds up the records itself": a scalar that needs a sum over the whole vector is
// formed by EACH block of the consuming launch from the per-block records the producing launch left
// behind -- at most kCgfBlocks = 512 from the loop's own launches and from a product with column
// groups, one per row range from a product without them (more than 512 only beyond 8.4 million rows
// at the full tile height: correct, the consumers just re-read more) -- 4 KB from L2, the same order
// in every block, so all blocks hold the same bits; block 0 also stores it for the host and for later launches.  No atomics, no fences, no
// "last block" -- an in-kernel finaliser behind a device counter was built first and measured:
// its agent-scope release / acquire fences (L2 write-back / invalidate on a multi-XCD part) cost
// 19 us per SpMV and its one-workgroup-per-row-range functor tail 10 us more.
//
// Every launch of the loop starts by reading S[kFcDone] (written by an earlier launch) and returns
// at once when the loop has ended.  The host enqueues as many steps as the previous projection
// took, then the closing launch -- which runs only if S[kFcDone] is set -- and the iteration's
// publish, and polls ONCE per ADMM iteration; if the loop had not ended it enqueues one more step
// and the closing launches again.
//
// y = A x without the product (projector_cgls.cpp:78): x_new = x_warm + sum_k alpha_k p_k, so
// A x_new = y_warm + sum_k alpha_k q_k with the q_k = A p_k the loop has already formed (y_warm =
// A x_warm is the previous iteration's y, see SparseSolver::cgls_project).  U1 accumulates it; the
// explicit product is still taken every POGS_AMD_YSYNC-th iteration (default 16), which bounds the
// rounding drift of the recurrence.
#pragma once
#include <hip/hip_runtime.h>

#include "reduce.h"
#include "vec_kernels.h"

namespace pogs_amd {
namespace {

constexpr int kCgfTpb = 1024;     // few, fat blocks: every block of a consumer re-adds the producer's records
constexpr int kCgfWaves = kCgfTpb / 64;
constexpr int kCgfBlocks = 512;   // upper bound on the blocks (= scalar records) of the loop's vector launches

inline int cgf_blocks(int n) { return std::max(1, std::min(kCgfBlocks, (n + kCgfTpb - 1) / kCgfTpb)); }

// Sum of `count` records of NS doubles (record b at p + b * NS): every thread of the block gets
// the totals, and every block that runs this on the same records gets the same bits (per-thread
// strided partial sums, wavefront butterfly, the four wavefront totals in order).
template <int NS>
__device__ __forceinline__ void cgf_sum(const double *p, int count, double (&out)[NS], double *smem /* [NS * kCgfWaves] */) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int k = 0; k < NS; ++k) out[k] = 0.0;
  for (int b = t; b < count; b += kCgfTpb) {
#pragma unroll
    for (int k = 0; k < NS; ++k) out[k] += p[static_cast<size_t>(b) * NS + k];
  }
  __syncthreads();   // smem may still be read from a previous call
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const double w = dev::wave_sum(out[k]);
    if (lane == 0) smem[k * kCgfWaves + wave] = w;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < kCgfWaves; ++w) tot += smem[k * kCgfWaves + w];
    out[k] = tot;
  }
}

// s_j = (A^T r)_j - shift xcg_j ; p_j = s_j ; |s|^2                        (cgls.h:236-245)
// u (row shards, else null): keeps (A^T r)_j, the start of the recurrence of CgfStepA
template <typename T>
struct SpCgInitOp {
  static constexpr int NS = 1;
  T shift;
  const T *x;
  T *s, *p;
  T *u = nullptr;
  template <int N>
  __device__ __forceinline__ void row(int j, T dot, double (&acc)[N]) const { apply(j, dot, load(j), acc); }
  struct In { T x; };
  __device__ __forceinline__ In load(int j) const { return In{x[j]}; }
  template <int N>
  __device__ __forceinline__ void apply(int j, T dot, const In &in, double (&acc)[N]) const {
    const T v = dot - shift * in.x;
    if (u) u[j] = dot;
    s[j] = v;
    p[j] = v;
    dev::prod_acc(acc[0], v, v);
  }
};

// guard convention of the loop's launches: run_if_done = 0 -> return when S[kFcDone] != 0 (the CG
// steps), 1 -> return when S[kFcDone] == 0 (the closing launches), -1 -> always run
__device__ __forceinline__ bool cgf_skip(const double *S, int run_if_done) {
  if (run_if_done < 0) return false;
  return (S[kFcDone] != 0.0) != (run_if_done != 0);
}

// R: row r = sum of its ncg partial sums in group order (what reduce_parts_kernel does), the row
// functor, one scalar record per block.  U rows x ncg loads are requested before any is used.
template <typename T, typename Op>
__global__ void __launch_bounds__(kCgfTpb) cgf_reduce_kernel(const T *__restrict__ part, int nrows, int ncg, Op op,
                                                             double *rec, const double *S, int run_if_done) {
  constexpr int NS = Op::NS > 0 ? Op::NS : 1;
  __shared__ double s_red[NS * (kCgfTpb / 64)];
  if (cgf_skip(S, run_if_done)) return;
  double sacc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) sacc[k] = 0.0;
  constexpr int U = 4;
  const int stride = gridDim.x * kCgfTpb;
  for (int r0 = blockIdx.x * kCgfTpb + threadIdx.x; r0 < nrows; r0 += stride * U) {
    T v[U];
    typename Op::In in[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = min(r0 + u * stride, nrows - 1);   // (clamped: past the end the last row again, not applied)
      v[u] = part[r];
      in[u] = op.load(r);
    }
    for (int g = 1; g < ncg; ++g) {
      T w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) w[u] = part[static_cast<size_t>(g) * nrows + min(r0 + u * stride, nrows - 1)];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] += w[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + u * stride;
      if (r < nrows) op.apply(r, v[u], in[u], sacc);
    }
  }
  dev::block_sum<NS, kCgfTpb>(sacc, s_red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NS; ++k) rec[static_cast<size_t>(blockIdx.x) * NS + k] = sacc[k];
  }
}

template <typename T>
struct CgfStepA {
  int n, m;
  double *S;
  int first;                 // step 0: gamma = |s_0|^2 from rec_s0 (and |p|^2 = gamma), else gamma from S
  int gslot;                 // gamma slot to read (not first)
  const double *rec_s0; int nrec_s0;
  const double *rec_p; int nrec_p;
  const double *rec_q; int nrec_q;
  double shift, eps;
  const T *p; T *x;
  const T *q; T *r;
  const T *ycur; T *ynew;    // ynew == nullptr: the y recurrence is off (explicit product at the end)
  double *rec_x;             // [blocks]
  // Row shards (u != nullptr): u = A^T r over ALL ranks is kept by recurrence -- t = A^T q (the
  // all-reduced product of this step, which travelled together with the |q|^2 records): u -= alpha t,
  // s = u - shift x, |s|^2 records.  One collective per CG step instead of two (|q|^2, then A^T r).
  T *u; const T *t; T *s;
  double *rec_s;             // [nb_n]
  // Blocks that take part in the n-sided loops and write their records (|x|^2, |s|^2): cgf_blocks(n),
  // whatever the launch's grid -- which follows max(n, m) with m this rank's LOCAL row count.  The
  // replicated sums |x|^2 and |s|^2 (beta, the stopping test, the number of steps and with it the number
  // of collectives a rank issues) must come out bit-identical on every rank of a row-sharded solve,
  // also with unequal shards: their split into records depends on n alone.
  int nb_n;
};
// U1: alpha = gamma / (|q|^2 + shift |p|^2) (cgls.h:262-271); x += alpha p, r -= alpha q (:274-277),
// y_new = (first ? y_warm : y_new) + alpha q; partial |x|^2 (:298)
template <typename T>
__global__ void __launch_bounds__(kCgfTpb) cgf_step_a_kernel(CgfStepA<T> a) {
  __shared__ double s_sum[kCgfWaves];
  __shared__ double s_red[kCgfTpb / 64];
  if (cgf_skip(a.S, 0)) return;
  double gamma, p2;
  if (a.first) {
    double g[1];
    cgf_sum<1>(a.rec_s0, a.nrec_s0, g, s_sum);
    gamma = g[0];
    p2 = g[0];
  } else {
    double g[1];
    cgf_sum<1>(a.rec_p, a.nrec_p, g, s_sum);
    gamma = a.S[kFcGamma0 + a.gslot];
    p2 = g[0];
  }
  if (a.first && sqrt(gamma) < a.eps) {   // flag 1 (cgls.h:247-252): nothing to do, x stays
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      a.S[kFcGamma0] = gamma;
      a.S[kFcNorms0] = gamma;
      a.S[kFcDone] = 1.0;   // (a block that starts after this store returns at the guard: same outcome)
    }
    return;
  }
  double q2[1];
  cgf_sum<1>(a.rec_q, a.nrec_q, q2, s_sum);
  double delta = q2[0] + a.shift * p2;
  const bool indef = delta <= 0.0;
  if (delta == 0.0) delta = a.eps;
  const double alpha_d = gamma / delta;
  if (blockIdx.x == 0 && threadIdx.x == 0) {   // (slots nobody reads in this launch)
    if (a.first) {
      a.S[kFcGamma0] = gamma;
      a.S[kFcNorms0] = gamma;
      a.S[kFcIndef] = 0.0;
    }
    if (indef) a.S[kFcIndef] = 1.0;
    a.S[kCgQ2] = q2[0];
    a.S[kCgP2] = p2;
    a.S[kFcDelta] = delta;
    a.S[kFcAlpha] = alpha_d;
  }
  const T alpha = static_cast<T>(alpha_d), neg_alpha = static_cast<T>(-alpha_d);
  const int stride = gridDim.x * kCgfTpb, t0 = blockIdx.x * kCgfTpb + threadIdx.x;
  const bool n_side = static_cast<int>(blockIdx.x) < a.nb_n;
  const int stride_n = a.nb_n * kCgfTpb, n_end = n_side ? a.n : 0;
  double acc[1] = {0.0};
  double acc_s[1] = {0.0};
  if (a.u) {
    const T sh = static_cast<T>(a.shift);
    for (int i = t0; i < n_end; i += stride_n) {
      const T v = a.x[i] + alpha * a.p[i];
      a.x[i] = v;
      dev::prod_acc(acc[0], v, v);
      const T un = a.u[i] + neg_alpha * a.t[i];
      a.u[i] = un;
      const T sv = un - sh * v;                  // cgls.h:281-286 with A^T r from the recurrence
      a.s[i] = sv;
      dev::prod_acc(acc_s[0], sv, sv);
    }
  } else {
    for (int i = t0; i < n_end; i += stride_n) {
      const T v = a.x[i] + alpha * a.p[i];
      a.x[i] = v;
      dev::prod_acc(acc[0], v, v);
    }
  }
  if (a.ynew) {
    const T *ysrc = a.first ? a.ycur : a.ynew;
    for (int i = t0; i < a.m; i += stride) {
      const T qi = a.q[i];
      a.r[i] += neg_alpha * qi;
      a.ynew[i] = ysrc[i] + alpha * qi;
    }
  } else {
    for (int i = t0; i < a.m; i += stride) a.r[i] += neg_alpha * a.q[i];
  }
  if (!n_side) return;   // (uniform per block) no record of this block
  dev::block_sum<1, kCgfTpb>(acc, s_red);
  if (threadIdx.x == 0) a.rec_x[blockIdx.x] = acc[0];
  if (a.u) {
    __syncthreads();
    dev::block_sum<1, kCgfTpb>(acc_s, s_red);
    if (threadIdx.x == 0) a.rec_s[blockIdx.x] = acc_s[0];
  }
}

template <typename T>
struct CgfStepB {
  int n;
  double *S;
  int k;                     // step index: gamma is read from slot k & 1 and written to the other
  const double *rec_s; int nrec_s;
  const double *rec_x; int nrec_x;
  double tol;
  int maxit;
  const T *s; T *p;
  double *rec_p;             // [blocks]
};
// U2: beta = |s|^2 / gamma, gamma = |s|^2 (cgls.h:288-292); the stopping test (:301-305);
// p = s + beta p, partial |p|^2 (:295-296)
template <typename T>
__global__ void __launch_bounds__(kCgfTpb) cgf_step_b_kernel(CgfStepB<T> a) {
  __shared__ double s_sum[kCgfWaves];
  __shared__ double s_red[kCgfTpb / 64];
  if (cgf_skip(a.S, 0)) return;
  double g[1], x2[1];
  cgf_sum<1>(a.rec_s, a.nrec_s, g, s_sum);
  cgf_sum<1>(a.rec_x, a.nrec_x, x2, s_sum);
  const double g1 = a.S[kFcGamma0 + (a.k & 1)];
  const double beta_d = g[0] / g1;
  const double norms = sqrt(g[0]), norms0 = sqrt(a.S[kFcNorms0]), normx = sqrt(x2[0]);
  const bool converged = (norms <= norms0 * a.tol) || (normx * a.tol >= 1.0);
  const bool done = converged || a.k + 1 >= a.maxit;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.S[kFcGamma0 + ((a.k + 1) & 1)] = g[0];
    a.S[kFcBeta] = beta_d;
    a.S[kCgS2] = g[0];
    a.S[kCgX2] = x2[0];
    a.S[kFcSteps] = static_cast<double>(a.k + 1);
    if (done) a.S[kFcDone] = 1.0;   // (a block that starts after this store returns at the guard: same outcome)
  }
  if (done) return;                 // p is not needed any more
  const T beta = static_cast<T>(beta_d);
  const int stride = gridDim.x * kCgfTpb;
  double acc[1] = {0.0};
  for (int i = blockIdx.x * kCgfTpb + threadIdx.x; i < a.n; i += stride) {
    const T v = a.s[i] + beta * a.p[i];
    a.p[i] = v;
    dev::prod_acc(acc[0], v, v);
  }
  dev::block_sum<1, kCgfTpb>(acc, s_red);
  if (threadIdx.x == 0) a.rec_p[blockIdx.x] = acc[0];
}

template <typename T>
struct CgfClose {
  int n, m;
  const double *S;
  T *x; const T *xprev, *x12; T *xtemp;           // x <- x + x0 (x0 = xtemp), the x-half bookkeeping
  T *ynew; const T *yprev, *y12; T *ytemp;        // y half (the recurrence's y_new); ynew == nullptr: skip
  double *part;                                    // [blocks_x + blocks_y][2], blocks = pre_blocks()
  int blocks_x;
};
// closing launch: x <- x + x0 (projector_cgls.cpp:75) and the element-wise projection tail of both
// halves (admm_tail_kernel / SpTailOp): sums of (z_prev - z)^2, (z12 - z)^2, ztemp <- ztemp - z.
// Zero CG steps: y_new is y_warm.
template <typename T>
__global__ void __launch_bounds__(kVecTpb) cgf_close_kernel(CgfClose<T> a) {
  __shared__ double s_red[2 * (kVecTpb / 64)];
  if (a.S[kFcDone] == 0.0) return;
  double acc[2] = {0.0, 0.0};
  if (static_cast<int>(blockIdx.x) < a.blocks_x) {
    T x0[kPreU], xc[kPreU], xp[kPreU], xh[kPreU];
#pragma unroll
    for (int u = 0; u < kPreU; ++u) {
      const int i = (blockIdx.x * kPreU + u) * kVecTpb + threadIdx.x;
      if (i < a.n) { x0[u] = a.xtemp[i]; xc[u] = a.x[i]; xp[u] = a.xprev[i]; xh[u] = a.x12[i]; }
    }
#pragma unroll
    for (int u = 0; u < kPreU; ++u) {
      const int i = (blockIdx.x * kPreU + u) * kVecTpb + threadIdx.x;
      if (i < a.n) {
        const T zn = xc[u] + x0[u];   // the reference: x <- 1 * x0 + x (blas_axpy)
        a.x[i] = zn;
        const T d1 = xp[u] - zn, d2 = xh[u] - zn;
        dev::prod_acc(acc[0], d1, d1);
        dev::prod_acc(acc[1], d2, d2);
        a.xtemp[i] = x0[u] - zn;
      }
    }
  } else {
    const bool no_step = a.S[kFcSteps] == 0.0;
    T yp[kPreU], yn[kPreU], yh[kPreU], yt[kPreU];
#pragma unroll
    for (int u = 0; u < kPreU; ++u) {
      const int i = ((blockIdx.x - a.blocks_x) * kPreU + u) * kVecTpb + threadIdx.x;
      if (i < a.m) { yp[u] = a.yprev[i]; yn[u] = no_step ? yp[u] : a.ynew[i]; yh[u] = a.y12[i]; yt[u] = a.ytemp[i]; }
    }
#pragma unroll
    for (int u = 0; u < kPreU; ++u) {
      const int i = ((blockIdx.x - a.blocks_x) * kPreU + u) * kVecTpb + threadIdx.x;
      if (i < a.m) {
        const T zn = yn[u];
        a.ynew[i] = zn;
        const T d1 = yp[u] - zn, d2 = yh[u] - zn;
        dev::prod_acc(acc[0], d1, d1);
        dev::prod_acc(acc[1], d2, d2);
        a.ytemp[i] = yt[u] - zn;
      }
    }
  }
  dev::block_sum<2, kVecTpb>(acc, s_red);
  if (threadIdx.x == 0) {
    a.part[static_cast<size_t>(blockIdx.x) * 2 + 0] = acc[0];
    a.part[static_cast<size_t>(blockIdx.x) * 2 + 1] = acc[1];
  }
}

}  // namespace
}  // namespace pogs_amd
