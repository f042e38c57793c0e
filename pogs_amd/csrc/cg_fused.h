// Device-resident CGLS loop of the sparse projector (single GPU, tiled lane-stream storage).
//
// Reference: ProjectorCgls::Project (src/cpu/projector/projector_cgls.cpp:52-88) -> cgls::Solve
// (src/cpu/include/cgls.h:200-323).  The reference's loop reads three scalars per step (alpha,
// beta, the stopping test); round 2 formed alpha and beta on the device but still polled the host
// once per step for the stopping test and spent ~9 launches per step.  Here a CG step is four
// launches that never involve the host:
//
//   L1  q = A p            spmv_sell_fin_kernel<SpAxpbyNormOp, FinAlpha>: |q|^2, |p|^2 -> alpha
//   U1  x += alpha p, r -= alpha q, partial |x|^2          (cgf_update_xr_kernel)
//   L2  s = A^T r - x      spmv_sell_fin_kernel<SpAxpbyNormOp, FinBeta>: |s|^2, |x|^2 -> beta,
//                          gamma, step count, the stopping test of cgls.h:301-305 -> S[kFcDone]
//   U2  p = s + beta p, partial |p|^2                      (cgf_update_p_kernel)
//
// every one of which starts by reading S[kFcDone] and returns at once when the loop has ended.
// The host enqueues as many steps as the previous projection took, then the closing launches
// (x += x0 with the x-half bookkeeping; y = A x with the y-half bookkeeping, whose last
// workgroup also publishes the scalar block to the host) -- which run only if S[kFcDone] is set --
// and polls ONCE per ADMM iteration; if the loop had not ended it enqueues one more step and the
// closing launches again.
#pragma once
#include <hip/hip_runtime.h>

#include "reduce.h"
#include "sell.h"
#include "vec_kernels.h"

namespace pogs_amd {
namespace {

// s_j = (A^T r)_j - shift xcg_j ; p_j = s_j ; |s|^2                        (cgls.h:236-245)
template <typename T>
struct SpCgInitOp {
  static constexpr int NS = 1;
  T shift;
  const T *x;
  T *s, *p;
  template <int N>
  __device__ __forceinline__ void row(int j, T dot, double (&acc)[N]) const {
    const T v = dot - shift * x[j];
    s[j] = v;
    p[j] = v;
    acc[0] += static_cast<double>(v) * v;
  }
  struct In { T x; };
  __device__ __forceinline__ In load(int j) const { return In{x[j]}; }
  template <int N>
  __device__ __forceinline__ void apply(int j, T dot, const In &in, double (&acc)[N]) const {
    const T v = dot - shift * in.x;
    s[j] = v;
    p[j] = v;
    acc[0] += static_cast<double>(v) * v;
  }
};

// gamma = |s_0|^2, loop state reset; and the sums of the prox step's partials (admm_pre_kernel),
// which nobody needs before the iteration's publish
struct FinCgInit {
  double *S;
  const double *pre;   // [bx + by][3]
  int bx, by;
  double eps;
  __device__ __forceinline__ bool skip() const { return false; }
  __device__ __forceinline__ void skipped() const {}
  __device__ __forceinline__ void run(const double *rec, int nrec, double *smem) const {
    double g[1], sx[3], sy[3];
    fin_sum_records<1>(rec, nrec, 1, g, smem);
    fin_sum_records<3>(pre, bx, 3, sx, smem);
    fin_sum_records<3>(pre + static_cast<size_t>(bx) * 3, by, 3, sy, smem);
    if (threadIdx.x == 0) {
      S[kFcGamma] = g[0];
      S[kFcNorms0] = g[0];
      S[kCgS2] = g[0];
      S[kCgP2] = g[0];
      S[kFcIndef] = 0.0;
      S[kFcSteps] = 0.0;
      S[kFcDone] = (sqrt(g[0]) < eps) ? 1.0 : 0.0;   // flag 1 (cgls.h:247-252): nothing to do
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        S[kGapX + k] = sx[k];
        S[kGapY + k] = sy[k];
      }
    }
  }
};

// alpha = gamma / (|q|^2 + shift |p|^2)                                    (cgls.h:262-271)
struct FinCgAlpha {
  double *S;
  const double *pp;   // partial |p|^2 of the preceding U2 (nullptr: first step, |p|^2 = |s_0|^2)
  int bp;
  double shift, eps;
  __device__ __forceinline__ bool skip() const { return S[kFcDone] != 0.0; }
  __device__ __forceinline__ void skipped() const {}
  __device__ __forceinline__ void run(const double *rec, int nrec, double *smem) const {
    double q2[1], p2[1];
    fin_sum_records<1>(rec, nrec, 1, q2, smem);
    if (pp) fin_sum_records<1>(pp, bp, 1, p2, smem);
    if (threadIdx.x == 0) {
      const double normp2 = pp ? p2[0] : S[kCgP2];
      S[kCgP2] = normp2;
      S[kCgQ2] = q2[0];
      double delta = q2[0] + shift * normp2;
      if (delta <= 0.0) S[kFcIndef] = 1.0;
      if (delta == 0.0) delta = eps;
      S[kFcDelta] = delta;
      S[kFcAlpha] = S[kFcGamma] / delta;
    }
  }
};

// beta = |s|^2 / gamma, gamma = |s|^2 (cgls.h:288-292); the stopping test (:301-305)
struct FinCgBeta {
  double *S;
  const double *px;   // partial |x|^2 of the preceding U1
  int bx;
  double tol;
  int maxit;
  __device__ __forceinline__ bool skip() const { return S[kFcDone] != 0.0; }
  __device__ __forceinline__ void skipped() const {}
  __device__ __forceinline__ void run(const double *rec, int nrec, double *smem) const {
    double g[1], x2[1];
    fin_sum_records<1>(rec, nrec, 1, g, smem);
    fin_sum_records<1>(px, bx, 1, x2, smem);
    if (threadIdx.x == 0) {
      const double g1 = S[kFcGamma];
      S[kFcGamma] = g[0];
      S[kFcBeta] = g[0] / g1;
      S[kCgS2] = g[0];
      S[kCgX2] = x2[0];
      const double steps = S[kFcSteps] + 1.0;
      S[kFcSteps] = steps;
      const double norms = sqrt(g[0]), norms0 = sqrt(S[kFcNorms0]), normx = sqrt(x2[0]);
      const bool converged = (norms <= norms0 * tol) || (normx * tol >= 1.0);
      if (converged || steps >= static_cast<double>(maxit)) S[kFcDone] = 1.0;
    }
  }
};

// (timing probe, SparseSolver::probe_spmv) a finaliser that only adds the records up
struct FinProbe {
  double *out;
  int probe_flags;
  __device__ __forceinline__ bool skip() const { return false; }
  __device__ __forceinline__ void skipped() const {}
  __device__ __forceinline__ void run(const double *rec, int nrec, double *smem) const {
    double g[1];
    fin_sum_records<1>(rec, nrec, 1, g, smem);
    if (threadIdx.x == 0) out[0] = g[0];
  }
};

// copies the scalar block to the host-mapped mirror and raises the sequence word (the body of
// publish_scalars_kernel); every thread of the workgroup calls it
__device__ __forceinline__ void fin_publish(double *S, double *host_S, unsigned long long *host_seq,
                                            unsigned long long seq) {
  __syncthreads();   // thread 0's stores into S
  const int t = threadIdx.x;
  if (t < kNumSlots) host_S[t] = __hip_atomic_load(S + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// closing launch y = A x: the y-half sums, the x-half sums left by cgf_close_x_kernel, publish.
// Runs only when the CG loop has ended; otherwise it just publishes (the host sees kFcDone == 0).
struct FinCgTail {
  double *S;
  const double *xpart;   // [bx][2]
  int bx;
  double *host_S;        // nullptr: no publish (the host copies the block itself)
  unsigned long long *host_seq;
  unsigned long long seq;
  __device__ __forceinline__ bool skip() const { return S[kFcDone] == 0.0; }
  __device__ __forceinline__ void skipped() const {
    if (host_S) fin_publish(S, host_S, host_seq, seq);
  }
  __device__ __forceinline__ void run(const double *rec, int nrec, double *smem) const {
    double sy[2], sx[2];
    fin_sum_records<2>(rec, nrec, 2, sy, smem);
    fin_sum_records<2>(xpart, bx, 2, sx, smem);
    if (threadIdx.x == 0) {
      S[kDYprev2] = sy[0];
      S[kDY12] = sy[1];
      S[kDXprev2] = sx[0];
      S[kDX12] = sx[1];
      __threadfence();
    }
    if (host_S) fin_publish(S, host_S, host_seq, seq);
  }
};

// U1: x += alpha p (n);  r -= alpha q (m);  partial |x|^2      (cgls.h:274-277, 298)
template <typename T>
__global__ void __launch_bounds__(kVecTpb) cgf_update_xr_kernel(int n, int m, const double *S, const T *p, T *x,
                                                                const T *q, T *r, double *partials, int blocks_x) {
  __shared__ double s_red[kVecTpb / 64];
  if (S[kFcDone] != 0.0) return;
  const T alpha = static_cast<T>(S[kFcAlpha]);
  const T neg_alpha = static_cast<T>(-S[kFcAlpha]);
  double acc[1] = {0.0};
  if (static_cast<int>(blockIdx.x) < blocks_x) {
    const int i = blockIdx.x * kVecTpb + threadIdx.x;
    if (i < n) {
      const T v = x[i] + alpha * p[i];
      x[i] = v;
      acc[0] = static_cast<double>(v) * v;
    }
  } else {
    const int i = (blockIdx.x - blocks_x) * kVecTpb + threadIdx.x;
    if (i < m) r[i] += neg_alpha * q[i];
    return;
  }
  dev::block_sum<1, kVecTpb>(acc, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

// U2: p = s + beta p; partial |p|^2      (cgls.h:295-296)
template <typename T>
__global__ void __launch_bounds__(kVecTpb) cgf_update_p_kernel(int n, const double *S, const T *s, T *p,
                                                               double *partials) {
  __shared__ double s_red[kVecTpb / 64];
  if (S[kFcDone] != 0.0) return;
  const T beta = static_cast<T>(S[kFcBeta]);
  const int i = blockIdx.x * kVecTpb + threadIdx.x;
  double acc[1] = {0.0};
  if (i < n) {
    const T v = s[i] + beta * p[i];
    p[i] = v;
    acc[0] = static_cast<double>(v) * v;
  }
  dev::block_sum<1, kVecTpb>(acc, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

// closing launch for the x half: x <- x + x0 (projector_cgls.cpp:75) and the element-wise
// projection tail (admm_tail_kernel): sums of (x_prev - x)^2, (x12 - x)^2, xtemp <- x0 - x
template <typename T>
__global__ void __launch_bounds__(kVecTpb) cgf_close_x_kernel(int n, const double *S, T *x, const T *xprev,
                                                              const T *x12, T *xtemp, double *partials) {
  __shared__ double s_red[2 * (kVecTpb / 64)];
  if (S[kFcDone] == 0.0) return;
  const int i = blockIdx.x * kVecTpb + threadIdx.x;
  double acc[2] = {0.0, 0.0};
  if (i < n) {
    const T x0 = xtemp[i];
    const T zn = x[i] + x0;   // the reference: x <- 1 * x0 + x (blas_axpy)
    x[i] = zn;
    const T a = xprev[i] - zn, b = x12[i] - zn;
    acc[0] = static_cast<double>(a) * a;
    acc[1] = static_cast<double>(b) * b;
    xtemp[i] = x0 - zn;
  }
  dev::block_sum<2, kVecTpb>(acc, s_red);
  if (threadIdx.x == 0) {
    partials[static_cast<size_t>(blockIdx.x) * 2 + 0] = acc[0];
    partials[static_cast<size_t>(blockIdx.x) * 2 + 1] = acc[1];
  }
}

}  // namespace
}  // namespace pogs_amd
