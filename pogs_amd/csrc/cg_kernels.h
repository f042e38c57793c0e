// CGLS vector kernels shared by the sparse and the dense CGLS projectors
// (reference: src/cpu/include/cgls.h:255-306).  The CG scalars (gamma, alpha, beta)
// live in a small device block so a CG iteration needs no host round trip to form them.
#pragma once
#include <hip/hip_runtime.h>

#include "reduce.h"
#include "vec_kernels.h"

namespace pogs_amd {
namespace {

enum CgSlot : int { kCgGamma = 0, kCgAlpha, kCgBeta, kCgDelta, kCgIndef, kCgNumSlots = 8 };

// alpha = gamma / (|q|^2 + shift |p|^2)   (cgls.h:262-271)
static __global__ void cg_alpha_kernel(double *S, double *cg, double shift, double eps) {
  const double normq2 = S[kCgQ2], normp2 = S[kCgP2];
  double delta = normq2 + shift * normp2;
  if (delta <= 0.0) cg[kCgIndef] = 1.0;
  if (delta == 0.0) delta = eps;
  cg[kCgDelta] = delta;
  cg[kCgAlpha] = cg[kCgGamma] / delta;
}
// beta = |s|^2 / gamma_prev; gamma = |s|^2   (cgls.h:288-292)
static __global__ void cg_beta_kernel(double *S, double *cg) {
  const double g1 = cg[kCgGamma];
  const double g = S[kCgS2];
  cg[kCgGamma] = g;
  cg[kCgBeta] = g / g1;
}

// x += alpha p (n);  r -= alpha q (m);  partial |x|^2      (cgls.h:274-277, 298)
// yacc (optional): y_acc = ysrc + alpha q -- the recurrence A x_new = A x_warm + sum alpha_k q_k that
// replaces the product y = A x at the end of the projection (projector_cgls.cpp:78; cg_fused.h);
// ysrc is A x_warm on the first step and yacc itself afterwards.
template <typename T>
__global__ void __launch_bounds__(kVecTpb) cg_update_xr_kernel(int n, int m, const double *cg, const T *p, T *x,
                                                               const T *q, T *r, double *partials, int blocks_x,
                                                               const T *ysrc, T *yacc) {
  __shared__ double s_red[kVecTpb / 64];
  const T alpha = static_cast<T>(cg[kCgAlpha]);
  const T neg_alpha = static_cast<T>(-cg[kCgAlpha]);
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
    if (i < m) {
      const T qi = q[i];
      r[i] += neg_alpha * qi;
      if (yacc) yacc[i] = ysrc[i] + alpha * qi;
    }
    return;   // (uniform per workgroup) only the x blocks carry a partial sum: partials has blocks_x entries
  }
  dev::block_sum<1, kVecTpb>(acc, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

// p = s + beta p; partial |p|^2      (cgls.h:295-296)
template <typename T>
__global__ void __launch_bounds__(kVecTpb) cg_update_p_kernel(int n, const double *cg, const T *s, T *p,
                                                              double *partials, bool first) {
  __shared__ double s_red[kVecTpb / 64];
  const T beta = first ? static_cast<T>(0) : static_cast<T>(cg[kCgBeta]);
  const int i = blockIdx.x * kVecTpb + threadIdx.x;
  double acc[1] = {0.0};
  if (i < n) {
    const T v = first ? s[i] : s[i] + beta * p[i];
    p[i] = v;
    acc[0] = static_cast<double>(v) * v;
  }
  dev::block_sum<1, kVecTpb>(acc, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

// out = a - b, partial |out|^2
template <typename T>
__global__ void __launch_bounds__(kVecTpb) sub_norm_kernel(int n, const T *a, const T *b, T *out, double *partials) {
  __shared__ double s_red[kVecTpb / 64];
  const int i = blockIdx.x * kVecTpb + threadIdx.x;
  double acc[1] = {0.0};
  if (i < n) {
    const T v = a[i] - b[i];
    out[i] = v;
    acc[0] = static_cast<double>(v) * v;
  }
  dev::block_sum<1, kVecTpb>(acc, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

static __global__ void set_gamma_kernel(const double *S, double *cg) {
  cg[kCgGamma] = S[kCgS2];
  cg[kCgIndef] = 0.0;
}

}  // namespace
}  // namespace pogs_amd
