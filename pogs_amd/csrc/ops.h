// Per-row / per-column functors plugged into stream_rows_kernel / reduce_cols_kernel.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace pogs_amd {

// ---- row functors ---------------------------------------------------------

// y[i] = alpha * dot + beta * y[i]   (Matrix::Mul 'n', matrix_dense.cpp:93-113)
template <typename T>
struct GemvNOp {
  static constexpr int NS = 0;
  T alpha, beta;
  T *y;
  template <int N>
  __device__ __forceinline__ T row(int i, T dot, double (&)[N]) const {
    T v = alpha * dot;
    if (beta != static_cast<T>(0)) v += beta * y[i];
    y[i] = v;
    return v;
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};

// ACC-only: u_i = alpha * x[i]   (Matrix::Mul 't')
template <typename T>
struct GemvTOp {
  static constexpr int NS = 0;
  T alpha;
  const T *x;
  template <int N>
  __device__ __forceinline__ T row(int, T, double (&)[N]) const { return 0; }
  __device__ __forceinline__ T u(int i) const { return alpha * x[i]; }
};

// ACC-only with u = 1: column sums (of squares) -- first Sinkhorn-Knopp half step
// with d = 1 (equil_helper.h:146-151).
template <typename T>
struct OnesOp {
  static constexpr int NS = 0;
  template <int N>
  __device__ __forceinline__ T row(int, T, double (&)[N]) const { return 0; }
  __device__ __forceinline__ T u(int) const { return 1; }
};

// Sinkhorn-Knopp row step: d_i = n / ((A.^2 e)_i + c)  (equil_helper.h:157-162);
// returns d_i so the same pass accumulates (A.^2)^T d for the next column step.
template <typename T>
struct SkRowOp {
  static constexpr int NS = 0;
  T nn, c;
  T *d;
  template <int N>
  __device__ __forceinline__ T row(int i, T dot, double (&)[N]) const {
    const T v = nn / (dot + c);
    d[i] = v;
    return v;
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};

// Power iteration: Sx_i = (A x)_i, accumulates |Sx|^2, returns Sx_i so the pass
// also accumulates A^T Sx (equil_helper.h:121-123).
template <typename T>
struct PowerRowOp {
  static constexpr int NS = 1;
  template <int N>
  __device__ __forceinline__ T row(int, T dot, double (&s)[N]) const {
    s[0] += static_cast<double>(dot) * dot;
    return dot;
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};

// Tail of the projection fused with the residual bookkeeping and the dual
// update, for one half (x or y) of z (pogs.cpp:342-348, 397-399):
//   znew_i = dot;  s0 += (zprev_i - znew_i)^2;  s1 += (z12_i - znew_i)^2;
//   ztemp_i <- ztemp_i - znew_i   ( = zt + alpha z12 + (1-alpha) zprev - znew )
template <typename T>
struct ProjTailOp {
  static constexpr int NS = 2;
  T *znew;
  const T *zprev, *z12;
  T *ztemp;
  template <int N>
  __device__ __forceinline__ T row(int i, T dot, double (&s)[N]) const {
    znew[i] = dot;
    const T a = zprev[i] - dot, b = z12[i] - dot;
    s[0] += static_cast<double>(a) * a;
    s[1] += static_cast<double>(b) * b;
    ztemp[i] -= dot;
    return dot;
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};

// Exact residuals in one pass (pogs.cpp:352-376):
//   r_i = (A x12)_i - y12_i, s0 += r_i^2;  returns y12_i + c yt_i - yprev_i, whose
//   A^T-image the pass accumulates.
template <typename T>
struct ExactRowOp {
  static constexpr int NS = 1;
  const T *y12, *yt, *yprev;
  T zt_scale;
  template <int N>
  __device__ __forceinline__ T row(int i, T dot, double (&s)[N]) const {
    const T r = dot - y12[i];
    s[0] += static_cast<double>(r) * r;
    return y12[i] + zt_scale * yt[i] - yprev[i];
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};

// out_i = dot - yin_i   (first step of the m <= n projection: A x0 - y0,
// projector_direct_dense.cpp:129)
template <typename T>
struct ResidOp {
  static constexpr int NS = 0;
  const T *yin;
  T *out;
  template <int N>
  __device__ __forceinline__ T row(int i, T dot, double (&)[N]) const {
    const T v = dot - yin[i];
    out[i] = v;
    return v;
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};

// m <= n projection, y half: t_i = dot, ynew_i = ytemp_i + t_i (projector_direct_dense.cpp:134),
// with the same residual sums / dual update as ProjTailOp.
template <typename T>
struct ProjTailAddOp {
  static constexpr int NS = 2;
  T *znew;
  const T *zprev, *z12;
  T *ztemp;
  T *tout;
  template <int N>
  __device__ __forceinline__ T row(int i, T dot, double (&s)[N]) const {
    const T zn = ztemp[i] + dot;
    tout[i] = dot;
    znew[i] = zn;
    const T a = zprev[i] - zn, b = z12[i] - zn;
    s[0] += static_cast<double>(a) * a;
    s[1] += static_cast<double>(b) * b;
    ztemp[i] -= zn;
    return dot;
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};

// ---- column functors ------------------------------------------------------

// out[j] = alpha * total + beta * out[j]
template <typename T>
struct StoreColOp {
  static constexpr int NS = 0;
  T alpha, beta;
  T *out;
  int n;  // columns >= n are padding: forced to zero
  template <int N>
  __device__ __forceinline__ void col(int j, T total, double (&)[N]) const {
    T v = alpha * total;
    if (beta != static_cast<T>(0)) v += beta * out[j];
    out[j] = (j < n) ? v : static_cast<T>(0);
  }
};

// Sinkhorn-Knopp column step: e_j = m / ((A.^2)^T d)_j + c)  (equil_helper.h:149-155)
template <typename T>
struct SkColOp {
  static constexpr int NS = 0;
  T mm, c;
  T *e;
  int n;
  template <int N>
  __device__ __forceinline__ void col(int j, T total, double (&)[N]) const {
    e[j] = (j < n) ? mm / (total + c) : static_cast<T>(0);
  }
};

// Power iteration: x'_j = total, accumulates |x'|^2.
template <typename T>
struct PowerColOp {
  static constexpr int NS = 1;
  T *x;
  int n;
  template <int N>
  __device__ __forceinline__ void col(int j, T total, double (&s)[N]) const {
    const T v = (j < n) ? total : static_cast<T>(0);
    x[j] = v;
    s[0] += static_cast<double>(v) * v;
  }
};

// Exact dual residual: s_j = total + x12_j + c xt_j - xprev_j; accumulates |s|^2.
template <typename T>
struct ExactColOp {
  static constexpr int NS = 1;
  const T *x12, *xt, *xprev;
  T zt_scale;
  int n;
  template <int N>
  __device__ __forceinline__ void col(int j, T total, double (&s)[N]) const {
    if (j < n) {
      const T v = total + x12[j] + zt_scale * xt[j] - xprev[j];
      s[0] += static_cast<double>(v) * v;
    }
  }
};

// m <= n projection, x half: xnew_j = xtemp_j - (A^T t)_j (projector_direct_dense.cpp:132-133)
// with the residual sums / dual update of ProjTailOp.
template <typename T>
struct ProjTailColOp {
  static constexpr int NS = 2;
  T *znew;
  const T *zprev, *z12;
  T *ztemp;
  int n;
  template <int N>
  __device__ __forceinline__ void col(int j, T total, double (&s)[N]) const {
    if (j < n) {
      const T zn = ztemp[j] - total;
      znew[j] = zn;
      const T a = zprev[j] - zn, b = z12[j] - zn;
      s[0] += static_cast<double>(a) * a;
      s[1] += static_cast<double>(b) * b;
      ztemp[j] -= zn;
    }
  }
};

}  // namespace pogs_amd
