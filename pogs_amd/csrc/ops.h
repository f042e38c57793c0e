// Per-row / per-column functors plugged into stream_rows_kernel / reduce_cols_kernel.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"
#include "prox.h"

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

// ---- transposed storage (m <= n: the solver keeps T = A^T, rows of length m) ----------------
// A x is then a column-sum pass over T (coefficient x_j per stored row j) and A^T u a
// row-dot pass, so the functors trade places: these adaptors let a row functor finish a
// column-sum pass and a column functor finish a row-dot pass.
template <typename T, typename Op>
struct RowAsColOp {
  static constexpr int NS = Op::NS;
  Op op;
  int n;   // valid entries (padding columns are skipped)
  template <int N>
  __device__ __forceinline__ void col(int j, T total, double (&s)[N]) const {
    if (j < n) (void)op.row(j, total, s);
  }
};
template <typename T, typename Op>
struct ColAsRowOp {
  static constexpr int NS = Op::NS;
  Op op;
  template <int N>
  __device__ __forceinline__ T row(int i, T dot, double (&s)[N]) const {
    op.col(i, dot, s);
    return 0;
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};

// ACC-only: u_j = (x_j + add_j) * sc, sc = 1 / sqrt(*x_nrm2) when given (lazy normalisation)
template <typename T>
struct VecCoefOp {
  static constexpr int NS = 0;
  const T *x, *add;
  const double *x_nrm2;
  template <int N>
  __device__ __forceinline__ T row(int, T, double (&)[N]) const { return 0; }
  __device__ __forceinline__ T u(int j) const {
    T v = add ? x[j] + add[j] : x[j];
    if (x_nrm2) v *= static_cast<T>(1.0 / sqrt(*x_nrm2));
    return v;
  }
};

// row-dot pass: out_j = dot, accumulates |out|^2
template <typename T>
struct StoreNormRowOp {
  static constexpr int NS = 1;
  T *out;
  template <int N>
  __device__ __forceinline__ T row(int j, T dot, double (&s)[N]) const {
    out[j] = dot;
    dev::prod_acc(s[0], dot, dot);
    return dot;
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};

// Exact residuals on T in one pass (pogs.cpp:352-376): the stored row j sees dot = (A^T u)_j,
// u = y12 + c yt - yprev given as a vector; dual residual s_j = dot + x12_j + c xt_j - xprev_j,
// and x12_j is returned so that the same pass accumulates A x12 for the primal residual.
template <typename T>
struct ExactTRowOp {
  static constexpr int NS = 1;
  const T *x12, *xt, *xprev;
  T zt_scale;
  template <int N>
  __device__ __forceinline__ T row(int j, T dot, double (&s)[N]) const {
    const T v = dot + x12[j] + zt_scale * xt[j] - xprev[j];
    dev::prod_acc(s[0], v, v);
    return x12[j];
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};
template <typename T>
struct ExactTColOp {   // r_i = (A x12)_i - y12_i
  static constexpr int NS = 1;
  const T *y12;
  int m;
  template <int N>
  __device__ __forceinline__ void col(int i, T total, double (&s)[N]) const {
    if (i < m) {
      const T r = total - y12[i];
      dev::prod_acc(s[0], r, r);
    }
  }
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
    dev::prod_acc(s[0], dot, dot);
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
    dev::prod_acc(s[0], a, a);
    dev::prod_acc(s[1], b, b);
    ztemp[i] -= dot;
    return dot;
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};

// t_i = dot, handed straight back as the coefficient of row i for the column sums of the same
// pass: x = W^T (W r) in one sweep over the triangular factor inverse.
template <typename T>
struct IdentRowOp {
  static constexpr int NS = 0;
  template <int N>
  __device__ __forceinline__ T row(int, T dot, double (&)[N]) const { return dot; }
  __device__ __forceinline__ T u(int) const { return 0; }
};

// ProjTailOp as the second stage of a column-sum pass (znew_j = total).
template <typename T>
struct ProjTailSumColOp {
  static constexpr int NS = 2;
  T *znew;
  const T *zprev, *z12;
  T *ztemp;
  int n;
  template <int N>
  __device__ __forceinline__ void col(int j, T total, double (&s)[N]) const {
    if (j >= n) return;
    znew[j] = total;
    const T a = zprev[j] - total, b = z12[j] - total;
    dev::prod_acc(s[0], a, a);
    dev::prod_acc(s[1], b, b);
    ztemp[j] -= total;
  }
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
    dev::prod_acc(s[0], r, r);
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
    dev::prod_acc(s[0], a, a);
    dev::prod_acc(s[1], b, b);
    ztemp[i] -= zn;
    return dot;
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};

// ProjTailAddOp as the second stage of a column-sum pass.
template <typename T>
struct ProjTailAddColOp {
  static constexpr int NS = 2;
  T *znew;
  const T *zprev, *z12;
  T *ztemp;
  T *tout;
  int n;
  template <int N>
  __device__ __forceinline__ void col(int j, T total, double (&s)[N]) const {
    if (j >= n) return;
    const T zn = ztemp[j] + total;
    tout[j] = total;
    znew[j] = zn;
    const T a = zprev[j] - zn, b = z12[j] - zn;
    dev::prod_acc(s[0], a, a);
    dev::prod_acc(s[1], b, b);
    ztemp[j] -= zn;
  }
};

// Power iteration on the Gram matrix stored as its lower triangle (strict upper part
// zero): one DOT+ACC pass gives L x (dot) and L^T x (column sums); G x = L x + L^T x - D x.
// x is held un-normalised; sc = 1/|x| comes from the device scalar written by the
// previous iteration.
template <typename T>
struct SymRowOp {
  static constexpr int NS = 0;
  const T *G;
  size_t ld;
  const T *x;
  const double *x_nrm2;   // nullable: first iteration uses x as is
  T *lowdot;              // out: (L x)_i - G_ii x_i
  template <int N>
  __device__ __forceinline__ T row(int i, T dot, double (&)[N]) const {
    const T sc = x_nrm2 ? static_cast<T>(1.0 / sqrt(*x_nrm2)) : static_cast<T>(1);
    const T xi = x[i] * sc;
    lowdot[i] = dot - G[static_cast<size_t>(i) * ld + i] * xi;
    return xi;
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};

// CGLS on a dense operator: q_i = (A p)_i, accumulates |q|^2 (cgls.h:257-263).
template <typename T>
struct CgQRowOp {
  static constexpr int NS = 1;
  T *q;
  template <int N>
  __device__ __forceinline__ T row(int i, T dot, double (&s)[N]) const {
    q[i] = dot;
    dev::prod_acc(s[0], dot, dot);
    return dot;
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};

// out_i = yin_i - dot   (b = y0 - A x0, r = b - A x; projector_cgls.cpp:68, cgls.h:229-233)
template <typename T>
struct SubDotOp {
  static constexpr int NS = 0;
  const T *yin;
  T *out;
  template <int N>
  __device__ __forceinline__ T row(int i, T dot, double (&)[N]) const {
    const T v = yin[i] - dot;
    out[i] = v;
    return v;
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};

// ---- functors of the one-pass iteration (stream_rows2_kernel) ---------------

// Column-sum-only pass before the projection when the previous pass could not
// speculate (rho changed / cold start): u0 = yhat_k (A^T yhat, projector_direct_dense.cpp:123),
// u1 = y12_k + c yt_k - yprev_k (exact dual residual, pogs.cpp:366-369).
template <typename T>
struct PreAccOp {
  static constexpr int NS = 0;
  struct Pre {};
  const T *ytemp, *y12, *yt, *ycur;
  T zt_scale;
  __device__ __forceinline__ Pre prefetch(int) const { return Pre{}; }
  template <int N, int ND, int NA>
  __device__ __forceinline__ void row(int, const Pre &, const T (&)[ND], double (&)[N], T (&)[NA]) const {}
  template <int NA>
  __device__ __forceinline__ void uonly(int i, T (&u)[NA]) const {
    u[0] = ytemp[i];
    u[1] = y12[i] + zt_scale * yt[i] - ycur[i];
  }
};

// Column-sum-only pass of the transposed storage (m <= n) when the previous pass could not
// speculate: u0 = xhat_k (A xhat, projector_direct_dense.cpp:130), u1 = x12_k (exact primal residual).
template <typename T>
struct PreAcc2Op {
  static constexpr int NS = 0;
  struct Pre {};
  const T *v0, *v1;
  __device__ __forceinline__ Pre prefetch(int) const { return Pre{}; }
  template <int N, int ND, int NA>
  __device__ __forceinline__ void row(int, const Pre &, const T (&)[ND], double (&)[N], T (&)[NA]) const {}
  template <int NA>
  __device__ __forceinline__ void uonly(int j, T (&u)[NA]) const {
    u[0] = v0[j];
    u[1] = v1[j];
  }
};

// out_i = total - yin_i (rhs of the m <= n projection: A xhat - yhat), padding forced to zero
template <typename T>
struct ResidColOp {
  static constexpr int NS = 0;
  const T *yin;
  T *out;
  int m;
  template <int N>
  __device__ __forceinline__ void col(int i, T total, double (&)[N]) const {
    out[i] = (i < m) ? total - yin[i] : static_cast<T>(0);
  }
};

// The single pass of iteration k (after x_{k+1} is known):
//   y_{k+1} = dot0; residual sums and dual update as ProjTailOp (pogs.cpp:342-348,397-399);
//   exact primal residual r_i = dot1 - y12_i with dot1 = (A x12_k)_i (pogs.cpp:353-364);
//   then, assuming rho stays, the y half of iteration k+1's prox / gap sums /
//   over-relaxation (pogs.cpp:257-278) and the two column-sum inputs of k+1.
// Scalars: [ |yprev-y|^2, |y12-y|^2, |A x12 - y12|^2, sum w h, |w|^2, |h|^2 ].
// LOGISTIC = true: every f_i is kLogistic (solve_logistic); the guarded-Newton prox
// (prox_lib.h:131-170) is inlined on its own so the register footprint stays small.
//
// XSIDE = true is the mirror image for m <= n on the transposed storage: the stored rows are
// the x coordinates, dot0 = (A^T t)_j gives x_{k+1,j} = xhat_j - dot0, dot1 = (A^T u)_j the
// exact DUAL residual dot1 + x12_j + c xt_j - x_j (pogs.cpp:366-373), the function view is g,
// and the two column sums are A xhat_{k+1} and A x12_{k+1}.
template <typename T, bool LOGISTIC = false, bool XSIDE = false>
struct FusedIterOp {
  static constexpr int NS = 6;
  struct Pre {
    T ycur, y12, ytemp, a, b, c, d, e, zt;
    int h;
  };
  T *ynew;
  const T *ycur, *y12;
  T *ytemp;        // in: yhat_k, out: ytilde_{k+1}
  FnView<T> f;     // scaled f
  T rho, alpha;    // rho: the PREDICTED rho of iteration k+1
  T zs;            // predicted lazy scale of ytilde_{k+1} (rho_k / rho_{k+1})
  T *y12s, *ytemps;  // speculative y12_{k+1}, yhat_{k+1}
  const T *zt_old = nullptr;   // XSIDE: xtilde_k and its lazy scale, for the exact dual residual
  T c_old = 0;
  __device__ __forceinline__ Pre prefetch(int i) const {
    Pre p;
    p.ycur = ycur[i]; p.y12 = y12[i]; p.ytemp = ytemp[i];
    p.zt = XSIDE ? zt_old[i] : static_cast<T>(0);
    p.h = f.h[i]; p.a = f.a[i]; p.b = f.b[i]; p.c = f.c[i]; p.d = f.d[i]; p.e = f.e[i];
    return p;
  }
  template <int N, int ND, int NA>
  __device__ __forceinline__ void row(int i, const Pre &p, const T (&dot)[ND], double (&s)[N], T (&u)[NA]) const {
    const T yn = XSIDE ? p.ytemp - dot[0] : dot[0];
    const T h0 = p.y12;
    ynew[i] = yn;
    const T a = p.ycur - yn, b = h0 - yn;
    dev::prod_acc(s[0], a, a);
    dev::prod_acc(s[1], b, b);
    if constexpr (ND > 1) {   // (ND = 1: the pass without the exact residuals, dense_solver.h: lean iterations)
      const T r = XSIDE ? dot[1] + h0 + c_old * p.zt - p.ycur : dot[1] - h0;
      dev::prod_acc(s[2], r, r);
    }
    const T ztn = p.ytemp - yn;
    ytemp[i] = ztn;
    const T zts = zs * ztn;
    const T v = yn - zts;
    T h;
    if (LOGISTIC) {
      // ProxEval wrapper (prox_lib.h:207-230) around ProxLogistic
      const T vv = p.a * (v * rho - p.d) / (p.e + rho) - p.b;
      const T rr = (p.e + rho) / (p.c * p.a * p.a);
      h = (dev::ProxLogistic(vv, rr) + p.b) / p.a;
    } else {
      h = dev::ProxEvalCheap(p.h, p.a, p.b, p.c, p.d, p.e, v, rho);
    }
    const T w = v - h;
    y12s[i] = h;
    const T yh = zts + alpha * h + (static_cast<T>(1) - alpha) * yn;
    ytemps[i] = yh;
    dev::prod_acc(s[3], w, h);
    dev::prod_acc(s[4], w, w);
    dev::prod_acc(s[5], h, h);
    u[0] = yh;
    if constexpr (NA > 1) u[1] = XSIDE ? h : h + zts - yn;
  }
  template <int NA>
  __device__ __forceinline__ void uonly(int, T (&)[NA]) const {}
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
// Probe: the ratio new / old of every entry is summed (s[0], its mean is the common growth factor
// of this iteration), and an entry whose ratio is further than tol (relative) from r_ref -- the
// mean of the previous iteration -- stamps *mark with the number of the pass (every writer stores
// the same value).  A pass that leaves no stamp moved the whole vector by one common factor.
template <typename T>
struct SkColOp {
  static constexpr int NS = 1;
  T mm, c;
  T *e;
  int n;
  double *mark = nullptr;
  double stamp = 0;
  T tol = 0;
  T r_ref = 0;
  template <int N>
  __device__ __forceinline__ void col(int j, T total, double (&s)[N]) const {
    const T v = (j < n) ? mm / (total + c) : static_cast<T>(0);
    if (j < n) {
      const T old = e[j];
      const T r = old > static_cast<T>(0) ? v / old : static_cast<T>(0);
      s[0] += static_cast<double>(r);
      if (mark && !(fabs(r - r_ref) <= tol * r_ref)) *mark = stamp;
    }
    e[j] = v;
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
    dev::prod_acc(s[0], v, v);
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
      dev::prod_acc(s[0], v, v);
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
      dev::prod_acc(s[0], a, a);
      dev::prod_acc(s[1], b, b);
      ztemp[j] -= zn;
    }
  }
};

// x'_j = (L^T x)_j + lowdot_j = (G x)_j; accumulates |x'|^2 and x^T G x (= |A x|^2).
template <typename T>
struct SymColOp {
  static constexpr int NS = 2;
  const T *lowdot, *x;
  const double *x_nrm2;
  T *xnext;
  int n;
  template <int N>
  __device__ __forceinline__ void col(int j, T total, double (&s)[N]) const {
    if (j < n) {
      const T sc = x_nrm2 ? static_cast<T>(1.0 / sqrt(*x_nrm2)) : static_cast<T>(1);
      const T v = total + lowdot[j];
      xnext[j] = v;
      dev::prod_acc(s[0], v, v);
      dev::prod_acc(s[1], x[j] * sc, v);
    } else {
      xnext[j] = 0;
    }
  }
};

// CGLS: s_j = (A^T r)_j - shift * x_j, accumulates |s|^2 (cgls.h:281-285).
template <typename T>
struct CgSColOp {
  static constexpr int NS = 1;
  const T *x;
  T shift;
  T *sout;
  int n;
  template <int N>
  __device__ __forceinline__ void col(int j, T total, double (&s)[N]) const {
    if (j < n) {
      const T v = total - shift * x[j];
      sout[j] = v;
      dev::prod_acc(s[0], v, v);
    } else {
      sout[j] = 0;
    }
  }
};

}  // namespace pogs_amd
