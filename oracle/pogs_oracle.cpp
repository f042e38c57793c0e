// =============================================================================
// oracle/pogs_oracle.cpp  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A CPU restatement of the reference POGS graph-form ADMM hot path
// (foges/pogs, /root/reference).  It exists so that tests/, bench.py's
// `cpu_baseline` leg and __graft_entry__.smoke() have something to check the
// HIP engine against on a box where /root/reference does not exist.  Nothing
// under pogs_amd/ may import, link or call this file.
//
// Pinning: validated in the build container against the compiled reference
// (oracle/_ref/libpogs_cpu.so built by oracle/Makefile from the reference's
// own sources + MKL) and against the golden fixtures in tests/golden/ that the
// compiled reference produced (tests/golden/make_golden.py).  See
// tests/test_oracle_golden.py and tests/test_oracle_vs_ref.py.
//
// The vendor BLAS/LAPACK arithmetic the reference delegates to
// (CMakeLists.txt:72-73, un-pinned) is restated as plain loops; results agree
// with the reference to rounding, not bitwise (summation order differs).
//
// Every function cites the reference file:line it follows.  Paths are relative
// to /root/reference/.
// =============================================================================
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "pogs_oracle.h"

namespace {

double now_s() {
  return std::chrono::duration<double>(
             std::chrono::steady_clock::now().time_since_epoch()).count();
}

// -----------------------------------------------------------------------------
// Collective hook for the row-sharded variant (no counterpart in the reference,
// which is single-process: SURVEY.md section 8(e)).  Single process: identity.
// -----------------------------------------------------------------------------
struct Comm {
  oracle_allreduce_fn fn = nullptr;
  void *ctx = nullptr;
  void sum(double *buf, size_t count) const {
    if (fn) fn(ctx, buf, count);
  }
  template <typename T>
  void sum_vec(T *v, size_t count) const {
    if (!fn) return;
    std::vector<double> tmp(v, v + count);
    fn(ctx, tmp.data(), count);
    for (size_t i = 0; i < count; ++i) v[i] = static_cast<T>(tmp[i]);
  }
  double sum1(double v) const { sum(&v, 1); return v; }
};

// -----------------------------------------------------------------------------
// prox_tools.h scalar helpers (src/include/prox_tools.h:12-95)
// -----------------------------------------------------------------------------
inline double Abs(double x) { return fabs(x); }
inline float Abs(float x) { return fabsf(x); }
inline double Exp(double x) { return exp(x); }
inline float Exp(float x) { return expf(x); }
inline double Log(double x) { return log(x); }
inline float Log(float x) { return logf(x); }
inline double Max(double x, double y) { return fmax(x, y); }
inline float Max(float x, float y) { return fmaxf(x, y); }
inline double Min(double x, double y) { return fmin(x, y); }
inline float Min(float x, float y) { return fminf(x, y); }
inline double Sqrt(double x) { return sqrt(x); }
inline float Sqrt(float x) { return sqrtf(x); }
inline double Pow(double x, double y) { return pow(x, y); }
inline float Pow(float x, float y) { return powf(x, y); }
inline double Acos(double x) { return acos(x); }
inline float Acos(float x) { return acosf(x); }
inline double Cos(double x) { return cos(x); }
inline float Cos(float x) { return cosf(x); }
template <typename T> inline T Epsilon();                 // prox_tools.h:49-54
template <> inline double Epsilon<double>() { return 4e-16; }
template <> inline float Epsilon<float>() { return 1e-7f; }
template <typename T> inline T Tol();                     // prox_tools.h:57-62
template <> inline double Tol<double>() { return 1e-10; }
template <> inline float Tol<float>() { return 1e-5f; }
template <typename T> inline T MaxPos(T x) { return Max(static_cast<T>(0), x); }
template <typename T> inline T MaxNeg(T x) { return Max(static_cast<T>(0), -x); }
template <typename T> inline T Sign(T x) { return x >= 0 ? 1 : -1; }

// LambertW(exp(x)), prox_tools.h:98-129.
template <typename T>
inline T LambertWExp(T x) {
  T w;
  if (x > static_cast<T>(100)) {
    T log_x = Log(x);
    return static_cast<T>(-0.36962844) + x - static_cast<T>(0.97284858) * log_x +
           static_cast<T>(1.3437973) / log_x;
  } else if (x < static_cast<T>(0)) {
    T p = Sqrt(static_cast<T>(2.0) * (Exp(x + static_cast<T>(1)) + static_cast<T>(1)));
    w = static_cast<T>(-1.0) +
        p * (static_cast<T>(1.0) +
             p * (static_cast<T>(-1.0 / 3.0) + p * static_cast<T>(11.0 / 72.0)));
  } else {
    w = x;
  }
  if (x > static_cast<T>(1.098612288668110)) w -= Log(w);
  for (unsigned int i = 0u; i < 10u; i++) {
    T e = Exp(w);
    T t = w * e - Exp(x);
    T p = w + static_cast<T>(1.);
    t /= e * p - static_cast<T>(0.5) * (p + static_cast<T>(1.0)) * t / p;
    w -= t;
    if (Abs(t) < Epsilon<T>() * (static_cast<T>(1) + Abs(w))) break;
  }
  return w;
}

// Single positive root of x^3 + p x^2 + q x + r, prox_tools.h:134-149.
template <typename T>
inline T CubicSolve(T p, T q, T r) {
  T s = p / 3, s2 = s * s, s3 = s2 * s;
  T a = -s2 + q / 3;
  T b = s3 - s * q / 2 + r / 2;
  T a3 = a * a * a;
  T b2 = b * b;
  if (a3 + b2 >= 0) {
    T A = Pow(Sqrt(a3 + b2) - b, static_cast<T>(1) / 3);
    return -s - a / A + A;
  } else {
    T A = Sqrt(-a3);
    T B = Acos(-b / A);
    T C = Pow(A, static_cast<T>(1) / 3);
    return -s + (C - a / C) * Cos(B / 3);
  }
}

// -----------------------------------------------------------------------------
// Function object + prox library (src/include/prox_lib.h)
// -----------------------------------------------------------------------------
enum Fn { kAbs, kExp, kHuber, kIdentity, kIndBox01, kIndEq0, kIndGe0, kIndLe0,
          kLogistic, kMaxNeg0, kMaxPos0, kNegEntr, kNegLog, kRecipr, kSquare,
          kZero };                                        // prox_lib.h:23-38

template <typename T>
struct FunctionObj {                                      // prox_lib.h:42-70
  int h;
  T a, b, c, d, e;
};

template <typename T>
std::vector<FunctionObj<T>> make_objs(size_t n, const T *a, const T *b, const T *c,
                                      const T *d, const T *e, const int *h) {
  std::vector<FunctionObj<T>> v(n);
  for (size_t i = 0; i < n; ++i) {
    // CheckConsts clamps c and e to be non-negative (prox_lib.h:62-69).
    v[i] = {h[i], a[i], b[i], std::max(c[i], static_cast<T>(0)), d[i],
            std::max(e[i], static_cast<T>(0))};
  }
  return v;
}

template <typename T> inline T ProxAbs(T v, T rho) {      // prox_lib.h:83-85
  return MaxPos(v - 1 / rho) - MaxNeg(v + 1 / rho);
}
template <typename T> inline T ProxNegEntr(T v, T rho) {  // prox_lib.h:88-94
  return static_cast<T>(LambertWExp<double>(
             static_cast<double>((rho * v - 1) + Log(rho)))) / rho;
}
template <typename T> inline T ProxExp(T v, T rho) {      // prox_lib.h:97-100
  return v - static_cast<T>(LambertWExp<double>(static_cast<double>(v - Log(rho))));
}
template <typename T> inline T ProxHuber(T v, T rho) {    // prox_lib.h:102-104
  return Abs(v) < 1 + 1 / rho ? v * rho / (1 + rho) : v - Sign(v) / rho;
}
template <typename T> inline T ProxIdentity(T v, T rho) { return v - 1 / rho; }
template <typename T> inline T ProxIndBox01(T v, T) { return v <= 0 ? 0 : v >= 1 ? 1 : v; }
template <typename T> inline T ProxIndEq0(T, T) { return 0; }
template <typename T> inline T ProxIndGe0(T v, T) { return v <= 0 ? 0 : v; }
template <typename T> inline T ProxIndLe0(T v, T) { return v >= 0 ? 0 : v; }
template <typename T> inline T ProxLogistic(T v, T rho) { // prox_lib.h:131-170
  T x;
  if (v < static_cast<T>(-2.5))
    x = v;
  else if (v > static_cast<T>(2.5) + 1 / rho)
    x = v - 1 / rho;
  else
    x = (rho * v - static_cast<T>(0.5)) / (static_cast<T>(0.2) + rho);
  T l = v - 1 / rho, u = v;
  for (unsigned int i = 0; i < 5; ++i) {
    T inv_ex = 1 / (1 + Exp(-x));
    T f = inv_ex + rho * (x - v);
    T g = inv_ex * (1 - inv_ex) + rho;
    if (f < 0) l = x; else u = x;
    x = x - f / g;
    x = Min(x, u);
    x = Max(x, l);
  }
  for (unsigned int i = 0; u - l > Tol<T>() && i < 100; ++i) {
    T g_rho = 1 / (rho * (1 + Exp(-x))) + (x - v);
    if (g_rho > 0) {
      l = Max(l, x - g_rho);
      u = x;
    } else {
      u = Min(u, x - g_rho);
      l = x;
    }
    x = (u + l) / 2;
  }
  return x;
}
template <typename T> inline T ProxMaxNeg0(T v, T rho) {  // prox_lib.h:173-176
  T z = v >= 0 ? v : 0;
  return v + 1 / rho <= 0 ? v + 1 / rho : z;
}
template <typename T> inline T ProxMaxPos0(T v, T rho) {  // prox_lib.h:179-182
  T z = v <= 0 ? v : 0;
  return v >= 1 / rho ? v - 1 / rho : z;
}
template <typename T> inline T ProxNegLog(T v, T rho) {   // prox_lib.h:185-187
  return (v + Sqrt(v * v + 4 / rho)) / 2;
}
template <typename T> inline T ProxRecipr(T v, T rho) {   // prox_lib.h:190-193
  v = Max(v, static_cast<T>(0));
  return CubicSolve(-v, static_cast<T>(0), -1 / rho);
}
template <typename T> inline T ProxSquare(T v, T rho) { return rho * v / (1 + rho); }
template <typename T> inline T ProxZero(T v, T) { return v; }

template <typename T>
inline T ProxEval(const FunctionObj<T> &f, T v, T rho) {  // prox_lib.h:207-230
  const T a = f.a, b = f.b, c = f.c, d = f.d, e = f.e;
  v = a * (v * rho - d) / (e + rho) - b;
  rho = (e + rho) / (c * a * a);
  switch (f.h) {
    case kAbs: v = ProxAbs(v, rho); break;
    case kNegEntr: v = ProxNegEntr(v, rho); break;
    case kExp: v = ProxExp(v, rho); break;
    case kHuber: v = ProxHuber(v, rho); break;
    case kIdentity: v = ProxIdentity(v, rho); break;
    case kIndBox01: v = ProxIndBox01(v, rho); break;
    case kIndEq0: v = ProxIndEq0(v, rho); break;
    case kIndGe0: v = ProxIndGe0(v, rho); break;
    case kIndLe0: v = ProxIndLe0(v, rho); break;
    case kLogistic: v = ProxLogistic(v, rho); break;
    case kMaxNeg0: v = ProxMaxNeg0(v, rho); break;
    case kMaxPos0: v = ProxMaxPos0(v, rho); break;
    case kNegLog: v = ProxNegLog(v, rho); break;
    case kRecipr: v = ProxRecipr(v, rho); break;
    case kSquare: v = ProxSquare(v, rho); break;
    case kZero: default: v = ProxZero(v, rho); break;
  }
  return (v + b) / a;
}

template <typename T>
inline T FuncEval(const FunctionObj<T> &f, T x) {         // prox_lib.h:326-349
  T dx = f.d * x;
  T ex = f.e * x * x / 2;
  x = f.a * x - f.b;
  switch (f.h) {
    case kAbs: x = Abs(x); break;
    case kNegEntr: x = x <= 0 ? 0 : x * Log(x); break;
    case kExp: x = Exp(x); break;
    case kHuber: {
      T xabs = Abs(x);
      x = xabs < static_cast<T>(1) ? xabs * xabs / 2 : xabs - static_cast<T>(0.5);
      break;
    }
    case kIdentity: break;
    case kIndBox01: case kIndEq0: case kIndGe0: case kIndLe0: x = 0; break;
    case kLogistic: x = Log(1 + Exp(x)); break;
    case kMaxNeg0: x = MaxNeg(x); break;
    case kMaxPos0: x = MaxPos(x); break;
    case kNegLog: x = -Log(Max(static_cast<T>(0), x)); break;
    case kRecipr: x = 1 / Max(static_cast<T>(0), x); break;
    case kSquare: x = x * x / 2; break;
    case kZero: default: x = 0; break;
  }
  return f.c * x + dx + ex;
}

// Projection onto the subdifferential (prox_lib.h:359-466 per function, :468-493 with the
// affine composition): restated case by case as the reference lists them.
template <typename T>
inline T ProjSubgradEval(const FunctionObj<T> &f, T v, T x) {
  const T a = f.a, b = f.b, c = f.c, d = f.d, e = f.e;
  if (a == static_cast<T>(0) || c == static_cast<T>(0)) return d + e * x;   // :471-472
  v = static_cast<T>(1) / (a * c) * (v - d - e * x);                          // :473
  const T u = a * x - b;                                                     // :474
  const T zero = 0, one = 1;
  switch (f.h) {
    case kAbs: v = u < zero ? -one : (u > zero ? one : Max(-one, Min(one, v))); break;            // :360-368
    case kNegEntr: v = -Log(u) - one; break;                                                       // :371-373
    case kExp: v = Exp(u); break;                                                                  // :376-378
    case kHuber: v = Max(-one, Min(one, u)); break;                                                // :381-383
    case kIdentity: v = one; break;                                                                // :386-388
    case kIndBox01: v = u <= zero ? Min(zero, v) : (u >= one ? Max(zero, v) : zero); break;        // :391-398
    case kIndEq0: break;                                                                           // :401-403
    case kIndGe0: v = u <= zero ? Min(zero, v) : zero; break;                                      // :406-411
    case kIndLe0: v = u >= zero ? Max(zero, v) : zero; break;                                      // :414-419
    case kLogistic: v = Exp(u) / (one + Exp(u)); break;                                            // :422-424
    case kMaxNeg0: v = u < zero ? -one : (u > zero ? zero : Min(zero, Max(-one, v))); break;       // :427-434
    case kMaxPos0: v = u < zero ? zero : (u > zero ? one : Min(one, Max(zero, v))); break;         // :437-444
    case kNegLog: v = -one / u; break;                                                             // :447-449
    case kRecipr: v = one / (u * u); break;                                                        // :452-454
    case kSquare: v = u; break;                                                                    // :457-459
    case kZero: default: v = zero; break;                                                          // :462-464
  }
  return a * c * v + d + e * x;                                                                    // :492
}

template <typename T>
void ProxEvalVec(const std::vector<FunctionObj<T>> &f, T rho, const T *in, T *out) {
  // prox_lib.h:503-511
  const long n = static_cast<long>(f.size());
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; ++i) out[i] = ProxEval(f[i], in[i], rho);
}

template <typename T>
T FuncEvalVec(const std::vector<FunctionObj<T>> &f, const T *in) {
  // prox_lib.h:520-529: sequential sum in T.
  T sum = 0;
  for (size_t i = 0; i < f.size(); ++i) sum += FuncEval(f[i], in[i]);
  return sum;
}

// -----------------------------------------------------------------------------
// gsl::rand (src/cpu/include/gsl/gsl_rand.h:8-16): a fresh
// std::default_random_engine (= minstd_rand0 in libstdc++, x <- 16807 x mod
// 2^31-1, seed 1) + uniform_real_distribution<T>(0,1) (= generate_canonical).
// Restated by hand so the start vector does not depend on the C++ library.
// -----------------------------------------------------------------------------
struct MinStd0 {
  uint64_t s = 1;
  uint32_t next() { s = (s * 16807ull) % 2147483647ull; return static_cast<uint32_t>(s); }
};
inline void rand_uniform(float *x, size_t n) {
  MinStd0 g;
  const float r = static_cast<float>(2147483646.0L);  // max-min+1, rounded to float
  for (size_t i = 0; i < n; ++i) {
    float v = static_cast<float>(g.next() - 1u) / r;
    if (v >= 1.0f) v = std::nextafter(1.0f, 0.0f);
    x[i] = v;
  }
}
inline void rand_uniform(double *x, size_t n) {
  MinStd0 g;
  const double r = 2147483646.0;
  for (size_t i = 0; i < n; ++i) {
    double lo = static_cast<double>(g.next() - 1u);
    double hi = static_cast<double>(g.next() - 1u);
    double v = (lo + hi * r) / (r * r);
    if (v >= 1.0) v = std::nextafter(1.0, 0.0);
    x[i] = v;
  }
}

// -----------------------------------------------------------------------------
// BLAS-1 restatements (src/cpu/include/gsl/gsl_blas.h:16-87 -> vendor CBLAS).
// -----------------------------------------------------------------------------
template <typename T>
double sumsq(const T *x, size_t n) {
  double s = 0;
  const long nn = static_cast<long>(n);
#pragma omp parallel for reduction(+ : s) schedule(static) if (nn > 100000)
  for (long i = 0; i < nn; ++i) s += static_cast<double>(x[i]) * x[i];
  return s;
}
template <typename T>
T nrm2(const T *x, size_t n) { return static_cast<T>(std::sqrt(sumsq(x, n))); }
template <typename T>
T dot(const T *x, const T *y, size_t n) {
  double s = 0;
  for (size_t i = 0; i < n; ++i) s += static_cast<double>(x[i]) * y[i];
  return static_cast<T>(s);
}
template <typename T>
void axpy(T alpha, const T *x, T *y, size_t n) {
  for (size_t i = 0; i < n; ++i) y[i] += alpha * x[i];
}
template <typename T>
void scal(T alpha, T *x, size_t n) {
  for (size_t i = 0; i < n; ++i) x[i] *= alpha;
}

// -----------------------------------------------------------------------------
// Operator interface (src/include/matrix/matrix.h): y <- alpha op(A) x + beta y.
// `square` applies the map a -> a*a on the fly, which is what the reference
// obtains by squaring A in place under the sign-bit trick
// (matrix_dense.cpp:126-172, equil_helper.h:62-102).
// -----------------------------------------------------------------------------
template <typename T>
struct Operator {
  size_t m = 0, n = 0;
  long n_mul = 0;  // operator applications (for the SpMV/GEMV count in reports)
  virtual ~Operator() {}
  virtual void MulImpl(char trans, T alpha, const T *x, T beta, T *y, bool square) = 0;
  void Mul(char trans, T alpha, const T *x, T beta, T *y, bool square = false) {
    MulImpl(trans, alpha, x, beta, y, square);
  }
};

// Dense, row- or column-major (src/cpu/matrix/matrix_dense.cpp:93-113).
template <typename T>
struct DenseOp : Operator<T> {
  std::vector<T> a;
  bool row_major;
  DenseOp(bool rm, size_t m_, size_t n_, const T *data) : row_major(rm) {
    this->m = m_; this->n = n_;
    a.assign(data, data + m_ * n_);                       // matrix_dense.cpp:85-87
  }
  inline T at(size_t i, size_t j) const { return row_major ? a[i * this->n + j] : a[j * this->m + i]; }
  inline T &at(size_t i, size_t j) { return row_major ? a[i * this->n + j] : a[j * this->m + i]; }

  void MulImpl(char trans, T alpha, const T *x, T beta, T *y, bool square) override {
    this->n_mul++;
    const size_t m = this->m, n = this->n;
    const bool tr = (trans == 't' || trans == 'T');
    // "contiguous-row" form: out[i] = sum_j M[i][j] x[j] over contiguous j.
    const bool dot_form = (row_major && !tr) || (!row_major && tr);
    const size_t rows = row_major ? m : n, cols = row_major ? n : m;
    const T *A = a.data();
    if (dot_form) {
      const long R = static_cast<long>(rows);
#pragma omp parallel for schedule(static)
      for (long i = 0; i < R; ++i) {
        const T *r = A + static_cast<size_t>(i) * cols;
        T acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        size_t j = 0;
        if (square) {
          for (; j + 8 <= cols; j += 8)
            for (int k = 0; k < 8; ++k) acc[k] += r[j + k] * r[j + k] * x[j + k];
          for (; j < cols; ++j) acc[0] += r[j] * r[j] * x[j];
        } else {
          for (; j + 8 <= cols; j += 8)
            for (int k = 0; k < 8; ++k) acc[k] += r[j + k] * x[j + k];
          for (; j < cols; ++j) acc[0] += r[j] * x[j];
        }
        T s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        y[i] = (beta == 0) ? alpha * s : alpha * s + beta * y[i];
      }
    } else {
      // out[j] = sum_i M[i][j] x[i]: column sums, threads own column ranges.
      const long C = static_cast<long>(cols);
#pragma omp parallel
      {
        std::vector<T> acc;
#pragma omp for schedule(static)
        for (long c0 = 0; c0 < C; c0 += 512) {
          const size_t c1 = std::min<size_t>(C, c0 + 512), w = c1 - c0;
          acc.assign(w, 0);
          for (size_t i = 0; i < rows; ++i) {
            const T *r = A + i * cols + c0;
            const T xi = x[i];
            if (square) for (size_t j = 0; j < w; ++j) acc[j] += r[j] * r[j] * xi;
            else        for (size_t j = 0; j < w; ++j) acc[j] += r[j] * xi;
          }
          for (size_t j = 0; j < w; ++j)
            y[c0 + j] = (beta == 0) ? alpha * acc[j] : alpha * acc[j] + beta * y[c0 + j];
        }
      }
    }
  }
};

// Sparse: CSR (or CSC) plus its transposed copy, both used as row-gather SpMVs
// (src/cpu/matrix/matrix_sparse.cpp:97-154, gsl_spblas.h:10-40, gsl_spmat.h:32-93).
template <typename T>
struct SparseOp : Operator<T> {
  size_t nnz;
  bool row_major;
  std::vector<T> val;      // 2*nnz
  std::vector<int> ind;    // 2*nnz
  std::vector<int> ptr;    // m+n+2
  SparseOp(bool rm, size_t m_, size_t n_, size_t nnz_, const T *v, const int *p, const int *id)
      : nnz(nnz_), row_major(rm) {
    this->m = m_; this->n = n_;
    val.resize(2 * nnz); ind.resize(2 * nnz); ptr.resize(m_ + n_ + 2);
    const size_t pl = ptr_len();
    std::memcpy(val.data(), v, nnz * sizeof(T));
    std::memcpy(ind.data(), id, nnz * sizeof(int));
    std::memcpy(ptr.data(), p, pl * sizeof(int));
    // csr2csc (gsl_spmat.h:32-55): stable counting sort by minor index.
    const size_t major = rm ? m_ : n_, minor = rm ? n_ : m_;
    int *cp = ptr.data() + pl;
    std::memset(cp, 0, (minor + 1) * sizeof(int));
    for (size_t i = 0; i < nnz; ++i) cp[ind[i] + 1]++;
    for (size_t i = 0; i < minor; ++i) cp[i + 1] += cp[i];
    for (size_t i = 0; i < major; ++i)
      for (int j = ptr[i]; j < ptr[i + 1]; ++j) {
        int k = ind[j];
        int l = cp[k]++;
        ind[nnz + l] = static_cast<int>(i);
        val[nnz + l] = val[j];
      }
    for (size_t i = minor; i > 0; --i) cp[i] = cp[i - 1];
    cp[0] = 0;
  }
  size_t ptr_len() const { return (row_major ? this->m : this->n) + 1; }

  void MulImpl(char trans, T alpha, const T *x, T beta, T *y, bool square) override {
    this->n_mul++;
    const bool tr = (trans == 't' || trans == 'T');
    const bool first = (row_major && !tr) || (!row_major && tr);  // gsl_spblas.h:17-26
    const T *d = first ? val.data() : val.data() + nnz;
    const int *ci = first ? ind.data() : ind.data() + nnz;
    const int *rp = first ? ptr.data() : ptr.data() + ptr_len();
    const long size = static_cast<long>(tr ? this->n : this->m);
#pragma omp parallel for schedule(dynamic, 1024)
    for (long i = 0; i < size; ++i) {
      T tmp = 0;
      if (square) for (int j = rp[i]; j < rp[i + 1]; ++j) tmp += d[j] * d[j] * x[ci[j]];
      else        for (int j = rp[i]; j < rp[i + 1]; ++j) tmp += d[j] * x[ci[j]];
      y[i] = alpha * tmp + beta * y[i];                   // gsl_spblas.h:37
    }
  }
};

// -----------------------------------------------------------------------------
// Sinkhorn-Knopp on A.^2 (src/cpu/include/equil_helper.h:140-164).
// With row shards: the column step all-reduces its n partial sums; m_glob is
// the global row count (== m for a single process).
// -----------------------------------------------------------------------------
template <typename T>
void SinkhornKnopp(Operator<T> *A, T *d, T *e, size_t m_glob, const Comm &comm) {
  const size_t m = A->m, n = A->n;
  const unsigned kEquilIter = 50u;
  const double kSinkhornConst = 1e-4;
  for (size_t i = 0; i < m; ++i) d[i] = 1;
  for (size_t j = 0; j < n; ++j) e[j] = 1;
  const T ce = static_cast<T>(kSinkhornConst) * (m_glob + n) / m_glob;  // equil_helper.h:152-153
  const T cd = static_cast<T>(kSinkhornConst) * (m_glob + n) / n;      // equil_helper.h:159-160
  for (unsigned k = 0; k < kEquilIter; ++k) {
    A->Mul('t', static_cast<T>(1), d, static_cast<T>(0), e, true);
    comm.sum_vec(e, n);
    for (size_t j = 0; j < n; ++j) e[j] = static_cast<T>(m_glob) / (e[j] + ce);
    A->Mul('n', static_cast<T>(1), e, static_cast<T>(0), d, true);
    for (size_t i = 0; i < m; ++i) d[i] = static_cast<T>(n) / (d[i] + cd);
  }
}

// Power iteration for ||A||_2 (equil_helper.h:107-135).
template <typename T>
T Norm2Est(Operator<T> *A, const Comm &comm, unsigned *iters_out) {
  const T kTol = static_cast<T>(1e-4);
  const unsigned kNormEstMaxIter = 50u;
  T norm_est = 0, norm_est_last;
  std::vector<T> x(A->n), Sx(A->m);
  rand_uniform(x.data(), x.size());
  unsigned i = 0;
  for (i = 0; i < kNormEstMaxIter; ++i) {
    norm_est_last = norm_est;
    A->Mul('n', static_cast<T>(1), x.data(), static_cast<T>(0), Sx.data());
    A->Mul('t', static_cast<T>(1), Sx.data(), static_cast<T>(0), x.data());
    comm.sum_vec(x.data(), x.size());
    T normx = nrm2(x.data(), x.size());
    T normSx = static_cast<T>(std::sqrt(comm.sum1(sumsq(Sx.data(), Sx.size()))));
    scal(1 / normx, x.data(), x.size());
    norm_est = normx / normSx;
    if (std::abs(norm_est_last - norm_est) < kTol * norm_est) break;
  }
  if (iters_out) *iters_out = (i < kNormEstMaxIter) ? i + 1 : i;
  return norm_est;
}

// A = sign(A) .* sqrt(A.^2): what is left in memory after SetSign/UnSetSign
// (matrix_dense.cpp:135,158; equil_helper.h:62-102).
template <typename T>
inline T sign_sqrt_square(T a) {
  const int neg = a < 0;
  return static_cast<T>(1 - 2 * neg) * Sqrt(a * a);
}

// MatrixDense::Equil (src/cpu/matrix/matrix_dense.cpp:116-200).
template <typename T>
void EquilDense(DenseOp<T> *A, T *d, T *e, size_t m_glob, const Comm &comm) {
  const size_t m = A->m, n = A->n;
  SinkhornKnopp<T>(A, d, e, m_glob, comm);
  for (size_t i = 0; i < m; ++i) d[i] = Sqrt(d[i]);       // :176
  for (size_t j = 0; j < n; ++j) e[j] = Sqrt(e[j]);       // :177
  double fro = 0;
  const long M = static_cast<long>(m);
#pragma omp parallel for reduction(+ : fro) schedule(static)
  for (long i = 0; i < M; ++i)
    for (size_t j = 0; j < n; ++j) {
      T v = sign_sqrt_square(A->at(i, j));
      v *= d[i] * e[j];                                   // :231-237 MultRow/MultCol
      A->at(i, j) = v;
      fro += static_cast<double>(v) * v;
    }
  fro = comm.sum1(fro);
  // ||A||_F / sqrt(min(m, n))  (:215-218)
  const T normA = static_cast<T>(std::sqrt(fro)) /
                  std::sqrt(static_cast<T>(std::min(m_glob, n)));
  const T inv = 1 / normA;                                // :186
  const long NE = static_cast<long>(m * n);
#pragma omp parallel for schedule(static)
  for (long t = 0; t < NE; ++t) A->a[t] *= inv;
  const T invs = 1 / std::sqrt(normA);                    // :191-192
  for (size_t i = 0; i < m; ++i) d[i] *= invs;
  for (size_t j = 0; j < n; ++j) e[j] *= invs;
}

// MatrixSparse::Equil (src/cpu/matrix/matrix_sparse.cpp:158-242, 249-302).
template <typename T>
void EquilSparse(SparseOp<T> *A, T *d, T *e, size_t m_glob = 0, const Comm &comm = Comm()) {
  const size_t m = A->m, n = A->n, nnz = A->nnz;
  if (m_glob == 0) m_glob = m;
  SinkhornKnopp<T>(A, d, e, m_glob, comm);
  for (size_t t = 0; t < 2 * nnz; ++t) A->val[t] = sign_sqrt_square(A->val[t]);
  for (size_t i = 0; i < m; ++i) d[i] = Sqrt(d[i]);
  for (size_t j = 0; j < n; ++j) e[j] = Sqrt(e[j]);
  // MultDiag on both copies (:291-302).
  const size_t pl = A->ptr_len();
  const int *p0 = A->ptr.data(), *p1 = A->ptr.data() + pl;
  const int *i0 = A->ind.data(), *i1 = A->ind.data() + nnz;
  T *v0 = A->val.data(), *v1 = A->val.data() + nnz;
  if (A->row_major) {
    for (size_t t = 0; t < m; ++t) for (int i = p0[t]; i < p0[t + 1]; ++i) v0[i] *= d[t] * e[i0[i]];
    for (size_t t = 0; t < n; ++t) for (int i = p1[t]; i < p1[t + 1]; ++i) v1[i] *= d[i1[i]] * e[t];
  } else {
    for (size_t t = 0; t < n; ++t) for (int i = p0[t]; i < p0[t + 1]; ++i) v0[i] *= d[i0[i]] * e[t];
    for (size_t t = 0; t < m; ++t) for (int i = p1[t]; i < p1[t + 1]; ++i) v1[i] *= d[t] * e[i1[i]];
  }
  // Frobenius norm over the first nnz only (:257), / sqrt(min(m,n)).
  const T normA = (comm.fn ? static_cast<T>(std::sqrt(comm.sum1(sumsq(v0, nnz)))) : nrm2(v0, nnz)) /
                  static_cast<T>(std::sqrt(static_cast<double>(std::min(m_glob, n))));
  const T inv = 1 / normA;
  for (size_t t = 0; t < 2 * nnz; ++t) A->val[t] *= inv;
  const T invs = 1 / std::sqrt(normA);
  for (size_t i = 0; i < m; ++i) d[i] *= invs;
  for (size_t j = 0; j < n; ++j) e[j] *= invs;
}

// -----------------------------------------------------------------------------
// Projectors (src/include/projector/projector.h).
// -----------------------------------------------------------------------------
template <typename T>
struct Projector {
  long cg_iters = 0, n_proj = 0;
  virtual ~Projector() {}
  virtual void Init() = 0;
  virtual void Project(const T *x0, const T *y0, T s, T *x, T *y, T tol) = 0;
};

// Direct: Gram + blocked Cholesky + two triangular solves
// (src/cpu/projector/projector_direct_dense.cpp:45-175, gsl_linalg.h:12-61).
template <typename T>
struct ProjectorDirect : Projector<T> {
  DenseOp<T> *A;
  Comm comm;
  size_t k;            // min(m, n)
  std::vector<T> AA, L;
  T s_fact = -1;
  bool tall;           // m > n  -> Gram = A^T A, else A A^T
  ProjectorDirect(DenseOp<T> *A_, const Comm &c, bool force_tall = false)
      : A(A_), comm(c) {
    tall = force_tall || A->m > A->n;
    k = tall ? A->n : A->m;
  }

  void Init() override {                                  // :45-84 (SYRK, lower)
    AA.assign(k * k, 0);
    L.assign(k * k, 0);
    const size_t m = A->m, n = A->n;
    const long K = static_cast<long>(k);
    if (tall && A->row_major) {
      // G[i][j] = sum_r A[r][i] A[r][j], j <= i, as rank-1 row updates summed
      // in row panels of 256 (a blocked SYRK's summation shape).
      const size_t PB = 256, IB = 32;
      const T *a = A->a.data();
#pragma omp parallel
      {
        std::vector<T> part(IB * k);
#pragma omp for schedule(dynamic, 1)
        for (long i0 = 0; i0 < K; i0 += IB) {
          const size_t i1 = std::min<size_t>(k, i0 + IB);
          for (size_t r0 = 0; r0 < m; r0 += PB) {
            const size_t r1 = std::min(m, r0 + PB);
            std::fill(part.begin(), part.end(), static_cast<T>(0));
            for (size_t r = r0; r < r1; ++r) {
              const T *row = a + r * n;
              for (size_t i = i0; i < i1; ++i) {
                const T ai = row[i];
                T *p = &part[(i - i0) * k];
                for (size_t j = 0; j <= i; ++j) p[j] += ai * row[j];
              }
            }
            for (size_t i = i0; i < i1; ++i)
              for (size_t j = 0; j <= i; ++j) AA[i * k + j] += part[(i - i0) * k + j];
          }
        }
      }
    } else if (tall) {
#pragma omp parallel for schedule(dynamic, 4)
      for (long i = 0; i < K; ++i)
        for (size_t j = 0; j <= static_cast<size_t>(i); ++j) {
          T acc = 0;
          for (size_t r = 0; r < m; ++r) acc += A->at(r, i) * A->at(r, j);
          AA[static_cast<size_t>(i) * k + j] = acc;
        }
    } else {
#pragma omp parallel for schedule(dynamic, 4)
      for (long i = 0; i < K; ++i)
        for (size_t j = 0; j <= static_cast<size_t>(i); ++j) {
          T acc = 0;
          for (size_t c = 0; c < n; ++c) acc += A->at(i, c) * A->at(j, c);
          AA[static_cast<size_t>(i) * k + j] = acc;
        }
    }
    if (comm.fn) comm.sum_vec(AA.data(), AA.size());
  }

  // gsl_linalg.h:36-55 (block 128) with the unblocked kernel :12-27; written
  // as a left-looking row Cholesky, which produces the same factor.
  void Factor(T s) {
    L = AA;
    for (size_t i = 0; i < k; ++i) L[i * k + i] += s;     // :118-119
    for (size_t j = 0; j < k; ++j) {
      double djj = L[j * k + j];
      for (size_t p = 0; p < j; ++p) djj -= static_cast<double>(L[j * k + p]) * L[j * k + p];
      const T ljj = static_cast<T>(std::sqrt(djj));
      L[j * k + j] = ljj;
      const long K = static_cast<long>(k);
#pragma omp parallel for schedule(static) if (k - j > 256)
      for (long i = static_cast<long>(j) + 1; i < K; ++i) {
        T v = L[i * k + j];
        const T *li = &L[i * k], *lj = &L[j * k];
        T acc = 0;
        for (size_t p = 0; p < j; ++p) acc += li[p] * lj[p];
        L[i * k + j] = (v - acc) / ljj;
      }
    }
    s_fact = s;
  }

  void CholSolve(T *x) const {                            // gsl_linalg.h:57-61
    for (size_t i = 0; i < k; ++i) {                      // L z = x
      T acc = x[i];
      const T *li = &L[i * k];
      for (size_t p = 0; p < i; ++p) acc -= li[p] * x[p];
      x[i] = acc / li[i];
    }
    for (size_t ii = k; ii > 0; --ii) {                   // L^T x = z (row-axpy form)
      const size_t i = ii - 1;
      const T *li = &L[i * k];
      const T xi = x[i] / li[i];
      x[i] = xi;
      for (size_t p = 0; p < i; ++p) x[p] -= li[p] * xi;
    }
  }

  void Project(const T *x0, const T *y0, T s, T *x, T *y, T) override {  // :87-175
    this->n_proj++;
    const size_t m = A->m, n = A->n;
    std::memcpy(x, x0, n * sizeof(T));
    std::memcpy(y, y0, m * sizeof(T));
    if (s != s_fact) Factor(s);                           // :116-121
    if (tall) {                                           // :122-127
      std::vector<T> t(n);
      A->Mul('t', static_cast<T>(1), y, static_cast<T>(0), t.data());
      comm.sum_vec(t.data(), n);
      for (size_t j = 0; j < n; ++j) x[j] += t[j];
      CholSolve(x);
      A->Mul('n', static_cast<T>(1), x, static_cast<T>(0), y);
    } else {                                              // :128-135
      A->Mul('n', static_cast<T>(1), x, static_cast<T>(-1), y);
      CholSolve(y);
      A->Mul('t', static_cast<T>(-1), y, static_cast<T>(1), x);
      axpy(static_cast<T>(1), y0, y, m);
    }
  }
};

// CGLS (src/cpu/include/cgls.h:200-323).
template <typename T>
int CglsSolve(Operator<T> *A, const T *b, T *x, double shift, double tol, int maxit,
              long *iters, const Comm &comm = Comm()) {
  // rows sharded: x-sized vectors are replicated, A^T products and |q|^2 are summed over ranks
  const size_t m = A->m, n = A->n;
  std::vector<T> p(n), q(m), r(b, b + m), s(x, x + n);
  double gamma, normp, normq, norms, norms0, normx, xmax;
  int k = 0, flag = 0, indefinite = 0;
  const T kNegShift = static_cast<T>(-shift);
  const double kEps = std::numeric_limits<T>::epsilon();
  normx = nrm2(x, n);
  if (normx > 0.) A->Mul('n', static_cast<T>(-1), x, static_cast<T>(1), r.data());   // :229-233
  if (comm.fn) {
    std::vector<T> t(n);
    A->Mul('t', static_cast<T>(1), r.data(), static_cast<T>(0), t.data());
    comm.sum_vec(t.data(), n);
    for (size_t j = 0; j < n; ++j) s[j] = t[j] + kNegShift * s[j];
  } else {
    A->Mul('t', static_cast<T>(1), r.data(), kNegShift, s.data());                   // :236
  }
  p = s;
  norms = nrm2(s.data(), n);
  norms0 = norms;
  gamma = norms0 * norms0;
  normx = nrm2(x, n);
  xmax = normx;
  if (norms < kEps) flag = 1;
  for (k = 0; k < maxit && !flag; ++k) {
    A->Mul('n', static_cast<T>(1), p.data(), static_cast<T>(0), q.data());           // :257
    normp = nrm2(p.data(), n);
    normq = comm.fn ? std::sqrt(comm.sum1(sumsq(q.data(), m))) : nrm2(q.data(), m);
    double delta = normq * normq + shift * normp * normp;                            // :266
    if (delta <= 0.) indefinite = 1;
    if (delta == 0.) delta = kEps;
    const T alpha = static_cast<T>(gamma / delta);
    const T neg_alpha = static_cast<T>(-gamma / delta);
    axpy(alpha, p.data(), x, n);
    axpy(neg_alpha, q.data(), r.data(), m);
    std::memcpy(s.data(), x, n * sizeof(T));                                         // :281
    if (comm.fn) {
      std::vector<T> t(n);
      A->Mul('t', static_cast<T>(1), r.data(), static_cast<T>(0), t.data());
      comm.sum_vec(t.data(), n);
      for (size_t j = 0; j < n; ++j) s[j] = t[j] + kNegShift * s[j];
    } else {
      A->Mul('t', static_cast<T>(1), r.data(), kNegShift, s.data());                 // :282
    }
    norms = nrm2(s.data(), n);
    const double gamma1 = gamma;
    gamma = norms * norms;
    const T beta = static_cast<T>(gamma / gamma1);
    axpy(beta, p.data(), s.data(), n);                                               // :295
    p = s;
    normx = nrm2(x, n);
    xmax = std::max(xmax, normx);
    const bool converged = (norms <= norms0 * tol) || (normx * tol >= 1.);           // :301
    if (converged) break;
  }
  if (iters) *iters += (k < maxit && !flag) ? k + 1 : k;
  const double shrink = normx / xmax;
  if (k == maxit) flag = 2;
  else if (indefinite) flag = 3;
  else if (shrink * shrink <= tol) flag = 4;
  return flag;
}

// src/cpu/projector/projector_cgls.cpp:52-88.
template <typename T>
struct ProjectorCgls : Projector<T> {
  Operator<T> *A;
  Comm comm;
  explicit ProjectorCgls(Operator<T> *A_, const Comm &c = Comm()) : A(A_), comm(c) {}
  void Init() override {}
  void Project(const T *x0, const T *y0, T s, T *x, T *y, T tol) override {
    this->n_proj++;
    const size_t m = A->m, n = A->n;
    axpy(static_cast<T>(-1), x0, x, n);                                   // :62
    std::memcpy(y, y0, m * sizeof(T));                                    // :65
    A->Mul('n', static_cast<T>(-1), x0, static_cast<T>(1), y);            // :68
    CglsSolve<T>(A, y, x, s, tol, 500, &this->cg_iters, comm);            // :71-72
    axpy(static_cast<T>(1), x0, x, n);                                    // :75
    A->Mul('n', static_cast<T>(1), x, static_cast<T>(0), y);              // :78
  }
};

// -----------------------------------------------------------------------------
// PogsImplementation::Solve (src/cpu/pogs.cpp:91-581) for a separable
// objective (kUseExactTol = false, :102-110), including _Init (:59-88) and
// PogsObjectiveSeparable::scale (:608-617).
//
// Row-sharded variant: y-sized quantities are local to the shard, x-sized ones
// replicated; sums over rows go through `comm`.
// -----------------------------------------------------------------------------
template <typename T>
struct SolveArgs {
  T rho, abs_tol, rel_tol;
  unsigned max_iter, verbose;
  bool adaptive_rho, gap_stop;
};

template <typename T>
int AdmmSolve(Operator<T> *A, Projector<T> *P, std::vector<FunctionObj<T>> f,
              std::vector<FunctionObj<T>> g, const T *de, T nrmA, size_t m_glob,
              const Comm &comm, const SolveArgs<T> &arg, T *x_out, T *y_out,
              T *l_out, T *mu_out, T *optval, unsigned *final_iter, OracleInfo *info) {
  const T kDeltaMin = static_cast<T>(1.05), kGamma = static_cast<T>(1.01);
  const T kTau = static_cast<T>(0.8), kRhoMin = static_cast<T>(1e-4);
  const T kRhoMax = static_cast<T>(1e4), kKappa = static_cast<T>(0.9);
  const T kOne = 1, kZero = 0;
  const T kProjTolMax = static_cast<T>(1e-8), kProjTolMin = static_cast<T>(1e-2);
  const T kAlpha = static_cast<T>(1.7);

  const size_t m = A->m, n = A->n;
  T rho = arg.rho;
  const T *d = de, *e = de + m;

  // z = [x | y] layout (pogs.cpp:129-138).
  std::vector<T> z(m + n, 0), zt(m + n, 0), zprev(m + n, 0), ztemp(m + n, 0), z12(m + n, 0);
  T *x = z.data(), *y = z.data() + n;
  T *x12 = z12.data(), *y12 = z12.data() + n;
  T *xprev = zprev.data(), *yprev = zprev.data() + n;
  T *xtemp = ztemp.data(), *ytemp = ztemp.data() + n;
  T *xt = zt.data(), *yt = zt.data() + n;

  // objective->scale(d, e)  (:141, :608-617)
  for (size_t i = 0; i < m; ++i) { f[i].a /= d[i]; f[i].d /= d[i]; f[i].e /= d[i] * d[i]; }
  for (size_t j = 0; j < n; ++j) { g[j].a *= e[j]; g[j].d *= e[j]; g[j].e *= e[j] * e[j]; }

  const T sqrtn_atol = std::sqrt(static_cast<T>(n)) * arg.abs_tol;             // :199-201
  const T sqrtm_atol = std::sqrt(static_cast<T>(m_glob)) * arg.abs_tol;
  const T sqrtmn_atol = std::sqrt(static_cast<T>(m_glob + n)) * arg.abs_tol;
  T delta = kDeltaMin, xi = 1;
  unsigned k = 0, kd = 0, ku = 0;
  bool converged = false;
  T nrm_r = 0, nrm_s = 0, gap = 0, eps_gap = 0, eps_pri = 0, eps_dua = 0;
  T prev_nrm_r = std::numeric_limits<T>::max();                                 // :251
  unsigned n_exact = 0;

  // Warm start from (x0, lambda0) (pogs.cpp:144-156); the reference only supports both
  // together (:159-179 assert otherwise).
  if (info && info->warm_x && info->warm_l) {
    const T *x0 = static_cast<const T *>(info->warm_x), *l0 = static_cast<const T *>(info->warm_l);
    for (size_t j = 0; j < n; ++j) xtemp[j] = x0[j] / e[j];                      // :145-146
    A->Mul('n', kOne, xtemp, kZero, ytemp);                                      // :147
    z = ztemp;                                                                   // :148
    for (size_t i = 0; i < m; ++i) ytemp[i] = l0[i] / d[i];                      // :151-152
    A->Mul('t', -kOne, ytemp, kZero, xtemp);                                     // :153
    comm.sum_vec(xtemp, n);                                                      // rows are sharded
    for (size_t i = 0; i < m + n; ++i) ztemp[i] *= -kOne / rho;                  // :154
    zt = ztemp;                                                                  // :155
  }

  // Norm helpers: x-part replicated, y-part sharded.
  auto nrm_xy = [&](const T *vx, const T *vy) {   // ||[vx|vy]||
    return static_cast<T>(std::sqrt(sumsq(vx, n) + comm.sum1(sumsq(vy, m))));
  };
  auto nrm_y = [&](const T *vy) { return static_cast<T>(std::sqrt(comm.sum1(sumsq(vy, m)))); };

  for (;; ++k) {
    zprev = z;                                                                  // :254
    axpy(-kOne, zt.data(), z.data(), m + n);                                    // :257
    ProxEvalVec(g, rho, x, x12);                                                // :263, :603-606
    ProxEvalVec(f, rho, y, y12);
    axpy(-kOne, z12.data(), z.data(), m + n);                                   // :267
    {
      double gx = 0, gy = 0;
      for (size_t j = 0; j < n; ++j) gx += static_cast<double>(x[j]) * x12[j];
      for (size_t i = 0; i < m; ++i) gy += static_cast<double>(y[i]) * y12[i];
      gap = std::abs(static_cast<T>(gx + comm.sum1(gy)));                       // :268-269
    }
    eps_gap = sqrtmn_atol + arg.rel_tol * nrm_xy(x, y) * nrm_xy(x12, y12);      // :270-271
    eps_pri = sqrtm_atol + arg.rel_tol * nrm_y(y12);                            // :272
    eps_dua = rho * (sqrtn_atol + arg.rel_tol * nrm2(x, n));                    // :273

    ztemp = zt;                                                                 // :276-278
    axpy(kAlpha, z12.data(), ztemp.data(), m + n);
    axpy(kOne - kAlpha, zprev.data(), ztemp.data(), m + n);

    std::memcpy(x, xprev, n * sizeof(T));                                       // :281

    T proj_tol = kProjTolMin * std::pow(std::min(prev_nrm_r, kOne), static_cast<T>(0.5));
    proj_tol = std::max(proj_tol, kProjTolMax);                                 // :287-290
    P->Project(xtemp, ytemp, kOne, x, y, proj_tol);                             // :296

    // Approximate residuals (:342-348).
    {
      double sx = 0, sy = 0, rx = 0, ry = 0;
      for (size_t j = 0; j < n; ++j) {
        const T a = xprev[j] - x[j]; sx += static_cast<double>(a) * a;
        const T b = x12[j] - x[j];   rx += static_cast<double>(b) * b;
      }
      for (size_t i = 0; i < m; ++i) {
        const T a = yprev[i] - y[i]; sy += static_cast<double>(a) * a;
        const T b = y12[i] - y[i];   ry += static_cast<double>(b) * b;
      }
      double yy[2] = {sy, ry};
      comm.sum(yy, 2);
      nrm_s = rho * (nrmA * static_cast<T>(std::sqrt(yy[0])) + static_cast<T>(std::sqrt(sx)));
      nrm_r = nrmA * static_cast<T>(std::sqrt(rx)) + static_cast<T>(std::sqrt(yy[1]));
    }

    bool exact = false;
    if (nrm_r < 10 * eps_pri && nrm_s < 10 * eps_dua) {                         // :352
      ztemp = z12;                                                              // :353
      A->Mul('n', kOne, x12, -kOne, ytemp);                                     // :354
      nrm_r = nrm_y(ytemp);                                                     // :364
      ztemp = z12;                                                              // :366-368
      axpy(kOne, zt.data(), ztemp.data(), m + n);
      axpy(-kOne, zprev.data(), ztemp.data(), m + n);
      {
        std::vector<T> t(n);
        A->Mul('t', kOne, ytemp, kZero, t.data());                              // :369
        comm.sum_vec(t.data(), n);
        for (size_t j = 0; j < n; ++j) xtemp[j] += t[j];
      }
      nrm_s = rho * nrm2(xtemp, n);                                             // :373
      exact = true;
      ++n_exact;
    }
    converged = exact && nrm_r < eps_pri && nrm_s < eps_dua &&
                (!arg.gap_stop || gap < eps_gap);                               // :379-380

    static const bool trace = std::getenv("POGS_ORACLE_ITERTRACE") != nullptr;   // debugging aid
    if (trace)
      std::printf("T %5u rho %.9e r %.9e s %.9e gap %.9e epri %.9e edua %.9e\n", k, (double)rho, (double)nrm_r,
                  (double)nrm_s, (double)gap, (double)eps_pri, (double)eps_dua);
    if (arg.verbose > 1 && (k % 100 == 0 || converged || (arg.verbose > 2 && k % 10 == 0)))
      std::printf("%5u : %.2e  %.2e  %.2e  %.2e  %.2e  %.2e\n", k, (double)nrm_r,
                  (double)eps_pri, (double)nrm_s, (double)eps_dua, (double)gap, (double)eps_gap);

    if (converged || k == arg.max_iter - 1) {                                   // :391-394
      *final_iter = k;
      break;
    }

    axpy(kAlpha, z12.data(), zt.data(), m + n);                                 // :397-399
    axpy(kOne - kAlpha, zprev.data(), zt.data(), m + n);
    axpy(-kOne, z.data(), zt.data(), m + n);

    if (arg.adaptive_rho) {                                                     // :402-466
      const unsigned kRhoUpdateFreq = 50u;
      const T kRhoChangeMax = static_cast<T>(1.5), kRhoChangeMin = static_cast<T>(0.67);
      const T kImbalanceThresh = static_cast<T>(10);
      if (k > 0 && k % kRhoUpdateFreq == 0 && eps_pri > kZero && eps_dua > kZero) {
        T pri_normalized = nrm_r / eps_pri, dua_normalized = nrm_s / eps_dua;
        if (pri_normalized > kZero && dua_normalized > kZero) {
          T imbalance = pri_normalized / dua_normalized;
          if (imbalance > kImbalanceThresh || imbalance < kOne / kImbalanceThresh) {
            T rho_ratio = std::sqrt(imbalance);
            rho_ratio = std::max(kRhoChangeMin, std::min(kRhoChangeMax, rho_ratio));
            T rho_new = rho * rho_ratio;
            rho_new = std::max(kRhoMin, std::min(kRhoMax, rho_new));
            if (std::abs(rho_new - rho) / rho > static_cast<T>(0.05)) {
              T scale = rho / rho_new;
              rho = rho_new;
              scal(scale, zt.data(), m + n);
            }
          }
        }
      } else if (nrm_s < xi * eps_dua && nrm_r > xi * eps_pri &&
                 kTau * static_cast<T>(k) > static_cast<T>(kd)) {
        if (rho < kRhoMax) {
          rho *= delta;
          scal(1 / delta, zt.data(), m + n);
          delta = kGamma * delta;
          ku = k;
        }
      } else if (nrm_s > xi * eps_dua && nrm_r < xi * eps_pri &&
                 kTau * static_cast<T>(k) > static_cast<T>(ku)) {
        if (rho > kRhoMin) {
          rho /= delta;
          scal(delta, zt.data(), m + n);
          delta = kGamma * delta;
          kd = k;
        }
      } else if (nrm_s < xi * eps_dua && nrm_r < xi * eps_pri) {
        xi *= kKappa;
      } else {
        delta = kDeltaMin;
      }
    }
    prev_nrm_r = nrm_r;                                                         // :469
  }

  // optval = f(y12) + g(x12)  (:473, :599-601): FuncEval(f) summed first.
  {
    T fy = FuncEvalVec(f, y12);
    if (comm.fn) fy = static_cast<T>(comm.sum1(static_cast<double>(fy)));
    *optval = fy + FuncEvalVec(g, x12);
  }

  int status;                                                                   // :476-482
  if (!converged && k == arg.max_iter - 1) status = 3;     // POGS_MAX_ITER
  else if (!converged && k < arg.max_iter - 1) status = 4; // POGS_NAN_FOUND
  else status = 0;

  // Un-scale (:510-518).
  ztemp = zt;
  axpy(-kOne, zprev.data(), ztemp.data(), m + n);
  axpy(kOne, z12.data(), ztemp.data(), m + n);
  scal(-rho, ztemp.data(), m + n);
  for (size_t i = 0; i < m; ++i) ytemp[i] *= d[i];
  for (size_t j = 0; j < n; ++j) xtemp[j] /= e[j];
  for (size_t i = 0; i < m; ++i) y12[i] /= d[i];
  for (size_t j = 0; j < n; ++j) x12[j] *= e[j];

  std::memcpy(x_out, x12, n * sizeof(T));                                       // :567-570
  std::memcpy(y_out, y12, m * sizeof(T));
  std::memcpy(l_out, ytemp, m * sizeof(T));
  if (mu_out) std::memcpy(mu_out, xtemp, n * sizeof(T));
  if (info) {
    info->rho_final = rho;
    info->exact_iters = n_exact;
    info->cg_iters = P->cg_iters;
    info->n_mul = A->n_mul;
  }
  (void)xt; (void)yt;
  return status;
}

template <typename T>
void fill_info_de(OracleInfo *info, const T *de, size_t m, size_t n) {
  if (!info) return;
  if (info->d_out) for (size_t i = 0; i < m; ++i) info->d_out[i] = de[i];
  if (info->e_out) for (size_t j = 0; j < n; ++j) info->e_out[j] = de[m + j];
}

// src/interface_c/pogs_c.cpp:9-55 (dense => direct projector, :19-20).
template <typename T>
int PogsDense(int ord, size_t m, size_t n, const T *A_, const T *f_a, const T *f_b,
              const T *f_c, const T *f_d, const T *f_e, const int *f_h, const T *g_a,
              const T *g_b, const T *g_c, const T *g_d, const T *g_e, const int *g_h,
              T rho, T abs_tol, T rel_tol, unsigned max_iter, unsigned verbose,
              int adaptive_rho, int gap_stop, T *x, T *y, T *l, T *optval,
              unsigned *final_iter, OracleInfo *info, size_t m_glob, const Comm &comm,
              int use_cgls) {
  const double t0 = now_s();
  DenseOp<T> A(ord == 1, m, n, A_);
  std::vector<T> de(m + n, 0);
  EquilDense<T>(&A, de.data(), de.data() + m, m_glob, comm);           // pogs.cpp:76
  unsigned kpow = 0;
  T nrmA = Norm2Est<T>(&A, comm, &kpow);                               // pogs.cpp:83
  Projector<T> *P;
  if (use_cgls) P = new ProjectorCgls<T>(&A);
  else P = new ProjectorDirect<T>(&A, comm, comm.fn != nullptr);
  P->Init();                                                           // pogs.cpp:85
  const double t1 = now_s();
  SolveArgs<T> arg{rho, abs_tol, rel_tol, max_iter, verbose, adaptive_rho != 0, gap_stop != 0};
  int st = AdmmSolve<T>(&A, P, make_objs(m, f_a, f_b, f_c, f_d, f_e, f_h),
                        make_objs(n, g_a, g_b, g_c, g_d, g_e, g_h), de.data(), nrmA, m_glob,
                        comm, arg, x, y, l, static_cast<T *>(nullptr), optval, final_iter, info);
  const double t2 = now_s();
  if (info) {
    info->nrmA = nrmA;
    info->norm_est_iters = kpow;
    info->t_init = t1 - t0;
    info->t_loop = t2 - t1;
  }
  fill_info_de(info, de.data(), m, n);
  delete P;
  return st;
}

// src/interface_c/pogs_c.cpp:57-108 (sparse => CGLS projector, :69-73).
template <typename T>
int PogsSparse(int ord, size_t m, size_t n, size_t nnz, const T *data, const int *ptr,
               const int *ind, const T *f_a, const T *f_b, const T *f_c, const T *f_d,
               const T *f_e, const int *f_h, const T *g_a, const T *g_b, const T *g_c,
               const T *g_d, const T *g_e, const int *g_h, T rho, T abs_tol, T rel_tol,
               unsigned max_iter, unsigned verbose, int adaptive_rho, int gap_stop, T *x,
               T *y, T *l, T *optval, unsigned *final_iter, OracleInfo *info, size_t m_glob = 0,
               const Comm &none = Comm()) {
  const double t0 = now_s();
  if (m_glob == 0) m_glob = m;
  SparseOp<T> A(ord == 1, m, n, nnz, data, ptr, ind);
  std::vector<T> de(m + n, 0);
  EquilSparse<T>(&A, de.data(), de.data() + m, m_glob, none);
  unsigned kpow = 0;
  T nrmA = Norm2Est<T>(&A, none, &kpow);
  ProjectorCgls<T> P(&A, none);
  const double t1 = now_s();
  SolveArgs<T> arg{rho, abs_tol, rel_tol, max_iter, verbose, adaptive_rho != 0, gap_stop != 0};
  int st = AdmmSolve<T>(&A, &P, make_objs(m, f_a, f_b, f_c, f_d, f_e, f_h),
                        make_objs(n, g_a, g_b, g_c, g_d, g_e, g_h), de.data(), nrmA, m_glob, none,
                        arg, x, y, l, static_cast<T *>(nullptr), optval, final_iter, info);
  const double t2 = now_s();
  if (info) {
    info->nrmA = nrmA;
    info->norm_est_iters = kpow;
    info->t_init = t1 - t0;
    info->t_loop = t2 - t1;
  }
  fill_info_de(info, de.data(), m, n);
  return st;
}

}  // namespace

// =============================================================================
// C entry points.
// =============================================================================
extern "C" {

#define ORACLE_DENSE(NAME, T)                                                              \
  int NAME(int ord, size_t m, size_t n, const T *A, const T *f_a, const T *f_b,            \
           const T *f_c, const T *f_d, const T *f_e, const int *f_h, const T *g_a,         \
           const T *g_b, const T *g_c, const T *g_d, const T *g_e, const int *g_h, T rho,  \
           T abs_tol, T rel_tol, unsigned max_iter, unsigned verbose, int adaptive_rho,    \
           int gap_stop, T *x, T *y, T *l, T *optval, unsigned *final_iter,                \
           OracleInfo *info) {                                                             \
    Comm none;                                                                             \
    return PogsDense<T>(ord, m, n, A, f_a, f_b, f_c, f_d, f_e, f_h, g_a, g_b, g_c, g_d,    \
                        g_e, g_h, rho, abs_tol, rel_tol, max_iter, verbose, adaptive_rho,  \
                        gap_stop, x, y, l, optval, final_iter, info, m, none,              \
                        info ? info->use_cgls : 0);                                        \
  }
ORACLE_DENSE(OraclePogsD, double)
ORACLE_DENSE(OraclePogsS, float)

#define ORACLE_SHARD(NAME, T)                                                              \
  int NAME(size_t m_local, size_t m_global, size_t n, const T *A, const T *f_a,            \
           const T *f_b, const T *f_c, const T *f_d, const T *f_e, const int *f_h,         \
           const T *g_a, const T *g_b, const T *g_c, const T *g_d, const T *g_e,           \
           const int *g_h, T rho, T abs_tol, T rel_tol, unsigned max_iter,                 \
           unsigned verbose, int adaptive_rho, int gap_stop, T *x, T *y, T *l, T *optval,  \
           unsigned *final_iter, OracleInfo *info, oracle_allreduce_fn fn, void *ctx) {    \
    Comm c;                                                                                \
    c.fn = fn;                                                                             \
    c.ctx = ctx;                                                                           \
    return PogsDense<T>(1, m_local, n, A, f_a, f_b, f_c, f_d, f_e, f_h, g_a, g_b, g_c,     \
                        g_d, g_e, g_h, rho, abs_tol, rel_tol, max_iter, verbose,           \
                        adaptive_rho, gap_stop, x, y, l, optval, final_iter, info,         \
                        m_global, c, 0);                                                   \
  }
ORACLE_SHARD(OraclePogsShardD, double)
ORACLE_SHARD(OraclePogsShardS, float)

#define ORACLE_SPARSE(NAME, T)                                                             \
  int NAME(int ord, size_t m, size_t n, size_t nnz, const T *data, const int *ptr,         \
           const int *ind, const T *f_a, const T *f_b, const T *f_c, const T *f_d,         \
           const T *f_e, const int *f_h, const T *g_a, const T *g_b, const T *g_c,         \
           const T *g_d, const T *g_e, const int *g_h, T rho, T abs_tol, T rel_tol,        \
           unsigned max_iter, unsigned verbose, int adaptive_rho, int gap_stop, T *x,      \
           T *y, T *l, T *optval, unsigned *final_iter, OracleInfo *info) {                \
    return PogsSparse<T>(ord, m, n, nnz, data, ptr, ind, f_a, f_b, f_c, f_d, f_e, f_h,     \
                         g_a, g_b, g_c, g_d, g_e, g_h, rho, abs_tol, rel_tol, max_iter,    \
                         verbose, adaptive_rho, gap_stop, x, y, l, optval, final_iter,     \
                         info);                                                            \
  }
ORACLE_SPARSE(OraclePogsSparseD, double)
ORACLE_SPARSE(OraclePogsSparseS, float)

// Row-sharded CSR + CGLS: this rank holds m_local consecutive rows.
#define ORACLE_SPARSE_SHARD(NAME, T)                                                       \
  int NAME(size_t m_local, size_t m_global, size_t n, size_t nnz, const T *data,           \
           const int *ptr, const int *ind, const T *f_a, const T *f_b, const T *f_c,       \
           const T *f_d, const T *f_e, const int *f_h, const T *g_a, const T *g_b,         \
           const T *g_c, const T *g_d, const T *g_e, const int *g_h, T rho, T abs_tol,     \
           T rel_tol, unsigned max_iter, unsigned verbose, int adaptive_rho, int gap_stop, \
           T *x, T *y, T *l, T *optval, unsigned *final_iter, OracleInfo *info,            \
           oracle_allreduce_fn fn, void *ctx) {                                            \
    Comm comm;                                                                             \
    comm.fn = fn;                                                                          \
    comm.ctx = ctx;                                                                        \
    return PogsSparse<T>(1, m_local, n, nnz, data, ptr, ind, f_a, f_b, f_c, f_d, f_e, f_h, \
                         g_a, g_b, g_c, g_d, g_e, g_h, rho, abs_tol, rel_tol, max_iter,    \
                         verbose, adaptive_rho, gap_stop, x, y, l, optval, final_iter,     \
                         info, m_global, comm);                                            \
  }
ORACLE_SPARSE_SHARD(OraclePogsSparseShardD, double)
ORACLE_SPARSE_SHARD(OraclePogsSparseShardS, float)

// Element-wise prox / function evaluation with SoA coefficients.
#define ORACLE_PROX(NAME, FNAME, T)                                                        \
  void NAME(size_t n, const int *h, const T *a, const T *b, const T *c, const T *d,        \
            const T *e, T rho, const T *in, T *out) {                                      \
    auto f = make_objs(n, a, b, c, d, e, h);                                               \
    ProxEvalVec(f, rho, in, out);                                                          \
  }                                                                                        \
  double FNAME(size_t n, const int *h, const T *a, const T *b, const T *c, const T *d,     \
               const T *e, const T *in) {                                                  \
    auto f = make_objs(n, a, b, c, d, e, h);                                               \
    return static_cast<double>(FuncEvalVec(f, in));                                        \
  }
ORACLE_PROX(OracleProxEvalD, OracleFuncEvalD, double)
ORACLE_PROX(OracleProxEvalS, OracleFuncEvalS, float)

// v_out[i] = ProjSubgradEval(f_i, v_in[i], x_in[i])   (prox_lib.h:538-546)
#define ORACLE_PROJSUB(NAME, T)                                                            \
  void NAME(size_t n, const int *h, const T *a, const T *b, const T *c, const T *d,        \
            const T *e, const T *x_in, const T *v_in, T *v_out) {                          \
    auto f = make_objs(n, a, b, c, d, e, h);                                               \
    for (size_t i = 0; i < n; ++i) v_out[i] = ProjSubgradEval(f[i], v_in[i], x_in[i]);     \
  }
ORACLE_PROJSUB(OracleProjSubgradEvalD, double)
ORACLE_PROJSUB(OracleProjSubgradEvalS, float)

// Raw prox_h(v, rho) (the functions tests/test_proximal.cpp of the reference pins).
double OracleProxRawD(int h, double v, double rho) {
  FunctionObj<double> f{h, 1, 0, 1, 0, 0};
  return ProxEval(f, v, rho);
}
float OracleProxRawS(int h, float v, float rho) {
  FunctionObj<float> f{h, 1, 0, 1, 0, 0};
  return ProxEval(f, v, rho);
}

void OracleRandS(float *x, size_t n) { rand_uniform(x, n); }
void OracleRandD(double *x, size_t n) { rand_uniform(x, n); }

// Projection onto {y = Ax} for a dense row-major A as-is (no equilibration):
// used to test projector kernels.  use_cgls selects ProjectorCgls.
#define ORACLE_PROJECT(NAME, T)                                                            \
  void NAME(size_t m, size_t n, const T *A, const T *x0, const T *y0, T s, T tol,          \
            int use_cgls, T *x, T *y) {                                                    \
    DenseOp<T> op(true, m, n, A);                                                          \
    Comm none;                                                                             \
    if (use_cgls) {                                                                        \
      ProjectorCgls<T> P(&op);                                                             \
      P.Project(x0, y0, s, x, y, tol);                                                     \
    } else {                                                                               \
      ProjectorDirect<T> P(&op, none);                                                     \
      P.Init();                                                                            \
      P.Project(x0, y0, s, x, y, tol);                                                     \
    }                                                                                      \
  }
ORACLE_PROJECT(OracleProjectD, double)
ORACLE_PROJECT(OracleProjectS, float)

// Equilibration only: returns the scaled matrix, d, e and the norm estimate.
#define ORACLE_EQUIL(NAME, T)                                                              \
  void NAME(size_t m, size_t n, T *A_inout, T *d, T *e, T *nrmA, unsigned *kpow) {         \
    DenseOp<T> op(true, m, n, A_inout);                                                    \
    Comm none;                                                                             \
    std::vector<T> de(m + n);                                                              \
    EquilDense<T>(&op, de.data(), de.data() + m, m, none);                                 \
    *nrmA = Norm2Est<T>(&op, none, kpow);                                                  \
    std::memcpy(A_inout, op.a.data(), m * n * sizeof(T));                                  \
    std::memcpy(d, de.data(), m * sizeof(T));                                              \
    std::memcpy(e, de.data() + m, n * sizeof(T));                                          \
  }
ORACLE_EQUIL(OracleEquilD, double)
ORACLE_EQUIL(OracleEquilS, float)

}  // extern "C"
