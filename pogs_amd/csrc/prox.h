// Device-side proximal-operator / function library.
//
// Behavioural spec: the reference's ProxEval / FuncEval
// (src/include/prox_lib.h:83-230, 240-349) and scalar helpers
// (src/include/prox_tools.h:49-149).  Written for the GPU: coefficients are
// struct-of-arrays (the ABI already supplies them that way,
// src/interface_c/pogs_c.h:75-91), every function is branch-light and the
// function id is a per-element runtime value (wave-uniform in every solve_*
// problem, so the switch does not diverge in practice).
#pragma once
#include <hip/hip_runtime.h>

namespace pogs_amd {

enum Fn : int { kAbs = 0, kExp, kHuber, kIdentity, kIndBox01, kIndEq0, kIndGe0, kIndLe0,
                kLogistic, kMaxNeg0, kMaxPos0, kNegEntr, kNegLog, kRecipr, kSquare, kZero };

// Struct-of-arrays view of m (or n) function objects c*h(a*v-b) + d*v + e*v^2/2.
template <typename T>
struct FnView {
  const int *h;
  const T *a, *b, *c, *d, *e;
};

namespace dev {

__device__ __forceinline__ float Exp(float x) { return expf(x); }
__device__ __forceinline__ double Exp(double x) { return exp(x); }
__device__ __forceinline__ float Log(float x) { return logf(x); }
__device__ __forceinline__ double Log(double x) { return log(x); }
__device__ __forceinline__ float Sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double Sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float Abs(float x) { return fabsf(x); }
__device__ __forceinline__ double Abs(double x) { return fabs(x); }
__device__ __forceinline__ float Max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double Max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float Min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double Min(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float Pow(float a, float b) { return powf(a, b); }
__device__ __forceinline__ double Pow(double a, double b) { return pow(a, b); }
__device__ __forceinline__ float Acos(float a) { return acosf(a); }
__device__ __forceinline__ double Acos(double a) { return acos(a); }
__device__ __forceinline__ float Cos(float a) { return cosf(a); }
__device__ __forceinline__ double Cos(double a) { return cos(a); }

// a * b rounded on its own: the empty asm makes the product opaque, so the compiler cannot contract
// it with a following add / subtract into one fused multiply-add (-ffp-contract=fast is the default)
__device__ __forceinline__ float MulRn(float a, float b) {
  float p = a * b;
  asm volatile("" : "+v"(p));
  return p;
}
__device__ __forceinline__ double MulRn(double a, double b) {
  double p = a * b;
  asm volatile("" : "+v"(p));
  return p;
}

template <typename T> __device__ __forceinline__ T BisectTol();   // prox_tools.h:57-62
template <> __device__ __forceinline__ float BisectTol<float>() { return 1e-5f; }
template <> __device__ __forceinline__ double BisectTol<double>() { return 1e-10; }

// ---- scalar root finders behind Exp / NegEntr / Recipr / Logistic ---------------------------
// Behavioural spec only (the parity tests compare every one of them with the reference's values:
// tests/golden/prox_table.npz is generated from the reference header): same start values, same
// number of refinement steps and the same stopping thresholds as src/include/prox_tools.h:98-149
// and prox_lib.h:131-170, so the iterates agree to rounding.

// Start value for w e^w = e^x: the asymptotic series for large x, a series in sqrt(2 (e^(x+1) + 1))
// around the branch point for negative x, x itself in between (less log x beyond x = log 3).
struct LambertSeed {
  double w;
  bool final;   // the asymptotic branch is returned as is
};
__device__ __forceinline__ LambertSeed lambert_seed(double x) {
  if (x > 100.0) {
    const double lx = log(x);
    return {x - 0.36962844 - 0.97284858 * lx + 1.3437973 / lx, true};
  }
  if (x < 0.0) {
    const double q = sqrt(2.0 * (exp(x + 1.0) + 1.0));
    const double series = 1.0 + q * (q * (11.0 / 72.0) - 1.0 / 3.0);   // Horner form of 1 - q/3 + 11 q^2 / 72
    return {q * series - 1.0, false};
  }
  return {x > 1.098612288668110 ? x - log(x) : x, false};
}
// W(exp(x)), principal branch, evaluated in double (callers cast as prox_lib.h:88-100 does):
// at most ten Halley steps on F(w) = w e^w - e^x.
__device__ inline double LambertWExp(double x) {
  const LambertSeed seed = lambert_seed(x);
  if (seed.final) return seed.w;
  const double target = exp(x);
  double w = seed.w;
  for (int step = 0; step < 10; ++step) {
    const double ew = exp(w), w1 = w + 1.0;
    const double F = w * ew - target;
    const double delta = F / (ew * w1 - 0.5 * (w1 + 1.0) * F / w1);   // Halley: F / (F' - F F'' / (2 F'))
    w -= delta;
    if (fabs(delta) < 4e-16 * (1.0 + fabs(w))) break;
  }
  return w;
}

// The positive root of x^3 + p x^2 + q x + r: substitute x = t - p/3 to get t^3 + 3 A t + 2 B = 0,
// then Cardano (one real root) or the trigonometric form (three real roots, the largest).
template <typename T>
__device__ inline T CubicSolve(T p, T q, T r) {
  const T third = static_cast<T>(1) / 3;
  const T shift = p / 3;
  const T shift2 = shift * shift;
  const T A = q / 3 - shift2;
  const T B = shift2 * shift - shift * q / 2 + r / 2;
  const T A3 = A * A * A;
  const T disc = A3 + B * B;
  if (disc >= 0) {
    const T root = Pow(Sqrt(disc) - B, third);
    return root - A / root - shift;
  }
  const T mod = Sqrt(-A3);
  const T angle = Acos(-B / mod);
  const T rad = Pow(mod, third);
  return (rad - A / rad) * Cos(angle / 3) - shift;
}

// x with sigma(x) + rho (x - v) = 0 (the prox of log(1 + e^x)).  The root lies in [v - 1/rho, v]:
// five Newton steps clipped to the bracket, then the bracket is shrunk with the fixed-point
// residual until it is narrower than BisectTol (at most 100 rounds).
template <typename T>
__device__ inline T ProxLogistic(T v, T rho) {
  const T one = 1;
  const T step = one / rho;
  T lo = v - step, hi = v;
  // start: the two asymptotes, or the tangent-line model sigma(x) ~ 0.5 + 0.2 x in between
  T x = (v < static_cast<T>(-2.5)) ? v
        : (v > static_cast<T>(2.5) + step) ? lo
        : (rho * v - static_cast<T>(0.5)) / (static_cast<T>(0.2) + rho);
#pragma unroll 1
  for (int it = 0; it < 5; ++it) {
    const T sig = one / (one + Exp(-x));
    const T resid = sig + rho * (x - v);
    const T slope = sig * (one - sig) + rho;
    if (resid < 0) lo = x; else hi = x;
    x = Max(Min(x - resid / slope, hi), lo);
  }
#pragma unroll 1
  for (int it = 0; it < 100 && hi - lo > BisectTol<T>(); ++it) {
    const T gap = one / (rho * (one + Exp(-x))) + (x - v);   // residual / rho
    if (gap > 0) {
      lo = Max(lo, x - gap);
      hi = x;
    } else {
      hi = Min(hi, x - gap);
      lo = x;
    }
    x = (hi + lo) / 2;
  }
  return x;
}

// prox_h(v, rho) for the 16 base functions (prox_lib.h:83-204).
template <typename T>
__device__ inline T ProxBase(int h, T v, T rho) {
  const T zero = 0;
  switch (h) {
    case kAbs: return Max(zero, v - 1 / rho) - Max(zero, -(v + 1 / rho));
    case kNegEntr:
      return static_cast<T>(LambertWExp(static_cast<double>((rho * v - 1) + Log(rho)))) / rho;
    case kExp: return v - static_cast<T>(LambertWExp(static_cast<double>(v - Log(rho))));
    case kHuber:
      return Abs(v) < 1 + 1 / rho ? v * rho / (1 + rho) : v - (v >= 0 ? static_cast<T>(1) : static_cast<T>(-1)) / rho;
    case kIdentity: return v - 1 / rho;
    case kIndBox01: return v <= 0 ? zero : (v >= 1 ? static_cast<T>(1) : v);
    case kIndEq0: return zero;
    case kIndGe0: return v <= 0 ? zero : v;
    case kIndLe0: return v >= 0 ? zero : v;
    case kLogistic: return ProxLogistic(v, rho);
    case kMaxNeg0: {
      const T z = v >= 0 ? v : zero;
      return v + 1 / rho <= 0 ? v + 1 / rho : z;
    }
    case kMaxPos0: {
      const T z = v <= 0 ? v : zero;
      return v >= 1 / rho ? v - 1 / rho : z;
    }
    case kNegLog: return (v + Sqrt(v * v + 4 / rho)) / 2;
    case kRecipr: return CubicSolve(-Max(v, zero), zero, -1 / rho);
    case kSquare: return rho * v / (1 + rho);
    case kZero:
    default: return v;
  }
}

// Prox of c*h(a*v-b) + d*v + e*v^2/2 (prox_lib.h:207-230).
template <typename T>
__device__ inline T ProxEval(int h, T a, T b, T c, T d, T e, T v, T rho) {
  v = a * (v * rho - d) / (e + rho) - b;
  rho = (e + rho) / (c * a * a);
  v = ProxBase(h, v, rho);
  return (v + b) / a;
}

// Subset used inside the one-pass streaming kernel, where registers are scarce:
// every base function whose prox is a few arithmetic operations.  The host
// only selects that kernel when all f_i are in this set (is_cheap_prox).
template <typename T>
__device__ __forceinline__ T ProxEvalCheap(int h, T a, T b, T c, T d, T e, T v, T rho) {
  v = a * (v * rho - d) / (e + rho) - b;
  rho = (e + rho) / (c * a * a);
  const T zero = 0, ir = 1 / rho;
  T r = v;
  switch (h) {
    case kAbs: r = Max(zero, v - ir) - Max(zero, -(v + ir)); break;
    case kHuber: r = Abs(v) < 1 + ir ? v * rho / (1 + rho) : v - (v >= 0 ? ir : -ir); break;
    case kIdentity: r = v - ir; break;
    case kIndBox01: r = v <= 0 ? zero : (v >= 1 ? static_cast<T>(1) : v); break;
    case kIndEq0: r = zero; break;
    case kIndGe0: r = v <= 0 ? zero : v; break;
    case kIndLe0: r = v >= 0 ? zero : v; break;
    case kMaxNeg0: r = v + ir <= 0 ? v + ir : (v >= 0 ? v : zero); break;
    case kMaxPos0: r = v >= ir ? v - ir : (v <= 0 ? v : zero); break;
    case kNegLog: r = (v + Sqrt(v * v + 4 / rho)) / 2; break;
    case kSquare: r = rho * v / (1 + rho); break;
    default: break;  // kZero
  }
  return (r + b) / a;
}

// c*h(a*x-b) + d*x + e*x^2/2 (prox_lib.h:240-349).
template <typename T>
__device__ inline T FuncEval(int h, T a, T b, T c, T d, T e, T x) {
  const T dx = d * x;
  const T ex = e * x * x / 2;
  const T zero = 0;
  x = a * x - b;
  switch (h) {
    case kAbs: x = Abs(x); break;
    case kNegEntr: x = x <= 0 ? zero : x * Log(x); break;
    case kExp: x = Exp(x); break;
    case kHuber: {
      const T xabs = Abs(x);
      x = xabs < static_cast<T>(1) ? xabs * xabs / 2 : xabs - static_cast<T>(0.5);
      break;
    }
    case kIdentity: break;
    case kIndBox01: case kIndEq0: case kIndGe0: case kIndLe0: x = zero; break;
    case kLogistic: x = Log(1 + Exp(x)); break;
    case kMaxNeg0: x = Max(zero, -x); break;
    case kMaxPos0: x = Max(zero, x); break;
    case kNegLog: x = -Log(Max(zero, x)); break;
    case kRecipr: x = 1 / Max(zero, x); break;
    case kSquare: x = x * x / 2; break;
    case kZero:
    default: x = zero; break;
  }
  return c * x + dx + ex;
}

// ---- projection onto the subdifferential (prox_lib.h:359-493) --------------------------------
// ProjSubgrad{h}(v, x): the point of the subdifferential of h at x closest to v.  Three families:
//   * differentiable h: the gradient, whatever v is (values as the reference's, including its
//     +1/x^2 for kRecipr);
//   * one kink at 0 with slopes lo (left) and hi (right) -- |x|, max(0, -x), max(0, x): the slope
//     away from the kink, v clipped to [lo, hi] on it;
//   * indicators of an interval: the normal cone -- 0 inside, v's non-positive part at the lower
//     end, its non-negative part at the upper end (kIndEq0: the whole line, i.e. v).
template <typename T>
__device__ __forceinline__ T kink_subgrad(T x, T lo, T hi, T v) {
  if (x < 0) return lo;
  if (x > 0) return hi;
  return Max(lo, Min(hi, v));
}
template <typename T>
__device__ inline T ProjSubgradBase(int h, T v, T x) {
  const T zero = 0, one = 1;
  switch (h) {
    case kAbs: return kink_subgrad(x, -one, one, v);
    case kMaxNeg0: return kink_subgrad(x, -one, zero, v);
    case kMaxPos0: return kink_subgrad(x, zero, one, v);
    case kIndGe0: return x <= zero ? Min(zero, v) : zero;
    case kIndLe0: return x >= zero ? Max(zero, v) : zero;
    case kIndBox01: return x <= zero ? Min(zero, v) : (x >= one ? Max(zero, v) : zero);
    case kIndEq0: return v;
    case kNegEntr: return -Log(x) - one;
    case kExp: return Exp(x);
    case kHuber: return Max(-one, Min(one, x));
    case kIdentity: return one;
    case kLogistic: {
      const T ex = Exp(x);
      return ex / (one + ex);
    }
    case kNegLog: return -one / x;
    case kRecipr: return one / (x * x);
    case kSquare: return x;
    case kZero:
    default: return zero;
  }
}
// The same for c*h(a*x-b) + d*x + e*x^2/2 (chain rule, prox_lib.h:468-493); c, e are expected
// clamped to >= 0 as FunctionObj's constructor does.
template <typename T>
__device__ inline T ProjSubgradEval(int h, T a, T b, T c, T d, T e, T v, T x) {
  const T lin = d + e * x;
  if (a == static_cast<T>(0) || c == static_cast<T>(0)) return lin;
  const T ac = a * c;
  // a x - b decides on which side of a kink x lies: the product is rounded on its own, as the
  // reference's host arithmetic does (no fused multiply-add), so points ON a kink stay on it
  const T inner = ProjSubgradBase(h, static_cast<T>(1) / ac * (v - lin), MulRn(a, x) - b);
  return ac * inner + lin;
}

}  // namespace dev

inline bool is_cheap_prox(int h) {
  return h != kExp && h != kLogistic && h != kNegEntr && h != kRecipr;
}

}  // namespace pogs_amd
