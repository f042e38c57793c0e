// extern "C" entry points (include/pogs_amd.h).  No exception crosses this file.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "engine.h"
#include "prox.h"
#include "stream.h"
#include "vec_kernels.h"

namespace pogs_amd {

// One factory per streaming shape and arithmetic type (dense_plan.hip, built once per entry of
// POGS_STREAM_PLANS plus the windowed form): the shape follows from the stored row length alone,
// so it is chosen here and only that translation unit's code object is ever loaded.
#define POGS_DECLARE_PLAN(TPB_, NV_)                                                                               \
  SolverBase *make_dense_solver_f32_p##TPB_##_##NV_(int, size_t, size_t, const void *, int, const PogsAmdOptions *, \
                                                    const PogsAmdDist *);                                          \
  SolverBase *make_dense_solver_f64_p##TPB_##_##NV_(int, size_t, size_t, const void *, int, const PogsAmdOptions *, \
                                                    const PogsAmdDist *);
POGS_STREAM_PLANS(POGS_DECLARE_PLAN)
#undef POGS_DECLARE_PLAN
SolverBase *make_dense_solver_f32_xl(int, size_t, size_t, const void *, int, const PogsAmdOptions *, const PogsAmdDist *);
SolverBase *make_dense_solver_f64_xl(int, size_t, size_t, const void *, int, const PogsAmdOptions *, const PogsAmdDist *);

template <typename T>
inline StreamPlan dense_plan_for(size_t m, size_t n, const PogsAmdOptions *opt, const PogsAmdDist *dist) {
  // the stored row length, as DenseSolver's constructor derives it
  const size_t m_global = (dist && dist->world >= 1) ? dist->m_global : m;
  const bool tall = m_global > n;
  const bool cgls = opt && opt->projector == POGS_AMD_PROJ_CGLS;
  const size_t stored_cols = (!tall && !cgls) ? m : n;
  return make_stream_plan<T>(static_cast<int>(round_up(stored_cols, Vec16<T>::N)), 256);
}

inline SolverBase *make_dense_solver(int dtype, int ord, size_t m, size_t n, const void *A, int mem,
                                     const PogsAmdOptions *opt, const PogsAmdDist *dist) {
  POGS_CHECK(dtype == POGS_AMD_F32 || dtype == POGS_AMD_F64, "unknown dtype");
  POGS_CHECK(m > 0 && n > 0 && m < (1u << 31) && n < (1u << 31), "bad dimensions");
  const bool f32 = dtype == POGS_AMD_F32;
  const StreamPlan p = f32 ? dense_plan_for<float>(m, n, opt, dist) : dense_plan_for<double>(m, n, opt, dist);
  POGS_CHECK(p.ok, "no streaming shape for this matrix");
  if (p.xl) return f32 ? make_dense_solver_f32_xl(ord, m, n, A, mem, opt, dist) : make_dense_solver_f64_xl(ord, m, n, A, mem, opt, dist);
#define POGS_PICK_PLAN(TPB_, NV_)                                                          \
  if (p.tpb == TPB_ && p.nv == NV_)                                                        \
    return f32 ? make_dense_solver_f32_p##TPB_##_##NV_(ord, m, n, A, mem, opt, dist)       \
               : make_dense_solver_f64_p##TPB_##_##NV_(ord, m, n, A, mem, opt, dist);
  POGS_STREAM_PLANS(POGS_PICK_PLAN)
#undef POGS_PICK_PLAN
  throw Error("no dense solver for this streaming shape");
}
SolverBase *make_sparse_solver(int dtype, int ord, size_t m, size_t n, size_t nnz, const void *data,
                               const int *ptr, const int *ind, int mem, const PogsAmdOptions *opt,
                               const PogsAmdDist *dist);

// gsl::rand (src/cpu/include/gsl/gsl_rand.h:8-16): a fresh
// std::default_random_engine (libstdc++: minstd_rand0, x <- 16807 x mod 2^31-1,
// seed 1) feeding uniform_real_distribution<T>(0,1) = generate_canonical: one
// draw per float, two per double.  Restated so the Norm2Est start vector does
// not depend on the C++ standard library in use.
namespace {
struct MinStd0 {
  uint64_t s = 1;
  uint32_t next() {
    s = (s * 16807ull) % 2147483647ull;
    return static_cast<uint32_t>(s);
  }
};
}  // namespace
void rand_uniform_host(float *x, size_t n) {
  MinStd0 g;
  const float r = static_cast<float>(2147483646.0L);
  for (size_t i = 0; i < n; ++i) {
    float v = static_cast<float>(g.next() - 1u) / r;
    if (v >= 1.0f) v = std::nextafter(1.0f, 0.0f);
    x[i] = v;
  }
}
void rand_uniform_host(double *x, size_t n) {
  MinStd0 g;
  const double r = 2147483646.0;
  for (size_t i = 0; i < n; ++i) {
    const double lo = static_cast<double>(g.next() - 1u);
    const double hi = static_cast<double>(g.next() - 1u);
    double v = (lo + hi * r) / (r * r);
    if (v >= 1.0) v = std::nextafter(1.0, 0.0);
    x[i] = v;
  }
}

namespace {

thread_local std::string g_last_error;

// `h` (may be null): the handle the entry point works on.  An error inside a row-sharded solve
// aborts the handle's communicator (SolverBase::on_error), so that the peers' next collective
// fails or times out instead of waiting for this rank for ever.
template <typename F>
int guarded(F &&fn, PogsAmdSolver *h = nullptr) {
  auto failed = [&](const char *what) {
    g_last_error = what;
    std::fprintf(stderr, "pogs_amd: %s\n", what);
    try {
      if (h && h->impl) h->impl->on_error();
    } catch (...) {}
    return POGS_ERROR;
  };
  try {
    g_last_error.clear();
    if (h && h->impl) h->impl->on_entry();
    return fn();
  } catch (const std::exception &e) {
    return failed(e.what());
  } catch (...) {
    return failed("unknown error");
  }
}

SolveParams make_params(double rho, double abs_tol, double rel_tol, unsigned max_iter, unsigned verbose,
                        int adaptive_rho, int gap_stop) {
  SolveParams p;
  p.rho = rho; p.abs_tol = abs_tol; p.rel_tol = rel_tol;
  p.max_iter = max_iter == 0 ? 1 : max_iter;
  p.verbose = verbose;
  p.adaptive_rho = adaptive_rho != 0;
  p.gap_stop = gap_stop != 0;
  return p;
}

// The array entry points (the reference's four and PogsAmdSolve / PogsAmdBeginRun) take TWELVE arrays: a null one is a
// caller's mistake there, not a broadcast field (that reading belongs to the *Fn entry points alone) -- without the
// check the solve would run on the defaults (a = c = 1, b = d = e = 0, h = kZero) and return a plausible wrong answer.
FnHost fn_arrays(const void *a, const void *b, const void *c, const void *d, const void *e, const int *h, const char *which) {
  if (!(a && b && c && d && e && h))
    throw Error(std::string("null coefficient array in the description of ") + which +
                " (a, b, c, d, e, h must all be given; broadcast fields: PogsAmdSolveFn)");
  return FnHost{a, b, c, d, e, h};
}

// One-shot dense solve: src/interface_c/pogs_c.cpp:9-55.
template <typename T>
int pogs_dense(enum ORD ord, size_t m, size_t n, const T *A, const T *f_a, const T *f_b, const T *f_c,
               const T *f_d, const T *f_e, const enum FUNCTION *f_h, const T *g_a, const T *g_b, const T *g_c,
               const T *g_d, const T *g_e, const enum FUNCTION *g_h, T rho, T abs_tol, T rel_tol,
               unsigned max_iter, unsigned verbose, int adaptive_rho, int gap_stop, T *x, T *y, T *l, T *optval,
               unsigned *final_iter) {
  return guarded([&]() {
    const FnHost f = fn_arrays(f_a, f_b, f_c, f_d, f_e, reinterpret_cast<const int *>(f_h), "f");
    const FnHost g = fn_arrays(g_a, g_b, g_c, g_d, g_e, reinterpret_cast<const int *>(g_h), "g");
    std::unique_ptr<SolverBase> s(make_dense_solver(sizeof(T) == 4 ? POGS_AMD_F32 : POGS_AMD_F64, ord, m, n, A,
                                                    POGS_AMD_HOST, nullptr, nullptr));
    double ov = 0;
    const int st = s->solve(f, g, make_params(rho, abs_tol, rel_tol, max_iter, verbose, adaptive_rho, gap_stop),
                            x, y, l, nullptr, &ov, final_iter);
    *optval = static_cast<T>(ov);
    return st;
  });
}

// One-shot sparse solve: src/interface_c/pogs_c.cpp:57-108.
template <typename T>
int pogs_sparse(enum ORD ord, size_t m, size_t n, size_t nnz, const T *data, const int *ptr, const int *ind,
                const T *f_a, const T *f_b, const T *f_c, const T *f_d, const T *f_e, const enum FUNCTION *f_h,
                const T *g_a, const T *g_b, const T *g_c, const T *g_d, const T *g_e, const enum FUNCTION *g_h,
                T rho, T abs_tol, T rel_tol, unsigned max_iter, unsigned verbose, int adaptive_rho, int gap_stop,
                T *x, T *y, T *l, T *optval, unsigned *final_iter) {
  return guarded([&]() {
    const FnHost f = fn_arrays(f_a, f_b, f_c, f_d, f_e, reinterpret_cast<const int *>(f_h), "f");
    const FnHost g = fn_arrays(g_a, g_b, g_c, g_d, g_e, reinterpret_cast<const int *>(g_h), "g");
    std::unique_ptr<SolverBase> s(make_sparse_solver(sizeof(T) == 4 ? POGS_AMD_F32 : POGS_AMD_F64, ord, m, n, nnz,
                                                     data, ptr, ind, POGS_AMD_HOST, nullptr, nullptr));
    double ov = 0;
    const int st = s->solve(f, g, make_params(rho, abs_tol, rel_tol, max_iter, verbose, adaptive_rho, gap_stop),
                            x, y, l, nullptr, &ov, final_iter);
    *optval = static_cast<T>(ov);
    return st;
  });
}

template <typename T>
void prox_eval_host(size_t n, const int *h, const void *a, const void *b, const void *c, const void *d,
                    const void *e, double rho, const void *in, void *out, double *fsum, const void *xin = nullptr) {
  POGS_CHECK(n < (1u << 31), "n too large");
  hipStream_t s = nullptr;
  const int cnt = static_cast<int>(n);
  FnBuf<T> fb;
  fb.alloc(n);
  DevBuf<T> vin(n), vout(n);
  auto up = [&](void *dst, const void *src, size_t bytes) {
    POGS_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
  };
  up(fb.h.p, h, n * sizeof(int));
  up(fb.a.p, a, n * sizeof(T)); up(fb.b.p, b, n * sizeof(T)); up(fb.c.p, c, n * sizeof(T));
  up(fb.d.p, d, n * sizeof(T)); up(fb.e.p, e, n * sizeof(T));
  up(vin.p, in, n * sizeof(T));
  {
    FnHost fh{a, b, c, d, e, h};
    warn_negative_coeffs<T>(fh, n);
  }
  // clamp c, e >= 0 as FunctionObj's constructor does (prox_lib.h:62-69): scale by 1.
  DevBuf<T> ones(n);
  launch_fill<T>(ones.p, static_cast<T>(1), n, s);
  launch_scale_objective<T>(fb.view(), fb.a.p, fb.c.p, fb.d.p, fb.e.p, ones.p, cnt, false, s);
  if (out && xin) {   // ProjSubgradEval: `in` is v, `xin` the point x
    DevBuf<T> xv(n);
    up(xv.p, xin, n * sizeof(T));
    launch_proj_subgrad<T>(cnt, fb.view(), xv.p, vin.p, vout.p, s);
    POGS_HIP_CHECK(hipMemcpyAsync(out, vout.p, n * sizeof(T), hipMemcpyDeviceToHost, s));
    POGS_HIP_CHECK(hipStreamSynchronize(s));   // xv is freed at scope exit
  } else if (out) {
    launch_prox_eval<T>(cnt, fb.view(), static_cast<T>(rho), vin.p, vout.p, s);
    POGS_HIP_CHECK(hipMemcpyAsync(out, vout.p, n * sizeof(T), hipMemcpyDeviceToHost, s));
  }
  if (fsum) {
    const int blocks = vec_blocks(cnt);
    DevBuf<double> part(blocks), tot(1);
    launch_func_eval<T>(cnt, fb.view(), vin.p, part.p, s);
    SumJob j{part.p, blocks, 1, tot.p};
    launch_sum_jobs(&j, 1, s);
    POGS_HIP_CHECK(hipMemcpyAsync(fsum, tot.p, sizeof(double), hipMemcpyDeviceToHost, s));
    POGS_HIP_CHECK(hipStreamSynchronize(s));
  }
  POGS_HIP_CHECK(hipStreamSynchronize(s));
}

}  // namespace
}  // namespace pogs_amd

using namespace pogs_amd;

extern "C" {

int PogsD(enum ORD ord, size_t m, size_t n, const double *A, const double *f_a, const double *f_b,
          const double *f_c, const double *f_d, const double *f_e, const enum FUNCTION *f_h, const double *g_a,
          const double *g_b, const double *g_c, const double *g_d, const double *g_e, const enum FUNCTION *g_h,
          double rho, double abs_tol, double rel_tol, unsigned int max_iter, unsigned int verbose,
          int adaptive_rho, int gap_stop, double *x, double *y, double *l, double *optval,
          unsigned int *final_iter) {
  return pogs_dense<double>(ord, m, n, A, f_a, f_b, f_c, f_d, f_e, f_h, g_a, g_b, g_c, g_d, g_e, g_h, rho, abs_tol,
                            rel_tol, max_iter, verbose, adaptive_rho, gap_stop, x, y, l, optval, final_iter);
}

int PogsS(enum ORD ord, size_t m, size_t n, const float *A, const float *f_a, const float *f_b, const float *f_c,
          const float *f_d, const float *f_e, const enum FUNCTION *f_h, const float *g_a, const float *g_b,
          const float *g_c, const float *g_d, const float *g_e, const enum FUNCTION *g_h, float rho, float abs_tol,
          float rel_tol, unsigned int max_iter, unsigned int verbose, int adaptive_rho, int gap_stop, float *x,
          float *y, float *l, float *optval, unsigned int *final_iter) {
  return pogs_dense<float>(ord, m, n, A, f_a, f_b, f_c, f_d, f_e, f_h, g_a, g_b, g_c, g_d, g_e, g_h, rho, abs_tol,
                           rel_tol, max_iter, verbose, adaptive_rho, gap_stop, x, y, l, optval, final_iter);
}

int PogsSparseD(enum ORD ord, size_t m, size_t n, size_t nnz, const double *data, const int *ptr, const int *ind,
                const double *f_a, const double *f_b, const double *f_c, const double *f_d, const double *f_e,
                const enum FUNCTION *f_h, const double *g_a, const double *g_b, const double *g_c,
                const double *g_d, const double *g_e, const enum FUNCTION *g_h, double rho, double abs_tol,
                double rel_tol, unsigned int max_iter, unsigned int verbose, int adaptive_rho, int gap_stop,
                double *x, double *y, double *l, double *optval, unsigned int *final_iter) {
  return pogs_sparse<double>(ord, m, n, nnz, data, ptr, ind, f_a, f_b, f_c, f_d, f_e, f_h, g_a, g_b, g_c, g_d, g_e,
                             g_h, rho, abs_tol, rel_tol, max_iter, verbose, adaptive_rho, gap_stop, x, y, l, optval,
                             final_iter);
}

int PogsSparseS(enum ORD ord, size_t m, size_t n, size_t nnz, const float *data, const int *ptr, const int *ind,
                const float *f_a, const float *f_b, const float *f_c, const float *f_d, const float *f_e,
                const enum FUNCTION *f_h, const float *g_a, const float *g_b, const float *g_c, const float *g_d,
                const float *g_e, const enum FUNCTION *g_h, float rho, float abs_tol, float rel_tol,
                unsigned int max_iter, unsigned int verbose, int adaptive_rho, int gap_stop, float *x, float *y,
                float *l, float *optval, unsigned int *final_iter) {
  return pogs_sparse<float>(ord, m, n, nnz, data, ptr, ind, f_a, f_b, f_c, f_d, f_e, f_h, g_a, g_b, g_c, g_d, g_e,
                            g_h, rho, abs_tol, rel_tol, max_iter, verbose, adaptive_rho, gap_stop, x, y, l, optval,
                            final_iter);
}

int PogsAmdDistUniqueId(char *out) {
  return guarded([&]() {
    DistComm::unique_id(out);
    return 0;
  });
}

int PogsAmdCreateDense(PogsAmdSolver **out, int dtype, enum ORD ord, size_t m, size_t n, const void *A, int mem,
                       const PogsAmdOptions *opt, const PogsAmdDist *dist) {
  return guarded([&]() {
    *out = nullptr;
    DeviceGuard guard(opt ? opt->device : -1);   // the caller's current device is put back on exit
    std::unique_ptr<PogsAmdSolver> h(new PogsAmdSolver);
    h->impl.reset(make_dense_solver(dtype, ord, m, n, A, mem, opt, dist));
    *out = h.release();
    return 0;
  });
}

int PogsAmdCreateSparse(PogsAmdSolver **out, int dtype, enum ORD ord, size_t m, size_t n, size_t nnz,
                        const void *data, const int *ptr, const int *ind, int mem, const PogsAmdOptions *opt,
                        const PogsAmdDist *dist) {
  return guarded([&]() {
    *out = nullptr;
    DeviceGuard guard(opt ? opt->device : -1);
    std::unique_ptr<PogsAmdSolver> h(new PogsAmdSolver);
    h->impl.reset(make_sparse_solver(dtype, ord, m, n, nnz, data, ptr, ind, mem, opt, dist));
    *out = h.release();
    return 0;
  });
}

int PogsAmdSolve(PogsAmdSolver *s, const void *f_a, const void *f_b, const void *f_c, const void *f_d,
                 const void *f_e, const int *f_h, const void *g_a, const void *g_b, const void *g_c,
                 const void *g_d, const void *g_e, const int *g_h, double rho, double abs_tol, double rel_tol,
                 unsigned int max_iter, unsigned int verbose, int adaptive_rho, int gap_stop, void *x, void *y,
                 void *l, void *mu, double *optval, unsigned int *final_iter) {
  return guarded([&]() {
    POGS_CHECK(s && s->impl, "null solver");
    DeviceGuard guard(s->impl->device());
    const FnHost f = fn_arrays(f_a, f_b, f_c, f_d, f_e, f_h, "f");
    const FnHost g = fn_arrays(g_a, g_b, g_c, g_d, g_e, g_h, "g");
    return s->impl->solve(f, g, make_params(rho, abs_tol, rel_tol, max_iter, verbose, adaptive_rho, gap_stop), x,
                          y, l, mu, optval, final_iter);
  }, s);
}

namespace {
FnHost fn_host(const PogsAmdFn *p) {
  POGS_CHECK(p != nullptr, "null function description");
  FnHost f{p->a, p->b, p->c, p->d, p->e, p->h};
  f.s0[0] = p->a0; f.s0[1] = p->b0; f.s0[2] = p->c0; f.s0[3] = p->d0; f.s0[4] = p->e0;
  f.h0 = p->h0;
  POGS_CHECK(p->h || (p->h0 >= 0 && p->h0 <= 15), "function code out of range");
  return f;
}
}  // namespace

int PogsAmdSolveFn(PogsAmdSolver *s, const PogsAmdFn *f, const PogsAmdFn *g, double rho, double abs_tol, double rel_tol,
                   unsigned int max_iter, unsigned int verbose, int adaptive_rho, int gap_stop, void *x, void *y, void *l,
                   void *mu, double *optval, unsigned int *final_iter) {
  return guarded([&]() {
    POGS_CHECK(s && s->impl, "null solver");
    DeviceGuard guard(s->impl->device());
    const FnHost fh = fn_host(f), gh = fn_host(g);
    return s->impl->solve(fh, gh, make_params(rho, abs_tol, rel_tol, max_iter, verbose, adaptive_rho, gap_stop), x, y, l, mu,
                          optval, final_iter);
  }, s);
}

int PogsAmdBeginRunFn(PogsAmdSolver *s, const PogsAmdFn *f, const PogsAmdFn *g, double rho, double abs_tol, double rel_tol,
                      unsigned int max_iter, int adaptive_rho, int gap_stop) {
  return guarded([&]() {
    POGS_CHECK(s && s->impl, "null solver");
    DeviceGuard guard(s->impl->device());
    const FnHost fh = fn_host(f), gh = fn_host(g);
    s->impl->begin_run(fh, gh, make_params(rho, abs_tol, rel_tol, max_iter, 0, adaptive_rho, gap_stop));
    return 0;
  }, s);
}

int PogsAmdBeginRun(PogsAmdSolver *s, const void *f_a, const void *f_b, const void *f_c, const void *f_d,
                    const void *f_e, const int *f_h, const void *g_a, const void *g_b, const void *g_c,
                    const void *g_d, const void *g_e, const int *g_h, double rho, double abs_tol, double rel_tol,
                    unsigned int max_iter, int adaptive_rho, int gap_stop) {
  return guarded([&]() {
    POGS_CHECK(s && s->impl, "null solver");
    DeviceGuard guard(s->impl->device());
    const FnHost f = fn_arrays(f_a, f_b, f_c, f_d, f_e, f_h, "f");
    const FnHost g = fn_arrays(g_a, g_b, g_c, g_d, g_e, g_h, "g");
    s->impl->begin_run(f, g, make_params(rho, abs_tol, rel_tol, max_iter, 0, adaptive_rho, gap_stop));
    return 0;
  }, s);
}

int PogsAmdIterate(PogsAmdSolver *s, unsigned int iters, double *seconds, unsigned int *solves_completed) {
  return guarded([&]() {
    POGS_CHECK(s && s->impl, "null solver");
    DeviceGuard guard(s->impl->device());
    s->impl->iterate(iters, seconds, solves_completed);
    return 0;
  }, s);
}

int PogsAmdSetWarmStart(PogsAmdSolver *s, const void *x0, const void *l0) {
  return guarded([&]() {
    POGS_CHECK(s && s->impl && x0 && l0, "warm start needs both x0 and l0 (pogs.cpp:159-179)");
    DeviceGuard guard(s->impl->device());
    s->impl->set_warm_start(x0, l0);
    return 0;
  });
}

int PogsAmdGetStats(const PogsAmdSolver *s, PogsAmdStats *out) {
  return guarded([&]() {
    POGS_CHECK(s && s->impl && out, "null argument");
    *out = const_cast<PogsAmdSolver *>(s)->impl->stats();
    return 0;
  });
}

int PogsAmdResetStats(PogsAmdSolver *s) {
  return guarded([&]() {
    POGS_CHECK(s && s->impl, "null solver");
    DeviceGuard guard(s->impl->device());
    PogsAmdStats &st = s->impl->stats();
    st.t_loop_s = 0; st.iterations = 0; st.exact_iters = 0; st.rho_updates = 0;
    st.cg_iters = 0; st.matvecs = 0;
    st.stream_ms = 0; st.stream_launches = 0; st.stream_bytes = 0;
    st.reserved[0] = 0; st.reserved[1] = 0;
    return 0;
  });
}

void PogsAmdDestroy(PogsAmdSolver *s) {
  try {
    if (!s) return;
    DeviceGuard guard(s->impl ? s->impl->device() : -1);   // buffers are freed on the handle's device
    delete s;
  } catch (...) {}
}

const char *PogsAmdLastError(void) { return g_last_error.c_str(); }

int PogsAmdPoolStats(int device, PogsAmdPoolInfo *out) {
  return guarded([&]() {
    POGS_CHECK(out, "null argument");
    int dev = device;
    if (dev < 0) POGS_HIP_CHECK(hipGetDevice(&dev));
    const PoolCounters c = DevicePool::get().counters(dev);
    out->mallocs = c.mallocs; out->reuses = c.reuses; out->frees = c.frees;
    out->malloc_ms = c.malloc_ms; out->free_ms = c.free_ms;
    out->cached_bytes = c.cached_bytes; out->live_bytes = c.live_bytes;
    out->peak_cached_bytes = c.peak_cached_bytes;
    return 0;
  });
}

int PogsAmdPoolTrim(int device, size_t *freed_bytes) {
  return guarded([&]() {
    const size_t b = DevicePool::get().trim(device);
    if (freed_bytes) *freed_bytes = b;
    return 0;
  });
}

int PogsAmdProxEval(int dtype, size_t n, const int *h, const void *a, const void *b, const void *c, const void *d,
                    const void *e, double rho, const void *in, void *out) {
  return guarded([&]() {
    if (dtype == POGS_AMD_F32) prox_eval_host<float>(n, h, a, b, c, d, e, rho, in, out, nullptr);
    else prox_eval_host<double>(n, h, a, b, c, d, e, rho, in, out, nullptr);
    return 0;
  });
}

int PogsAmdFuncEval(int dtype, size_t n, const int *h, const void *a, const void *b, const void *c, const void *d,
                    const void *e, const void *in, double *out) {
  return guarded([&]() {
    if (dtype == POGS_AMD_F32) prox_eval_host<float>(n, h, a, b, c, d, e, 1.0, in, nullptr, out);
    else prox_eval_host<double>(n, h, a, b, c, d, e, 1.0, in, nullptr, out);
    return 0;
  });
}

int PogsAmdProjSubgradEval(int dtype, size_t n, const int *h, const void *a, const void *b, const void *c,
                           const void *d, const void *e, const void *x_in, const void *v_in, void *v_out) {
  return guarded([&]() {
    POGS_CHECK(x_in && v_in && v_out, "null argument");
    if (dtype == POGS_AMD_F32) prox_eval_host<float>(n, h, a, b, c, d, e, 1.0, v_in, v_out, nullptr, x_in);
    else prox_eval_host<double>(n, h, a, b, c, d, e, 1.0, v_in, v_out, nullptr, x_in);
    return 0;
  });
}

int PogsAmdGetEquil(const PogsAmdSolver *s, void *A_eq, void *d, void *e, double *nrmA) {
  return guarded([&]() {
    POGS_CHECK(s && s->impl, "null solver");
    DeviceGuard guard(s->impl->device());
    const_cast<PogsAmdSolver *>(s)->impl->get_equil(A_eq, d, e, nrmA);
    return 0;
  });
}

int PogsAmdProject(PogsAmdSolver *s, const void *x0, const void *y0, double tol, void *x, void *y) {
  return guarded([&]() {
    POGS_CHECK(s && s->impl, "null solver");
    DeviceGuard guard(s->impl->device());
    s->impl->project(x0, y0, tol, x, y);
    return 0;
  });
}

int PogsAmdMul(PogsAmdSolver *s, char trans, double alpha, const void *x, double beta, void *y) {
  return guarded([&]() {
    POGS_CHECK(s && s->impl, "null solver");
    DeviceGuard guard(s->impl->device());
    s->impl->mul(trans, alpha, x, beta, y);
    return 0;
  });
}

int PogsAmdReadBandwidth(int device, size_t bytes, int reps, double *gb_per_s, int *pattern) {
  return guarded([&]() {
    POGS_CHECK(gb_per_s && bytes >= (1u << 20), "null argument / fewer than 1 MiB");
    DeviceGuard guard(device);
    int pat = 0;
    *gb_per_s = measure_read_bandwidth_gbs(bytes, reps, &pat);
    if (pattern) *pattern = pat;
    return 0;
  });
}

int PogsAmdWaveSumCheck(int dtype, size_t n, const void *in_host, void *alu_host, void *lds_host) {
  return guarded([&]() {
    POGS_CHECK(in_host && alu_host && lds_host, "null argument");
    if (dtype == POGS_AMD_F32) wave_sum_check(static_cast<const float *>(in_host), n, static_cast<float *>(alu_host), static_cast<float *>(lds_host));
    else wave_sum_check(static_cast<const double *>(in_host), n, static_cast<double *>(alu_host), static_cast<double *>(lds_host));
    return 0;
  });
}

int PogsAmdRandUniform(int dtype, size_t n, void *out_host) {
  return guarded([&]() {
    if (dtype == POGS_AMD_F32) rand_uniform_host(static_cast<float *>(out_host), n);
    else rand_uniform_host(static_cast<double *>(out_host), n);
    return 0;
  });
}

}  // extern "C"
