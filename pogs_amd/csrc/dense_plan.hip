// One dense solver class per streaming shape (stream.h: POGS_STREAM_PLANS) and arithmetic type.
// pogs_amd/build.py compiles this file once per (type, shape) with
//   -DPOGS_PLAN_T=float|double  -DPOGS_PLAN_NAME=f32_p256_10|...  and
//   -DPOGS_PLAN_TPB=256 -DPOGS_PLAN_NV=10      (one plain shape)   or   -DPOGS_PLAN_XL=1  (the windowed form),
// so that a solve loads the ~0.3 MB code object of its own shape instead of a 4 MB one with all of them.
#include "dense_solver.h"

#define POGS_CAT2(a, b) a##b
#define POGS_CAT(a, b) POGS_CAT2(a, b)

namespace pogs_amd {
#if defined(POGS_PLAN_XL)
using PlanTag = WindowPlans;
#else
using PlanTag = OnePlan<POGS_PLAN_TPB, POGS_PLAN_NV>;
#endif
SolverBase *POGS_CAT(make_dense_solver_, POGS_PLAN_NAME)(int ord, size_t m, size_t n, const void *A, int mem,
                                                          const PogsAmdOptions *opt, const PogsAmdDist *dist) {
  return make_dense_solver_t<POGS_PLAN_T, PlanTag>(ord, m, n, A, mem, opt, dist);
}
}  // namespace pogs_amd
