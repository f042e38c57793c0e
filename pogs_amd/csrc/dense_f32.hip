// DenseSolver<float> (see dense_solver.h).
#include "dense_solver.h"

namespace pogs_amd {
SolverBase *make_dense_solver_f32(int ord, size_t m, size_t n, const void *A, int mem, const PogsAmdOptions *opt,
                                  const PogsAmdDist *dist) {
  return make_dense_solver_t<float>(ord, m, n, A, mem, opt, dist);
}
}  // namespace pogs_amd
