// Sparse graph-form ADMM solver with the CGLS projector (placeholder until built).
#include "engine.h"

namespace pogs_amd {

SolverBase *make_sparse_solver(int, int, size_t, size_t, size_t, const void *, const int *, const int *, int,
                               const PogsAmdOptions *) {
  throw Error("sparse path not built yet");
}

}  // namespace pogs_amd
