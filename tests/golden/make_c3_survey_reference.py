"""Generates tests/golden/c3_survey_reference.npz: what the compiled reference does on configs[2]
(solve_logistic, dense fp32, 200000 x 5000, lambda = 0.01) with SURVEY.md section 8(d)'s OWN generator --
w_true ~ N(0,1) on 30 % of the entries, unscaled, so the logits have a spread of ~ sqrt(0.3 n) = 39
and the labels are all but separable.  bench.py and the other C3 tests rescale w_true to a logit
spread of 2 ("known deviation"); this pins the un-rescaled problem once:

  * PogsS with the default max_iter = 2500: status, final_iter, optval (does it converge?);
  * PogsS with max_iter = 300: x, a slice of y and l after exactly 300 iterations -- a trajectory
    that an engine following the reference's algorithm has to reproduce to rounding.

Run in the build container (8 cores, about ten minutes):  python tests/golden/make_c3_survey_reference.py
The matrix is not stored (4 GB): pogs_amd.synth.dense_logistic_rows(200000, 5000, seed=33) regenerates
it bit for bit (numpy PCG64, no BLAS); the fixture carries checksums."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np  # noqa: E402

import oracle_binding as ob  # noqa: E402
from pogs_amd import graph as G  # noqa: E402
from pogs_amd import synth  # noqa: E402

M, N, SEED, LAM = 200000, 5000, 33, 0.01


def checksums(A, lab):
    return np.array([float(A[::997].astype(np.float64).sum()), float(np.abs(A[:, ::113]).astype(np.float64).sum()),
                     float(lab.sum()), float(lab[::101].sum())])


def main():
    t0 = time.time()
    A, lab, _ = synth.dense_logistic_rows(M, N, seed=SEED)
    print("generated in %.1f s" % (time.time() - t0), flush=True)
    f, g = G.logistic_functions(lab, LAM, N)
    soa = lambda fv: {k: getattr(fv, k) for k in "habcde"}  # noqa: E731
    os.environ.pop("OMP_WAIT_POLICY", None)
    short = ob.ref_solve(A, soa(f), soa(g), dtype=np.float32, verbose=1, threads=os.cpu_count(), max_iter=300)
    print(short["stdout"][-400:], flush=True)
    full = ob.ref_solve(A, soa(f), soa(g), dtype=np.float32, verbose=1, threads=os.cpu_count())
    print(full["stdout"][-400:], flush=True)
    np.savez_compressed(os.path.join(HERE, "c3_survey_reference.npz"), shape=np.array([M, N]), seed=SEED, lam=LAM,
                        checksums=checksums(A, lab),
                        status=full["status"], iterations=full["iterations"], optval=full["optval"],
                        t_total=full.get("t_total", 0.0), t_init=full.get("t_init", 0.0),
                        x_full=full["x"].astype(np.float32),
                        status_300=short["status"], iterations_300=short["iterations"], optval_300=short["optval"],
                        x_300=short["x"].astype(np.float32), y_head_300=short["y"][:4096].astype(np.float32),
                        l_head_300=short["l"][:4096].astype(np.float32),
                        y_norm_300=float(np.linalg.norm(short["y"].astype(np.float64))), threads=os.cpu_count())
    print("c3_survey_reference.npz: status %d after %d iterations (optval %.6g, %.0f s); 300-iteration run status %d"
          % (full["status"], full["iterations"] + 1, full["optval"], full.get("t_total", 0), short["status"]))


if __name__ == "__main__":
    main()
