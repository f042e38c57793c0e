"""Generates tests/golden/c3_reference.npz: the compiled reference's solutions of bench.py's configs[2]
problem -- solve_logistic, dense, 200000 x 5000, lambda = 0.01, default tolerances, labels from logits
of spread 2 (pogs_amd.synth.dense_logistic_rows(200000, 5000, seed=35, logit_std=2.0); the recipe of
SURVEY.md section 8(d) unscaled is all but separable and the reference itself runs into max_iter on
it: tests/golden/make_c3_survey_reference.py pins that case).

Two runs of oracle/_ref/libpogs_cpu.so (the reference's own six sources against the image's MKL):
PogsS on the fp32 matrix and PogsD on the same matrix widened to fp64.  bench.py regenerates the
matrix from the seed (numpy PCG64, row chunks, no BLAS in the generator), checks it against the
checksums stored here and reports `parity_vs_reference` for c3 from this file, as it does for c2.

Run in the build container (8 cores, about ten minutes):  python tests/golden/make_c3_reference.py"""
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

M, N, SEED, LAM, LOGIT_STD = 200000, 5000, 35, 0.01, 2.0


def checksums(A, lab):
    return np.array([float(A[::997].astype(np.float64).sum()), float(np.abs(A[:, ::113]).astype(np.float64).sum()),
                     float(lab.sum()), float(lab[::101].sum())])


def main():
    t0 = time.time()
    A, lab, _ = synth.dense_logistic_rows(M, N, seed=SEED, logit_std=LOGIT_STD)
    print("generated in %.1f s" % (time.time() - t0), flush=True)
    f, g = G.logistic_functions(lab, LAM, N)
    soa = lambda fv: {k: getattr(fv, k) for k in "habcde"}  # noqa: E731
    os.environ.pop("OMP_WAIT_POLICY", None)
    r32 = ob.ref_solve(A, soa(f), soa(g), dtype=np.float32, verbose=1, threads=os.cpu_count())
    print(r32["stdout"][-400:], flush=True)
    r64 = ob.ref_solve(A, soa(f), soa(g), dtype=np.float64, verbose=1, threads=os.cpu_count())
    print(r64["stdout"][-400:], flush=True)
    np.savez_compressed(os.path.join(HERE, "c3_reference.npz"), shape=np.array([M, N]), seed=SEED, lam=LAM,
                        logit_std=LOGIT_STD, checksums=checksums(A, lab),
                        status=r32["status"], iterations=r32["iterations"], optval=r32["optval"],
                        t_total=r32.get("t_total", 0.0), t_init=r32.get("t_init", 0.0), x=r32["x"].astype(np.float32),
                        y_head=r32["y"][:4096].astype(np.float32), l_head=r32["l"][:4096].astype(np.float32),
                        status_fp64=r64["status"], iterations_fp64=r64["iterations"], optval_fp64=r64["optval"],
                        t_total_fp64=r64.get("t_total", 0.0), t_init_fp64=r64.get("t_init", 0.0), x_fp64=r64["x"],
                        y_head_fp64=r64["y"][:4096], l_head_fp64=r64["l"][:4096], threads=os.cpu_count())
    rel = np.linalg.norm(r32["x"].astype(np.float64) - r64["x"]) / np.linalg.norm(r64["x"])
    print("c3_reference.npz: fp32 status %d, %d iterations, optval %.8g, %.0f s | fp64 status %d, %d iterations, "
          "optval %.10g, %.0f s | fp32 vs fp64 rel x %.2e"
          % (r32["status"], r32["iterations"] + 1, r32["optval"], r32.get("t_total", 0), r64["status"],
             r64["iterations"] + 1, r64["optval"], r64.get("t_total", 0), rel))


if __name__ == "__main__":
    main()
