"""Generates tests/golden/c2_reference.npz: the solution of the FULL-SIZE configs[1] problem
(dense fp32 lasso, 100000 x 10000, lambda = 0.1, default tolerances) by the compiled reference
(oracle/_ref/libpogs_cpu.so = the reference's own src/cpu + src/interface_c sources built by
oracle/Makefile).  Run in the build container (8 cores: ~3 minutes; the GPU box's host needs more
than 15 minutes for the same call, which is why this is a fixture and not a live comparison):

    python tests/golden/make_c2_reference.py

The matrix is not stored (4 GB): pogs_amd.synth.dense_lasso_rows(100000, 10000, seed=2024)
regenerates it bit for bit (numpy PCG64); the fixture carries checksums of A and b so that a test
can tell if the generator ever changes."""
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

M, N, SEED, LAM = 100000, 10000, 2024, 0.1


def checksums(A, b):
    return np.array([float(A[::997].astype(np.float64).sum()), float(np.abs(A[:, ::113]).astype(np.float64).sum()),
                     float(np.linalg.norm(b)), float(b[::101].sum())])


def main():
    t0 = time.time()
    A, b, _ = synth.dense_lasso_rows(M, N, seed=SEED)
    print("generated in %.1f s" % (time.time() - t0), flush=True)
    f, g = G.lasso_functions(b, LAM, N)
    soa = lambda fv: {k: getattr(fv, k) for k in "habcde"}  # noqa: E731
    r = ob.ref_solve(A, soa(f), soa(g), dtype=np.float32, verbose=1, threads=os.cpu_count())
    print(r["stdout"][-600:])
    assert r["status"] == 0
    # 0.5 |A x - b|^2 + lambda |x|_1 at the returned x, in fp64 (optval is taken at the prox point
    # y12 != A x12 and moves by percents between two runs that stop a few iterations apart)
    x64 = r["x"].astype(np.float64)
    y64 = np.concatenate([A[r0:r0 + 10000].astype(np.float64) @ x64 for r0 in range(0, M, 10000)])
    obj = 0.5 * float(np.sum((y64 - b) ** 2)) + LAM * float(np.abs(x64).sum())
    np.savez_compressed(os.path.join(HERE, "c2_reference.npz"), x=r["x"].astype(np.float32), optval=r["optval"],
                        iterations=r["iterations"], status=r["status"], shape=np.array([M, N]), seed=SEED, lam=LAM,
                        checksums=checksums(A, b), y_norm=float(np.linalg.norm(r["y"].astype(np.float64))),
                        y_head=r["y"][:4096].astype(np.float32), l_head=r["l"][:4096].astype(np.float32),
                        objective_at_x=obj, t_total=r.get("t_total", 0.0), t_init=r.get("t_init", 0.0), threads=os.cpu_count())
    print("c2_reference.npz written: iterations %d, optval %.6f, total %.1f s (init %.1f s) on %d threads"
          % (r["iterations"] + 1, r["optval"], r.get("t_total", 0), r.get("t_init", 0), os.cpu_count()))


if __name__ == "__main__":
    main()
