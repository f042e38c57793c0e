"""Generates tests/golden/wide_reference.npz: the compiled reference's solutions (PogsS fp32 and
PogsD fp64) of a FULL-SIZE wide lasso, 10000 x 100000 -- the m <= n path of the direct projector
(src/cpu/projector/projector_direct_dense.cpp:128-135, A A^T) that the engine runs on transposed
storage.  x_true is 1 % dense, lambda = 0.2 max|A^T b| (computed in fp64 and stored in the fixture).
The matrix is pogs_amd.synth.dense_lasso_rows(10000, 100000, seed=2025, density=0.01): regenerated
bit for bit from the seed; checksums in the fixture.  Build container, ~15 minutes, 20 GB:

    python tests/golden/make_wide_reference.py"""
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

M, N, SEED = 10000, 100000, 2025


def checksums(A, b):
    return np.array([float(A[::97].astype(np.float64).sum()), float(np.abs(A[:, ::1013]).astype(np.float64).sum()),
                     float(np.linalg.norm(b)), float(b[::11].sum())])


def main():
    A, b, _ = synth.dense_lasso_rows(M, N, seed=SEED, density=0.01, chunk=500)
    atb = np.zeros(N)
    for r0 in range(0, M, 500):
        atb += A[r0:r0 + 500].astype(np.float64).T @ b[r0:r0 + 500]
    lam = 0.2 * float(np.max(np.abs(atb)))
    f, g = G.lasso_functions(b, lam, N)
    soa = lambda fv: {k: getattr(fv, k) for k in "habcde"}  # noqa: E731
    out = dict(shape=np.array([M, N]), seed=SEED, lam=lam, checksums=checksums(A, b))

    def objective(x):
        x = x.astype(np.float64)
        y = np.concatenate([A[r0:r0 + 500].astype(np.float64) @ x for r0 in range(0, M, 500)])
        return 0.5 * float(np.sum((y - b) ** 2)) + lam * float(np.abs(x).sum())

    for tag, dt, Ain in (("", np.float32, A), ("_fp64", np.float64, None)):
        t0 = time.time()
        r = ob.ref_solve(A.astype(np.float64) if Ain is None else Ain, soa(f), soa(g), dtype=dt, verbose=1, threads=os.cpu_count())
        print(r["stdout"][-400:])
        assert r["status"] == 0
        out.update({"x" + tag: r["x"].astype(dt), "optval" + tag: r["optval"], "iterations" + tag: r["iterations"],
                    "objective_at_x" + tag: objective(r["x"]), "y" + tag: r["y"].astype(dt), "l_norm" + tag: float(np.linalg.norm(r["l"])),
                    "seconds" + tag: time.time() - t0})
        print("%s reference: iterations %d, optval %.6f, %.0f s" % (dt.__name__, r["iterations"] + 1, r["optval"], time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, "wide_reference.npz"), **out)
    x32, x64 = out["x"].astype(np.float64), out["x_fp64"]
    print("rel_x fp32 vs fp64 reference: %.3e" % (np.linalg.norm(x32 - x64) / np.linalg.norm(x64)))


if __name__ == "__main__":
    main()
