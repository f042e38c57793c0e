"""Generates tests/golden/wide_sparse_reference.npz: the compiled reference's solutions (PogsSparseS fp32 and
PogsSparseD fp64) of a FULL-SIZE wide sparse lasso, CSR 200000 x 1000000 with 50 uniformly drawn column indices
per row (nnz ~ 1e7) -- `PogsSparse*` takes any shape through its CGLS projector (src/interface_c/pogs_c.cpp:69-73,
src/cpu/projector/projector_cgls.cpp:52-88), and until round 5 every sparse parity solve of this repository was
tall.  x_true is 0.5 % dense; lambda = 0.2 max|A^T b| (computed in fp64, stored in the fixture).  The matrix is
pogs_amd.synth.csr_lasso(200000, 1000000, 50, seed=7, density=0.005): the same on every machine (numpy PCG64;
checksums in the fixture).  Build container, a few minutes:

    python tests/golden/make_wide_sparse_reference.py"""
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

M, N, K, SEED, DENS = 200000, 1000000, 50, 7, 0.005


def checksums(A, b):
    return np.array([float(A.nnz), float(A.data[::1009].astype(np.float64).sum()), float(A.indices[::1013].astype(np.float64).sum()),
                     float(np.linalg.norm(b)), float(b[::101].sum())])


def main():
    A, b, _ = synth.csr_lasso(M, N, K, seed=SEED, dtype=np.float32, density=DENS)
    A64 = A.astype(np.float64)
    lam = 0.2 * float(np.max(np.abs(A64.T @ b)))
    f, g = G.lasso_functions(b, lam, N)
    soa = lambda fv: {k: getattr(fv, k) for k in "habcde"}  # noqa: E731
    out = dict(shape=np.array([M, N, K]), seed=SEED, density=DENS, lam=lam, checksums=checksums(A, b))
    for tag, dt, Ain in (("", np.float32, A), ("_fp64", np.float64, A64)):
        t0 = time.time()
        r = ob.ref_solve(Ain, soa(f), soa(g), dtype=dt, verbose=1, threads=os.cpu_count(), timeout=3 * 3600)
        print(r["stdout"][-400:])
        assert r["status"] == 0
        x64 = r["x"].astype(np.float64)
        obj = 0.5 * float(np.sum((A64 @ x64 - b) ** 2)) + lam * float(np.abs(x64).sum())
        nz = np.flatnonzero(r["x"])
        # x is sparse (lasso): its support and values, not a million mostly-zero entries
        out.update({"x_idx" + tag: nz.astype(np.int32), "x_val" + tag: r["x"][nz].astype(dt), "optval" + tag: r["optval"],
                    "iterations" + tag: r["iterations"], "objective_at_x" + tag: obj, "y" + tag: r["y"].astype(dt),
                    "l_norm" + tag: float(np.linalg.norm(r["l"])), "seconds" + tag: time.time() - t0})
        print("%s reference: iterations %d, optval %.6f, nnz(x) %d, %.0f s"
              % (dt.__name__, r["iterations"] + 1, r["optval"], len(nz), time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, "wide_sparse_reference.npz"), **out)


if __name__ == "__main__":
    main()
