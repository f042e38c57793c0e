"""Generates tests/golden/c4_reference.npz: the compiled reference's solution (PogsSparseS, fp32) of the
FULL-SIZE configs[3] problem -- CSR 2000000 x 500000, 50 uniformly drawn column indices per row with
duplicates summed (nnz ~ 1e8), lambda = 0.1 -- for pogs_amd.synth.csr_lasso(2000000, 500000, 50,
seed=4, dtype=float32) (numpy PCG64: the same matrix on every machine; checksums in the fixture).
The reference is single-threaded on its sparse path: this takes the better part of an hour in the
build container, which is why it is a fixture.

    python tests/golden/make_c4_reference.py"""
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

M, N, K, SEED, LAM = 2000000, 500000, 50, 4, 0.1


def checksums(A, b):
    return np.array([float(A.nnz), float(A.data[::1009].astype(np.float64).sum()), float(A.indices[::1013].astype(np.float64).sum()),
                     float(np.linalg.norm(b)), float(b[::101].sum())])


def main():
    t0 = time.time()
    A, b, _ = synth.csr_lasso(M, N, K, seed=SEED, dtype=np.float32)
    print("generated in %.0f s, nnz %d" % (time.time() - t0, A.nnz), flush=True)
    f, g = G.lasso_functions(b, LAM, N)
    soa = lambda fv: {k: getattr(fv, k) for k in "habcde"}  # noqa: E731
    t0 = time.time()
    r = ob.ref_solve(A, soa(f), soa(g), dtype=np.float32, verbose=1, threads=os.cpu_count(), timeout=3 * 3600)
    print(r["stdout"][-500:])
    assert r["status"] == 0
    x64, y64 = r["x"].astype(np.float64), r["y"].astype(np.float64)
    obj = 0.5 * float(np.sum((A.astype(np.float64) @ x64 - b) ** 2)) + LAM * float(np.abs(x64).sum())
    np.savez_compressed(os.path.join(HERE, "c4_reference.npz"), x=r["x"].astype(np.float32), optval=r["optval"],
                        iterations=r["iterations"], status=r["status"], shape=np.array([M, N, K]), seed=SEED, lam=LAM,
                        checksums=checksums(A, b), y_norm=float(np.linalg.norm(y64)), y_head=r["y"][:4096].astype(np.float32),
                        l_head=r["l"][:4096].astype(np.float32), objective_at_x=obj, seconds=time.time() - t0)
    print("c4_reference.npz written: iterations %d, optval %.6f, %.0f s" % (r["iterations"] + 1, r["optval"], time.time() - t0))


if __name__ == "__main__":
    main()
