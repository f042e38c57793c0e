"""Adds the compiled reference's FP64 run (PogsD) of the full-size C2 problem to
tests/golden/c2_reference.npz (made by make_c2_reference.py, which holds the fp32 run, PogsS).

Why both: on this problem the reference's fp32 build needs 154 iterations, its fp64 build 106 --
same algorithm, same inputs (the fp32 matrix, widened), solutions 1e-5 apart.  The 48 extra
iterations are the reference's own fp32 rounding (sequential fp32 sums over 1e5 rows next to a
1e-4 stopping rule), not the algorithm.  The engine stores fp32 but sums in blocks with fp64 scalar
reductions and walks the fp64 trajectory: 106 iterations, optval 485.0442 vs 485.0448.  The fixture
therefore pins the iteration count and optval to the fp64 run and x to both.

Build container, ~8 minutes, 20 GB:   python tests/golden/make_c2_reference_fp64.py"""
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


def main():
    path = os.path.join(HERE, "c2_reference.npz")
    fx = dict(np.load(path))
    m, n = (int(v) for v in fx["shape"])
    A, b, _ = synth.dense_lasso_rows(m, n, seed=int(fx["seed"]))
    f, g = G.lasso_functions(b, float(fx["lam"]), n)
    soa = lambda fv: {k: getattr(fv, k) for k in "habcde"}  # noqa: E731
    t0 = time.time()
    r = ob.ref_solve(A.astype(np.float64), soa(f), soa(g), dtype=np.float64, verbose=1, threads=os.cpu_count())
    print(r["stdout"][-500:])
    assert r["status"] == 0
    x64 = r["x"].astype(np.float64)
    y64 = np.concatenate([A[r0:r0 + 10000].astype(np.float64) @ x64 for r0 in range(0, m, 10000)])
    obj = 0.5 * float(np.sum((y64 - b) ** 2)) + float(fx["lam"]) * float(np.abs(x64).sum())
    xr = fx["x"].astype(np.float64)
    fx.update(x_fp64=x64, optval_fp64=r["optval"], iterations_fp64=r["iterations"], objective_at_x_fp64=obj,
              y_norm_fp64=float(np.linalg.norm(r["y"])), y_head_fp64=r["y"][:4096], l_head_fp64=r["l"][:4096])
    np.savez_compressed(path, **fx)
    print("fp64 reference: iterations %d optval %.6f  rel_x vs the fp32 reference %.3e  (%.0f s)"
          % (r["iterations"] + 1, r["optval"], np.linalg.norm(x64 - xr) / np.linalg.norm(xr), time.time() - t0))


if __name__ == "__main__":
    main()
