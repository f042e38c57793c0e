"""Small problems through the one-shot ABI: wall time per call (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import pogs_amd
from pogs_amd import synth

for (m, n, dt) in [(500, 300, np.float64), (2000, 300, np.float32), (5000, 1000, np.float32), (20000, 2000, np.float32)]:
    A, b, _ = synth.dense_lasso(m, n, seed=0, dtype=dt)
    pogs_amd.solve_lasso(A, b, 0.1, dtype=dt)
    t0 = time.time()
    for _ in range(3):
        r = pogs_amd.solve_lasso(A, b, 0.1, dtype=dt)
    dt_call = (time.time() - t0) / 3
    with pogs_amd.Solver(A, dtype=dt) as s:
        f, g = pogs_amd.graph.lasso_functions(b, 0.1, n)
        s.solve(f, g)
        t0 = time.time()
        r2 = s.solve(f, g)
        t_solve = time.time() - t0
        st = s.stats()
    print("%6d x %5d %s: one-shot %.1f ms (%d iterations); handle solve %.1f ms = %.1f us/iteration; init %.1f ms" % (
        m, n, np.dtype(dt).name, dt_call * 1e3, r["iterations"] + 1, t_solve * 1e3, t_solve * 1e6 / (r2["iterations"] + 1), st["t_init_s"] * 1e3))
