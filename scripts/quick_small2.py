"""Where does a one-shot call on a small problem spend its time (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import pogs_amd
from pogs_amd import synth

for (m, n, dt) in [(2000, 300, np.float32), (5000, 1000, np.float32)]:
    A, b, _ = synth.dense_lasso(m, n, seed=0, dtype=dt)
    f, g = pogs_amd.graph.lasso_functions(b, 0.1, n)
    pogs_amd.solve_lasso(A, b, 0.1, dtype=dt)
    for rep in range(4):
        t0 = time.time(); s = pogs_amd.Solver(A, dtype=dt); t1 = time.time()
        r = s.solve(f, g); t2 = time.time()
        r = s.solve(f, g); t3 = time.time()
        s.close(); t4 = time.time()
        print("%dx%d rep %d: create %.1f  solve1 %.1f  solve2 %.1f  close %.1f ms  (init %.1f)" % (
            m, n, rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, 0))
