"""C2 through the reference's one-shot ABI shape: host float32 A (PCIe-inclusive), development aid."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import pogs_amd
from pogs_amd import synth, graph as G

m, n = 100000, 10000
rng = np.random.default_rng(0)
A = rng.standard_normal((m, n), dtype=np.float32)
xt = rng.standard_normal(n) * (rng.random(n) < 0.1)
b = A @ xt.astype(np.float32) + 0.1 * rng.standard_normal(m)
f, g = G.lasso_functions(b, 0.1, n)
for rep in range(3):
    t0 = time.time()
    s = pogs_amd.Solver(A, dtype=np.float32)
    t1 = time.time()
    r = s.solve(f, g)
    t2 = time.time()
    st = s.stats()
    s.close()
    print("rep %d: create %.3f s (h2d %.3f s, init %.3f s)  solve %.3f s  iterations %d  -> total %.3f s" % (
        rep, t1 - t0, st["t_h2d_s"], st["t_init_s"], t2 - t1, r["iterations"] + 1, t2 - t0))
