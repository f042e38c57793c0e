"""Stress (development aid): many solver set-ups of changing shape / type in one process."""
import faulthandler, os, sys, time
faulthandler.enable()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import pogs_amd
from pogs_amd import graph as G

rng = np.random.default_rng(0)
sizes = [120, 240, 480, 500, 1000, 1520, 2000, 2540, 3000, 3040, 4000, 5000]
mats = {}
for n in sizes:
    mats[n] = rng.standard_normal((n + 64, n))
t0 = time.time()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for i in range(reps):
    n = sizes[int(rng.integers(len(sizes)))]
    dt = np.float32 if rng.random() < 0.5 else np.float64
    wide = rng.random() < 0.3
    A = mats[n].T.copy() if wide else mats[n]
    b = rng.standard_normal(A.shape[0])
    f, g = G.lasso_functions(b, 0.5 * np.max(np.abs(A.T @ b)), A.shape[1])
    with pogs_amd.Solver(A, dtype=dt) as s:
        r = s.solve(f, g, max_iter=30)
    if i % 50 == 0:
        print(i, n, np.dtype(dt).name, wide, r["status"], "%.1f s" % (time.time() - t0), flush=True)
print("done", reps, "%.1f s" % (time.time() - t0))
