"""Sinkhorn-Knopp early exit vs the full 50 iterations at C2 (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import pogs_amd
from pogs_amd import graph as G
import bench

m, n = 100000, 10000
DT = np.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else np.float32
TDT = torch.float64 if DT == np.float64 else torch.float32
dev = torch.device("cuda:0")
A, b = bench.make_problem(m, n, 0, dev)
A = A.to(TDT)
out = {}
for mode in ("early", "full"):
    if mode == "full":
        os.environ["POGS_AMD_SK_FULL"] = "1"
    s = pogs_amd.Solver(A.data_ptr(), dtype=DT, shape=(m, n), device_ptr=True, profile=True)
    f, g = G.lasso_functions(b, 0.1, n)
    r = s.solve(f, g)
    st = s.stats()
    _, d, e, nrm = s.equilibrated(want_matrix=False)
    out[mode] = (d.astype(np.float64), e.astype(np.float64), nrm, r["x"].astype(np.float64), st)
    print(mode, "passes", st["matvecs_init"], "equil_ms %.1f" % st["equil_ms"], "iterations", st["iterations"], "optval", r["optval"], flush=True)
    s.close()
d0, e0, n0, x0, _ = out["early"]; d1, e1, n1, x1, _ = out["full"]
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
print("d rel diff %.2e, e rel diff %.2e, outer(d,e) scale-mode: mean(d0/d1)-1 = %.2e, mean(e0/e1)-1 = %.2e" % (rel(d0, d1), rel(e0, e1), np.mean(d0 / d1) - 1, np.mean(e0 / e1) - 1))
print("d*e product rel diff (first entries) %.2e" % rel(np.outer(d0[:200], e0[:200]), np.outer(d1[:200], e1[:200])))
print("nrmA %.7f %.7f ; x rel diff %.2e" % (n0, n1, rel(x0, x1)))
