"""fp32 degenerate inputs with the Sinkhorn-Knopp shortcut active, against the oracle (development aid)."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import pogs_amd as pogs
import oracle_binding as ob
from helpers import relerr, soa

rng = np.random.default_rng(0)
A = rng.standard_normal((600, 300))
An = A.copy(); An[3, 4] = np.nan
Az = A.copy(); Az[:, 7] = 0; Az[11, :] = 0
W = rng.standard_normal((200, 900)); Wz = W.copy(); Wz[:, 5] = 0; Wz[3, :] = 0
cases = [
    ("zero tall", np.zeros((50, 20)), 2500), ("zero wide", np.zeros((20, 50)), 2500),
    ("nan entry", An, 50), ("zero row+col tall", Az, 2500), ("zero row+col wide", Wz, 2500),
    ("rank one", np.outer(rng.standard_normal(800), rng.standard_normal(250)), 2500),
    ("duplicate columns", np.hstack([A, A]), 2500),
    ("huge range", A * np.exp(rng.uniform(-8, 8, (600, 1))) * np.exp(rng.uniform(-8, 8, (1, 300))), 2500),
]
for tag, M, max_iter in cases:
    m, n = M.shape
    b = rng.standard_normal(m)
    f, g = pogs.graph.lasso_functions(b, 0.1, n)
    got = pogs.graph._solve_graph_form(M, f, g, 1e-4, 1e-4, max_iter, 0, 1.0, dtype=np.float32)
    want = ob.oracle_solve(M, soa(f), soa(g), dtype=np.float32, max_iter=max_iter)
    fin = np.array_equal(np.isfinite(got["x"]), np.isfinite(want["x"]))
    err = relerr(got["x"], want["x"]) if np.all(np.isfinite(want["x"])) and np.linalg.norm(want["x"]) > 0 else float("nan")
    print("%-20s status %d/%d iterations %d/%d finite-pattern %s relerr %.2e" % (tag, got["status"], want["status"], got["iterations"], want["iterations"], fin, err), flush=True)
