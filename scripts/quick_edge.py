"""Edge inputs through the ABI (development aid): zero matrix, empty sparse matrix, max_iter 1, NaN."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, scipy.sparse as sp
import pogs_amd
import oracle_binding as ob
from helpers import soa

def both(tag, A, b, lam=0.1, **kw):
    n = A.shape[1]
    f, g = pogs_amd.graph.lasso_functions(b, lam, n)
    try:
        r = pogs_amd.graph._solve_graph_form(A, f, g, kw.get("abs_tol", 1e-4), kw.get("rel_tol", 1e-4), kw.get("max_iter", 2500), 0, 1.0, dtype=np.float64)
        got = (r["status"], r["iterations"], float(np.linalg.norm(r["x"])))
    except Exception as e:
        got = ("EXC", str(e)[:80])
    try:
        w = ob.oracle_solve(A, soa(f), soa(g), dtype=np.float64, max_iter=kw.get("max_iter", 2500))
        want = (w["status"], w["iterations"], float(np.linalg.norm(w["x"])))
    except Exception as e:
        want = ("EXC", str(e)[:80])
    print("%-28s engine %s   oracle %s" % (tag, got, want))

rng = np.random.default_rng(0)
both("zero dense 50x20", np.zeros((50, 20)), rng.standard_normal(50))
both("zero dense wide 20x50", np.zeros((20, 50)), rng.standard_normal(20))
both("empty csr 50x20", sp.csr_matrix((50, 20)), rng.standard_normal(50))
A = rng.standard_normal((60, 30))
both("max_iter 1", A, rng.standard_normal(60), max_iter=1)
both("max_iter 2", A, rng.standard_normal(60), max_iter=2)
An = A.copy(); An[3, 4] = np.nan
both("nan in A", An, rng.standard_normal(60), max_iter=50)
both("1x1", np.array([[2.0]]), np.array([1.0]))
both("rank-1 dense", np.outer(rng.standard_normal(80), rng.standard_normal(25)), rng.standard_normal(80))
both("duplicate columns", np.hstack([A, A]), rng.standard_normal(60))
