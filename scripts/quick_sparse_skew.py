"""Skewed sparse structure through the column-blocked SpMV (development aid): adjacent nearly dense
rows (window splitting at the 16-bit offset limit), a dense column, empty leading/trailing rows."""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import scipy.sparse as sp
import pogs_amd
from pogs_amd import _lib

def build(m, n, seed):
    rng = np.random.default_rng(seed)
    k = 5
    rows = np.repeat(np.arange(m), k); cols = rng.integers(0, n, m * k); vals = rng.standard_normal(m * k)
    A = sp.csr_matrix((vals, (rows, cols)), shape=(m, n)).tolil()
    for r in range(100, 106):                      # six adjacent rows, 26000 entries each inside block 0
        c = rng.choice(28000, 26000, replace=False); A[r, c] = rng.standard_normal(26000)
    A = A.tocsr()
    dense_col = sp.csr_matrix((rng.standard_normal(m), (np.arange(m), np.full(m, n - 7))), shape=(m, n))
    A = (A + dense_col).tocsr()
    A = sp.vstack([sp.csr_matrix((3, n)), A, sp.csr_matrix((2, n))]).tocsr()
    A.sum_duplicates(); A.sort_indices()
    return A

for dtype in (np.float64, np.float32):
    A = build(40000, 61000, 3).astype(dtype)
    m, n = A.shape
    rng = np.random.default_rng(1)
    x = rng.standard_normal(n).astype(dtype); y = rng.standard_normal(m).astype(dtype)
    with pogs_amd.Solver(A, dtype=dtype) as s:
        buf = np.zeros(A.nnz, dtype); nrm = ctypes.c_double()
        assert _lib.lib.PogsAmdGetEquil(s._h, buf.ctypes.data_as(ctypes.c_void_p), None, None, ctypes.byref(nrm)) == 0
        As = sp.csr_matrix((buf.astype(np.float64), A.indices, A.indptr), shape=(m, n))
        got = s.mul("n", 1.0, x, 0.0, np.zeros(m, dtype))
        ref = As @ x.astype(np.float64)
        e1 = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        got_t = s.mul("t", 1.0, y, 0.0, np.zeros(n, dtype))
        ref_t = As.T @ y.astype(np.float64)
        e2 = np.linalg.norm(got_t - ref_t) / np.linalg.norm(ref_t)
        print(np.dtype(dtype).name, "nnz", A.nnz, "Ax err %.2e  A^T y err %.2e" % (e1, e2), flush=True)
