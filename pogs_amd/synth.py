"""Synthetic problem generators for tests and bench.py (numpy default_rng / PCG64).

Recipes follow the reference's own generators:
dense lasso -- python/benchmarks/problems/lasso.py:40-53; logistic --
python/benchmarks/problems/logistic.py:27-37 (without its bias term); the
README problem -- README.md:55-59.  CSR: 50 uniformly drawn column indices per
row, N(0,1) values, duplicates summed.
"""
import numpy as np


def readme_lasso():
    """C1: the README recipe verbatim (legacy numpy RandomState)."""
    np.random.seed(0)
    A = np.random.randn(500, 300)
    b = np.random.randn(500)
    return A, b, 0.1


def dense_lasso(m, n, seed=0, dtype=np.float64, density=0.1, noise=0.1):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, n), dtype=dtype)
    x_true = rng.standard_normal(n) * (rng.random(n) < density)
    b = A.astype(np.float64) @ x_true + noise * rng.standard_normal(m)
    return A, b, x_true


def dense_logistic(m, n, seed=0, dtype=np.float64, density=0.3, logit_std=None):
    """logit_std=None is the reference recipe (w_true ~ N(0,1) on 30% of the entries).  For
    large n that makes the logits' spread ~ sqrt(0.3 n), i.e. nearly separable labels, on
    which the reference algorithm itself runs into max_iter (checked with the compiled
    reference / oracle at 20000 x 500); logit_std rescales w_true so std(A w) = logit_std."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, n), dtype=dtype)
    w = rng.standard_normal(n) * (rng.random(n) < density)
    if logit_std is not None:
        w *= logit_std / np.sqrt(max(np.sum(w * w), 1e-300))
    p = 1.0 / (1.0 + np.exp(-(A.astype(np.float64) @ w)))
    y = 2.0 * (rng.random(m) < p) - 1.0
    return A, y, w


def csr_lasso(m, n, nnz_per_row=50, seed=0, dtype=np.float64, density=0.05, noise=0.1):
    import scipy.sparse as sp

    rng = np.random.default_rng(seed)
    cols = rng.integers(0, n, size=(m, nnz_per_row))
    vals = rng.standard_normal((m, nnz_per_row))
    rows = np.repeat(np.arange(m), nnz_per_row)
    A = sp.coo_matrix((vals.ravel(), (rows, cols.ravel())), shape=(m, n)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    A = A.astype(dtype)
    x_true = rng.standard_normal(n) * (rng.random(n) < density)
    b = A.astype(np.float64) @ x_true + noise * rng.standard_normal(m)
    return A, b, x_true


def dense_lasso_rows(m, n, seed=0, density=0.1, noise=0.1, chunk=4000):
    """The dense-lasso recipe generated in row chunks (fp32 A, no fp64 copy of the matrix): the
    same bits on every machine for a given (m, n, seed, chunk).  Used for the full-size C2 problem
    whose reference solution is a committed fixture (tests/golden/c2_reference.npz)."""
    rng = np.random.default_rng(seed)
    x_true = rng.standard_normal(n) * (rng.random(n) < density)
    A = np.empty((m, n), np.float32)
    b = np.empty(m, np.float64)
    nz = np.flatnonzero(x_true)
    for r0 in range(0, m, chunk):
        r1 = min(m, r0 + chunk)
        blk = rng.standard_normal((r1 - r0, n), dtype=np.float32)
        A[r0:r1] = blk
        # (element-wise product + numpy's own pairwise sum: no BLAS, so no dependence on its threads)
        b[r0:r1] = (blk[:, nz].astype(np.float64) * x_true[nz]).sum(axis=1) + noise * rng.standard_normal(r1 - r0)
    return A, b, x_true


def dense_logistic_rows(m, n, seed=0, density=0.3, logit_std=None, chunk=4000):
    """The logistic recipe (python/benchmarks/problems/logistic.py:27-37 without its bias term:
    w_true ~ N(0,1) on 30 % of the entries, labels 2 (U < sigma(A w)) - 1) generated in row chunks
    with no BLAS in the generator: the same bits on every machine for a given (m, n, seed, chunk).
    logit_std=None is the recipe as SURVEY.md section 8(d) states it (nearly separable at large n);
    a value rescales w_true so that std(A w) = logit_std."""
    rng = np.random.default_rng(seed)
    w = rng.standard_normal(n) * (rng.random(n) < density)
    if logit_std is not None:
        w *= logit_std / np.sqrt(max(np.sum(w * w), 1e-300))
    A = np.empty((m, n), np.float32)
    lab = np.empty(m, np.float64)
    nz = np.flatnonzero(w)
    for r0 in range(0, m, chunk):
        r1 = min(m, r0 + chunk)
        blk = rng.standard_normal((r1 - r0, n), dtype=np.float32)
        A[r0:r1] = blk
        z = (blk[:, nz].astype(np.float64) * w[nz]).sum(axis=1)
        lab[r0:r1] = 2.0 * (rng.random(r1 - r0) < 1.0 / (1.0 + np.exp(-z))) - 1.0
    return A, lab, w
