"""GPU parity tests for the sparse path (CSR + transposed CSR SpMV, CGLS projector)
against the CPU oracle and the golden fixtures of the compiled reference."""
import ctypes
import os

import numpy as np
import pytest
import scipy.sparse as sp

import oracle_binding as ob
from helpers import relerr, soa

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_outputs.npz"))


def _pogs():
    import pogs_amd

    return pogs_amd


def _tol(dtype, f64, f32):
    return f64 if dtype == np.float64 else f32


def _rand_csr(m, n, per_row, seed, long_row=None, empty_rows=()):
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    for i in range(m):
        if i in empty_rows:
            continue
        k = long_row[1] if (long_row and i == long_row[0]) else int(rng.integers(1, 2 * per_row))
        c = rng.choice(n, size=min(k, n), replace=False)
        rows += [i] * len(c)
        cols += list(c)
        vals += list(rng.standard_normal(len(c)))
    A = sp.csr_matrix((vals, (rows, cols)), shape=(m, n))
    A.sort_indices()
    return A


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_spmv_both_directions_and_structure(dtype):
    """A and its device-built transpose, incl. empty rows, a row longer than one LDS tile
    (> 4096 non-zeros) and a column longer than the sort tile."""
    pogs = _pogs()
    m, n = 1500, 6000
    A = _rand_csr(m, n, 12, seed=1, long_row=(7, 5000), empty_rows=(0, 3, 1499))
    # make column 5 dense-ish (3000 entries) so its transposed segment exceeds the LDS sort tile
    extra = sp.csr_matrix((np.ones(1400), (np.arange(50, 1450), np.full(1400, 5))), shape=(m, n))
    A = (A + extra).tocsr()
    A.sort_indices()
    rng = np.random.default_rng(2)
    with pogs.Solver(A, dtype=dtype) as s:
        vals, d, e, nrmA = s.equilibrated(want_matrix=False) if False else (None, None, None, None)
        # equilibrated values come back in CSR order of the input
        buf = np.zeros(A.nnz, dtype)
        dd, ee = np.zeros(m, dtype), np.zeros(n, dtype)
        nrm = ctypes.c_double()
        from pogs_amd import _lib

        assert _lib.lib.PogsAmdGetEquil(s._h, buf.ctypes.data_as(ctypes.c_void_p), dd.ctypes.data_as(ctypes.c_void_p),
                                        ee.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nrm)) == 0
        A_eq = sp.csr_matrix((buf.astype(np.float64), A.indices, A.indptr), shape=(m, n))
        # D A E / normA structure
        A_chk = sp.diags(dd.astype(np.float64)) @ A.astype(np.float64) @ sp.diags(ee.astype(np.float64))
        assert relerr(A_eq.data, A_chk.data) < _tol(dtype, 1e-10, 1e-5)
        assert np.sum(A_eq.data ** 2) == pytest.approx(min(m, n), rel=_tol(dtype, 1e-10, 1e-4))
        x, y = rng.standard_normal(n), rng.standard_normal(m)
        assert relerr(s.mul("n", 1.0, x, 0.0, y), A_eq @ x) < _tol(dtype, 1e-12, 2e-5)
        assert relerr(s.mul("n", -2.0, x, 0.5, y), -2.0 * (A_eq @ x) + 0.5 * y) < _tol(dtype, 1e-12, 2e-5)
        assert relerr(s.mul("t", 1.0, y, 0.0, x), A_eq.T @ y) < _tol(dtype, 1e-12, 2e-5)
        assert relerr(s.mul("t", 1.5, y, -1.0, x), 1.5 * (A_eq.T @ y) - x) < _tol(dtype, 1e-12, 2e-5)
        sig = np.linalg.norm(A_eq.toarray(), 2)
        assert nrm.value <= sig * 1.001 and nrm.value >= 0.85 * sig


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_cgls_projection_kkt(dtype):
    pogs = _pogs()
    from pogs_amd import _lib

    m, n = 900, 400
    A = _rand_csr(m, n, 10, seed=3)
    rng = np.random.default_rng(4)
    x0, y0 = rng.standard_normal(n), rng.standard_normal(m)
    with pogs.Solver(A, dtype=dtype) as s:
        buf = np.zeros(A.nnz, dtype)
        assert _lib.lib.PogsAmdGetEquil(s._h, buf.ctypes.data_as(ctypes.c_void_p), None, None, None) == 0
        x, y = s.project(x0, y0, tol=_tol(dtype, 1e-10, 1e-6))
    A64 = sp.csr_matrix((buf.astype(np.float64), A.indices, A.indptr), shape=(m, n))
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    eps = _tol(dtype, 1e-8, 2e-4)
    assert np.linalg.norm(A64 @ x64 - y64) / np.sqrt(m) < eps
    assert np.linalg.norm(A64.T @ (A64 @ x64 - y0) + (x64 - x0)) / np.sqrt(n) < eps


def _check(got, want, xtol, it_slack):
    assert got["status"] == want["status"] == 0
    assert abs(got["iterations"] - want["iterations"]) <= it_slack, (got["iterations"], want["iterations"])
    assert relerr(got["x"], want["x"]) < xtol
    assert relerr(got["y"], want["y"]) < xtol
    assert got["optval"] == pytest.approx(want["optval"], rel=max(xtol, 1e-7))


def test_sparse_lasso_fp64_follows_oracle_and_golden():
    pogs = _pogs()
    from pogs_amd import synth

    A, b, _ = synth.csr_lasso(3000, 800, 20, seed=4, dtype=np.float64)
    got = pogs.solve_lasso(A, b, 0.1)
    f, g = pogs.graph.lasso_functions(b, 0.1, 800)
    want = ob.oracle_solve(A, soa(f), soa(g), dtype=np.float64)
    _check(got, want, 1e-6, 2)
    gold = {k: GOLD["csr3000_f64_" + k] for k in ("x", "y", "optval", "iterations", "status")}
    assert got["status"] == int(gold["status"]) and abs(got["iterations"] - int(gold["iterations"])) <= 2
    assert relerr(got["x"], gold["x"]) < 1e-6


def test_sparse_lasso_fp32_scaled_c4():
    """Scaled-down C4: CSR fp32 20000 x 5000, ~50 nnz per row."""
    pogs = _pogs()
    from pogs_amd import synth

    A, b, _ = synth.csr_lasso(20000, 5000, 50, seed=3, dtype=np.float32)
    got = pogs.solve_lasso(A, b, 0.1, dtype=np.float32)
    f, g = pogs.graph.lasso_functions(b, 0.1, 5000)
    want = ob.oracle_solve(A, soa(f), soa(g), dtype=np.float32)
    # measured (scripts/parity_report.py): 128 iterations like the oracle, ||dx|| / ||x|| = 2.0e-7
    _check(got, want, 2e-5, max(5, int(0.1 * want["iterations"])))
    assert relerr(got["x"], GOLD["csr20000_f32_x"]) < 2e-5


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("order", ["csr", "csc"])
def test_sparse_wide_matrix_follows_oracle(dtype, order):
    """m < n through the sparse entry points: `PogsSparse*` takes any shape through CGLS
    (src/interface_c/pogs_c.cpp:69-73, projector_cgls.cpp:52-88; the shift keeps A^T A + I definite whatever
    the rank of A).  5000 x 20000, ~30 non-zeros per row: several column blocks for A, several row ranges for
    A^T, every x-sized vector four times as long as the y-sized ones."""
    pogs = _pogs()
    from pogs_amd import _lib, synth

    m, n = 5000, 20000
    A, b, _ = synth.csr_lasso(m, n, 30, seed=21, dtype=dtype)
    f, g = pogs.graph.lasso_functions(b, 0.1, n)
    want = ob.oracle_solve(A, soa(f), soa(g), dtype=dtype)
    if order == "csr":
        got = pogs.graph._solve_graph_form(A, f, g, dtype=dtype)
    else:   # the same matrix handed over as CSC through the raw ABI (COL_MAJ)
        C = A.tocsc()
        C.sort_indices()
        fa, ga = f.arrays(dtype), g.arrays(dtype)
        x, y, l = np.zeros(n, dtype), np.zeros(m, dtype), np.zeros(m, dtype)
        real = ctypes.c_double if dtype == np.float64 else ctypes.c_float
        optval, it = real(), ctypes.c_uint()
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        data, indptr, ind = (np.ascontiguousarray(C.data, dtype), np.ascontiguousarray(C.indptr, np.int32),
                             np.ascontiguousarray(C.indices, np.int32))
        fn = _lib.lib.PogsSparseD if dtype == np.float64 else _lib.lib.PogsSparseS
        st = fn(_lib.COL_MAJ, m, n, C.nnz, ptr(data), ptr(indptr), ptr(ind),
                *[ptr(fa[k]) for k in "abcdeh"], *[ptr(ga[k]) for k in "abcdeh"],
                real(1.0), real(1e-4), real(1e-4), 2500, 0, 1, 1, ptr(x), ptr(y), ptr(l),
                ctypes.cast(ctypes.byref(optval), ctypes.c_void_p), ctypes.cast(ctypes.byref(it), ctypes.c_void_p))
        got = {"x": x, "y": y, "l": l, "optval": optval.value, "iterations": it.value, "status": st}
    assert want["status"] == 0 and want["iterations"] > 50
    _check(got, want, _tol(dtype, 1e-6, 2e-5), _tol(dtype, 2, max(5, int(0.05 * want["iterations"]))))
    assert relerr(got["l"], want["l"]) < _tol(dtype, 1e-5, 2e-3)
    # a solution of the problem, not just of the oracle: the objective in fp64 with numpy
    from helpers import objective

    assert objective(A.astype(np.float64), f, g, got["x"]) == pytest.approx(
        objective(A.astype(np.float64), f, g, want["x"]), rel=_tol(dtype, 1e-7, 1e-5))


@pytest.mark.parametrize("case", ["uniform", "c_varies", "all_vary", "h_varies"])
def test_sparse_prox_step_with_uniform_and_per_element_coefficients(case):
    """The sparse prox step does not stream coefficient arrays that hold one value throughout (vec_kernels.h:
    FnUniform, probed on the device per solve: h, c, d, e of a lasso) -- every mix of uniform and per-element
    arrays must give the oracle's solve (prox_lib.h:207-230: the coefficients enter per element)."""
    pogs = _pogs()
    from pogs_amd import synth

    m, n = 3000, 800
    A, b, _ = synth.csr_lasso(m, n, 20, seed=31, dtype=np.float64)
    f, g = pogs.graph.lasso_functions(b, 0.1, n)
    rng = np.random.default_rng(5)
    if case in ("c_varies", "all_vary"):
        f.c = rng.uniform(0.5, 2.0, m)
        g.c = 0.1 * rng.uniform(0.5, 2.0, n)
    if case == "all_vary":
        f.d = 0.01 * rng.standard_normal(m)
        f.e = rng.uniform(0.0, 0.2, m)
        g.d = 0.01 * rng.standard_normal(n)
        g.e = rng.uniform(0.0, 0.2, n)
    if case == "h_varies":
        g.h = np.where(rng.random(n) < 0.5, int(pogs.graph.Function.kAbs), int(pogs.graph.Function.kHuber)).astype(np.int32)
        f.h = np.where(rng.random(m) < 0.7, int(pogs.graph.Function.kSquare), int(pogs.graph.Function.kHuber)).astype(np.int32)
    want = ob.oracle_solve(A, soa(f), soa(g), dtype=np.float64)
    got = pogs.graph._solve_graph_form(A, f, g, dtype=np.float64)
    _check(got, want, 1e-6, 2)


def test_stamps_diagnostic_prints_per_xcd_times_and_changes_nothing(monkeypatch, capfd):
    """POGS_AMD_SELL_STAMPS=1: per-workgroup time stamps of both SpMVs, summed per XCC id, on stderr once per handle
    (sell.h / profiles/NOTES_r05.md: the evidence that the XCDs stream at the same rate); the solve is the same."""
    pogs = _pogs()
    from pogs_amd import synth

    A, b, _ = synth.csr_lasso(20000, 5000, 50, seed=3, dtype=np.float32)
    base = pogs.solve_lasso(A, b, 0.1, dtype=np.float32)
    capfd.readouterr()
    monkeypatch.setenv("POGS_AMD_SELL_STAMPS", "1")
    got = pogs.solve_lasso(A, b, 0.1, dtype=np.float32)
    err = capfd.readouterr().err
    lines = [ln for ln in err.splitlines() if ln.startswith("[pogs_amd stamps]")]
    assert len(lines) == 2 and "20000 x 5000" in lines[0] and "5000 x 20000" in lines[1], err[-800:]
    assert got["iterations"] == base["iterations"] and np.array_equal(got["x"], base["x"])


def _format_line(err, shape):
    lines = [ln for ln in err.splitlines() if ln.startswith("[pogs_amd trace] tiled copy") and shape in ln]
    assert lines, err[-600:]
    return lines[-1]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", ["two_per_tile", "single_elements", "mixed_long_rows"])
def test_two_slot_storage_gives_the_tag_storages_bits(dtype, shape, monkeypatch, capfd):
    """The tiled copy in its two storage formats (sell.h: a row tag per element, 8 B per stored fp32 element; a
    row-end bit in the local column + two id slots per batch of 4, 7 B) holds every row's elements in the same
    order, so both SpMVs return the same bits from either -- checked with each format pinned
    (POGS_AMD_SELL_FORMAT) on structures where the two-slot planner (a) needs no padding, (b) has to pad nearly
    every batch (single-element rows: three or four row ends per batch otherwise), (c) mixes rows far longer than a
    batch with single elements, with empty rows and one row that spans several classes.  Left to itself the
    planner keeps the smaller matrix: two slots for (a), tags for (b)."""
    pogs = _pogs()
    rng = np.random.default_rng(11)
    if shape == "two_per_tile":      # ~2.2 elements per (row, column block), as at C4
        m, n, per = 50000, 150000, 18
        rows = np.arange(m).repeat(per)
        A = sp.csr_matrix((rng.standard_normal(m * per), (rows, rng.integers(0, n, m * per))), shape=(m, n))
    elif shape == "single_elements":   # exactly one element per (row, fp32 column block of 18432): 3-4 single-element rows per stream
        m, n = 200000, 4 * 18432
        rows = np.arange(m).repeat(4)
        cols = (rng.integers(0, 18432, (m, 4)) + 18432 * np.arange(4)).ravel()
        A = sp.csr_matrix((rng.standard_normal(4 * m), (rows, cols)), shape=(m, n))
    else:
        m, n = 40000, 30000
        k = rng.choice([0, 1, 1, 1, 2, 3, 7, 40, 300], size=m)
        k[77] = 20000
        rows = np.arange(m).repeat(k)
        cols = np.concatenate([rng.choice(n, size=int(c), replace=False) for c in k if c > 0])
        A = sp.csr_matrix((rng.standard_normal(len(rows)), (rows, cols)), shape=(m, n))
    A = A.astype(dtype)
    A.sum_duplicates()
    A.sort_indices()
    x = rng.standard_normal(n).astype(dtype)
    y = rng.standard_normal(m).astype(dtype)
    monkeypatch.setenv("POGS_AMD_TRACE", "1")
    got = {}
    for fmt in ("tags", "two", "auto"):
        monkeypatch.setenv("POGS_AMD_SELL_FORMAT", fmt)
        capfd.readouterr()
        with pogs.Solver(A, dtype=dtype) as s:
            r = (s.mul("n", 1.0, x, 0.0, np.zeros(m, dtype)), s.mul("t", 1.0, y, 0.0, np.zeros(n, dtype)))
        err = capfd.readouterr().err
        got[fmt] = (r, _format_line(err, "%d x %d" % (m, n)), _format_line(err, "%d x %d" % (n, m)))
    assert "a row tag per element" in got["tags"][1] and "two id slots" in got["two"][1], (got["tags"][1], got["two"][1])
    for fmt in ("two", "auto"):
        assert np.array_equal(got[fmt][0][0], got["tags"][0][0]) and np.array_equal(got[fmt][0][1], got["tags"][0][1]), fmt
    # left alone the planner keeps the copy with fewer bytes: stored elements x (value + column + half an id | a tag)
    size = np.dtype(dtype).itemsize
    for k in (1, 2):
        stored = {fmt: float(got[fmt][k].split(",")[-1].split()[0]) for fmt in ("tags", "two")}
        want = "two id slots" if stored["two"] * (size + 3) < stored["tags"] * (size + 4) else "a row tag per element"
        assert want in got["auto"][k], (stored, got["auto"][k])
    if dtype == np.float32 and shape != "mixed_long_rows":
        assert ("two id slots" if shape == "two_per_tile" else "a row tag per element") in got["auto"][1], got["auto"][1]
    # and the product itself (the equilibrated matrix the handle holds)
    from pogs_amd import _lib

    monkeypatch.setenv("POGS_AMD_SELL_FORMAT", "two")
    with pogs.Solver(A, dtype=dtype) as s:
        buf = np.zeros(A.nnz, dtype)
        nrm = ctypes.c_double()
        assert _lib.lib.PogsAmdGetEquil(s._h, buf.ctypes.data_as(ctypes.c_void_p), None, None, ctypes.byref(nrm)) == 0
        As = sp.csr_matrix((buf.astype(np.float64), A.indices, A.indptr), shape=(m, n))
    tol = 1e-12 if dtype == np.float64 else 2e-5
    assert relerr(got["two"][0][0], As @ x.astype(np.float64)) < tol
    assert relerr(got["two"][0][1], As.T @ y.astype(np.float64)) < tol


def test_solve_is_the_same_in_both_storage_formats(monkeypatch):
    """A whole solve (Sinkhorn-Knopp passes over squared entries, CGLS loop, exact residuals) on the tag storage and
    on the two-slot storage: the same iterations, the same bits."""
    pogs = _pogs()
    from pogs_amd import synth

    A, b, _ = synth.csr_lasso(60000, 150000, 20, seed=5, dtype=np.float32)
    res = {}
    for fmt in ("tags", "two"):
        monkeypatch.setenv("POGS_AMD_SELL_FORMAT", fmt)
        res[fmt] = pogs.solve_lasso(A, b, 0.1, dtype=np.float32)
    assert res["two"]["status"] == 0 and res["two"]["iterations"] == res["tags"]["iterations"]
    assert np.array_equal(res["two"]["x"], res["tags"]["x"]) and np.array_equal(res["two"]["y"], res["tags"]["y"])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_cg_loop_variants_walk_the_same_trajectory(dtype, monkeypatch):
    """The device-resident CGLS loop (cg_fused.h) with y = A x from the CG recurrence every iteration
    (POGS_AMD_YSYNC=1000000), with the explicit product every 16th (default) and every iteration
    (POGS_AMD_YSYNC=0, the reference's projector_cgls.cpp:78), and round 2's host-polled loop
    (POGS_AMD_CG=host): the same iteration counts, the same CG step total, x within rounding -- and all
    of them on the oracle's trajectory."""
    pogs = _pogs()
    from pogs_amd import synth

    A, b, _ = synth.csr_lasso(20000, 5000, 50, seed=3, dtype=dtype)
    f, g = pogs.graph.lasso_functions(b, 0.1, 5000)
    want = ob.oracle_solve(A, soa(f), soa(g), dtype=dtype)
    res = {}
    for name, env in (("ysync16", {}), ("never", {"POGS_AMD_YSYNC": "1000000"}), ("always", {"POGS_AMD_YSYNC": "0"}),
                      ("host", {"POGS_AMD_CG": "host"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with pogs.Solver(A, dtype=dtype) as s:
            r = s.solve(f, g)
            st = s.stats()
        for k in env:
            monkeypatch.delenv(k)
        res[name] = (r, st)
    ref = res["always"][0]
    tol = 1e-9 if dtype == np.float64 else 2e-6
    for name, (r, st) in res.items():
        assert r["status"] == 0 and r["iterations"] == ref["iterations"], name
        assert st["cg_iters"] == res["always"][1]["cg_iters"], (name, st["cg_iters"])
        assert relerr(r["x"], ref["x"]) < tol, (name, relerr(r["x"], ref["x"]))
        assert relerr(r["y"], ref["y"]) < 10 * tol, name
    assert abs(int(ref["iterations"]) - int(want["iterations"])) <= (1 if dtype == np.float64 else 3)
    assert relerr(ref["x"], want["x"]) < (1e-6 if dtype == np.float64 else 2e-5)
    # fewer products with the recurrence: one per iteration less, except the synchronising ones
    assert res["never"][1]["matvecs"] < res["always"][1]["matvecs"] - 0.9 * (ref["iterations"] - 1)


@pytest.mark.parametrize("problem", ["ridge", "logistic", "svm", "nonneg_ls"])
def test_sparse_other_families(problem):
    pogs = _pogs()
    from helpers import PROBLEMS

    rng = np.random.default_rng(8)
    m, n = 1200, 300
    A = _rand_csr(m, n, 8, seed=9)
    b = A @ (rng.standard_normal(n) * (rng.random(n) < 0.2)) + 0.1 * rng.standard_normal(m)
    f, g = PROBLEMS[problem](b, n)
    got = pogs._solve_graph_form(A, f, g)
    want = ob.oracle_solve(A, soa(f), soa(g), dtype=np.float64)
    _check(got, want, 1e-5, 3)


def test_csc_input_through_raw_abi():
    """ORD = COL_MAJ: data/ptr/ind describe CSC (src/interface_c/pogs_c.h:92-98)."""
    pogs = _pogs()
    from pogs_amd import _lib, synth

    A, b, _ = synth.csr_lasso(1000, 300, 10, seed=12, dtype=np.float64)
    f, g = pogs.graph.lasso_functions(b, 0.1, 300)
    want = pogs._solve_graph_form(A, f, g)
    C = A.tocsc()
    C.sort_indices()
    fa, ga = f.arrays(np.float64), g.arrays(np.float64)
    m, n = A.shape
    x, y, l = np.zeros(n), np.zeros(m), np.zeros(m)
    optval, fi = ctypes.c_double(), ctypes.c_uint()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    data = np.ascontiguousarray(C.data)
    ptr = np.ascontiguousarray(C.indptr, np.int32)
    ind = np.ascontiguousarray(C.indices, np.int32)
    st = _lib.lib.PogsSparseD(0, m, n, C.nnz, p(data), p(ptr), p(ind), *[p(fa[k]) for k in "abcdeh"],
                              *[p(ga[k]) for k in "abcdeh"], 1.0, 1e-4, 1e-4, 2500, 0, 1, 1, p(x), p(y), p(l),
                              ctypes.cast(ctypes.byref(optval), ctypes.c_void_p),
                              ctypes.cast(ctypes.byref(fi), ctypes.c_void_p))
    assert st == 0 == want["status"]
    assert abs(fi.value - want["iterations"]) <= 1
    assert relerr(x, want["x"]) < 1e-6


def _messy_csr(m, n, per_row, seed):
    """CSR arrays the reference accepts and scipy would canonicalise away: column indices in random
    order inside each row, and repeated (row, column) entries left un-summed (the reference's product
    is a plain gather over ptr / ind, src/cpu/include/gsl/gsl_spblas.h:16-40: indifferent to both)."""
    rng = np.random.default_rng(seed)
    ptr, ind, val = [0], [], []
    for i in range(m):
        k = int(rng.integers(0, 2 * per_row))
        c = rng.integers(0, n, size=k)
        if k >= 3:
            c[1] = c[0]                      # a duplicate
            if i % 5 == 0:
                c[2] = c[0]                  # ... sometimes a triple
        rng.shuffle(c)
        ind += list(c)
        val += list(rng.standard_normal(k))
        ptr.append(len(ind))
    return np.array(val), np.array(ptr, np.int32), np.array(ind, np.int32)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("order", ["csr", "csc"])
def test_unsorted_indices_and_unsummed_duplicates_through_raw_abi(dtype, order):
    """"int32 indices (unsorted allowed)".  PogsSparseD/S fed a matrix with
    shuffled column indices and repeated entries, as CSR and as CSC, against the oracle run on the
    very same arrays (and the compiled reference where it is available)."""
    pogs = _pogs()
    from pogs_amd import _lib

    m, n = 2500, 700
    if order == "csr":
        val, ptr, ind = _messy_csr(m, n, 9, seed=21)
        A_raw = sp.csr_matrix((val.astype(dtype), ind, ptr), shape=(m, n))       # kept as given: no sum, no sort
        assert not A_raw.has_canonical_format
    else:
        val, ptr, ind = _messy_csr(n, m, 30, seed=22)                             # the rows of A^T
        A_raw = sp.csc_matrix((val.astype(dtype), ind, ptr), shape=(m, n))
    A_canon = sp.csr_matrix(A_raw.astype(np.float64)).copy()
    A_canon.sum_duplicates()
    rng = np.random.default_rng(23)
    b = A_canon @ (rng.standard_normal(n) * (rng.random(n) < 0.2)) + 0.1 * rng.standard_normal(m)
    f, g = pogs.graph.lasso_functions(b, 0.1, n)
    fa, ga = f.arrays(dtype), g.arrays(dtype)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    x, y, l = np.zeros(n, dtype), np.zeros(m, dtype), np.zeros(m, dtype)
    real = ctypes.c_double if dtype == np.float64 else ctypes.c_float
    optval, fi = real(), ctypes.c_uint()
    data = np.ascontiguousarray(val.astype(dtype))
    fn = _lib.lib.PogsSparseD if dtype == np.float64 else _lib.lib.PogsSparseS
    st = fn(1 if order == "csr" else 0, m, n, len(data), p(data), p(ptr), p(ind), *[p(fa[k]) for k in "abcdeh"],
            *[p(ga[k]) for k in "abcdeh"], 1.0, 1e-4, 1e-4, 2500, 0, 1, 1, p(x), p(y), p(l),
            ctypes.cast(ctypes.byref(optval), ctypes.c_void_p), ctypes.cast(ctypes.byref(fi), ctypes.c_void_p))
    assert st == 0, _lib.last_error()
    # the oracle on the same arrays (CSR as given; for CSC the oracle takes the equivalent raw CSR of A)
    A_or = A_raw if order == "csr" else sp.csr_matrix((val.astype(dtype), ind, ptr), shape=(n, m)).T.tocsr()
    want = ob.oracle_solve(A_or, soa(f), soa(g), dtype=dtype)
    assert want["status"] == 0
    d_it = abs(int(fi.value) - int(want["iterations"]))
    assert d_it <= (1 if dtype == np.float64 else 3)
    tol = 1e-6 if dtype == np.float64 else (2e-5 if d_it == 0 else 1e-4 * (1 + d_it))
    assert relerr(x, want["x"]) < tol
    assert float(optval.value) == pytest.approx(want["optval"], rel=1e-6 if dtype == np.float64 else 2e-4)
    # The canonical form of the same matrix is the same PROBLEM (the product adds repeated entries), but
    # not the same run: equilibration and the Frobenius norm work on the stored entries (a^2 + b^2 for a
    # repeated pair where the summed matrix has (a + b)^2; matrix_sparse.cpp:158-242), so the scaled
    # problem and the stopping point differ -- measured 1.2e-3 in x at the default tolerances.  Both are
    # solutions of the same lasso to the stopping rule's accuracy:
    canon = pogs._solve_graph_form(A_canon.astype(dtype), f, g, dtype=dtype)
    assert canon["status"] == 0 and relerr(x, canon["x"]) < 1e-2
    obj = lambda v: 0.5 * np.sum((A_canon @ v.astype(np.float64) - b) ** 2) + 0.1 * np.abs(v).sum()  # noqa: E731
    assert obj(x) == pytest.approx(obj(canon["x"]), rel=1e-3)
    if ob.ref_available() and order == "csr":
        ref = ob.ref_solve(A_raw, soa(f), soa(g), dtype=dtype)
        assert ref["status"] == 0 and abs(int(ref["iterations"]) - int(fi.value)) <= (1 if dtype == np.float64 else 3)
        assert relerr(x, ref["x"]) < (1e-6 if dtype == np.float64 else 1e-4)


def test_duplicate_entries_give_the_same_bits_run_to_run():
    """The device transpose delivers repeated entries of a column in no particular order; the sort
    breaks those ties by value, so two handles on the same input hold the same bits."""
    pogs = _pogs()
    from pogs_amd import _lib

    m, n = 3000, 500
    val, ptr, ind = _messy_csr(m, n, 12, seed=31)
    A = sp.csr_matrix((val.astype(np.float32), ind, ptr), shape=(m, n))
    rng = np.random.default_rng(32)
    v = rng.standard_normal(m).astype(np.float32)
    outs = []
    for _ in range(3):
        with pogs.Solver(A, dtype=np.float32) as s:   # (scipy keeps the arrays as given: no sum, no sort)
            outs.append(s.mul("t", 1.0, v, 0.0, np.zeros(n, np.float32)))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_device_resident_csr_gives_the_host_paths_bits():
    """PogsAmdCreateSparse with mem = POGS_AMD_DEVICE (bench.py hands the CSR arrays over in HBM, as it does
    the dense matrices): same handle contents, same solve, bit for bit."""
    import torch

    pogs = _pogs()
    from pogs_amd import synth

    A, b, _ = synth.csr_lasso(6000, 1500, 25, seed=14, dtype=np.float32)
    f, g = pogs.graph.lasso_functions(b, 0.1, 1500)
    with pogs.Solver(A, dtype=np.float32) as s:
        host = s.solve(f, g)
    dev = torch.device("cuda:0")
    data = torch.from_numpy(np.ascontiguousarray(A.data, np.float32)).to(dev)
    ptr = torch.from_numpy(np.ascontiguousarray(A.indptr, np.int32)).to(dev)
    ind = torch.from_numpy(np.ascontiguousarray(A.indices, np.int32)).to(dev)
    keep = (data.clone(), ptr.clone(), ind.clone())
    with pogs.Solver((data.data_ptr(), ptr.data_ptr(), ind.data_ptr(), A.nnz), dtype=np.float32, shape=A.shape,
                     device_ptr=True) as s:
        got = s.solve(f, g)
    assert got["status"] == host["status"] == 0 and got["iterations"] == host["iterations"]
    for k in "xyl":
        assert np.array_equal(got[k], host[k]), k
    # the caller's arrays are never written (inputs are copied)
    assert torch.equal(data, keep[0]) and torch.equal(ptr, keep[1]) and torch.equal(ind, keep[2])


def test_sparse_warm_start_matches_oracle():
    """Warm start (pogs.cpp:144-156) on the CSR + CGLS path."""
    pogs = _pogs()
    from pogs_amd import synth

    A, b, _ = synth.csr_lasso(3000, 800, 20, seed=8, dtype=np.float64)
    f1, g1 = pogs.graph.lasso_functions(b, 0.3, 800)
    f2, g2 = pogs.graph.lasso_functions(b, 0.2, 800)
    w1 = ob.oracle_solve(A, soa(f1), soa(g1), dtype=np.float64)
    cold = ob.oracle_solve(A, soa(f2), soa(g2), dtype=np.float64)
    w2 = ob.oracle_solve(A, soa(f2), soa(g2), dtype=np.float64, x0=w1["x"], l0=w1["l"])
    assert w2["iterations"] < cold["iterations"]
    with pogs.Solver(A, dtype=np.float64) as s:
        r2 = s.solve(f2, g2, x0=w1["x"], l0=w1["l"])
        r3 = s.solve(f2, g2)
    _check(r2, w2, 1e-6, 2)
    _check(r3, cold, 1e-6, 2)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_column_blocked_spmv_several_blocks(dtype):
    """Shapes past one LDS column block (28672 fp32 / 10240 fp64 columns) in both directions:
    partial sums per (block, row) + ordered reduction; an empty row, a long row, duplicates."""
    pogs = _pogs()
    rng = np.random.default_rng(5)
    m, n = 70000, 61000
    k = 6
    rows = np.repeat(np.arange(m), k)
    cols = rng.integers(0, n, m * k)
    vals = rng.standard_normal(m * k)
    # one long row (> 4096 non-zeros in a single column block) and one empty row
    long_cols = rng.choice(20000, 9000, replace=False)
    rows = np.concatenate([rows[rows != 17], np.full(9000, 12345)])
    cols = np.concatenate([cols[: len(rows) - 9000], long_cols])
    vals = np.concatenate([vals[: len(rows) - 9000], rng.standard_normal(9000)])
    A = sp.csr_matrix((vals, (rows, cols)), shape=(m, n)).astype(dtype)
    A.sum_duplicates()
    assert A.indptr[18] == A.indptr[17]
    x = rng.standard_normal(n).astype(dtype)
    y = rng.standard_normal(m).astype(dtype)
    tol = 1e-12 if dtype == np.float64 else 2e-5
    A.sort_indices()
    with pogs.Solver(A, dtype=dtype) as s:
        from pogs_amd import _lib

        buf = np.zeros(A.nnz, dtype)
        nrm = ctypes.c_double()
        assert _lib.lib.PogsAmdGetEquil(s._h, buf.ctypes.data_as(ctypes.c_void_p), None, None, ctypes.byref(nrm)) == 0
        As = sp.csr_matrix((buf.astype(np.float64), A.indices, A.indptr), shape=(m, n))
        got = s.mul("n", 1.0, x, 0.0, np.zeros(m, dtype))
        assert relerr(got, As @ x.astype(np.float64)) < tol
        got_t = s.mul("t", 2.0, y, 0.5, x.copy())
        assert relerr(got_t, 2.0 * (As.T @ y.astype(np.float64)) + 0.5 * x) < tol


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_column_blocked_spmv_skewed_structure(dtype):
    """Adjacent nearly dense rows inside one column block (their fragment has to be split at the
    16-bit offset limit), a fully dense column (a 40000-entry row of the transpose), empty rows at
    both ends."""
    pogs = _pogs()
    from pogs_amd import _lib

    rng = np.random.default_rng(3)
    m0, n, k = 40000, 61000, 5
    A = sp.csr_matrix((rng.standard_normal(m0 * k), (np.repeat(np.arange(m0), k), rng.integers(0, n, m0 * k))),
                      shape=(m0, n)).tolil()
    for r in range(100, 106):
        A[r, rng.choice(28000, 26000, replace=False)] = rng.standard_normal(26000)
    A = A.tocsr() + sp.csr_matrix((rng.standard_normal(m0), (np.arange(m0), np.full(m0, n - 7))), shape=(m0, n))
    A = sp.vstack([sp.csr_matrix((3, n)), A, sp.csr_matrix((2, n))]).tocsr().astype(dtype)
    A.sum_duplicates()
    A.sort_indices()
    m = A.shape[0]
    x, y = rng.standard_normal(n).astype(dtype), rng.standard_normal(m).astype(dtype)
    tol = 1e-12 if dtype == np.float64 else 2e-5
    with pogs.Solver(A, dtype=dtype) as s:
        buf = np.zeros(A.nnz, dtype)
        nrm = ctypes.c_double()
        assert _lib.lib.PogsAmdGetEquil(s._h, buf.ctypes.data_as(ctypes.c_void_p), None, None, ctypes.byref(nrm)) == 0
        As = sp.csr_matrix((buf.astype(np.float64), A.indices, A.indptr), shape=(m, n))
        assert relerr(s.mul("n", 1.0, x, 0.0, np.zeros(m, dtype)), As @ x.astype(np.float64)) < tol
        assert relerr(s.mul("t", 1.0, y, 0.0, np.zeros(n, dtype)), As.T @ y.astype(np.float64)) < tol


def test_plain_csr_kernel_still_matches(monkeypatch):
    """POGS_AMD_SPMV=plain keeps the un-blocked CSR-stream kernel (the fallback for shapes whose
    per-(block, row) bookkeeping would outweigh the non-zeros): same solve, same iterations."""
    pogs = _pogs()
    from pogs_amd import synth

    A, b, _ = synth.csr_lasso(3000, 800, 20, seed=4, dtype=np.float64)
    blocked = pogs.solve_lasso(A, b, 0.1)
    monkeypatch.setenv("POGS_AMD_SPMV", "plain")
    plain = pogs.solve_lasso(A, b, 0.1)
    assert plain["status"] == blocked["status"] == 0
    assert abs(int(plain["iterations"]) - int(blocked["iterations"])) <= 1
    assert relerr(plain["x"], blocked["x"]) < 1e-8


@pytest.mark.parametrize("dtype,world", [(np.float64, 2), (np.float32, 3)])
def test_row_sharded_sparse_engine_matches_single_rank(dtype, world):
    """CSR row blocks over several ranks (threads + the in-process,
    stream-ordered test communicator, one GPU), the device-resident CGLS loop with its two all-reduces
    per step.  Held against the ORACLE -- the unsharded restatement (oracle_solve, pinned to the
    compiled reference) and its row-sharded entry with an in-test sum as the collective
    (projector_cgls.cpp:59-78, cgls.h:200-323 with the A^T products and |q|^2 summed over the ranks) --
    and, as a consistency check, against the unsharded engine."""
    pogs = _pogs()
    from helpers import run_row_sharded, run_sharded_oracle
    from pogs_amd import synth

    m, n = 4001, 900
    A, b, _ = synth.csr_lasso(m, n, 15, seed=12, dtype=dtype)
    f, g = pogs.graph.lasso_functions(b, 0.1, n)
    want = ob.oracle_solve(A, soa(f), soa(g), dtype=dtype)
    want_sh, _ = run_sharded_oracle(A.tocsr(), f, g, world, dtype)
    with pogs.Solver(A, dtype=dtype) as s:
        one = s.solve(f, g)
    res, bounds = run_row_sharded(pogs, A.tocsr(), f, g, world, dtype)
    assert want["status"] == 0 and all(w["status"] == 0 for w in want_sh)

    def tol_for(it_a, it_b):
        # fp32: 2e-5 on the same trajectory, tolerance-sized when the inexact CGLS projections (whose
        # A^T products are summed in another order across shards) stop the solves a few iterations apart
        d_it = abs(int(it_a) - int(it_b))
        return 1e-7 if dtype == np.float64 else (2e-5 if d_it == 0 else min(5e-4, 1e-4 * (1 + d_it)))

    it_slack = 1 if dtype == np.float64 else 5
    for r, out in enumerate(res):
        lo, hi = bounds[r], bounds[r + 1]
        assert out["status"] == 0
        # the oracle, unsharded: the same tolerances as the unsharded engine tests of this file
        tol = tol_for(out["iterations"], want["iterations"])
        assert abs(int(out["iterations"]) - int(want["iterations"])) <= it_slack
        assert relerr(out["x"], want["x"]) < tol
        assert relerr(out["y"], want["y"][lo:hi]) < tol * 10
        assert relerr(out["l"], want["l"][lo:hi]) < tol * 100
        assert out["optval"] == pytest.approx(want["optval"], rel=max(tol, 1e-6 if dtype == np.float64 else 2e-4))
        # the oracle's own row-sharded run, rank by rank
        ws = want_sh[r]
        tol = tol_for(out["iterations"], ws["iterations"])
        assert abs(int(out["iterations"]) - int(ws["iterations"])) <= it_slack
        assert relerr(out["x"], ws["x"]) < tol
        assert relerr(out["y"], ws["y"]) < tol * 10
        # and the unsharded engine
        tol = tol_for(out["iterations"], one["iterations"])
        assert abs(int(out["iterations"]) - int(one["iterations"])) <= it_slack
        assert relerr(out["x"], one["x"]) < tol
        assert relerr(out["y"], one["y"][lo:hi]) < tol * 10
    for out in res[1:]:
        assert out["iterations"] == res[0]["iterations"]
        assert np.array_equal(out["x"], res[0]["x"])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_row_sharded_sparse_unequal_shards(dtype):
    """Clearly unequal row shards (2000 / 3100 rows, and a 300 / 4800 split whose two ranks launch different
    grids): the replicated sums of the device-resident CG loop (|x|^2, |s|^2 -> beta, the stopping test, the
    number of steps and with it the number of collectives a rank issues) must be bit-identical on every rank
    whatever its local row count (cg_fused.h: CgfStepA::nb_n) -- a rank that took one step more
    than its peer would hang in a collective nobody else joins."""
    pogs = _pogs()
    from helpers import run_row_sharded, run_sharded_oracle
    from pogs_amd import synth

    m, n = 5100, 1200
    A, b, _ = synth.csr_lasso(m, n, 15, seed=13, dtype=dtype)
    f, g = pogs.graph.lasso_functions(b, 0.1, n)
    want = ob.oracle_solve(A, soa(f), soa(g), dtype=dtype)
    for bounds in ([0, 2000, 5100], [0, 300, 5100]):
        res, bd = run_row_sharded(pogs, A.tocsr(), f, g, 2, dtype, bounds=bounds)
        want_sh, _ = run_sharded_oracle(A.tocsr(), f, g, 2, dtype, bounds=bounds)
        assert list(bd) == bounds
        assert res[0]["iterations"] == res[1]["iterations"] and np.array_equal(res[0]["x"], res[1]["x"])
        it_slack = 1 if dtype == np.float64 else 5
        for r, out in enumerate(res):
            d_it = abs(int(out["iterations"]) - int(want["iterations"]))
            tol = 1e-7 if dtype == np.float64 else (2e-5 if d_it == 0 else min(5e-4, 1e-4 * (1 + d_it)))
            assert out["status"] == 0 and d_it <= it_slack
            assert relerr(out["x"], want["x"]) < tol
            assert relerr(out["y"], want["y"][bounds[r]:bounds[r + 1]]) < tol * 10
            assert abs(int(out["iterations"]) - int(want_sh[r]["iterations"])) <= it_slack
            assert relerr(out["x"], want_sh[r]["x"]) < (1e-7 if dtype == np.float64 else 5e-4)


@pytest.mark.parametrize("mode", ["host", "cg_h"])
def test_row_sharded_sparse_other_transport_and_loop(monkeypatch, mode):
    """The same sharded solve with the host-staged form of the test transport (stream order then plays
    no role: a difference from the stream-ordered run would be an ordering bug) and with round 2's
    host-polled CG loop (POGS_AMD_CG=h, also what the plain-CSR fallback runs): same solution."""
    pogs = _pogs()
    from helpers import run_row_sharded
    from pogs_amd import synth

    m, n = 4001, 900
    A, b, _ = synth.csr_lasso(m, n, 15, seed=12, dtype=np.float64)
    f, g = pogs.graph.lasso_functions(b, 0.1, n)
    base, _ = run_row_sharded(pogs, A.tocsr(), f, g, 2, np.float64)
    if mode == "host":
        monkeypatch.setenv("POGS_AMD_TEST_TRANSPORT", "host")
        other, _ = run_row_sharded(pogs, A.tocsr(), f, g, 2, np.float64, transport="host")
        assert other[0]["iterations"] == base[0]["iterations"]
        assert np.array_equal(other[0]["x"], base[0]["x"])      # both sum in rank order: the same bits
    else:
        monkeypatch.setenv("POGS_AMD_CG", "h")
        other, _ = run_row_sharded(pogs, A.tocsr(), f, g, 2, np.float64)
        assert abs(int(other[0]["iterations"]) - int(base[0]["iterations"])) <= 1
        assert relerr(other[0]["x"], base[0]["x"]) < 1e-7


def test_sparse_one_rank_rccl_path():
    """The sharded sparse code path over RCCL itself (1-rank communicator)."""
    pogs = _pogs()
    from pogs_amd import synth

    A, b, _ = synth.csr_lasso(3000, 800, 20, seed=4, dtype=np.float32)
    f, g = pogs.graph.lasso_functions(b, 0.1, 800)
    with pogs.Solver(A, dtype=np.float32) as s:
        r0 = s.solve(f, g)
    with pogs.Solver(A, dtype=np.float32, dist=(0, 1, 3000, pogs.dist_unique_id())) as s:
        r1 = s.solve(f, g)
    assert r0["status"] == r1["status"] == 0
    assert abs(int(r0["iterations"]) - int(r1["iterations"])) <= 2
    assert relerr(r1["x"], r0["x"]) < 1e-4
