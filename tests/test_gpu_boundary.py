"""Boundary behaviours of the C ABI that the reference shows to its callers:
the non-convex-coefficient warning and clamp, the verbose summary, error returns instead of
faults, and the device a handle lives on."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_binding as ob
from helpers import relerr, soa

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _pogs():
    import pogs_amd

    return pogs_amd


def test_negative_c_and_e_are_clamped_with_the_reference_warning(capfd):
    """FunctionObj::CheckConsts (src/include/prox_lib.h:62-69): c < 0 or e < 0 prints
    'WARNING c < 0. Function not convex. Using c = 0' (resp. e) and the object is used with 0.
    The solve must equal the oracle's (which restates the clamp) and the solve with the
    coefficients already clamped."""
    pogs = _pogs()
    from pogs_amd import synth

    A, b, _ = synth.dense_lasso(400, 60, seed=3, dtype=np.float64)
    f, g = pogs.graph.lasso_functions(b, 0.1, 60)
    g.e[:] = 0.05
    g.h[5] = int(pogs.Function.kZero)
    g.c[5] = -1.0       # c < 0 on a function whose prox ignores c: the clamp must not break the solve
    g.e[7] = -0.5       # two negative e
    f.e[3] = -1.0
    got = pogs._solve_graph_form(A, f, g)
    out = capfd.readouterr().out
    assert out.count("WARNING c < 0. Function not convex. Using c = 0") == 1
    assert out.count("WARNING e < 0. Function not convex. Using e = 0") == 2
    fc, gc = f.slice(0, 400), g.slice(0, 60)
    gc.c[5] = 0.0
    gc.e[7] = 0.0
    fc.e[3] = 0.0
    clamped = pogs._solve_graph_form(A, fc, gc)
    assert "WARNING" not in capfd.readouterr().out
    want = ob.oracle_solve(A, soa(f), soa(g))     # 59 iterations, as the compiled reference
    assert got["status"] == clamped["status"] == want["status"] == 0
    assert got["iterations"] == clamped["iterations"]
    assert np.array_equal(got["x"], clamped["x"])
    assert abs(int(got["iterations"]) - int(want["iterations"])) <= 2
    assert relerr(got["x"], want["x"]) < 1e-6
    # many offenders: the first eight are printed, then a count
    f.e[:100] = -1.0
    pogs._solve_graph_form(A, f, g, max_iter=3)
    out = capfd.readouterr().out
    assert out.count("WARNING e < 0. Function not convex. Using e = 0") == 8 + 1   # f: the first eight of 100; g: its one
    assert "WARNING e < 0 in 92 more function objects" in out
    # a negative c where it matters makes the problem degenerate in the reference as well
    # (c = 0 turns the prox parameter into infinity): same status and count as the oracle
    f2, g2 = pogs.graph.lasso_functions(b, 0.1, 60)
    f2.c[3] = -2.0
    got2 = pogs._solve_graph_form(A, f2, g2, max_iter=200)
    want2 = ob.oracle_solve(A, soa(f2), soa(g2), max_iter=200)
    assert got2["status"] == want2["status"] and got2["iterations"] == want2["iterations"]


@pytest.mark.parametrize("sparse", [False, True])
def test_verbose_summary_in_the_reference_format(capfd, sparse):
    """verbose >= 1: status / timing / iteration summary and the three normalised error metrics
    (src/cpu/pogs.cpp:485-500); verbose >= 2: header and per-iteration lines with the primal
    objective (pogs.cpp:193-195, 382-388)."""
    pogs = _pogs()
    from pogs_amd import synth

    if sparse:
        A, b, _ = synth.csr_lasso(1500, 300, 12, seed=4, dtype=np.float64)
    else:
        A, b, _ = synth.dense_lasso(600, 80, seed=4, dtype=np.float64)
    r = pogs.solve_lasso(A, b, 0.1, verbose=2)
    out = capfd.readouterr().out
    assert "Status: Solved" in out
    assert "Timing: Total = " in out and ", Init = " in out
    assert "Iter  : %u" % r["iterations"] in out
    for key in ("Error Metrics:", "Pri: |Ax - y|    / (abs_tol sqrt(m)     / rel_tol + |y|)          = ",
                "Dua: |A'l + u|   / (abs_tol sqrt(n)     / rel_tol + |u|)          = ",
                "Gap: |x'u + y'l| / (abs_tol sqrt(m + n) / rel_tol + |x,u| |y,l|)  = "):
        assert key in out, key
    assert " Iter | pri res | pri tol | dua res | dua tol |   gap   | eps gap | pri obj" in out
    lines = [ln for ln in out.splitlines() if ln[:5].strip().isdigit() and " : " in ln]
    assert lines and lines[0].split(":")[0].strip() == "0"
    last = lines[-1].split()
    assert int(last[0]) == r["iterations"]
    assert float(last[-1]) == pytest.approx(r["optval"], rel=1e-2)   # printed with 3 digits
    # the normalised metrics are below rel_tol at convergence
    for tag in ("Pri:", "Dua:", "Gap:"):
        val = float([ln for ln in out.splitlines() if ln.startswith(tag)][0].split("=")[-1])
        assert 0 <= val < 1e-4
    pogs.solve_lasso(A, b, 0.1, verbose=0)
    assert capfd.readouterr().out == ""
    # verbose > 3 (pogs.cpp:206, 432-459, 501-506): the adaptive-rho messages, a line every 10 iterations,
    # the per-iteration timing breakdown; the oracle prints the same rho messages for the same run
    r4 = pogs.solve_lasso(A, b, 0.1, verbose=4)
    out4 = capfd.readouterr().out
    rho_lines = [ln for ln in out4.splitlines() if ln.startswith(("+ rho ", "- rho ", "spectral rho update: "))]
    assert rho_lines, "no adaptive-rho message at verbose = 4"
    assert all(float(ln.split()[2 if ln[0] in "+-" else 3]) > 0 for ln in rho_lines)
    tb = [ln for ln in out4.splitlines() if ln.startswith("Timing breakdown (per-iter avg): prox = ")]
    assert len(tb) == 1 and ", proj = " in tb[0] and ", residual = " in tb[0]
    assert r4["iterations"] == r["iterations"]


def test_iterate_before_begin_run_is_an_error_not_a_garbage_run():
    pogs = _pogs()
    from pogs_amd import synth

    A, b, _ = synth.dense_lasso(300, 40, seed=5, dtype=np.float32)
    with pogs.Solver(A, dtype=np.float32) as s:
        with pytest.raises(RuntimeError, match="BeginRun"):
            s.iterate(3)
        f, g = pogs.graph.lasso_functions(b, 0.1, 40)
        s.begin_run(f, g)
        s.iterate(3)
    As, bs, _ = synth.csr_lasso(500, 100, 8, seed=5, dtype=np.float32)
    with pogs.Solver(As, dtype=np.float32) as s:
        with pytest.raises(RuntimeError, match="BeginRun"):
            s.iterate(1)


def test_malformed_sparse_input_returns_pogs_error():
    """Indices out of range / a decreasing ptr must come back as POGS_ERROR (6), not as
    out-of-bounds device writes."""
    pogs = _pogs()
    from pogs_amd import _lib, synth

    A, b, _ = synth.csr_lasso(200, 50, 5, seed=6, dtype=np.float64)
    f, g = pogs.graph.lasso_functions(b, 0.1, 50)
    fa, ga = f.arrays(np.float64), g.arrays(np.float64)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731

    def call(data, ptr, ind):
        x, y, l = np.zeros(50), np.zeros(200), np.zeros(200)
        ov, it = ctypes.c_double(), ctypes.c_uint()
        return _lib.lib.PogsSparseD(1, 200, 50, len(data), P(data), P(ptr), P(ind),
                                    *[P(fa[k]) for k in "abcdeh"], *[P(ga[k]) for k in "abcdeh"],
                                    1.0, 1e-4, 1e-4, 100, 0, 1, 1, P(x), P(y), P(l), ctypes.byref(ov), ctypes.byref(it))

    data, ptr, ind = A.data.copy(), A.indptr.astype(np.int32), A.indices.astype(np.int32)
    assert call(data, ptr, ind) == 0
    bad = ind.copy()
    bad[17] = 50
    assert call(data, ptr, bad) == 6 and "index" in _lib.last_error()
    bad[17] = -1
    assert call(data, ptr, bad) == 6
    badp = ptr.copy()
    badp[10], badp[11] = ptr[11], ptr[10]
    if badp[11] < badp[10]:
        assert call(data, badp, ind) == 6 and "ptr" in _lib.last_error()


def test_handle_keeps_its_device_and_leaves_the_callers_device_alone():
    """The device is selected per entry point, not once at creation (HIP's current device is
    per thread): a handle used from another thread works, and creation with an explicit device
    leaves the caller's current device as it was."""
    import threading

    import torch

    pogs = _pogs()
    from pogs_amd import synth

    A, b, _ = synth.dense_lasso(500, 64, seed=7, dtype=np.float32)
    f, g = pogs.graph.lasso_functions(b, 0.1, 64)
    before = torch.cuda.current_device()
    s = pogs.Solver(A, dtype=np.float32, device=0)
    assert torch.cuda.current_device() == before
    want = s.solve(f, g)
    got = {}

    def other_thread():
        got["r"] = s.solve(f, g)
        got["eq"] = s.equilibrated(want_matrix=False)[3]

    t = threading.Thread(target=other_thread)
    t.start()
    t.join(120)
    assert got["r"]["status"] == 0 and np.array_equal(got["r"]["x"], want["x"])
    s.close()
    assert torch.cuda.current_device() == before


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dense64", "dense32", "colmajor64", "csr64", "handle64"])
def test_plain_c_caller_gets_the_python_bindings_answer(tmp_path, kind):
    """tests/c_caller/graph_form_driver.c (gcc, C99, linked with -lpogs_amd only) calls PogsD /
    PogsS / PogsSparseD like examples/c/lasso.c:102-106 on inputs written by this test (handle64: the additive
    handle entry points with broadcast coefficients, PogsAmdCreateDense + PogsAmdSolveFn); what it
    prints must be what the python binding returns for the same arrays (same library, same
    entry points: bit-identical) and within the usual tolerance of the oracle."""
    import subprocess

    import oracle_binding as ob
    import scipy.sparse as sp
    from test_abi import build_c_driver

    pogs = _pogs()
    m, n, lam = 600, 90, 0.3
    rng = np.random.default_rng(77)
    A = rng.standard_normal((m, n))
    if kind == "csr64":
        A *= rng.random((m, n)) < 0.25
    b = rng.standard_normal(m)
    data = tmp_path / "in.bin"
    with open(data, "wb") as fh:
        fh.write(np.ascontiguousarray(A, np.float64).tobytes())
        fh.write(np.ascontiguousarray(b, np.float64).tobytes())
    exe = build_c_driver(tmp_path)
    r = subprocess.run([exe, kind, str(m), str(n), repr(lam), str(data)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-2000:]
    out = {ln.split()[0]: np.array(ln.split()[1:], dtype=np.float64) for ln in r.stdout.splitlines() if ln.strip()}
    assert int(out["status"][0]) == 0
    dtype = np.float32 if kind == "dense32" else np.float64
    Ain = sp.csr_matrix(A) if kind == "csr64" else (np.asfortranarray(A) if kind == "colmajor64" else A)
    want = pogs.solve_lasso(Ain, b, lam, dtype=dtype)
    assert int(out["final_iter"][0]) == want["iterations"]
    assert np.array_equal(out["x"].astype(dtype), want["x"])
    assert np.array_equal(out["y"].astype(dtype), want["y"])
    assert np.array_equal(out["l"].astype(dtype), want["l"])
    assert dtype(out["optval"][0]) == dtype(want["optval"])
    f, g = pogs.graph.lasso_functions(b, lam, n)
    soa = lambda fv: {k: getattr(fv, k) for k in "habcde"}  # noqa: E731
    orc = ob.oracle_solve(Ain if kind == "csr64" else A, soa(f), soa(g), dtype=dtype)
    assert orc["iterations"] == want["iterations"]
    assert np.linalg.norm(out["x"] - orc["x"]) <= (2e-5 if dtype == np.float32 else 1e-9) * np.linalg.norm(orc["x"])


@pytest.mark.gpu
def test_cvxpy_front_end_runs_the_engine():
    """pogs_amd.pogs_solve (reference: python/pogs/cvxpy.py:33-92) on stand-in CVXPY problems: the
    detected lasso / ridge / non-negative least squares are solved by the HIP engine, the variable
    receives x, and the returned value is the CVXPY objective (optimum scaled back by 2 s)."""
    import cvxpy_standins as S

    pogs = _pogs()
    cases = S.cases()
    A = cases["lasso"].objective.expr.args[0].args[0].args[0].args[0].value
    b = -cases["lasso"].objective.expr.args[0].args[0].args[1].value
    for name, objective in (("lasso", lambda x: np.sum((A @ x - b) ** 2) + 0.3 * np.abs(x).sum()),
                            ("ridge", lambda x: 2 * np.sum((A @ x - b) ** 2) + 0.6 * np.sum(x * x)),
                            ("nnls", lambda x: 0.5 * np.sum((A @ x - b) ** 2))):     # (reference quirk: 1/2 |.|^2, unscaled)
        p = cases[name]
        value = pogs.pogs_solve(p, abs_tol=1e-6, rel_tol=1e-6)
        x = p.variables()[0].value
        assert p._status == "optimal" and p._value == value and not p.fallback_calls, name
        assert value == pytest.approx(objective(np.asarray(x, np.float64)), rel=1e-4), name
        if name == "nnls":
            assert np.all(x >= -1e-9)
    want = pogs.solve_lasso(A, b, 0.15, abs_tol=1e-6, rel_tol=1e-6)
    assert np.array_equal(cases["lasso"].variables()[0].value, want["x"])


_REF_LOADER_SCRIPT = r"""
# What python/pogs/graph.py does, restated: find `libpogs_cpu.so` next to the package file
# (graph.py:29-67, first candidate), CDLL it, declare PogsD as graph.py:167-198 does, and make the call
# of graph.py:352-381 -- nothing of this repository is imported.
import ctypes, json, os, sys
import numpy as np
pkg_dir = sys.argv[1]
path = os.path.join(pkg_dir, "libpogs_cpu.so")
assert os.path.exists(path)
lib = ctypes.CDLL(path)
D = ctypes.POINTER(ctypes.c_double); I = ctypes.POINTER(ctypes.c_int)
lib.PogsD.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, D] + [D] * 5 + [I] + [D] * 5 + [I] + [
    ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.c_int,
    D, D, D, D, ctypes.POINTER(ctypes.c_uint)]
lib.PogsD.restype = ctypes.c_int
np.random.seed(0)                                   # README.md:55-59 (C1)
A = np.ascontiguousarray(np.random.randn(500, 300)); b = np.random.randn(500); lam = 0.1
m, n = A.shape
def arr(v, k): return np.full(k, v, np.float64)
f = [arr(1, m), b.copy(), arr(1, m), arr(0, m), arr(0, m)]; fh = np.full(m, 14, np.int32)   # kSquare (graph.py:428)
g = [arr(1, n), arr(0, n), arr(lam, n), arr(0, n), arr(0, n)]; gh = np.full(n, 0, np.int32)  # kAbs    (graph.py:431)
x = np.zeros(n); y = np.zeros(m); l = np.zeros(m); ov = ctypes.c_double(); it = ctypes.c_uint()
p = lambda a: a.ctypes.data_as(D)
st = lib.PogsD(1, m, n, p(A), *[p(a) for a in f], fh.ctypes.data_as(I), *[p(a) for a in g], gh.ctypes.data_as(I),
               1.0, 1e-4, 1e-4, 2500, 0, 1, 1, p(x), p(y), p(l), ctypes.byref(ov), ctypes.byref(it))
print(json.dumps(dict(status=st, iterations=it.value, optval=ov.value, x=x.tolist())))
"""


def test_the_reference_loader_restated_picks_up_the_alias_and_solves_c1(tmp_path):
    """A `pogs/` package directory that holds nothing but the alias `libpogs_cpu.so`: the reference's
    loader logic (restated, no import of this repository) finds it, binds PogsD with the reference's
    own argtypes and gets configs[0]'s golden answer -- status 0, 100 iterations, optval
    91.76711931681265 (tests/golden/reference_outputs.npz)."""
    import json
    import os
    import subprocess
    import sys

    ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    pkg = tmp_path / "pogs"
    pkg.mkdir()
    os.symlink(os.path.join(ROOT, "pogs_amd", "libpogs_cpu.so"), pkg / "libpogs_cpu.so")
    script = tmp_path / "caller.py"
    script.write_text(_REF_LOADER_SCRIPT)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, str(script), str(pkg)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["status"] == 0 and out["iterations"] == 100      # final_iter as the reference returns it
    assert out["optval"] == pytest.approx(91.76711931681265, rel=1e-9)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "reference_outputs.npz"))
    assert out["iterations"] == int(gold["c1_f64_iterations"]) and out["status"] == int(gold["c1_f64_status"])
    assert relerr(np.array(out["x"]), gold["c1_f64_x"]) < 1e-9


@pytest.mark.parametrize("kind", ["dense", "sparse"])
def test_broadcast_coefficients_give_the_array_paths_bits(kind):
    """`Solver.solve` hands fields that are still scalars to `PogsAmdSolveFn` as broadcast values (filled on the device;
    include/pogs_amd.h) instead of building, converting and uploading an array each: the solve must come out bit for bit
    as with all twelve arrays spelled out (`PogsAmdSolve`), for a lasso (one per-element field) and for a vector whose
    every field is per-element; a negative broadcast `c` is clamped with the reference's warning like an array's."""
    import ctypes

    import pogs_amd
    from pogs_amd import _lib, synth
    from pogs_amd import graph as G

    if kind == "dense":
        A, b, _ = synth.dense_lasso(3000, 400, seed=61, dtype=np.float32)
    else:
        A, b, _ = synth.csr_lasso(6000, 1500, 20, seed=61, dtype=np.float32)
    m, n = A.shape
    f, g = G.lasso_functions(b, 0.1, n)
    assert not isinstance(f._v["c"], np.ndarray) and isinstance(f._v["b"], np.ndarray)     # scalars stayed scalars
    with pogs_amd.Solver(A, dtype=np.float32) as s:
        got = s.solve(f, g)                                                              # PogsAmdSolveFn, 1 array of 12
        fa, ga = f.arrays(np.float32), g.arrays(np.float32)
        x, y, l, mu = np.zeros(n, np.float32), np.zeros(m, np.float32), np.zeros(m, np.float32), np.zeros(n, np.float32)
        optval, it = ctypes.c_double(), ctypes.c_uint()
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        st = _lib.lib.PogsAmdSolve(s._h, *[ptr(fa[k]) for k in "abcdeh"], *[ptr(ga[k]) for k in "abcdeh"], 1.0, 1e-4, 1e-4,
                                   2500, 0, 1, 1, ptr(x), ptr(y), ptr(l), ptr(mu), ctypes.byref(optval), ctypes.byref(it))
        assert st == got["status"] == 0 and it.value == got["iterations"]
        assert np.array_equal(x, got["x"]) and np.array_equal(y, got["y"]) and np.array_equal(l, got["l"])
        assert optval.value == got["optval"]
        # every field per element (the scalars read as arrays once): the same bits again
        f2, g2 = G.lasso_functions(b, 0.1, n)
        for fv in (f2, g2):
            for k in "habcde":
                getattr(fv, k)
            assert all(isinstance(fv._v[k], np.ndarray) for k in "habcde")
        again = s.solve(f2, g2)
        assert again["iterations"] == got["iterations"] and np.array_equal(again["x"], got["x"])
        # a negative broadcast c: clamped to 0 (prox_lib.h:62-69), i.e. g = 0 -> plain least squares, like the array form
        g3 = G.FunctionVector(n, G.Function.kAbs, 1.0, 0.0, -0.5)
        g4 = G.FunctionVector(n, G.Function.kAbs, 1.0, 0.0, -0.5)
        g4.c
        r3, r4 = s.solve(f, g3), s.solve(f, g4)
        assert r3["status"] == r4["status"] and np.array_equal(r3["x"], r4["x"])


def test_a_test_transport_id_is_refused_without_the_plug_in():
    """The in-process / shared-memory communicators are NOT in libpogs_amd.so (round 6): a unique id that starts with
    `POGS` is served by the shared object POGS_AMD_TRANSPORT_PLUGIN names and refused -- with a message that says so --
    when the variable is not set.  (A fresh process: the plug-in table is loaded once per process.)"""
    code = r"""
import os, sys
sys.path.insert(0, %r)
os.environ.pop("POGS_AMD_TRANSPORT_PLUGIN", None)
import numpy as np
import pogs_amd
A = np.random.default_rng(0).standard_normal((64, 8)).astype(np.float32)
uid = b"POGSLOCAL:refused".ljust(128, b"\0")
try:
    pogs_amd.Solver(A, dtype=np.float32, dist=(0, 1, 64, uid))
    print("CREATED")
except Exception as e:
    print("REFUSED", str(e))
""" % ROOT
    env = dict(os.environ)
    env.pop("POGS_AMD_TRANSPORT_PLUGIN", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    out = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:]
    assert out.startswith("REFUSED") and "POGS_AMD_TRANSPORT_PLUGIN" in out, (out, r.stderr[-800:])
