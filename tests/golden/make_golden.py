"""Generates the golden fixtures in tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference and oracle/_ref built by
`make -C oracle ref`).  It must run in a clean process: numpy + ctypes only, no
torch (see oracle_binding.ref_solve for why).

    python tests/golden/make_golden.py

What is stored is data only: seeds/recipes are re-run by the tests through
pogs_amd/synth.py, the fixtures hold the reference's outputs (x, y, l, optval,
final_iter, status), the 12 coefficient arrays the reference's own Python layer
(python/pogs/graph.py) hands to PogsD for each solve_* call, and a table of
ProxEval / FuncEval values produced by a small driver (prox_driver.cpp, ours)
compiled against the reference's src/include/prox_lib.h.
"""
import ctypes
import importlib.util
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libpogs_cpu.so")
REF_PY = "/root/reference/python"
assert "torch" not in sys.modules


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


synth = _load(os.path.join(ROOT, "pogs_amd", "synth.py"), "synth")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_runner  # noqa: E402,F401  (only for REF_SO consistency)


# ---------------------------------------------------------------- raw C ABI of the reference
lib = ctypes.CDLL(REF_SO)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def ref_call(A, f, g, dtype, **kw):
    """f, g: dicts h,a,b,c,d,e.  A dense ndarray or scipy CSR."""
    c = ctypes.c_double if dtype == np.float64 else ctypes.c_float
    rho, abs_tol, rel_tol = kw.get("rho", 1.0), kw.get("abs_tol", 1e-4), kw.get("rel_tol", 1e-4)
    max_iter, adaptive_rho, gap_stop = kw.get("max_iter", 2500), kw.get("adaptive_rho", 1), kw.get("gap_stop", 1)
    co = {}
    for pre, src in (("f", f), ("g", g)):
        for k in "abcde":
            co[pre + k] = np.ascontiguousarray(src[k], dtype=dtype)
        co[pre + "h"] = np.ascontiguousarray(src["h"], dtype=np.int32)
    sparse = hasattr(A, "indptr")
    if sparse:
        m, n = A.shape
        data = np.ascontiguousarray(A.data, dtype=dtype)
        ptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
        ind = np.ascontiguousarray(A.indices, dtype=np.int32)
        head = [ctypes.c_int(1), ctypes.c_size_t(m), ctypes.c_size_t(n), ctypes.c_size_t(len(data)), _p(data),
                _p(ptr), _p(ind)]
        fn = lib.PogsSparseD if dtype == np.float64 else lib.PogsSparseS
    else:
        A = np.ascontiguousarray(A, dtype=dtype)
        m, n = A.shape
        head = [ctypes.c_int(1), ctypes.c_size_t(m), ctypes.c_size_t(n), _p(A)]
        fn = lib.PogsD if dtype == np.float64 else lib.PogsS
    x, y, l = np.zeros(n, dtype), np.zeros(m, dtype), np.zeros(m, dtype)
    optval, fi = c(), ctypes.c_uint()
    st = fn(*head, *[_p(co[k]) for k in ("fa", "fb", "fc", "fd", "fe", "fh", "ga", "gb", "gc", "gd", "ge", "gh")],
            c(rho), c(abs_tol), c(rel_tol), ctypes.c_uint(max_iter), ctypes.c_uint(0), ctypes.c_int(adaptive_rho),
            ctypes.c_int(gap_stop), _p(x), _p(y), _p(l), ctypes.byref(optval), ctypes.byref(fi))
    return {"x": x, "y": y, "l": l, "optval": np.float64(optval.value), "iterations": np.int64(fi.value),
            "status": np.int64(st)}


def fv(n, h, a=1.0, b=0.0, c=1.0, d=0.0, e=0.0):
    bc = lambda v: np.broadcast_to(np.asarray(v, np.float64), (n,)).copy()  # noqa: E731
    return {"h": np.full(n, int(h), np.int32), "a": bc(a), "b": bc(b), "c": bc(c), "d": bc(d), "e": bc(e)}


K = dict(kAbs=0, kHuber=2, kIndGe0=6, kLogistic=8, kMaxPos0=10, kSquare=14, kZero=15)


def problems(b, n):
    """The same eight encodings as tests/helpers.PROBLEMS (reference: graph.py:428-705)."""
    m = len(b)
    sgn = np.sign(b) + (b == 0)
    return {
        "lasso": (fv(m, K["kSquare"], 1, b, 1), fv(n, K["kAbs"], 1, 0, 0.1)),
        "ridge": (fv(m, K["kSquare"], 1, b, 1), fv(n, K["kSquare"], 1, 0, 0.5)),
        "elastic_net": (fv(m, K["kSquare"], 1, b, 1), fv(n, K["kAbs"], 1, 0, 0.1, 0, 0.2 / 2)),
        "logistic": (fv(m, K["kLogistic"], -sgn, 0, 1), fv(n, K["kAbs"], 1, 0, 0.01)),
        "logistic0": (fv(m, K["kLogistic"], -sgn, 0, 1), fv(n, K["kZero"])),
        "huber": (fv(m, K["kHuber"], 1.0, b, 1.0), fv(n, K["kAbs"], 1, 0, 0.05)),
        "svm": (fv(m, K["kMaxPos0"], -sgn, -1.0, 1), fv(n, K["kSquare"], 1, 0, 1.0)),
        "nonneg_ls": (fv(m, K["kSquare"], 1, b, 1), fv(n, K["kIndGe0"])),
    }


def pack(prefix, r, out):
    for k, v in r.items():
        out[prefix + k] = v


def main():
    out = {}
    # C1: README recipe
    A, b, lam = synth.readme_lasso()
    f, g = fv(500, K["kSquare"], 1, b, 1), fv(300, K["kAbs"], 1, 0, lam)
    pack("c1_f64_", ref_call(A, f, g, np.float64), out)
    pack("c1_f32_", ref_call(A, f, g, np.float32), out)
    pack("c1_maxiter5_", ref_call(A, f, g, np.float64, max_iter=5), out)
    out["c1_A_checksum"] = np.float64(np.sum(A * np.arange(1, A.size + 1).reshape(A.shape) % 7))
    # the eight problem families on 200 x 100
    rng = np.random.default_rng(7)
    m, n = 200, 100
    A = rng.standard_normal((m, n))
    b = A @ (rng.standard_normal(n) * (rng.random(n) < 0.2)) + 0.1 * rng.standard_normal(m)
    for name, (f, g) in problems(b, n).items():
        for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
            pack("fam_%s_%s_" % (name, tag), ref_call(A, f, g, dt), out)
    # dense fp32 2000 x 300 lasso
    A, b, _ = synth.dense_lasso(2000, 300, seed=11, dtype=np.float32)
    pack("lasso2000_f32_", ref_call(A, fv(2000, K["kSquare"], 1, b, 1), fv(300, K["kAbs"], 1, 0, 0.1), np.float32), out)
    # dense fp32 4000 x 200 logistic (lambda 0.01 and 0)
    A, y, _ = synth.dense_logistic(4000, 200, seed=5, dtype=np.float32)
    pack("logit4000_f32_", ref_call(A, fv(4000, K["kLogistic"], -y, 0, 1), fv(200, K["kAbs"], 1, 0, 0.01), np.float32), out)
    pack("logit4000_l0_f32_", ref_call(A, fv(4000, K["kLogistic"], -y, 0, 1), fv(200, K["kZero"]), np.float32), out)
    # CSR 20000 x 5000, ~50 nnz/row
    A, b, _ = synth.csr_lasso(20000, 5000, 50, seed=3, dtype=np.float32)
    pack("csr20000_f32_", ref_call(A, fv(20000, K["kSquare"], 1, b, 1), fv(5000, K["kAbs"], 1, 0, 0.1), np.float32), out)
    A, b, _ = synth.csr_lasso(3000, 800, 20, seed=4, dtype=np.float64)
    pack("csr3000_f64_", ref_call(A, fv(3000, K["kSquare"], 1, b, 1), fv(800, K["kAbs"], 1, 0, 0.1), np.float64), out)
    # wide dense (m <= n) fp64
    A, b, _ = synth.dense_lasso(120, 300, seed=6)
    pack("wide120_f64_", ref_call(A, fv(120, K["kSquare"], 1, b, 1), fv(300, K["kAbs"], 1, 0, 0.1), np.float64), out)
    np.savez_compressed(os.path.join(HERE, "reference_outputs.npz"), **out)
    print("reference_outputs.npz:", len(out), "arrays")

    python_layer()
    prox_table()


def python_layer():
    """Coefficient arrays + return dict of the reference's own python/pogs solve_* functions."""
    real_exists, real_cdll = os.path.exists, ctypes.CDLL

    def fake_exists(p):
        return True if str(p).endswith(os.path.join("pogs", "libpogs_cpu.so")) else real_exists(p)

    def fake_cdll(p, *a, **k):
        return real_cdll(REF_SO if str(p).endswith("libpogs_cpu.so") else p, *a, **k)

    os.path.exists, ctypes.CDLL = fake_exists, fake_cdll
    try:
        sys.path.insert(0, REF_PY)
        import pogs.graph as rg  # the reference's module, imported from where it lies
    finally:
        os.path.exists, ctypes.CDLL = real_exists, real_cdll
        sys.path.remove(REF_PY)

    captured = {}

    class Proxy:
        def __init__(self, real):
            self._real = real

        def __getattr__(self, name):
            return getattr(self._real, name)

        def PogsD(self, *args):
            m, n = args[1], args[2]
            names = ["f_a", "f_b", "f_c", "f_d", "f_e", "f_h", "g_a", "g_b", "g_c", "g_d", "g_e", "g_h"]
            for i, nm in enumerate(names):
                cnt = m if nm.startswith("f") else n
                ptr = args[4 + i]
                typ = ctypes.c_int if nm.endswith("_h") else ctypes.c_double
                arr = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(typ)), shape=(cnt,)).copy()
                captured[nm] = arr
            captured["scalars"] = np.array([float(args[16]), float(args[17]), float(args[18]), float(args[19]),
                                            float(args[20]), float(args[21]), float(args[22])])
            return self._real.PogsD(*args)

    rg._lib = Proxy(rg._lib)
    rng = np.random.default_rng(21)
    m, n = 40, 15
    A = rng.standard_normal((m, n))
    b = rng.standard_normal(m)
    lab = np.sign(b)
    calls = {
        "lasso": lambda: rg.solve_lasso(A, b, 0.1),
        "ridge": lambda: rg.solve_ridge(A, b, 0.5),
        "elastic_net": lambda: rg.solve_elastic_net(A, b, 0.1, 0.2),
        "logistic": lambda: rg.solve_logistic(A, lab, 0.01),
        "logistic0": lambda: rg.solve_logistic(A, lab),
        "huber": lambda: rg.solve_huber(A, b, 1.5, 0.05),
        "svm": lambda: rg.solve_svm(A, lab, 1.0),
        "nonneg_ls": lambda: rg.solve_nonneg_ls(A, b),
    }
    out = {"A": A, "b": b}
    for name, fn in calls.items():
        captured.clear()
        r = fn()
        for k, v in captured.items():
            out["%s_%s" % (name, k)] = v
        for k in ("x", "y", "l"):
            out["%s_ret_%s" % (name, k)] = r[k]
        out["%s_ret_optval" % name] = np.float64(r["optval"])
        out["%s_ret_iterations" % name] = np.int64(r["iterations"])
        out["%s_ret_status" % name] = np.int64(r["status"])
    np.savez_compressed(os.path.join(HERE, "python_layer.npz"), **out)
    print("python_layer.npz:", len(out), "arrays")


def prox_table():
    """ProxEval / FuncEval of the reference header on a grid, via prox_driver.cpp."""
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "prox_driver")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I/root/reference/src/include",
                               os.path.join(HERE, "prox_driver.cpp"), "-o", exe])
        raw = subprocess.check_output([exe])
    arr = np.frombuffer(raw, dtype=np.float64)
    # layout written by the driver: rows of [is_float, h, a, b, c, d, e, rho, v, prox, func]
    tab = arr.reshape(-1, 11)
    np.savez_compressed(os.path.join(HERE, "prox_table.npz"), table=tab)
    print("prox_table.npz:", tab.shape)


def projsub_table():
    """ProjSubgradEval of the reference header on a grid, via projsub_driver.cpp."""
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "projsub_driver")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I/root/reference/src/include",
                               os.path.join(HERE, "projsub_driver.cpp"), "-o", exe])
        raw = subprocess.check_output([exe])
    # rows of [is_float, h, a, b, c, d, e, x, v, result]
    tab = np.frombuffer(raw, dtype=np.float64).reshape(-1, 10)
    np.savez_compressed(os.path.join(HERE, "projsub_table.npz"), table=tab)
    print("projsub_table.npz:", tab.shape)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "projsub":
        projsub_table()   # only this fixture (the others are unchanged)
    else:
        main()
        projsub_table()
