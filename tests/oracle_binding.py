"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when it has been
built in the build container, the compiled reference (oracle/_ref/libpogs_cpu.so).

TEST INFRASTRUCTURE: importable from tests/, bench.py (cpu_baseline leg) and
__graft_entry__.smoke() only.  Nothing under pogs_amd/ may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_ORACLE_SO = os.path.join(_ORACLE_DIR, "liboracle.so")
_REF_SO = os.path.join(_ORACLE_DIR, "_ref", "libpogs_cpu.so")
# the same six reference sources against scipy's OpenBLAS (oracle/Makefile: ref_openblas): TIMING ONLY,
# bench.py's "best configuration" CPU column -- parity stays pinned to the MKL build above
_REF_SO_OPENBLAS = os.path.join(_ORACLE_DIR, "_ref", "libpogs_cpu_openblas.so")


class OracleInfo(ctypes.Structure):
    _fields_ = [
        ("use_cgls", ctypes.c_int),
        ("d_out", ctypes.POINTER(ctypes.c_double)),
        ("e_out", ctypes.POINTER(ctypes.c_double)),
        ("warm_x", ctypes.c_void_p),
        ("warm_l", ctypes.c_void_p),
        ("nrmA", ctypes.c_double),
        ("norm_est_iters", ctypes.c_uint),
        ("rho_final", ctypes.c_double),
        ("exact_iters", ctypes.c_uint),
        ("cg_iters", ctypes.c_long),
        ("n_mul", ctypes.c_long),
        ("t_init", ctypes.c_double),
        ("t_loop", ctypes.c_double),
    ]


ALLREDUCE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_size_t)


def build_oracle(force=False):
    src = os.path.join(_ORACLE_DIR, "pogs_oracle.cpp")
    if force or not os.path.exists(_ORACLE_SO) or os.path.getmtime(_ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _ORACLE_SO


def build_ref():
    """Build the real reference if its sources are present (build container only)."""
    if os.path.isdir("/root/reference/src") and not os.path.exists(_REF_SO_OPENBLAS):
        subprocess.call(["make", "-C", _ORACLE_DIR, "ref_openblas"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if os.path.exists(_REF_SO):
        return _REF_SO
    if not os.path.isdir("/root/reference/src"):
        return None
    r = subprocess.call(["make", "-C", _ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return _REF_SO if r == 0 and os.path.exists(_REF_SO) else None


_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        _oracle = ctypes.CDLL(build_oracle())
        _oracle.OracleFuncEvalD.restype = ctypes.c_double
        _oracle.OracleFuncEvalS.restype = ctypes.c_double
        _oracle.OracleProxRawD.restype = ctypes.c_double
        _oracle.OracleProxRawD.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double]
        _oracle.OracleProxRawS.restype = ctypes.c_float
        _oracle.OracleProxRawS.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_float]
    return _oracle


def _ct(dtype):
    return ctypes.c_double if dtype == np.float64 else ctypes.c_float


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _coef_arrays(objs, dtype):
    """objs = dict(h=int array, a,b,c,d,e arrays) -> contiguous typed arrays."""
    out = {k: np.ascontiguousarray(objs[k], dtype=dtype) for k in "abcde"}
    out["h"] = np.ascontiguousarray(objs["h"], dtype=np.int32)
    return out


def _result(x, y, l, optval, final_iter, status, info=None):
    r = {"x": x, "y": y, "l": l, "optval": optval.value, "iterations": final_iter.value, "status": status}
    if info is not None:
        r["info"] = {k: getattr(info, k) for k, _ in OracleInfo._fields_ if k not in ("d_out", "e_out")}
    return r


def _solve_dense(fn, A, f, g, dtype, rho, abs_tol, rel_tol, max_iter, verbose, adaptive_rho, gap_stop, order,
                 info=None):
    c = _ct(dtype)
    A = np.ascontiguousarray(A, dtype=dtype) if order == 1 else np.asfortranarray(A, dtype=dtype)
    m, n = A.shape
    f = _coef_arrays(f, dtype)
    g = _coef_arrays(g, dtype)
    x = np.zeros(n, dtype)
    y = np.zeros(m, dtype)
    l = np.zeros(m, dtype)
    optval = c()
    final_iter = ctypes.c_uint()
    args = [ctypes.c_int(order), ctypes.c_size_t(m), ctypes.c_size_t(n), _p(A),
            _p(f["a"]), _p(f["b"]), _p(f["c"]), _p(f["d"]), _p(f["e"]), _p(f["h"]),
            _p(g["a"]), _p(g["b"]), _p(g["c"]), _p(g["d"]), _p(g["e"]), _p(g["h"]),
            c(rho), c(abs_tol), c(rel_tol), ctypes.c_uint(max_iter), ctypes.c_uint(verbose),
            ctypes.c_int(int(adaptive_rho)), ctypes.c_int(int(gap_stop)),
            _p(x), _p(y), _p(l), ctypes.byref(optval), ctypes.byref(final_iter)]
    if info is not None:
        args.append(ctypes.byref(info))
    status = fn(*args)
    return _result(x, y, l, optval, final_iter, status, info)


def _solve_sparse(fn, A_csr, f, g, dtype, rho, abs_tol, rel_tol, max_iter, verbose, adaptive_rho, gap_stop,
                  info=None):
    c = _ct(dtype)
    m, n = A_csr.shape
    data = np.ascontiguousarray(A_csr.data, dtype=dtype)
    ptr = np.ascontiguousarray(A_csr.indptr, dtype=np.int32)
    ind = np.ascontiguousarray(A_csr.indices, dtype=np.int32)
    f = _coef_arrays(f, dtype)
    g = _coef_arrays(g, dtype)
    x = np.zeros(n, dtype)
    y = np.zeros(m, dtype)
    l = np.zeros(m, dtype)
    optval = c()
    final_iter = ctypes.c_uint()
    args = [ctypes.c_int(1), ctypes.c_size_t(m), ctypes.c_size_t(n), ctypes.c_size_t(len(data)),
            _p(data), _p(ptr), _p(ind),
            _p(f["a"]), _p(f["b"]), _p(f["c"]), _p(f["d"]), _p(f["e"]), _p(f["h"]),
            _p(g["a"]), _p(g["b"]), _p(g["c"]), _p(g["d"]), _p(g["e"]), _p(g["h"]),
            c(rho), c(abs_tol), c(rel_tol), ctypes.c_uint(max_iter), ctypes.c_uint(verbose),
            ctypes.c_int(int(adaptive_rho)), ctypes.c_int(int(gap_stop)),
            _p(x), _p(y), _p(l), ctypes.byref(optval), ctypes.byref(final_iter)]
    if info is not None:
        args.append(ctypes.byref(info))
    status = fn(*args)
    return _result(x, y, l, optval, final_iter, status, info)


def oracle_solve(A, f, g, dtype=np.float64, rho=1.0, abs_tol=1e-4, rel_tol=1e-4, max_iter=2500, verbose=0,
                 adaptive_rho=True, gap_stop=True, order=1, use_cgls=False, want_de=False, x0=None, l0=None):
    """Run the oracle through its PogsD/S-shaped entry.  A dense ndarray or scipy CSR.
    (x0, l0): warm start, the reference's SetInitX/SetInitLambda (pogs.cpp:144-156)."""
    lib = oracle_lib()
    info = OracleInfo()
    info.use_cgls = int(use_cgls)
    if x0 is not None:
        warm = (np.ascontiguousarray(x0, dtype=dtype), np.ascontiguousarray(l0, dtype=dtype))
        info.warm_x, info.warm_l = warm[0].ctypes.data, warm[1].ctypes.data
    sparse = hasattr(A, "indptr")
    m, n = A.shape
    keep = None
    if want_de:
        keep = (np.zeros(m), np.zeros(n))
        info.d_out = keep[0].ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        info.e_out = keep[1].ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    if sparse:
        fn = lib.OraclePogsSparseD if dtype == np.float64 else lib.OraclePogsSparseS
        r = _solve_sparse(fn, A, f, g, dtype, rho, abs_tol, rel_tol, max_iter, verbose, adaptive_rho, gap_stop, info)
    else:
        fn = lib.OraclePogsD if dtype == np.float64 else lib.OraclePogsS
        r = _solve_dense(fn, A, f, g, dtype, rho, abs_tol, rel_tol, max_iter, verbose, adaptive_rho, gap_stop, order,
                         info)
    if keep is not None:
        r["d"], r["e"] = keep
    return r


def oracle_set_threads(n=None):
    """OpenMP threads of the oracle port (its loops are `#pragma omp parallel for`): the container's
    CPU quota by default -- the GPU box shows 256 hardware threads but grants 16 cores, and 256
    OpenMP threads on 16 cores are several times slower than 16."""
    n = int(n or cpu_quota())
    os.environ["OMP_NUM_THREADS"] = str(n)
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(n)
    except OSError:
        pass
    return n


def ref_available(blas="mkl"):
    return os.path.exists(_REF_SO_OPENBLAS if blas == "openblas" else _REF_SO)


class RefRun:
    """A reference solve running in its own process (ref_start); finish() waits for it."""

    def __init__(self, td, proc, outp, t0):
        self.td, self.proc, self.outp, self.t0 = td, proc, outp, t0

    def finish(self, timeout=None):
        import re
        import time

        try:
            try:
                out, err = self.proc.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                self.proc.kill()
                self.proc.communicate()
                raise
            if self.proc.returncode != 0 or not os.path.exists(self.outp):
                raise RuntimeError("reference runner failed:\n" + out[-2000:] + err[-2000:])
            z = np.load(self.outp)
            r = {"x": z["x"], "y": z["y"], "l": z["l"], "optval": float(z["optval"]),
                 "iterations": int(z["iterations"]), "status": int(z["status"]), "wall_s": float(z["wall_s"]),
                 "stdout": out, "elapsed_s": time.time() - self.t0}
            mt = re.search(r"Total = ([0-9.eE+-]+) s, Init = ([0-9.eE+-]+) s", out)
            if mt:
                r["t_total"], r["t_init"] = float(mt.group(1)), float(mt.group(2))
            return r
        finally:
            self.td.cleanup()


REF_THREADS = 16   # BLAS threads of the reference subprocess: the fastest setting measured on the GPU box
                   # (scripts/ref_threads_probe.py, 20000 x 10000 fp32: 16 -> 12.8 it/s, 32 -> 9.2, 64 -> 11.1,
                   # 128 -> 7.2, 256 -> 8.8) -- its container has a CPU quota of 16 cores (cgroup cpu.max) although
                   # 256 hardware threads are visible
# Environment of the reference subprocess.  MKL_DYNAMIC=FALSE: use the threads it is given.  Passive
# waiting (KMP_BLOCKTIME=0, OMP_WAIT_POLICY=PASSIVE): MKL's idle OpenMP workers otherwise spin for
# 200 ms after every parallel region, which under the container's CPU quota competes with the one
# thread that runs the reference's gemv (profiles/r03_ref_cpu_diagnosis.md: on the GPU box's AMD
# host MKL runs cblas_sgemv on ONE thread whatever the settings; 70000 x 10000: 77 s passive, 88 s
# spinning).
REF_ENV = {"MKL_DYNAMIC": "FALSE", "KMP_BLOCKTIME": "0", "OMP_WAIT_POLICY": "PASSIVE"}


def cpu_quota():
    """CPUs this container may use: the cgroup quota if there is one, else the visible count."""
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(round(int(q) / int(per)))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(round(q / per))))
        except Exception:
            pass
    return n


def ref_threads():
    return max(1, min(REF_THREADS, cpu_quota()))


def ref_env_note():
    return "MKL_NUM_THREADS=OMP_NUM_THREADS=%d %s" % (ref_threads(), " ".join("%s=%s" % kv for kv in sorted(REF_ENV.items())))


def ref_start(A, f, g, dtype=np.float64, rho=1.0, abs_tol=1e-4, rel_tol=1e-4, max_iter=2500, verbose=0,
              adaptive_rho=True, gap_stop=True, order=1, threads=None, blas="mkl"):
    """Start the compiled reference (PogsD/S, PogsSparseD/S) in a CLEAN subprocess and return a
    RefRun (None if the reference is not available); RefRun.finish() collects the result.

    The reference links MKL; in a process that has PyTorch loaded (its own OpenMP /
    BLAS symbols) MKL's threading layer mis-binds and the reference returns NaN, so
    it never shares a process with torch.  The result carries 't_total' / 't_init' parsed
    from the reference's own verbose=1 summary (src/cpu/pogs.cpp:485-490) when verbose >= 1.
    """
    import sys
    import tempfile
    import time

    if not ref_available(blas):
        return None
    sparse = hasattr(A, "indptr")
    td = tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        inp, outp = os.path.join(td.name, "in.npz"), os.path.join(td.name, "out.npz")
        payload = {"dtype": np.dtype(dtype).name, "params": np.array([rho, abs_tol, rel_tol, max_iter, verbose,
                                                                      int(adaptive_rho), int(gap_stop), order],
                                                                     dtype=np.float64)}
        for k in "habcde":
            payload["f_" + k] = np.asarray(f[k])
            payload["g_" + k] = np.asarray(g[k])
        if sparse:
            payload.update(sp_data=np.asarray(A.data, dtype=dtype), sp_ptr=np.asarray(A.indptr, np.int32),
                           sp_ind=np.asarray(A.indices, np.int32), sp_shape=np.array(A.shape))
        else:
            np.save(os.path.join(td.name, "A.npy"), np.asarray(A, dtype=dtype))
        np.savez(inp, **payload)
        env = dict(os.environ)
        env.pop("PYTHONPATH", None)
        nthr = str(threads if threads else ref_threads())
        env.update(MKL_NUM_THREADS=nthr, OMP_NUM_THREADS=nthr, OPENBLAS_NUM_THREADS=nthr)
        env.update(REF_ENV)
        if blas == "openblas":
            env["POGS_REF_SO"] = _REF_SO_OPENBLAS
        else:
            env.pop("POGS_REF_SO", None)
        cmd = [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_runner.py"), td.name]
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    except Exception:
        td.cleanup()
        raise
    return RefRun(td, proc, outp, time.time())


def ref_solve(A, f, g, dtype=np.float64, rho=1.0, abs_tol=1e-4, rel_tol=1e-4, max_iter=2500, verbose=0,
              adaptive_rho=True, gap_stop=True, order=1, timeout=None, threads=None, blas="mkl"):
    """ref_start + finish.  Returns None if the reference is not available; raises
    subprocess.TimeoutExpired on timeout."""
    run = ref_start(A, f, g, dtype=dtype, rho=rho, abs_tol=abs_tol, rel_tol=rel_tol, max_iter=max_iter,
                    verbose=verbose, adaptive_rho=adaptive_rho, gap_stop=gap_stop, order=order, threads=threads,
                    blas=blas)
    return None if run is None else run.finish(timeout)


def oracle_solve_shard(A_local, m_global, f_local, g, allreduce, dtype=np.float64, rho=1.0, abs_tol=1e-4,
                       rel_tol=1e-4, max_iter=2500, verbose=0, adaptive_rho=True, gap_stop=True):
    """Row-sharded oracle: `allreduce(np.ndarray float64)` sums in place across ranks.
    A_local: dense ndarray (direct projector) or scipy CSR (CGLS projector) holding this rank's rows."""
    lib = oracle_lib()
    c = _ct(dtype)
    sparse = hasattr(A_local, "indptr")
    m, n = A_local.shape
    f = _coef_arrays(f_local, dtype)
    gg = _coef_arrays(g, dtype)
    x = np.zeros(n, dtype)
    y = np.zeros(m, dtype)
    l = np.zeros(m, dtype)
    optval = c()
    final_iter = ctypes.c_uint()
    info = OracleInfo()

    def _cb(_ctx, buf, count):
        arr = np.ctypeslib.as_array(buf, shape=(count,))
        allreduce(arr)

    cb = ALLREDUCE_FN(_cb)
    if sparse:
        data = np.ascontiguousarray(A_local.data, dtype=dtype)
        ptr = np.ascontiguousarray(A_local.indptr, dtype=np.int32)
        ind = np.ascontiguousarray(A_local.indices, dtype=np.int32)
        head = [ctypes.c_size_t(m), ctypes.c_size_t(m_global), ctypes.c_size_t(n), ctypes.c_size_t(len(data)),
                _p(data), _p(ptr), _p(ind)]
        fn = lib.OraclePogsSparseShardD if dtype == np.float64 else lib.OraclePogsSparseShardS
    else:
        A = np.ascontiguousarray(A_local, dtype=dtype)
        head = [ctypes.c_size_t(m), ctypes.c_size_t(m_global), ctypes.c_size_t(n), _p(A)]
        fn = lib.OraclePogsShardD if dtype == np.float64 else lib.OraclePogsShardS
    status = fn(*head,
                _p(f["a"]), _p(f["b"]), _p(f["c"]), _p(f["d"]), _p(f["e"]), _p(f["h"]),
                _p(gg["a"]), _p(gg["b"]), _p(gg["c"]), _p(gg["d"]), _p(gg["e"]), _p(gg["h"]),
                c(rho), c(abs_tol), c(rel_tol), ctypes.c_uint(max_iter), ctypes.c_uint(verbose),
                ctypes.c_int(int(adaptive_rho)), ctypes.c_int(int(gap_stop)),
                _p(x), _p(y), _p(l), ctypes.byref(optval), ctypes.byref(final_iter), ctypes.byref(info),
                cb, None)
    return _result(x, y, l, optval, final_iter, status, info)


def oracle_prox(objs, rho, v, dtype=np.float64):
    lib = oracle_lib()
    c = _ct(dtype)
    o = _coef_arrays(objs, dtype)
    v = np.ascontiguousarray(v, dtype=dtype)
    out = np.zeros_like(v)
    fn = lib.OracleProxEvalD if dtype == np.float64 else lib.OracleProxEvalS
    fn(ctypes.c_size_t(len(v)), _p(o["h"]), _p(o["a"]), _p(o["b"]), _p(o["c"]), _p(o["d"]), _p(o["e"]), c(rho),
       _p(v), _p(out))
    return out


def oracle_func(objs, v, dtype=np.float64):
    lib = oracle_lib()
    o = _coef_arrays(objs, dtype)
    v = np.ascontiguousarray(v, dtype=dtype)
    fn = lib.OracleFuncEvalD if dtype == np.float64 else lib.OracleFuncEvalS
    return fn(ctypes.c_size_t(len(v)), _p(o["h"]), _p(o["a"]), _p(o["b"]), _p(o["c"]), _p(o["d"]), _p(o["e"]), _p(v))


def oracle_proj_subgrad(objs, x, v, dtype=np.float64):
    """ProjSubgradEval of the oracle (prox_lib.h:468-493): v projected onto the subdifferential at x."""
    lib = oracle_lib()
    o = _coef_arrays(objs, dtype)
    x = np.ascontiguousarray(x, dtype=dtype)
    v = np.ascontiguousarray(v, dtype=dtype)
    out = np.zeros_like(v)
    fn = lib.OracleProjSubgradEvalD if dtype == np.float64 else lib.OracleProjSubgradEvalS
    fn(ctypes.c_size_t(len(v)), _p(o["h"]), _p(o["a"]), _p(o["b"]), _p(o["c"]), _p(o["d"]), _p(o["e"]), _p(x), _p(v),
       _p(out))
    return out


def oracle_prox_raw(h, v, rho, dtype=np.float64):
    lib = oracle_lib()
    if dtype == np.float64:
        return lib.OracleProxRawD(int(h), float(v), float(rho))
    return lib.OracleProxRawS(int(h), float(v), float(rho))


def oracle_rand(n, dtype=np.float64):
    lib = oracle_lib()
    x = np.zeros(n, dtype)
    (lib.OracleRandD if dtype == np.float64 else lib.OracleRandS)(_p(x), ctypes.c_size_t(n))
    return x


def oracle_project(A, x0, y0, s=1.0, tol=1e-8, use_cgls=False, dtype=np.float64):
    lib = oracle_lib()
    c = _ct(dtype)
    A = np.ascontiguousarray(A, dtype=dtype)
    m, n = A.shape
    x0 = np.ascontiguousarray(x0, dtype=dtype)
    y0 = np.ascontiguousarray(y0, dtype=dtype)
    x = np.zeros(n, dtype)
    y = np.zeros(m, dtype)
    fn = lib.OracleProjectD if dtype == np.float64 else lib.OracleProjectS
    fn(ctypes.c_size_t(m), ctypes.c_size_t(n), _p(A), _p(x0), _p(y0), c(s), c(tol), ctypes.c_int(int(use_cgls)),
       _p(x), _p(y))
    return x, y


def oracle_equil(A, dtype=np.float64):
    lib = oracle_lib()
    c = _ct(dtype)
    A = np.array(A, dtype=dtype, order="C", copy=True)
    m, n = A.shape
    d = np.zeros(m, dtype)
    e = np.zeros(n, dtype)
    nrm = c()
    kpow = ctypes.c_uint()
    fn = lib.OracleEquilD if dtype == np.float64 else lib.OracleEquilS
    fn(ctypes.c_size_t(m), ctypes.c_size_t(n), _p(A), _p(d), _p(e), ctypes.byref(nrm), ctypes.byref(kpow))
    return A, d, e, nrm.value, kpow.value
