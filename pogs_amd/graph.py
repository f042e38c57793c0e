"""Graph-form POGS interface backed by the MI355X HIP engine.

Host-side mirror of the reference's ``python/pogs/graph.py``: the same seven
``solve_*`` entry points with the same signatures, defaults, function encodings
and return dictionary (``x, y, l, optval, iterations, status``), talking to
``libpogs_amd.so`` through the same C ABI the reference exposes
(``PogsD`` / ``PogsSparseD``; reference ``python/pogs/graph.py:167-233``), plus
what the reference's Python layer lacks for a GPU (SURVEY.md finding 6):

* ``dtype=`` to route to the fp32 entry points ``PogsS`` / ``PogsSparseS``
  (the reference wrapper forces float64, ``graph.py:281-288``);
* vectorised coefficient construction (``FunctionVector``) instead of m + n
  Python objects (``graph.py:428,431`` builds them one by one);
* ``Solver``: a persistent handle that keeps the equilibrated matrix and its
  factorisation on the GPU across solves, accepts device-resident matrices and
  row-shards over several GPUs (one process per GPU, RCCL).

Solves problems of the form

    minimize    sum_i f_i(y_i) + sum_j g_j(x_j)
    subject to  y = A x

with ``f_i(v) = c h(a v - b) + d v + e v^2 / 2`` and ``h`` one of `Function`.
"""
import atexit
import ctypes
import weakref
from enum import IntEnum

import numpy as np

from . import _lib
from ._lib import lib

try:
    import scipy.sparse as sp

    HAS_SCIPY = True
except ImportError:  # pragma: no cover
    HAS_SCIPY = False


class Ordering(IntEnum):
    """Matrix ordering (reference: python/pogs/graph.py:107-111)."""

    COL_MAJ = 0
    ROW_MAJ = 1


class Function(IntEnum):
    """Function types for f_i and g_j (reference: python/pogs/graph.py:114-132)."""

    kAbs = 0  # f(x) = |x|
    kExp = 1  # f(x) = e^x
    kHuber = 2  # f(x) = huber(x)
    kIdentity = 3  # f(x) = x
    kIndBox01 = 4  # f(x) = I(0 <= x <= 1)
    kIndEq0 = 5  # f(x) = I(x = 0)
    kIndGe0 = 6  # f(x) = I(x >= 0)
    kIndLe0 = 7  # f(x) = I(x <= 0)
    kLogistic = 8  # f(x) = log(1 + e^x)
    kMaxNeg0 = 9  # f(x) = max(0, -x)
    kMaxPos0 = 10  # f(x) = max(0, x)
    kNegEntr = 11  # f(x) = x log(x)
    kNegLog = 12  # f(x) = -log(x)
    kRecipr = 13  # f(x) = 1/x
    kSquare = 14  # f(x) = (1/2) x^2
    kZero = 15  # f(x) = 0


# Status codes (reference: src/include/pogs.h:31-37).  3, not 1, is "max_iter".
POGS_SUCCESS, POGS_INFEASIBLE, POGS_UNBOUNDED, POGS_MAX_ITER, POGS_NAN_FOUND, POGS_INVALID_CONE, POGS_ERROR = range(7)


class FunctionObj:
    """One function ``c * h(a * x - b) + d * x + e * x^2`` (reference: graph.py:135-164)."""

    def __init__(self, h=Function.kZero, a=1.0, b=0.0, c=1.0, d=0.0, e=0.0):
        self.h = h
        self.a = float(a)
        self.b = float(b)
        self.c = float(c)
        self.d = float(d)
        self.e = float(e)


class FunctionVector:
    """Struct-of-arrays of ``n`` function objects; every field broadcasts from a scalar.

    A field given as a scalar STAYS a scalar until somebody reads it as an array (``fv.c``, ``fv.c = ...``):
    `arrays()` -- what a solve hands to the C ABI -- then fills the typed array directly instead of converting
    a float64 copy.  A lasso on 2e6 rows has one per-element field out of twelve; building and marshalling
    its two vectors costs a few milliseconds this way instead of ~0.1 s (the reference builds m + n Python
    objects, python/pogs/graph.py:428,431)."""

    _FIELDS = ("h", "a", "b", "c", "d", "e")

    def __init__(self, n, h=Function.kZero, a=1.0, b=0.0, c=1.0, d=0.0, e=0.0):
        object.__setattr__(self, "n", int(n))
        object.__setattr__(self, "_v", {})
        object.__setattr__(self, "_typed", {})   # (field, dtype) -> the filled array of a SCALAR field (cannot go stale:
                                                 # a scalar has no elements to edit in place; dropped when the field is set)
        for k, v in zip(self._FIELDS, (h, a, b, c, d, e)):
            self._store(k, v)

    def _store(self, k, v):
        dt = np.int32 if k == "h" else np.float64
        for key in [key for key in self._typed if key[0] == k]:
            del self._typed[key]
        if np.ndim(v) == 0:
            self._v[k] = int(v) if k == "h" else float(v)          # scalar: materialised on demand
        else:
            self._v[k] = np.broadcast_to(np.asarray(v, dtype=dt), (self.n,)).copy()

    def _materialise(self, k):
        v = self._v[k]
        if not isinstance(v, np.ndarray):
            v = np.full(self.n, v, dtype=np.int32 if k == "h" else np.float64)
            self._v[k] = v                      # from here on an array the caller may edit: converted per call
            for key in [key for key in self._typed if key[0] == k]:
                del self._typed[key]
        return v

    def __getattr__(self, k):        # only reached for names that are not instance attributes: the six fields
        if k in FunctionVector._FIELDS:
            return self._materialise(k)
        raise AttributeError(k)

    def __setattr__(self, k, v):
        if k in self._FIELDS:
            self._store(k, v)
        else:
            object.__setattr__(self, k, v)

    def __len__(self):
        return self.n

    @staticmethod
    def from_objs(objs):
        fv = FunctionVector(len(objs))
        fv.h = np.array([int(o.h) for o in objs], dtype=np.int32)
        for k in "abcde":
            setattr(fv, k, np.array([getattr(o, k) for o in objs], dtype=np.float64))
        return fv

    def arrays(self, dtype):
        """The six contiguous arrays the C ABI takes (reference: graph.py:296-308)."""
        out = {}
        for k in self._FIELDS:
            v, dt = self._v[k], (np.int32 if k == "h" else dtype)
            if isinstance(v, np.ndarray):
                out[k] = np.ascontiguousarray(v, dtype=dt)
            else:
                key = (k, np.dtype(dt).str)
                if key not in self._typed:
                    # (zeros come as untouched zero pages: no fill, no page faults until somebody writes)
                    self._typed[key] = np.zeros(self.n, dtype=dt) if v == 0 else np.full(self.n, v, dtype=dt)
                out[k] = self._typed[key]       # read-only to the library (the ABI never writes its inputs)
        return out

    def slice(self, lo, hi):
        fv = FunctionVector(hi - lo)
        for k in self._FIELDS:
            v = self._v[k]
            fv._v[k] = v[lo:hi].copy() if isinstance(v, np.ndarray) else v
        return fv


def _as_vector(f):
    return f if isinstance(f, FunctionVector) else FunctionVector.from_objs(list(f))


def _resolve_dtype(dtype):
    dt = np.dtype(np.float64 if dtype is None else dtype)
    if dt not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise ValueError("dtype must be float32 or float64")
    return dt


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _solve_graph_form(A, f, g, abs_tol=1e-4, rel_tol=1e-4, max_iter=2500, verbose=0, rho=1.0,
                      adaptive_rho=True, gap_stop=True, dtype=None):
    """Solve one graph-form problem through the one-shot C ABI (reference: graph.py:236-390).

    ``f`` / ``g``: lists of `FunctionObj` (as in the reference) or a `FunctionVector`.
    Returns dict with 'x', 'y', 'l', 'optval', 'iterations', 'status'.
    """
    dt = _resolve_dtype(dtype)
    real = ctypes.c_double if dt == np.float64 else ctypes.c_float
    is_sparse = HAS_SCIPY and sp.issparse(A)
    if is_sparse:
        A_csr = sp.csr_matrix(A, dtype=dt)
        m, n = A_csr.shape
        nnz = A_csr.nnz
        data = np.ascontiguousarray(A_csr.data, dtype=dt)
        ptr = np.ascontiguousarray(A_csr.indptr, dtype=np.int32)
        ind = np.ascontiguousarray(A_csr.indices, dtype=np.int32)
    else:
        A = np.asarray(A, dtype=dt, order="C")
        m, n = A.shape

    assert len(f) == m, f"f should have length {m}, got {len(f)}"
    assert len(g) == n, f"g should have length {n}, got {len(g)}"
    fa = _as_vector(f).arrays(dt)
    ga = _as_vector(g).arrays(dt)

    x = np.zeros(n, dtype=dt)
    y = np.zeros(m, dtype=dt)
    dual = np.zeros(m, dtype=dt)
    optval = real()
    final_iter = ctypes.c_uint()
    tail = [_ptr(fa["a"]), _ptr(fa["b"]), _ptr(fa["c"]), _ptr(fa["d"]), _ptr(fa["e"]), _ptr(fa["h"]),
            _ptr(ga["a"]), _ptr(ga["b"]), _ptr(ga["c"]), _ptr(ga["d"]), _ptr(ga["e"]), _ptr(ga["h"]),
            real(rho), real(abs_tol), real(rel_tol), int(max_iter), int(verbose), int(adaptive_rho), int(gap_stop),
            _ptr(x), _ptr(y), _ptr(dual), ctypes.cast(ctypes.byref(optval), ctypes.c_void_p),
            ctypes.cast(ctypes.byref(final_iter), ctypes.c_void_p)]
    if is_sparse:
        fn = lib.PogsSparseD if dt == np.float64 else lib.PogsSparseS
        status = fn(int(Ordering.ROW_MAJ), m, n, nnz, _ptr(data), _ptr(ptr), _ptr(ind), *tail)
    else:
        fn = lib.PogsD if dt == np.float64 else lib.PogsS
        status = fn(int(Ordering.ROW_MAJ), m, n, _ptr(A), *tail)
    return {"x": x, "y": y, "l": dual, "optval": optval.value, "iterations": final_iter.value, "status": status}


# ---------------------------------------------------------------------------
# Function encodings of the seven solve_* problems (reference: graph.py:428-705)
# ---------------------------------------------------------------------------
def lasso_functions(b, lambd, n):
    """f_i = 0.5 (y_i - b_i)^2 ; g_j = lambda |x_j|   (graph.py:428,431)"""
    b = np.asarray(b, dtype=np.float64).ravel()
    return (FunctionVector(len(b), Function.kSquare, 1.0, b, 1.0),
            FunctionVector(n, Function.kAbs, 1.0, 0.0, lambd))


def ridge_functions(b, lambd, n):
    """g_j = 0.5 lambda x_j^2 as kSquare with c = lambda   (graph.py:471,474)"""
    b = np.asarray(b, dtype=np.float64).ravel()
    return (FunctionVector(len(b), Function.kSquare, 1.0, b, 1.0),
            FunctionVector(n, Function.kSquare, 1.0, 0.0, lambd))


def elastic_net_functions(b, lambda1, lambda2, n):
    """g_j = kAbs with c = lambda1, e = lambda2 / 2   (graph.py:518,522; quirk kept:
    the engine's quadratic term is e x^2 / 2, so this penalises lambda2/4 x^2)"""
    b = np.asarray(b, dtype=np.float64).ravel()
    return (FunctionVector(len(b), Function.kSquare, 1.0, b, 1.0),
            FunctionVector(n, Function.kAbs, 1.0, 0.0, lambda1, 0.0, lambda2 / 2))


def logistic_functions(b, lambd, n):
    """f_i = log(1 + exp(-b_i y_i)) as kLogistic with a = -b_i   (graph.py:562-568)"""
    b = np.asarray(b, dtype=np.float64).ravel()
    f = FunctionVector(len(b), Function.kLogistic, -b, 0.0, 1.0)
    g = FunctionVector(n, Function.kAbs, 1.0, 0.0, lambd) if lambd > 0 else FunctionVector(n, Function.kZero)
    return f, g


def huber_functions(b, delta, lambd, n):
    """f_i = delta^2 huber((y_i - b_i) / delta)   (graph.py:614-620)"""
    b = np.asarray(b, dtype=np.float64).ravel()
    f = FunctionVector(len(b), Function.kHuber, 1.0 / delta, b / delta, delta * delta)
    g = FunctionVector(n, Function.kAbs, 1.0, 0.0, lambd) if lambd > 0 else FunctionVector(n, Function.kZero)
    return f, g


def svm_functions(b, lambd, n):
    """f_i = max(0, 1 - b_i y_i) as kMaxPos0 with a = -b_i, b = -1   (graph.py:660,663)"""
    b = np.asarray(b, dtype=np.float64).ravel()
    return (FunctionVector(len(b), Function.kMaxPos0, -b, -1.0, 1.0),
            FunctionVector(n, Function.kSquare, 1.0, 0.0, lambd))


def nonneg_ls_functions(b, n):
    """g_j = I(x_j >= 0)   (graph.py:702,705)"""
    b = np.asarray(b, dtype=np.float64).ravel()
    return (FunctionVector(len(b), Function.kSquare, 1.0, b, 1.0), FunctionVector(n, Function.kIndGe0))


def _shape(A):
    if HAS_SCIPY and sp.issparse(A):
        return A.shape
    return np.asarray(A).shape


def solve_lasso(A, b, lambd, abs_tol=1e-4, rel_tol=1e-4, max_iter=2500, verbose=0, rho=1.0, dtype=None):
    """minimize 0.5 ||A x - b||^2 + lambda ||x||_1   (reference: graph.py:393-433)"""
    m, n = _shape(A)
    f, g = lasso_functions(b, lambd, n)
    return _solve_graph_form(A, f, g, abs_tol, rel_tol, max_iter, verbose, rho, dtype=dtype)


def solve_ridge(A, b, lambd, abs_tol=1e-4, rel_tol=1e-4, max_iter=2500, verbose=0, rho=1.0, dtype=None):
    """minimize 0.5 ||A x - b||^2 + 0.5 lambda ||x||^2   (reference: graph.py:436-476)"""
    m, n = _shape(A)
    f, g = ridge_functions(b, lambd, n)
    return _solve_graph_form(A, f, g, abs_tol, rel_tol, max_iter, verbose, rho, dtype=dtype)


def solve_elastic_net(A, b, lambda1, lambda2, abs_tol=1e-4, rel_tol=1e-4, max_iter=2500, verbose=0, rho=1.0,
                      dtype=None):
    """minimize 0.5 ||A x - b||^2 + lambda1 ||x||_1 + 0.5 lambda2 ||x||^2   (reference: graph.py:479-524)"""
    m, n = _shape(A)
    f, g = elastic_net_functions(b, lambda1, lambda2, n)
    return _solve_graph_form(A, f, g, abs_tol, rel_tol, max_iter, verbose, rho, dtype=dtype)


def solve_logistic(A, b, lambd=0.0, abs_tol=1e-4, rel_tol=1e-4, max_iter=2500, verbose=0, rho=1.0, dtype=None):
    """minimize sum_i log(1 + exp(-b_i a_i' x)) + lambda ||x||_1   (reference: graph.py:527-570)"""
    m, n = _shape(A)
    f, g = logistic_functions(b, lambd, n)
    return _solve_graph_form(A, f, g, abs_tol, rel_tol, max_iter, verbose, rho, dtype=dtype)


def solve_huber(A, b, delta=1.0, lambd=0.0, abs_tol=1e-4, rel_tol=1e-4, max_iter=2500, verbose=0, rho=1.0,
                dtype=None):
    """minimize sum_i huber(A x - b, delta) + lambda ||x||_1   (reference: graph.py:573-622)"""
    m, n = _shape(A)
    f, g = huber_functions(b, delta, lambd, n)
    return _solve_graph_form(A, f, g, abs_tol, rel_tol, max_iter, verbose, rho, dtype=dtype)


def solve_svm(A, b, lambd=1.0, abs_tol=1e-4, rel_tol=1e-4, max_iter=2500, verbose=0, rho=1.0, dtype=None):
    """minimize sum_i max(0, 1 - b_i a_i' x) + 0.5 lambda ||x||^2   (reference: graph.py:625-665)"""
    m, n = _shape(A)
    f, g = svm_functions(b, lambd, n)
    return _solve_graph_form(A, f, g, abs_tol, rel_tol, max_iter, verbose, rho, dtype=dtype)


def solve_nonneg_ls(A, b, abs_tol=1e-4, rel_tol=1e-4, max_iter=2500, verbose=0, rho=1.0, dtype=None):
    """minimize 0.5 ||A x - b||^2 subject to x >= 0   (reference: graph.py:668-707)"""
    m, n = _shape(A)
    f, g = nonneg_ls_functions(b, n)
    return _solve_graph_form(A, f, g, abs_tol, rel_tol, max_iter, verbose, rho, dtype=dtype)


# ---------------------------------------------------------------------------
# Persistent handle (no counterpart in the reference's Python layer; its C++ API
# reuses the factorisation through _done_init, src/cpu/pogs.cpp:113-114)
# ---------------------------------------------------------------------------
# Handles still alive when the interpreter shuts down are destroyed by an atexit callback -- while the
# HIP runtime is certainly still there -- instead of by whatever order garbage collection and the C
# runtime's own exit handlers happen to take (a handle's destructor synchronises its stream).
_LIVE_SOLVERS = weakref.WeakSet()


def _close_live_solvers():
    for s in list(_LIVE_SOLVERS):
        try:
            s.close()
        except Exception:
            pass


atexit.register(_close_live_solvers)


class Solver:
    """Keeps the equilibrated matrix and its factorisation on the GPU across solves.

    ``A``: numpy array / scipy sparse matrix (host), an integer device pointer (dense), or a tuple
    (data_ptr, indptr_ptr, indices_ptr, nnz) of device addresses of a CSR matrix (sparse)
    with ``shape=(m, n)`` and ``device_ptr=True`` (e.g. ``tensor.data_ptr()``).
    ``dist``: None, or ``(rank, world, m_global, unique_id_bytes)`` for a
    row-sharded solve where this process holds ``m`` consecutive rows.
    """

    def __init__(self, A, dtype=None, shape=None, device_ptr=False, order=Ordering.ROW_MAJ, device=-1,
                 profile=False, projector=_lib.PROJ_DEFAULT, dist=None):
        self.dtype = _resolve_dtype(dtype)
        self._h = ctypes.c_void_p()
        code = _lib.F64 if self.dtype == np.float64 else _lib.F32
        opt = _lib.PogsAmdOptions(device=device, projector=projector, profile=int(profile))
        dist_s = None
        if dist is not None:
            rank, world, m_global, uid = dist
            dist_s = _lib.PogsAmdDist(rank=rank, world=world, m_global=m_global)
            uid = bytes(uid)
            assert len(uid) == _lib.UNIQUE_ID_BYTES
            # (a c_char array field reads back as an immutable bytes copy: write through the address)
            ctypes.memmove(ctypes.addressof(dist_s) + _lib.PogsAmdDist.unique_id.offset, uid, _lib.UNIQUE_ID_BYTES)
        if device_ptr:
            _lib.check_device_pointer_interop()
        self.sparse = (not device_ptr) and HAS_SCIPY and sp.issparse(A)
        if device_ptr and isinstance(A, (tuple, list)):
            # CSR already resident in HBM: (data_ptr, indptr_ptr, indices_ptr, nnz) device addresses
            # (data of the solver's dtype, int32 indices), with shape=(m, n)
            self.sparse = True
            self.m, self.n = shape
            dptr, pptr, iptr, nnz = (int(v) for v in A)
            st = lib.PogsAmdCreateSparse(ctypes.byref(self._h), code, int(Ordering.ROW_MAJ), self.m, self.n, nnz,
                                         ctypes.c_void_p(dptr), ctypes.c_void_p(pptr), ctypes.c_void_p(iptr), _lib.DEVICE,
                                         ctypes.byref(opt), ctypes.byref(dist_s) if dist_s is not None else None)
        elif self.sparse:
            A_csr = sp.csr_matrix(A, dtype=self.dtype)
            self.m, self.n = A_csr.shape
            data = np.ascontiguousarray(A_csr.data, dtype=self.dtype)
            ptr = np.ascontiguousarray(A_csr.indptr, dtype=np.int32)
            ind = np.ascontiguousarray(A_csr.indices, dtype=np.int32)
            st = lib.PogsAmdCreateSparse(ctypes.byref(self._h), code, int(Ordering.ROW_MAJ), self.m, self.n,
                                         A_csr.nnz, _ptr(data), _ptr(ptr), _ptr(ind), _lib.HOST, ctypes.byref(opt),
                                         ctypes.byref(dist_s) if dist_s is not None else None)
        else:
            if device_ptr:
                self.m, self.n = shape
                ptr_val, mem = ctypes.c_void_p(int(A)), _lib.DEVICE
            else:
                A = np.asarray(A, dtype=self.dtype, order="C" if order == Ordering.ROW_MAJ else "F")
                self.m, self.n = A.shape
                ptr_val, mem = _ptr(A), _lib.HOST
            st = lib.PogsAmdCreateDense(ctypes.byref(self._h), code, int(order), self.m, self.n, ptr_val, mem,
                                        ctypes.byref(opt), ctypes.byref(dist_s) if dist_s is not None else None)
        if st != 0:
            raise RuntimeError("pogs_amd: solver creation failed: " + _lib.last_error())
        _LIVE_SOLVERS.add(self)

    def _coef(self, f, g):
        """(f, g) as two PogsAmdFn structures: per-element fields as typed host arrays, fields that are still scalars
        as broadcast values (filled on the device; include/pogs_amd.h: PogsAmdSolveFn).  Returns (structs, keep-alive)."""
        assert len(f) == self.m and len(g) == self.n
        keep, structs = [], []
        for fv in (_as_vector(f), _as_vector(g)):
            st = _lib.PogsAmdFn()
            for k in FunctionVector._FIELDS:
                v = fv._v[k]
                if isinstance(v, np.ndarray):
                    arr = np.ascontiguousarray(v, dtype=np.int32 if k == "h" else self.dtype)
                    keep.append(arr)
                    setattr(st, k, arr.ctypes.data)
                else:
                    setattr(st, k, None)
                    setattr(st, k + "0", int(v) if k == "h" else float(v))
            structs.append(st)
        return structs, keep

    def warm_start(self, x0, l0):
        """Start the next solve from (x0, lambda0) instead of zero (reference: SetInitX /
        SetInitLambda, src/include/pogs.h:112-119; both are required, pogs.cpp:159-179)."""
        x0 = np.ascontiguousarray(x0, self.dtype)
        l0 = np.ascontiguousarray(l0, self.dtype)
        assert x0.shape == (self.n,) and l0.shape == (self.m,)
        if lib.PogsAmdSetWarmStart(self._h, _ptr(x0), _ptr(l0)) != 0:
            raise RuntimeError("pogs_amd: " + _lib.last_error())

    def solve(self, f, g, abs_tol=1e-4, rel_tol=1e-4, max_iter=2500, verbose=0, rho=1.0, adaptive_rho=True,
              gap_stop=True, x0=None, l0=None):
        if x0 is not None or l0 is not None:
            self.warm_start(x0, l0)
        args, keep = self._coef(f, g)
        x = np.zeros(self.n, self.dtype)
        y = np.zeros(self.m, self.dtype)
        l = np.zeros(self.m, self.dtype)
        mu = np.zeros(self.n, self.dtype)
        optval = ctypes.c_double()
        final_iter = ctypes.c_uint()
        status = lib.PogsAmdSolveFn(self._h, ctypes.byref(args[0]), ctypes.byref(args[1]), rho, abs_tol, rel_tol,
                                    int(max_iter), int(verbose), int(adaptive_rho), int(gap_stop), _ptr(x), _ptr(y),
                                    _ptr(l), _ptr(mu), ctypes.byref(optval), ctypes.byref(final_iter))
        del keep
        if status == POGS_ERROR:
            raise RuntimeError("pogs_amd: solve failed: " + _lib.last_error())
        return {"x": x, "y": y, "l": l, "mu": mu, "optval": optval.value, "iterations": final_iter.value,
                "status": status}

    def begin_run(self, f, g, abs_tol=1e-4, rel_tol=1e-4, max_iter=2500, rho=1.0, adaptive_rho=True, gap_stop=True):
        args, keep = self._coef(f, g)
        st = lib.PogsAmdBeginRunFn(self._h, ctypes.byref(args[0]), ctypes.byref(args[1]), rho, abs_tol, rel_tol,
                                   int(max_iter), int(adaptive_rho), int(gap_stop))
        del keep
        if st != 0:
            raise RuntimeError("pogs_amd: begin_run failed: " + _lib.last_error())

    def iterate(self, iters):
        """Advance exactly `iters` ADMM iterations; returns (seconds, solves_completed)."""
        sec = ctypes.c_double()
        solves = ctypes.c_uint()
        st = lib.PogsAmdIterate(self._h, int(iters), ctypes.byref(sec), ctypes.byref(solves))
        if st != 0:
            raise RuntimeError("pogs_amd: iterate failed: " + _lib.last_error())
        return sec.value, solves.value

    def stats(self):
        s = _lib.PogsAmdStats()
        lib.PogsAmdGetStats(self._h, ctypes.byref(s))
        return s.as_dict()

    def reset_stats(self):
        lib.PogsAmdResetStats(self._h)

    def equilibrated(self, want_matrix=True):
        """(A_eq, d, e, nrmA) as the engine holds them (dense solvers)."""
        A_eq = np.zeros((self.m, self.n), self.dtype) if want_matrix else None
        d = np.zeros(self.m, self.dtype)
        e = np.zeros(self.n, self.dtype)
        nrm = ctypes.c_double()
        st = lib.PogsAmdGetEquil(self._h, _ptr(A_eq) if want_matrix else None, _ptr(d), _ptr(e), ctypes.byref(nrm))
        if st != 0:
            raise RuntimeError("pogs_amd: " + _lib.last_error())
        return A_eq, d, e, nrm.value

    def project(self, x0, y0, tol=1e-8):
        x0 = np.ascontiguousarray(x0, self.dtype)
        y0 = np.ascontiguousarray(y0, self.dtype)
        x = np.zeros(self.n, self.dtype)
        y = np.zeros(self.m, self.dtype)
        st = lib.PogsAmdProject(self._h, _ptr(x0), _ptr(y0), tol, _ptr(x), _ptr(y))
        if st != 0:
            raise RuntimeError("pogs_amd: " + _lib.last_error())
        return x, y

    def mul(self, trans, alpha, x, beta, y):
        x = np.ascontiguousarray(x, self.dtype)
        y = np.array(y, dtype=self.dtype, copy=True)
        st = lib.PogsAmdMul(self._h, trans.encode()[0:1], alpha, _ptr(x), beta, _ptr(y))
        if st != 0:
            raise RuntimeError("pogs_amd: " + _lib.last_error())
        return y

    def close(self):
        if self._h:
            lib.PogsAmdDestroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def prox_eval(f, rho, v, dtype=None):
    """Element-wise ProxEval on the GPU (reference: src/include/prox_lib.h:207-230)."""
    dt = _resolve_dtype(dtype)
    fa = _as_vector(f).arrays(dt)
    v = np.ascontiguousarray(v, dt)
    out = np.zeros_like(v)
    st = lib.PogsAmdProxEval(_lib.F64 if dt == np.float64 else _lib.F32, len(v), _ptr(fa["h"]), _ptr(fa["a"]),
                             _ptr(fa["b"]), _ptr(fa["c"]), _ptr(fa["d"]), _ptr(fa["e"]), rho, _ptr(v), _ptr(out))
    if st != 0:
        raise RuntimeError("pogs_amd: " + _lib.last_error())
    return out


def func_eval(f, v, dtype=None):
    """sum_i f_i(v_i) on the GPU (reference: src/include/prox_lib.h:326-349,520-529)."""
    dt = _resolve_dtype(dtype)
    fa = _as_vector(f).arrays(dt)
    v = np.ascontiguousarray(v, dt)
    out = ctypes.c_double()
    st = lib.PogsAmdFuncEval(_lib.F64 if dt == np.float64 else _lib.F32, len(v), _ptr(fa["h"]), _ptr(fa["a"]),
                             _ptr(fa["b"]), _ptr(fa["c"]), _ptr(fa["d"]), _ptr(fa["e"]), _ptr(v), ctypes.byref(out))
    if st != 0:
        raise RuntimeError("pogs_amd: " + _lib.last_error())
    return out.value


def proj_subgrad_eval(f, x, v, dtype=None):
    """Element-wise projection of v onto the subdifferential of f_i at x_i, on the GPU
    (reference: ProjSubgradEval, src/include/prox_lib.h:468-493, 538-546)."""
    dt = _resolve_dtype(dtype)
    fa = _as_vector(f).arrays(dt)
    x = np.ascontiguousarray(x, dt)
    v = np.ascontiguousarray(v, dt)
    assert x.shape == v.shape
    out = np.zeros_like(v)
    st = lib.PogsAmdProjSubgradEval(_lib.F64 if dt == np.float64 else _lib.F32, len(v), _ptr(fa["h"]), _ptr(fa["a"]),
                                    _ptr(fa["b"]), _ptr(fa["c"]), _ptr(fa["d"]), _ptr(fa["e"]), _ptr(x), _ptr(v),
                                    _ptr(out))
    if st != 0:
        raise RuntimeError("pogs_amd: " + _lib.last_error())
    return out


def rand_uniform(n, dtype=None):
    """The Norm2Est start vector (reference: src/cpu/include/gsl/gsl_rand.h:8-16)."""
    dt = _resolve_dtype(dtype)
    out = np.zeros(n, dt)
    lib.PogsAmdRandUniform(_lib.F64 if dt == np.float64 else _lib.F32, n, _ptr(out))
    return out


def dist_unique_id():
    buf = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
    st = lib.PogsAmdDistUniqueId(buf)
    if st != 0:
        raise RuntimeError("pogs_amd: " + _lib.last_error())
    return buf.raw
