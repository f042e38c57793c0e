"""Loads libpogs_amd.so (the HIP engine) and declares its C ABI (include/pogs_amd.h).

There is no CPU fallback: if the library is missing the import fails loudly,
exactly like the reference package does when libpogs_cpu.so is absent
(python/pogs/graph.py:69-76).
"""
import ctypes
import os
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libpogs_amd.so")

# PyTorch ships its own libamdhip64 / librccl.  A process that hands torch DEVICE pointers to this
# library (Solver(device_ptr=True): bench.py, the full-size tests) must have torch load its HIP
# runtime first, otherwise two runtimes end up in one address space and the pointers mean nothing
# to the other one.  The package itself never imports torch: a pure ctypes / C caller and this
# package behave alike.  Either import torch before pogs_amd, or set POGS_AMD_TORCH_PRELOAD=1 to
# have it done here; Solver(device_ptr=True) refuses to run if torch appeared in the process only
# after this library was loaded (torch_loaded_first below).
TORCH_LOADED_FIRST = "torch" in sys.modules
if not TORCH_LOADED_FIRST and os.environ.get("POGS_AMD_TORCH_PRELOAD", "0") == "1":
    try:
        import torch  # noqa: F401

        TORCH_LOADED_FIRST = True
    except ImportError:   # a machine without torch: nothing to preload, pure ctypes callers still work
        pass


def check_device_pointer_interop():
    """Called before a caller-supplied device pointer is used: raises if torch was imported AFTER
    this library (its HIP runtime is then a second one)."""
    if "torch" in sys.modules and not TORCH_LOADED_FIRST:
        raise RuntimeError(
            "pogs_amd was imported before torch: the two then run on separate HIP runtimes and a torch device "
            "pointer is not valid here.  Import torch first, or set POGS_AMD_TORCH_PRELOAD=1.")


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "pogs_amd: %s not found. Build it with:\n"
        "  python pogs_amd/build.py        (needs hipcc; cross-compiles for gfx950 without a GPU)\n" % LIB_PATH
    )

lib = ctypes.CDLL(LIB_PATH)

c_int, c_uint, c_size_t, c_double, c_float, c_void_p, c_char = (
    ctypes.c_int, ctypes.c_uint, ctypes.c_size_t, ctypes.c_double, ctypes.c_float, ctypes.c_void_p, ctypes.c_char)

UNIQUE_ID_BYTES = 128
F32, F64 = 0, 1
HOST, DEVICE = 0, 1
COL_MAJ, ROW_MAJ = 0, 1
PROJ_DEFAULT, PROJ_DIRECT, PROJ_CGLS = 0, 1, 2


class PogsAmdDist(ctypes.Structure):
    _fields_ = [("rank", c_int), ("world", c_int), ("m_global", c_size_t), ("unique_id", c_char * UNIQUE_ID_BYTES)]


class PogsAmdOptions(ctypes.Structure):
    _fields_ = [("device", c_int), ("projector", c_int), ("profile", c_int), ("reserved", c_int * 5)]


class PogsAmdStats(ctypes.Structure):
    _fields_ = [
        ("t_total_s", c_double), ("t_init_s", c_double), ("t_loop_s", c_double), ("t_h2d_s", c_double),
        ("iterations", c_uint), ("exact_iters", c_uint), ("norm_est_iters", c_uint), ("rho_updates", c_uint),
        ("cg_iters", ctypes.c_ulonglong), ("matvecs", ctypes.c_ulonglong), ("matvecs_init", ctypes.c_ulonglong),
        ("rho_final", c_double), ("nrmA", c_double),
        ("stream_ms", c_double), ("stream_launches", ctypes.c_ulonglong), ("stream_bytes", c_double),
        ("equil_ms", c_double), ("normest_ms", c_double), ("gram_ms", c_double), ("chol_ms", c_double),
        ("trtri_ms", c_double), ("gram_flops", c_double), ("reserved", c_double * 8),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}
        d["spec_hits"], d["spec_misses"] = self.reserved[0], self.reserved[1]
        d["collectives"] = int(self.reserved[2])   # all-reduce calls issued by the handle so far
        d["comm_nranks"] = int(self.reserved[3])   # ranks of the communicator as RCCL reports them (0: no row shards)
        return d


def _dense_sig(real):
    return [c_int, c_size_t, c_size_t, c_void_p] + [c_void_p] * 5 + [c_void_p] + [c_void_p] * 5 + [c_void_p] + \
        [real, real, real, c_uint, c_uint, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]


def _sparse_sig(real):
    return [c_int, c_size_t, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p] + [c_void_p] * 5 + [c_void_p] + \
        [c_void_p] * 5 + [c_void_p] + [real, real, real, c_uint, c_uint, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]


lib.PogsD.argtypes = _dense_sig(c_double)
lib.PogsS.argtypes = _dense_sig(c_float)
lib.PogsSparseD.argtypes = _sparse_sig(c_double)
lib.PogsSparseS.argtypes = _sparse_sig(c_float)
for _f in (lib.PogsD, lib.PogsS, lib.PogsSparseD, lib.PogsSparseS):
    _f.restype = c_int

lib.PogsAmdDistUniqueId.argtypes = [c_void_p]
lib.PogsAmdCreateDense.argtypes = [ctypes.POINTER(c_void_p), c_int, c_int, c_size_t, c_size_t, c_void_p, c_int,
                                   ctypes.POINTER(PogsAmdOptions), ctypes.POINTER(PogsAmdDist)]
lib.PogsAmdCreateSparse.argtypes = [ctypes.POINTER(c_void_p), c_int, c_int, c_size_t, c_size_t, c_size_t, c_void_p,
                                    c_void_p, c_void_p, c_int, ctypes.POINTER(PogsAmdOptions),
                                    ctypes.POINTER(PogsAmdDist)]
lib.PogsAmdSolve.argtypes = [c_void_p] + [c_void_p] * 12 + [c_double, c_double, c_double, c_uint, c_uint, c_int, c_int,
                                                            c_void_p, c_void_p, c_void_p, c_void_p,
                                                            ctypes.POINTER(c_double), ctypes.POINTER(c_uint)]
class PogsAmdFn(ctypes.Structure):
    """include/pogs_amd.h: a function vector whose NULL fields are broadcast scalars."""
    _fields_ = [("a", c_void_p), ("b", c_void_p), ("c", c_void_p), ("d", c_void_p), ("e", c_void_p), ("h", c_void_p),
                ("a0", c_double), ("b0", c_double), ("c0", c_double), ("d0", c_double), ("e0", c_double), ("h0", c_int)]


lib.PogsAmdSolveFn.argtypes = [c_void_p, ctypes.POINTER(PogsAmdFn), ctypes.POINTER(PogsAmdFn), c_double, c_double, c_double,
                               c_uint, c_uint, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                               ctypes.POINTER(c_double), ctypes.POINTER(c_uint)]
lib.PogsAmdBeginRunFn.argtypes = [c_void_p, ctypes.POINTER(PogsAmdFn), ctypes.POINTER(PogsAmdFn), c_double, c_double, c_double,
                                  c_uint, c_int, c_int]
lib.PogsAmdBeginRun.argtypes = [c_void_p] + [c_void_p] * 12 + [c_double, c_double, c_double, c_uint, c_int, c_int]
lib.PogsAmdIterate.argtypes = [c_void_p, c_uint, ctypes.POINTER(c_double), ctypes.POINTER(c_uint)]
lib.PogsAmdSetWarmStart.argtypes = [c_void_p, c_void_p, c_void_p]
lib.PogsAmdGetStats.argtypes = [c_void_p, ctypes.POINTER(PogsAmdStats)]
lib.PogsAmdResetStats.argtypes = [c_void_p]
lib.PogsAmdDestroy.argtypes = [c_void_p]
lib.PogsAmdDestroy.restype = None
lib.PogsAmdLastError.restype = ctypes.c_char_p
lib.PogsAmdProxEval.argtypes = [c_int, c_size_t] + [c_void_p] * 6 + [c_double, c_void_p, c_void_p]
lib.PogsAmdFuncEval.argtypes = [c_int, c_size_t] + [c_void_p] * 6 + [c_void_p, ctypes.POINTER(c_double)]
lib.PogsAmdProjSubgradEval.argtypes = [c_int, c_size_t] + [c_void_p] * 6 + [c_void_p, c_void_p, c_void_p]
lib.PogsAmdGetEquil.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_double)]
lib.PogsAmdProject.argtypes = [c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p]
lib.PogsAmdMul.argtypes = [c_void_p, c_char, c_double, c_void_p, c_double, c_void_p]
lib.PogsAmdRandUniform.argtypes = [c_int, c_size_t, c_void_p]
lib.PogsAmdReadBandwidth.argtypes = [c_int, c_size_t, c_int, ctypes.POINTER(c_double), ctypes.POINTER(c_int)]
lib.PogsAmdWaveSumCheck.argtypes = [c_int, c_size_t, c_void_p, c_void_p, c_void_p]


class PogsAmdPoolInfo(ctypes.Structure):
    _fields_ = [("mallocs", ctypes.c_ulonglong), ("reuses", ctypes.c_ulonglong), ("frees", ctypes.c_ulonglong),
                ("malloc_ms", c_double), ("free_ms", c_double), ("cached_bytes", c_size_t), ("live_bytes", c_size_t),
                ("peak_cached_bytes", c_size_t)]


lib.PogsAmdPoolStats.argtypes = [c_int, ctypes.POINTER(PogsAmdPoolInfo)]
lib.PogsAmdPoolTrim.argtypes = [c_int, ctypes.POINTER(c_size_t)]


def pool_stats(device=-1):
    """Counters of the library's device memory pool (include/pogs_amd.h: PogsAmdPoolInfo)."""
    info = PogsAmdPoolInfo()
    if lib.PogsAmdPoolStats(device, ctypes.byref(info)) != 0:
        raise RuntimeError(last_error())
    return {k: getattr(info, k) for k, _ in info._fields_}


def pool_trim(device=-1):
    """Give the idle device blocks of the pool back to the HIP runtime; returns the bytes freed."""
    freed = c_size_t(0)
    if lib.PogsAmdPoolTrim(device, ctypes.byref(freed)) != 0:
        raise RuntimeError(last_error())
    return freed.value

# Every symbol include/pogs_amd.h declares (checked by tests/test_abi.py).
ABI_SYMBOLS = [
    "PogsD", "PogsS", "PogsSparseD", "PogsSparseS",
    "PogsAmdDistUniqueId", "PogsAmdCreateDense", "PogsAmdCreateSparse", "PogsAmdSolve", "PogsAmdSolveFn", "PogsAmdBeginRun", "PogsAmdBeginRunFn",
    "PogsAmdIterate", "PogsAmdSetWarmStart", "PogsAmdGetStats", "PogsAmdResetStats", "PogsAmdDestroy", "PogsAmdLastError",
    "PogsAmdPoolStats", "PogsAmdPoolTrim",
    "PogsAmdProxEval", "PogsAmdFuncEval", "PogsAmdProjSubgradEval", "PogsAmdGetEquil", "PogsAmdProject", "PogsAmdMul", "PogsAmdRandUniform",
    "PogsAmdReadBandwidth", "PogsAmdWaveSumCheck",
]


def read_bandwidth(device=-1, nbytes=4 << 30, reps=10):
    """(GB/s, pattern) of the device's read-bandwidth probe (include/pogs_amd.h: PogsAmdReadBandwidth)."""
    gbs, pat = c_double(0.0), c_int(0)
    if lib.PogsAmdReadBandwidth(device, nbytes, reps, ctypes.byref(gbs), ctypes.byref(pat)) != 0:
        raise RuntimeError(last_error())
    return gbs.value, ("side-by-side grid stride", "row blocks")[pat.value]


def wave_sum_check(values):
    """(alu, lds): per input value its wavefront's total by the engine's ALU reduction and by the __shfl_xor
    butterfly (include/pogs_amd.h: PogsAmdWaveSumCheck); `values`: float32 / float64, a multiple of 64 long."""
    import numpy as np
    v = np.ascontiguousarray(values)
    assert v.dtype in (np.float32, np.float64) and v.size % 64 == 0
    a, b = np.empty_like(v), np.empty_like(v)
    if lib.PogsAmdWaveSumCheck(0 if v.dtype == np.float32 else 1, v.size, v.ctypes.data, a.ctypes.data, b.ctypes.data) != 0:
        raise RuntimeError(last_error())
    return a, b


def last_error():
    msg = lib.PogsAmdLastError()
    return msg.decode() if msg else ""
