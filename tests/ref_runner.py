"""Runs the compiled reference (oracle/_ref/libpogs_cpu.so) in a clean process.

TEST INFRASTRUCTURE.  Invoked by oracle_binding.ref_solve as
    python ref_runner.py <dir>        (dir holds in.npz [+ A.npy]; writes out.npz)
Only numpy + ctypes are imported here -- in particular not torch (see ref_solve).
"""
import ctypes
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# POGS_REF_SO: another build of the same reference sources (oracle/_ref/libpogs_cpu_openblas.so, timing only)
REF_SO = os.environ.get("POGS_REF_SO") or os.path.join(HERE, "..", "oracle", "_ref", "libpogs_cpu.so")


def main(td):
    z = np.load(os.path.join(td, "in.npz"))
    dtype = np.dtype(str(z["dtype"]))
    rho, abs_tol, rel_tol, max_iter, verbose, adaptive_rho, gap_stop, order = z["params"]
    lib = ctypes.CDLL(REF_SO)
    c = ctypes.c_double if dtype == np.float64 else ctypes.c_float
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    coef = {}
    for pre in "fg":
        for k in "abcde":
            coef[pre + k] = np.ascontiguousarray(z[pre + "_" + k], dtype=dtype)
        coef[pre + "h"] = np.ascontiguousarray(z[pre + "_h"], dtype=np.int32)
    sparse = "sp_data" in z
    if sparse:
        m, n = (int(v) for v in z["sp_shape"])
        data = np.ascontiguousarray(z["sp_data"], dtype=dtype)
        ptr = np.ascontiguousarray(z["sp_ptr"], dtype=np.int32)
        ind = np.ascontiguousarray(z["sp_ind"], dtype=np.int32)
        head = [ctypes.c_int(1), ctypes.c_size_t(m), ctypes.c_size_t(n), ctypes.c_size_t(len(data)), p(data), p(ptr),
                p(ind)]
        fn = lib.PogsSparseD if dtype == np.float64 else lib.PogsSparseS
    else:
        # (memory-mapped: the reference makes its own copy of A, matrix_dense.cpp:85-87, so the file's pages
        # are read once and never duplicated in this process)
        A = np.load(os.path.join(td, "A.npy"), mmap_mode="r")
        A = np.ascontiguousarray(A, dtype=dtype) if int(order) == 1 else np.asfortranarray(A, dtype=dtype)
        m, n = A.shape
        head = [ctypes.c_int(int(order)), ctypes.c_size_t(m), ctypes.c_size_t(n), p(A)]
        fn = lib.PogsD if dtype == np.float64 else lib.PogsS
    x = np.zeros(n, dtype)
    y = np.zeros(m, dtype)
    l = np.zeros(m, dtype)
    optval = c()
    final_iter = ctypes.c_uint()
    args = head + [p(coef[k]) for k in ("fa", "fb", "fc", "fd", "fe", "fh", "ga", "gb", "gc", "gd", "ge", "gh")] + [
        c(rho), c(abs_tol), c(rel_tol), ctypes.c_uint(int(max_iter)), ctypes.c_uint(int(verbose)),
        ctypes.c_int(int(adaptive_rho)), ctypes.c_int(int(gap_stop)), p(x), p(y), p(l), ctypes.byref(optval),
        ctypes.byref(final_iter)]
    t0 = time.time()
    status = fn(*args)
    wall = time.time() - t0
    sys.stdout.flush()
    np.savez(os.path.join(td, "out.npz"), x=x, y=y, l=l, optval=optval.value, iterations=final_iter.value,
             status=status, wall_s=wall)


if __name__ == "__main__":
    main(sys.argv[1])
