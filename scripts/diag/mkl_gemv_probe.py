import ctypes, numpy as np, time, os, resource, sys
mkl = ctypes.CDLL("/opt/conda/lib/libmkl_rt.so")
m, n = int(sys.argv[1]), 10000
A = np.empty((m, n), np.float32)
rng = np.random.default_rng(0)
for r in range(0, m, 5000): A[r:r+5000] = rng.standard_normal((min(5000, m-r), n), dtype=np.float32)
x = np.ones(n, np.float32); y = np.zeros(m, np.float32); xt = np.zeros(n, np.float32)
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
RowMajor, NoTrans, Trans = 101, 111, 112
res = []
for it in range(8):
    for tr in (Trans, NoTrans):
        r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.time()
        if tr == NoTrans:
            mkl.cblas_sgemv(RowMajor, NoTrans, m, n, ctypes.c_float(1), p(A), n, p(x), 1, ctypes.c_float(0), p(y), 1)
        else:
            mkl.cblas_sgemv(RowMajor, Trans, m, n, ctypes.c_float(1), p(A), n, p(y), 1, ctypes.c_float(1), p(xt), 1)
        t = time.time() - t0; r1 = resource.getrusage(resource.RUSAGE_SELF)
        cpu = (r1.ru_utime + r1.ru_stime) - (r0.ru_utime + r0.ru_stime)
        res.append("%s %.0fms/%.1f" % ("T" if tr == Trans else "N", 1e3 * t, cpu / t))
print(" ".join(res))
