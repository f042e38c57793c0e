"""Per-workgroup start/end stamps of the C4 SpMVs (debug variant of the library)."""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pogs_amd
from pogs_amd import synth, _lib
A, b, _ = synth.csr_lasso(2000000, 500000, 50, seed=4, dtype=np.float32)
rng = np.random.default_rng(0)
x = rng.standard_normal(500000).astype(np.float32); y = rng.standard_normal(2000000).astype(np.float32)
f = _lib.lib.PogsAmdDebugSellTimes
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
def dump(tag, nwg):
    buf = np.zeros(8192, np.uint64)
    assert f(buf.ctypes.data_as(ctypes.c_void_p), 8192) == 0
    d = buf.reshape(-1, 4)[:nwg].astype(np.int64)
    t0 = d[:, 0].min()
    st = (d[:, 0] - t0) / 100.0; en = (d[:, 1] - t0) / 100.0   # us (100 MHz)
    dur = en - st
    print("%s: nwg %d  start spread %.1f us  end: min %.1f med %.1f max %.1f us   dur: min %.1f med %.1f max %.1f" %
          (tag, nwg, st.max(), en.min(), np.median(en), en.max(), dur.min(), np.median(dur), dur.max()))
    xcc = d[:, 2] & 0xF
    for k in range(8):
        m = xcc == k
        if m.any(): print("   xcc %d: n %d  dur med %.1f max %.1f  end max %.1f" % (k, m.sum(), np.median(dur[m]), dur[m].max(), en[m].max()))
    o = np.argsort(en)[-8:]
    print("   slowest wgs:", [(int(i), round(float(st[i]), 1), round(float(en[i]), 1), int(xcc[i])) for i in o])
with pogs_amd.Solver(A, dtype=np.float32) as s:
    for it in range(3):
        s.mul("n", 1.0, x, 0.0, y); dump("A   (246 wgs)", 246)
        s.mul("t", 1.0, y, 0.0, x); dump("A^T (248 wgs)", 248)
