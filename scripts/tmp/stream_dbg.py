"""Per-workgroup start/end stamps of the C2 pass over A (debug variant of the library)."""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pogs_amd
from pogs_amd import synth, _lib
m, n = 100000, 10000
rng = np.random.default_rng(0)
A = rng.standard_normal((m, n), dtype=np.float32)
b = rng.standard_normal(m).astype(np.float32)
f = _lib.lib.PogsAmdDebugStreamTimes
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
def dump(tag, nwg=512):
    buf = np.zeros(4096, np.uint64)
    assert f(buf.ctypes.data_as(ctypes.c_void_p), 4096) == 0
    d = buf.reshape(-1, 4)[:nwg].astype(np.int64)
    t0 = d[:, 0].min()
    st = (d[:, 0] - t0) / 100.0; en = (d[:, 1] - t0) / 100.0
    dur = en - st
    print("%s: start spread %.1f us  end: min %.1f med %.1f max %.1f us   dur: min %.1f med %.1f max %.1f" %
          (tag, st.max(), en.min(), np.median(en), en.max(), dur.min(), np.median(dur), dur.max()))
    xcc = d[:, 2] & 0xF
    for k in range(8):
        msk = xcc == k
        if msk.any(): print("   xcc %d: n %d  end med %.1f max %.1f min %.1f" % (k, msk.sum(), np.median(en[msk]), en[msk].max(), en[msk].min()))
for max_iter in (20, 40, 60):
    r = pogs_amd.solve_lasso(A, b, 0.1, dtype=np.float32, max_iter=max_iter)
    dump("after %d iterations (status %s)" % (max_iter, r["status"]))
