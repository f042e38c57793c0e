"""Development aid: does the package work in a process that never imports torch (system HIP runtime)?"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
t0 = time.time()
import numpy as np
import pogs_amd as pogs
from pogs_amd import synth
print("import %.1f s, torch in process: %s" % (time.time() - t0, "torch" in sys.modules), flush=True)
A, b, lam = synth.readme_lasso()
r = pogs.solve_lasso(A, b, lam)
print("readme lasso: status %d iterations %d optval %.12g  (%.1f s)" % (r["status"], r["iterations"], r["optval"], time.time() - t0), flush=True)
A, b, _ = synth.dense_lasso(3000, 400, seed=1, dtype=np.float32)
r = pogs.solve_lasso(A, b, 0.1, dtype=np.float32)
print("3000 x 400 fp32: status %d iterations %d  (%.1f s)" % (r["status"], r["iterations"], time.time() - t0), flush=True)
