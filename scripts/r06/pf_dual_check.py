"""Development aid (round 6): the 256 x 5 logistic test problem on the library in place; prints what _check_solution compares."""
import sys, os, faulthandler, time
faulthandler.dump_traceback_later(90, repeat=False)
T0 = time.time()
def say(*a):
    print("[%.1f s]" % (time.time() - T0), *a, flush=True)
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import oracle_binding as ob
ob.oracle_set_threads()
import pogs_amd as pogs
from pogs_amd import synth
soa = lambda fv: {k: getattr(fv, k) for k in "habcde"}
m, n = int(sys.argv[1]), int(sys.argv[2])
A, y, _ = synth.dense_logistic(m, n, seed=m % 97, dtype=np.float32, logit_std=2.0)
lam = 0.05 * float(np.max(np.abs(A.T @ y)))
f, g = pogs.graph.logistic_functions(y, lam, n)
say("problem ready")
got = pogs.solve_logistic(A, y, lam, dtype=np.float32)
say("gpu solve done", got["iterations"])
want = ob.oracle_solve(A, soa(f), soa(g), dtype=np.float32)
w64 = want
say("oracle done")
rel = lambda a, b: np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b)
print("iterations gpu %d oracle32 %d oracle64 %d" % (got["iterations"], want["iterations"], w64["iterations"]))
for k in ("x", "y", "l"):
    print(k, "gpu vs oracle32 %.2e   gpu vs oracle64 %.2e   oracle32 vs oracle64 %.2e" % (rel(got[k], want[k]), rel(got[k], w64[k]), rel(want[k], w64[k])))
print("optval", got["optval"], want["optval"], w64["optval"])
