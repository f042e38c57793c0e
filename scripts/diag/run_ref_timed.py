import os, sys, subprocess, tempfile, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from pogs_amd import graph as G, synth
m, n, maxit = int(sys.argv[1]), 10000, int(sys.argv[2])
A, b, _ = synth.dense_lasso_rows(m, n, seed=2024)
f, g = G.lasso_functions(b, 0.1, n)
td = tempfile.mkdtemp(dir="/dev/shm")
np.save(td + "/A.npy", A)
payload = {"dtype": "float32", "params": np.array([1.0, 1e-4, 1e-4, maxit, 1, 1, 1, 1], dtype=np.float64)}
for k in "habcde":
    payload["f_" + k] = np.asarray(getattr(f, k)); payload["g_" + k] = np.asarray(getattr(g, k))
np.savez(td + "/in.npz", **payload)
del A
env = dict(os.environ); env.update(MKL_NUM_THREADS=os.environ.get("NT", "16"), OMP_NUM_THREADS=os.environ.get("NT", "16"), MKL_DYNAMIC="FALSE", LD_PRELOAD="/tmp/blas_timer.so")
for kv in sys.argv[3:]:
    k, v = kv.split("="); env[k] = v
t0 = time.time()
r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "ref_runner.py"), td], env=env, capture_output=True, text=True)
print(r.stdout[-600:]); print(r.stderr[-2500:]); print("wall", time.time() - t0)
subprocess.call(["rm", "-rf", td])
