"""How the compiled reference's speed depends on the BLAS thread count on this box (development
aid for bench.py's cpu_baseline: SURVEY.md 8(d) asks for the CPU path in its best configuration).

    python scripts/ref_threads_probe.py rows cols "8,16,32,64" """
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle_binding as ob  # noqa: E402
from pogs_amd import graph as G  # noqa: E402
from pogs_amd import synth  # noqa: E402

m, n = int(sys.argv[1]), int(sys.argv[2])
threads = [int(v) for v in sys.argv[3].split(",")]
A, b, _ = synth.dense_lasso(m, n, seed=0, dtype=np.float32)
f, g = G.lasso_functions(b, 0.1, n)
fs = {k: getattr(f, k) for k in "habcde"}
gs = {k: getattr(g, k) for k in "habcde"}
for t in threads:
    os.environ["MKL_NUM_THREADS"] = str(t)
    os.environ["OMP_NUM_THREADS"] = str(t)
    t0 = time.time()
    try:
        r = ob.ref_solve(A, fs, gs, dtype=np.float32, verbose=1, timeout=float(sys.argv[4]) if len(sys.argv) > 4 else 600)
        print(json.dumps({"threads": t, "iters": r["iterations"] + 1, "t_total": r.get("t_total"), "t_init": r.get("t_init"),
                          "its": (r["iterations"] + 1) / max(r.get("t_total", 0) - r.get("t_init", 0), 1e-9),
                          "wall": time.time() - t0}), flush=True)
    except Exception as e:
        print(json.dumps({"threads": t, "error": repr(e)[:200], "wall": time.time() - t0}), flush=True)
