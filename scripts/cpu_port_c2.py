"""The OpenMP oracle port (oracle/pogs_oracle.cpp, pinned to the reference in tests/) on the whole C2
workload with every core the container grants -- the "all host cores" side of the comparison that the
compiled reference cannot give on the GPU box's host, where its MKL runs sgemv on one thread
(profiles/r03_ref_cpu_diagnosis.md).  One-off measurement for DESIGN.md section 6 (its plain-loop Gram
product and Cholesky take minutes; not part of bench.py's default run).

    python scripts/cpu_port_c2.py [max_iter]"""
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

max_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
threads = ob.oracle_set_threads()
A, b, _ = synth.dense_lasso_rows(100000, 10000, seed=2024)
f, g = G.lasso_functions(b, 0.1, 10000)
soa = lambda fv: {k: getattr(fv, k) for k in "habcde"}  # noqa: E731
t0 = time.time()
r = ob.oracle_solve(A, soa(f), soa(g), dtype=np.float32, max_iter=max_iter)
wall = time.time() - t0
it = r["iterations"] + 1
print(json.dumps({"threads": threads, "status": r["status"], "iterations": it, "t_init_s": r["info"]["t_init"],
                  "t_loop_s": r["info"]["t_loop"], "it_per_s": it / r["info"]["t_loop"], "wall_s": wall}))
