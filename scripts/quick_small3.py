"""Per-call wall times of the one-shot ABI on small problems (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import pogs_amd
from pogs_amd import synth

mode = sys.argv[1] if len(sys.argv) > 1 else "host"
keep = []
for (m, n, dt) in [(2000, 300, np.float32), (5000, 1000, np.float32), (8000, 1000, np.float32), (20000, 2000, np.float32)]:
    A, b, _ = synth.dense_lasso(m, n, seed=0, dtype=dt)
    keep.append(A)
    if mode == "device":
        A = torch.from_numpy(A).cuda()
        torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        t0 = time.time()
        r = pogs_amd.solve_lasso(A, b, 0.1, dtype=dt)
        ts.append((time.time() - t0) * 1e3)
    print(mode, m, n, " ".join("%.1f" % t for t in ts), "ms; iterations", r["iterations"] + 1, flush=True)
