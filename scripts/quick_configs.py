"""Full-size C3 (logistic 200000x5000) and C4 (CSR 2e6 x 5e5, nnz 1e8) on one GPU (development aid)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import pogs_amd
from pogs_amd import graph as G

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
if which == "c3":
    m, n = 200000, 5000
    A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float32)
    w = torch.randn(n, generator=g, device=dev) * (torch.rand(n, generator=g, device=dev) < 0.3)
    w = w * (2.0 / torch.linalg.norm(w))  # logits with std 2 (see pogs_amd/synth.py:dense_logistic)
    p = torch.sigmoid(A @ w)
    y = (2.0 * (torch.rand(m, generator=g, device=dev) < p).double() - 1.0).cpu().numpy()
    t0 = time.time()
    s = pogs_amd.Solver(A.data_ptr(), dtype=np.float32, shape=(m, n), device_ptr=True, profile=True)
    t1 = time.time()
    f, gg = G.logistic_functions(y, 0.01, n)
    r = s.solve(f, gg, verbose=2)
    t2 = time.time()
else:
    import scipy.sparse as sp

    m, n, k = 2000000, 500000, 50
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    m, n = int(m * scale), int(n * scale)
    cols = torch.randint(0, n, (m, k), generator=g, device=dev, dtype=torch.int32)
    cols, _ = torch.sort(cols, dim=1)
    vals = torch.randn((m, k), generator=g, device=dev, dtype=torch.float32)
    ptr = np.arange(0, m * k + 1, k, dtype=np.int32)
    A = sp.csr_matrix((vals.cpu().numpy().ravel(), cols.cpu().numpy().ravel(), ptr), shape=(m, n))
    A.sum_duplicates()
    rng = np.random.default_rng(0)
    xt = rng.standard_normal(n) * (rng.random(n) < 0.05)
    b = A @ xt + 0.1 * rng.standard_normal(m)
    print("nnz", A.nnz, flush=True)
    t0 = time.time()
    s = pogs_amd.Solver(A, dtype=np.float32, profile=True)
    t1 = time.time()
    f, gg = G.lasso_functions(b, 0.1, n)
    r = s.solve(f, gg, verbose=2)
    t2 = time.time()
st = s.stats()
print(json.dumps({"create_s": t1 - t0, "solve_s": t2 - t1, "status": r["status"], "iters": r["iterations"],
                  "optval": r["optval"], **st}, indent=1))
it = st["iterations"]
print("it/s", it / st["t_loop_s"], "ms/iter", 1e3 * st["t_loop_s"] / it)
if st["stream_launches"]:
    avg = st["stream_ms"] / st["stream_launches"]
    print("stream kernel avg ms", avg, "GB/s", st["stream_bytes"] / st["stream_launches"] / (avg * 1e-3) / 1e9,
          "launches/iter", st["stream_launches"] / it)
