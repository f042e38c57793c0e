"""Cholesky with and without look-ahead (csrc/gemm.hip: cholesky_lower) at a C2-shaped problem:
setup phase times and bitwise equality of the solve (development aid).

    python scripts/chol_lookahead_probe.py [m n]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import pogs_amd
from pogs_amd import graph as G

m = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float32)
xt = torch.randn(n, generator=g, device=dev) * (torch.rand(n, generator=g, device=dev) < 0.1)
b = (A @ xt + 0.1 * torch.randn(m, generator=g, device=dev)).cpu().numpy().astype(np.float64)
f, gg = G.lasso_functions(b, 0.1, n)
res = {}
for mode in ("1", "0", "1", "0"):
    os.environ["POGS_AMD_CHOL_LOOKAHEAD"] = mode
    with pogs_amd.Solver(A.data_ptr(), dtype=np.float32, shape=(m, n), device_ptr=True) as s:
        r = s.solve(f, gg)
        st = s.stats()
    print(json.dumps({"lookahead": mode, "chol_ms": st["chol_ms"], "trtri_ms": st["trtri_ms"], "gram_ms": st["gram_ms"],
                      "init_s": st["t_init_s"], "iters": r["iterations"] + 1}), flush=True)
    res.setdefault(mode, []).append(r)
same = all(np.array_equal(res["1"][0][k], res["0"][0][k]) for k in "xyl")
print("bitwise identical with and without look-ahead:", same, "| repeat identical:",
      all(np.array_equal(res["1"][0][k], res["1"][1][k]) for k in "xyl"))
