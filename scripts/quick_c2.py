"""Quick look at the C2-shaped solve on one GPU (development aid, not the bench)."""
import json
import sys
import time

import os
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
torch.cuda.synchronize()
t0 = time.time()
s = pogs_amd.Solver(A.data_ptr(), dtype=np.float32, shape=(m, n), device_ptr=True, profile=True)
t1 = time.time()
f, gg = G.lasso_functions(b, 0.1, n)
r = s.solve(f, gg, verbose=int(os.environ.get("QC_VERBOSE", "0")))
t2 = time.time()
st = s.stats()
print(json.dumps({"create_s": t1 - t0, "solve_s": t2 - t1, "status": r["status"], "iters": r["iterations"],
                  "optval": r["optval"], **st}, indent=1))
it = st["iterations"]
print("fronts_ahead", st.get("fronts_ahead"), "noops", st.get("fronts_ahead_noops")); print("it/s", it / st["t_loop_s"], "ms/iter", 1e3 * st["t_loop_s"] / it, "spec hits/misses", st.get("spec_hits"), st.get("spec_misses"))
if st["stream_launches"]:
    avg = st["stream_ms"] / st["stream_launches"]
    print("stream kernel avg ms", avg, "GB/s", st["stream_bytes"] / st["stream_launches"] / (avg * 1e-3) / 1e9)
