"""Setup breakdown of C2 in fp64 (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import pogs_amd
from pogs_amd import graph as G

m, n = 100000, 10000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float64)
xt = torch.randn(n, generator=g, device=dev, dtype=torch.float64) * (torch.rand(n, generator=g, device=dev) < 0.1)
b = (A @ xt + 0.1 * torch.randn(m, generator=g, device=dev, dtype=torch.float64)).cpu().numpy()
torch.cuda.synchronize()
t0 = time.time()
s = pogs_amd.Solver(A.data_ptr(), dtype=np.float64, shape=(m, n), device_ptr=True, profile=True)
t1 = time.time()
f, gg = G.lasso_functions(b, 0.1, n)
r = s.solve(f, gg)
st = s.stats()
print("create %.3f s; iterations %d; %.1f it/s" % (t1 - t0, st["iterations"], st["iterations"] / st["t_loop_s"]))
print({k: round(v, 1) for k, v in st.items() if k.endswith("_ms")})
