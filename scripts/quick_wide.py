"""Wide dense problem on one GPU (development aid): m x n with m <= n."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import pogs_amd
from pogs_amd import graph as G

m = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float32)
xt = torch.randn(n, generator=g, device=dev) * (torch.rand(n, generator=g, device=dev) < 0.01)
b = (A @ xt + 0.1 * torch.randn(m, generator=g, device=dev))
lam = 0.2 * float(torch.max(torch.abs(A.T @ b)))
b = b.cpu().numpy().astype(np.float64)
torch.cuda.synchronize()
t0 = time.time()
s = pogs_amd.Solver(A.data_ptr(), dtype=np.float32, shape=(m, n), device_ptr=True, profile=True)
t1 = time.time()
f, gg = G.lasso_functions(b, lam, n)
r = s.solve(f, gg, verbose=1)
t2 = time.time()
st = s.stats()
it = st["iterations"]
print("create %.3f s, solve %.3f s, status %d, iterations %d, %.1f it/s, %.3f ms/iter" % (t1 - t0, t2 - t1, r["status"], it, it / st["t_loop_s"], 1e3 * st["t_loop_s"] / it))
print({k: round(v, 2) for k, v in st.items() if k.endswith("_ms")})
if st["stream_launches"]:
    avg = st["stream_ms"] / st["stream_launches"]
    print("stream kernel avg ms %.3f, launches/iter %.2f" % (avg, st["stream_launches"] / it))
