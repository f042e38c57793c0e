"""C2 shape in float64 (the reference's default dtype) and fp32 n = 16384 on one GPU (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import pogs_amd
from pogs_amd import graph as G

def run(m, n, tdtype, ndtype):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(0)
    A = torch.randn((m, n), generator=g, device=dev, dtype=tdtype)
    xt = (torch.randn(n, generator=g, device=dev) * (torch.rand(n, generator=g, device=dev) < 0.1)).to(tdtype)
    b = (A @ xt + 0.1 * torch.randn(m, generator=g, device=dev, dtype=tdtype)).cpu().numpy().astype(np.float64)
    torch.cuda.synchronize()
    s = pogs_amd.Solver(A.data_ptr(), dtype=ndtype, shape=(m, n), device_ptr=True, profile=True)
    f, gg = G.lasso_functions(b, 0.1, n)
    r = s.solve(f, gg)
    st = s.stats(); it = st["iterations"]
    print("%d x %d %s: status %d, %d iterations, %.1f it/s (%.3f ms/iter), init %.3f s, passes/iter %.2f, spec hits %d" % (
        m, n, np.dtype(ndtype).name, r["status"], it, it / st["t_loop_s"], 1e3 * st["t_loop_s"] / it, st["t_init_s"],
        st["stream_launches"] / it, st["spec_hits"]))
    s.close(); del A

run(100000, 10000, torch.float64, np.float64)
run(60000, 16384, torch.float32, np.float32)
