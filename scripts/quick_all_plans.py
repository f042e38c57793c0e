"""Every streaming shape x arithmetic type once (development aid): square-ish lasso problems whose
row length selects each entry of POGS_STREAM_PLANS and the windowed form; fp32 and fp64 solve the
same data on different shapes, so agreement of the two checks both code objects."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import pogs_amd
from pogs_amd import graph as G

dev = torch.device("cuda:0")
vprs = [60, 120, 250, 500, 760, 1000, 1270, 1500, 2000, 2500, 3000, 4000, 5000, 6000, 8000, 9000]
sizes = sorted(set([2 * v for v in vprs] + [4 * v for v in vprs]))
wide = "--wide" in sys.argv     # m < n: transposed storage, the mirrored iteration
args = [a for a in sys.argv[1:] if a != "--wide"]
if args:
    sizes = [int(a) for a in args]
for k in sizes:
    m, n = (k, k + 64) if wide else (k + 64, k)
    g = torch.Generator(device=dev); g.manual_seed(k)
    A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float64)
    xt = torch.randn(n, generator=g, device=dev, dtype=torch.float64) * (torch.rand(n, generator=g, device=dev) < 0.05)
    b = A @ xt + 0.1 * torch.randn(m, generator=g, device=dev, dtype=torch.float64)
    lam = 0.3 * float(torch.max(torch.abs(A.T @ b)))
    bh = b.cpu().numpy()
    f, gg = G.lasso_functions(bh, lam, n)
    res = {}
    for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
        At = A.to(tdt)
        torch.cuda.synchronize()
        t0 = time.time()
        with pogs_amd.Solver(At.data_ptr(), dtype=dt, shape=(m, n), device_ptr=True) as s:
            r = s.solve(f, gg)
        res[dt] = (r, time.time() - t0)
        del At
    r64, t64 = res[np.float64]; r32, t32 = res[np.float32]
    err = np.linalg.norm(r32["x"].astype(np.float64) - r64["x"]) / max(np.linalg.norm(r64["x"]), 1e-300)
    ok = r64["status"] == 0 and r32["status"] == 0 and err < 5e-3
    print("k %6d: fp64 status %d it %4d (%.2f s)   fp32 status %d it %4d (%.2f s)   x rel diff %.1e  %s" % (
        k, r64["status"], r64["iterations"], t64, r32["status"], r32["iterations"], t32, err, "ok" if ok else "MISMATCH"), flush=True)
    del A
    torch.cuda.empty_cache()
