"""Gram accuracy / time against the MFMA chain length and tile (tuning aid; GPU box).
Each variant solves C2 in its own process; prints gram_ms, iterations and ||x - x_default|| / ||x||."""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import torch
    import pogs_amd
    from pogs_amd import graph as G
    m, n = 100000, 10000
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(0)
    A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float32)
    xt = torch.randn(n, generator=g, device=dev) * (torch.rand(n, generator=g, device=dev) < 0.1)
    b = (A @ xt + 0.1 * torch.randn(m, generator=g, device=dev)).cpu().numpy().astype(np.float64)
    s = pogs_amd.Solver(A.data_ptr(), dtype=np.float32, shape=(m, n), device_ptr=True, profile=True)
    f, gg = G.lasso_functions(b, 0.1, n)
    r = s.solve(f, gg)
    st = s.stats()
    np.save(sys.argv[1], r["x"])
    print(json.dumps(dict(gram_ms=st["gram_ms"], iters=int(r["iterations"]) + 1, optval=float(r["optval"]), status=int(r["status"]))))
else:
    variants = [("default", {}), ("fp32", {"POGS_AMD_GRAM": "fp32"}),
                ("t128 c4 noflush", {"POGS_AMD_GRAM_FLUSH": "0"}),
                ("t128 c16 flush1024", {"POGS_AMD_GRAM_CHAINS": "16"}),
                ("t256 c1", {"POGS_AMD_GRAM_TILE": "256"}),
                ("t256 c4", {"POGS_AMD_GRAM_TILE": "256", "POGS_AMD_GRAM_CHAINS": "4"}),
                ("t256 c8", {"POGS_AMD_GRAM_TILE": "256", "POGS_AMD_GRAM_CHAINS": "8"}),
                ("t256 c16", {"POGS_AMD_GRAM_TILE": "256", "POGS_AMD_GRAM_CHAINS": "16"})]
    x0 = None
    for i, (tag, env) in enumerate(variants):
        e = dict(os.environ); e.update(env)
        out = "/tmp/gcp_%d.npy" % i
        r = subprocess.run([sys.executable, __file__, out], env=e, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(tag, "FAILED", r.stderr[-300:]); continue
        d = json.loads(line[-1]); x = np.load(out).astype(np.float64)
        if x0 is None: x0 = x
        print("%-22s gram %.1f ms  iters %d  optval %.4f  rel_x vs default %.2e" % (tag, d["gram_ms"], d["iters"], d["optval"], np.linalg.norm(x - x0) / np.linalg.norm(x0)))
