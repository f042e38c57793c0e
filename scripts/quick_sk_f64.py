"""fp64 Sinkhorn-Knopp shortcut vs the full 50 passes on small and medium shapes (development aid)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import pogs_amd

rng = np.random.default_rng(3)
for (m, n) in [(500, 300), (3000, 1100), (300, 2500), (20000, 2000), (60000, 3000)]:
    A = rng.standard_normal((m, n))
    if m == 3000:
        A *= rng.uniform(0.1, 10.0, (m, 1)) * rng.uniform(0.1, 10.0, (1, n))
    out = {}
    for mode in ("short", "full"):
        if mode == "full":
            os.environ["POGS_AMD_SK_FULL"] = "1"
        else:
            os.environ.pop("POGS_AMD_SK_FULL", None)
        with pogs_amd.Solver(A, dtype=np.float64) as s:
            _, d, e, nrm = s.equilibrated(want_matrix=False)
            out[mode] = (d, e, s.stats()["matvecs_init"])
    d0, e0, p0 = out["short"]; d1, e1, p1 = out["full"]
    print("%6d x %5d: passes %d / %d, max rel diff d %.2e e %.2e" % (m, n, p0, p1, np.max(np.abs(d0 / d1 - 1)), np.max(np.abs(e0 / e1 - 1))), flush=True)
