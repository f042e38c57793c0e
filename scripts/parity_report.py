"""Measured deviation of the engine from the oracle on the parity problems whose tests use an
fp32 tolerance above 1e-4 (development aid for the tolerance table of DESIGN.md section 5).

    python scripts/parity_report.py        (on the GPU box)"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle_binding as ob  # noqa: E402
import pogs_amd  # noqa: E402
from pogs_amd import _lib, synth  # noqa: E402
from pogs_amd import graph as G  # noqa: E402


def soa(fv):
    return {k: getattr(fv, k) for k in "habcde"}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def report(name, got, want, st=None):
    out = {"problem": name, "iters": [int(got["iterations"]), int(want["iterations"])], "status": [got["status"], want["status"]],
           "rel_x": rel(got["x"], want["x"]), "rel_y": rel(got["y"], want["y"]),
           "rel_optval": abs(got["optval"] - want["optval"]) / abs(want["optval"])}
    if st:
        out["cg_iters"] = [st.get("cg_iters"), want.get("info", {}).get("cg_iters")]
    print(json.dumps(out), flush=True)


dt = np.float32
A, b, _ = synth.csr_lasso(20000, 5000, 50, seed=3, dtype=dt)
f, g = G.lasso_functions(b, 0.1, 5000)
with pogs_amd.Solver(A, dtype=dt) as s:
    got = s.solve(f, g)
    st = s.stats()
report("csr 20000x5000 lasso", got, ob.oracle_solve(A, soa(f), soa(g), dtype=dt), st)
for tol in (1e-6,):
    with pogs_amd.Solver(A, dtype=dt) as s:
        got = s.solve(f, g, abs_tol=tol, rel_tol=tol)
    report("csr 20000x5000 lasso, tolerances %g" % tol, got, ob.oracle_solve(A, soa(f), soa(g), dtype=dt, abs_tol=tol, rel_tol=tol))
for (m, n, seed) in ((100, 240, 9), (300, 900, 31), (120, 300, 6)):
    A, b, _ = synth.dense_lasso(m, n, seed=seed, dtype=dt)
    f, g = G.lasso_functions(b, 0.1, n)
    report("dense %dx%d lasso (m <= n)" % (m, n), pogs_amd.solve_lasso(A, b, 0.1, dtype=dt), ob.oracle_solve(A, soa(f), soa(g), dtype=dt))
A, b, _ = synth.dense_lasso(900, 200, seed=23, dtype=dt)
f, g = G.lasso_functions(b, 0.1, 200)
with pogs_amd.Solver(A, dtype=dt, projector=_lib.PROJ_CGLS) as s:
    got = s.solve(f, g)
report("dense 900x200 lasso, CGLS projector", got, ob.oracle_solve(A, soa(f), soa(g), dtype=dt, use_cgls=True))
A, b, _ = synth.dense_lasso(2000, 300, seed=1, dtype=dt)
f, g = G.lasso_functions(b, 0.1, 300)
report("dense 2000x300 lasso", pogs_amd.solve_lasso(A, b, 0.1, dtype=dt), ob.oracle_solve(A, soa(f), soa(g), dtype=dt))
os.environ["POGS_AMD_XL_LIMIT"] = "16"
report("dense 2000x300 lasso, windowed passes", pogs_amd.solve_lasso(A, b, 0.1, dtype=dt), ob.oracle_solve(A, soa(f), soa(g), dtype=dt))
os.environ.pop("POGS_AMD_XL_LIMIT")
