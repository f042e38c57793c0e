"""How far apart two CORRECT runs of the reference lie in fp32 at the default tolerances
(abs_tol = rel_tol = 1e-4): the compiled reference (oracle/_ref) against itself in fp64, and with a
different BLAS thread count (another summation order), on the problems whose GPU parity tests use
a tolerance above 1e-4.  DESIGN.md section 5 quotes these numbers; run in the build container.

    python scripts/fp32_spread.py"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle_binding as ob  # noqa: E402
from pogs_amd import graph as G  # noqa: E402
from pogs_amd import synth  # noqa: E402


def soa(fv):
    return {k: getattr(fv, k) for k in "habcde"}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def spread(name, A, f, g, use_oracle_cgls=False):
    out = {"problem": name}
    r32 = ob.ref_solve(A, soa(f), soa(g), dtype=np.float32, threads=1)
    r32b = ob.ref_solve(A, soa(f), soa(g), dtype=np.float32, threads=8)
    r64 = ob.ref_solve(A, soa(f), soa(g), dtype=np.float64, threads=8)
    o32 = ob.oracle_solve(A, soa(f), soa(g), dtype=np.float32, use_cgls=use_oracle_cgls)
    out.update(iters_f32=r32["iterations"], iters_f32_8thr=r32b["iterations"], iters_f64=r64["iterations"],
               ref32_vs_ref64_x=rel(r32["x"], r64["x"]), ref32_1thr_vs_8thr_x=rel(r32["x"], r32b["x"]),
               oracle32_vs_ref32_x=rel(o32["x"], r32["x"]), oracle32_vs_ref64_x=rel(o32["x"], r64["x"]),
               ref32_vs_ref64_optval=abs(r32["optval"] - r64["optval"]) / abs(r64["optval"]))
    print(json.dumps(out), flush=True)


A, b, _ = synth.csr_lasso(20000, 5000, 50, seed=3, dtype=np.float32)
f, g = G.lasso_functions(b, 0.1, 5000)
spread("csr 20000x5000 fp32 lasso (scaled C4)", A, f, g)
A, b, _ = synth.dense_lasso(100, 240, seed=9, dtype=np.float32)
f, g = G.lasso_functions(b, 0.1, 240)
spread("dense 100x240 lasso (m <= n)", A, f, g)
A, b, _ = synth.dense_lasso(2000, 300, seed=1, dtype=np.float32)
f, g = G.lasso_functions(b, 0.1, 300)
spread("dense 2000x300 lasso", A, f, g)
A, b, _ = synth.dense_lasso(900, 200, seed=23, dtype=np.float32)
f, g = G.lasso_functions(b, 0.1, 200)
spread("dense 900x200 lasso", A, f, g)
