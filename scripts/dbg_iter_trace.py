"""Per-iteration residual trace of the sixteen-decade fp32 test problem (tests/test_gpu_dense.py::
test_degenerate_inputs_fp32_with_equilibration_shortcut): the engine's one-pass iteration, its
three-pass iteration (POGS_AMD_FUSED=0) and the CPU oracle side by side, printed at full precision
(POGS_AMD_ITERTRACE / POGS_ORACLE_ITERTRACE).  Shows that both engine paths carry the same fp32
noise in the dual-residual bound (1e-4 .. 2e-3 relative to the oracle) and where the rho schedules
part (one threshold comparison near iteration 630).  GPU box: python scripts/dbg_iter_trace.py"""
import os, sys, subprocess
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
def problem():
    import pogs_amd as pogs
    rng = np.random.default_rng(0)
    A = rng.standard_normal((600, 300))
    rng.standard_normal((200, 900)); rng.standard_normal(800); rng.standard_normal(250)
    M = A * np.exp(rng.uniform(-8, 8, (600, 1))) * np.exp(rng.uniform(-8, 8, (1, 300)))
    for s in (50, 20, 600, 600, 200, 800, 600): rng.standard_normal(s)
    b = rng.standard_normal(600)
    f, g = pogs.graph.lasso_functions(b, 0.1, 300)
    return pogs, M, f, g
if len(sys.argv) > 1:
    pogs, M, f, g = problem()
    if sys.argv[1] == "oracle":
        import oracle_binding as ob
        soa = lambda fv: {k: getattr(fv, k) for k in "habcde"}
        ob.oracle_solve(M, soa(f), soa(g), dtype=np.float32, max_iter=2500)
    else:
        pogs.graph._solve_graph_form(M, f, g, 1e-4, 1e-4, 2500, 0, 1.0, dtype=np.float32)
else:
    outs = {}
    for tag, env in [("fused1", {"POGS_AMD_ITERTRACE": "1"}), ("fused0", {"POGS_AMD_FUSED": "0", "POGS_AMD_ITERTRACE": "1"}),
                     ("oracle", {"POGS_ORACLE_ITERTRACE": "1"})]:
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, tag], env=e, capture_output=True, text=True)
        outs[tag] = [l.split() for l in r.stdout.splitlines() if l.startswith("T ")]
        if not outs[tag]: print(tag, r.stderr[-500:])
    n = min(len(v) for v in outs.values())
    shown = 0
    for i in range(n):
        a, b, c = outs["fused1"][i], outs["fused0"][i], outs["oracle"][i]
        rs = [abs(float(a[7]) / float(c[7]) - 1), abs(float(b[7]) / float(c[7]) - 1)]
        rr = [abs(float(a[5]) / float(c[5]) - 1), abs(float(b[5]) / float(c[5]) - 1)]
        if i < 12 or i % 20 == 0 or (145 <= i <= 175):
            print("%4d rho %s/%s/%s  s: fused %.1e unfused %.1e   r: fused %.1e unfused %.1e  s=%s" % (i, a[3], b[3], c[3], rs[0], rs[1], rr[0], rr[1], c[7]))
