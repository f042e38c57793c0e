"""C4: the loop of a cold solve() against the same iterations through iterate() (development aid, round 5)."""
import os, sys, time
os.environ.setdefault("POGS_AMD_TORCH_PRELOAD", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pogs_amd
from pogs_amd import synth, graph as G
A, b, _ = synth.csr_lasso(2000000, 500000, 50, seed=4, dtype=np.float32)
f, g = G.lasso_functions(b, 0.1, 500000)
with pogs_amd.Solver(A, dtype=np.float32) as s:
    for rep in range(2):
        t0 = time.time(); r = s.solve(f, g); t1 = time.time()
        st = s.stats()
        print("solve %d: wall %.4f s, loop %.4f s, iterations %d, cg %d, spmv %d, exact %d -> %.4f ms / iteration" % (
            rep, t1 - t0, st["t_loop_s"], st["iterations"], st["cg_iters"], st["matvecs"], st["exact_iters"], 1e3 * st["t_loop_s"] / st["iterations"]))
    s.begin_run(f, g)
    s.reset_stats()
    for chunk in (20, 100, 120, 119):
        sec, solves = s.iterate(chunk)
        st = s.stats()
        print("iterate(%d): %.4f s = %.4f ms / iteration; cumulative cg %d spmv %d, solves completed %d" % (chunk, sec, 1e3 * sec / chunk, st["cg_iters"], st["matvecs"], solves))
