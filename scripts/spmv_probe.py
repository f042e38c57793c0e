"""Timing probe: SpMVs of the C4 matrix through PogsAmdMul (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
import numpy as np
os.environ.setdefault("POGS_AMD_TORCH_PRELOAD", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pogs_amd
from pogs_amd import synth
A, b, _ = synth.csr_lasso(2000000, 500000, 50, seed=4, dtype=np.float32)
rng = np.random.default_rng(0)
x = rng.standard_normal(500000).astype(np.float32); y = rng.standard_normal(2000000).astype(np.float32)
with pogs_amd.Solver(A, dtype=np.float32) as s:
    for _ in range(12):
        s.mul("n", 1.0, x, 0.0, y)
        s.mul("t", 1.0, y, 0.0, x)
print("done")
