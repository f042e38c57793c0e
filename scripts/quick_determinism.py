"""Race screen (development aid): repeated C2 set-ups and solves must agree bit for bit."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import pogs_amd
from pogs_amd import graph as G
import bench

m, n = 100000, 10000
dev = torch.device("cuda:0")
A, b = bench.make_problem(m, n, 0, dev)
f, g = G.lasso_functions(b, 0.1, n)
ref = None
for rep in range(6):
    with pogs_amd.Solver(A.data_ptr(), dtype=np.float32, shape=(m, n), device_ptr=True) as s:
        r = s.solve(f, g)
    key = (r["iterations"], r["x"].tobytes(), r["y"].tobytes())
    if ref is None:
        ref = key
    print(rep, r["iterations"], "identical" if key == ref else "DIFFERENT", flush=True)
