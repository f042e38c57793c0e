"""Full-size agreement of the engine with the compiled reference (development aid; the test is
tests/test_gpu_fullsize.py::test_c2_solution_matches_compiled_reference).

    python scripts/parity_fullsize.py c2|c3 [rows] [cols]

Generates the configuration on the GPU, hands the SAME (A, b, lambda) to
oracle/_ref/libpogs_cpu.so (clean subprocess, all host cores) and to the engine
(defaults, then POGS_AMD_SK_FULL=1 POGS_AMD_GRAM=fp32), prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle_binding as ob  # noqa: E402
import pogs_amd  # noqa: E402
from pogs_amd import graph as G  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
m = int(sys.argv[2]) if len(sys.argv) > 2 else (100000 if cfg == "c2" else 200000)
n = int(sys.argv[3]) if len(sys.argv) > 3 else (10000 if cfg == "c2" else 5000)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float32)
if cfg == "c2":
    xt = torch.randn(n, generator=g, device=dev) * (torch.rand(n, generator=g, device=dev) < 0.1)
    b = (A @ xt + 0.1 * torch.randn(m, generator=g, device=dev)).double().cpu().numpy()
    f, gg = G.lasso_functions(b, 0.1, n)
else:
    w = torch.randn(n, generator=g, device=dev) * (torch.rand(n, generator=g, device=dev) < 0.3)
    w = w * (2.0 / torch.sqrt((w * w).sum()))
    p = torch.sigmoid(A @ w)
    lab = (2.0 * (torch.rand(m, generator=g, device=dev) < p) - 1.0).double().cpu().numpy()
    f, gg = G.logistic_functions(lab, 0.01, n)
torch.cuda.synchronize()
t0 = time.time()
A_host = A.cpu().numpy()
fs = {k: getattr(f, k) for k in "habcde"}
gs = {k: getattr(gg, k) for k in "habcde"}
run = ob.ref_start(A_host, fs, gs, dtype=np.float32, verbose=1)
t_start = time.time() - t0
out = {"config": cfg, "m": m, "n": n, "ref_launch_s": t_start}


def engine(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with pogs_amd.Solver(A.data_ptr(), dtype=np.float32, shape=(m, n), device_ptr=True) as s:
            r = s.solve(f, gg)
            st = s.stats()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return r, st


runs = {"default": engine({}), "sk_full_gram_fp32": engine({"POGS_AMD_SK_FULL": "1", "POGS_AMD_GRAM": "fp32"})}
ref = run.finish(timeout=1500)
out["ref"] = {k: ref.get(k) for k in ("status", "iterations", "optval", "wall_s", "t_total", "t_init", "elapsed_s")}
xr = ref["x"].astype(np.float64)
for name, (r, st) in runs.items():
    x = r["x"].astype(np.float64)
    out[name] = {"status": r["status"], "iterations": r["iterations"], "optval": r["optval"],
                 "rel_x": float(np.linalg.norm(x - xr) / np.linalg.norm(xr)),
                 "rel_optval": abs(r["optval"] - ref["optval"]) / abs(ref["optval"]),
                 "rel_y": float(np.linalg.norm(r["y"].astype(np.float64) - ref["y"]) / np.linalg.norm(ref["y"])),
                 "rel_l": float(np.linalg.norm(r["l"].astype(np.float64) - ref["l"]) / np.linalg.norm(ref["l"])),
                 "init_s": st["t_init_s"], "loop_s": st["t_loop_s"]}
xd = runs["default"][0]["x"].astype(np.float64)
xf = runs["sk_full_gram_fp32"][0]["x"].astype(np.float64)
out["default_vs_full"] = float(np.linalg.norm(xd - xf) / np.linalg.norm(xf))
print(json.dumps(out))
