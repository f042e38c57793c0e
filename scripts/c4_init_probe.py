import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
import pogs_amd
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["c4"]
A, b, _ = bench.make_problem(cfg, cfg["m"], cfg["n"], 0, dev)
data = torch.from_numpy(np.ascontiguousarray(A.data, np.float32)).to(dev)
ptr = torch.from_numpy(np.ascontiguousarray(A.indptr, np.int32)).to(dev)
ind = torch.from_numpy(np.ascontiguousarray(A.indices, np.int32)).to(dev)
torch.cuda.synchronize()
for rep in range(3):
    for mode in ("host", "device"):
        t0 = time.time()
        if mode == "host":
            s = pogs_amd.Solver(A, dtype=np.float32)
        else:
            s = pogs_amd.Solver((data.data_ptr(), ptr.data_ptr(), ind.data_ptr(), A.nnz), dtype=np.float32, shape=A.shape, device_ptr=True)
        t = time.time() - t0
        st = s.stats()
        print(mode, "create %.4f s  t_h2d %.4f  t_init %.4f  equil %.1f ms normest %.1f ms" % (t, st["t_h2d_s"], st["t_init_s"], st["equil_ms"], st["normest_ms"]), flush=True)
        s.close()
