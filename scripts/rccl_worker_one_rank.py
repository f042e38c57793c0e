"""Runs the worker of tests/test_gpu_rccl.py under torch.distributed.run with ONE rank (a 1-GPU box):
the launcher, the unique-id broadcast over torch's nccl backend, the engine's own RCCL communicator and
the result files are exercised as in the 2-rank test, which needs a second GPU.  GPU box:
    python scripts/rccl_worker_one_rank.py"""
import os, sys, subprocess, tempfile, json
sys.path.insert(0, "tests")
import test_gpu_rccl as T
out = tempfile.mkdtemp()
for kind, dtype in (("dense", "float32"), ("sparse", "float32")):
    script = os.path.join(out, "worker_%s.py" % kind)
    open(script, "w").write(T.WORKER.format(root=T.ROOT, kind=kind, dtype=dtype, out=out))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", script], capture_output=True, text=True, timeout=600, env=env)
    print(kind, "rc", p.returncode, p.stderr[-300:] if p.returncode else "")
    r = json.load(open(os.path.join(out, "rank0.json")))
    import numpy as np
    x, x1 = np.array(r["x"]), np.array(r["one"]["x"])
    print(kind, "iterations", r["iterations"], r["one"]["iterations"], "rel_x", float(np.linalg.norm(x - x1) / np.linalg.norm(x1)), "collectives", r["collectives"])
