"""Development aid: the worker script of tests/test_gpu_rccl.py run with ONE rank under torch.distributed.run (a box
with one GPU cannot run the test itself): catches anything in the worker that is not about the second rank."""
import json, os, socket, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_rccl as T
for kind, dtype in (("dense", "float32"), ("dense", "float64"), ("sparse", "float32")):
    tmp = tempfile.mkdtemp(dir="/tmp")
    script = os.path.join(tmp, "worker.py")
    open(script, "w").write(T.WORKER.format(root=ROOT, kind=kind, dtype=dtype, out=tmp))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), script], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    if p.returncode != 0:
        print(kind, dtype, "FAILED", p.stdout[-1500:], p.stderr[-1500:]); continue
    r = json.load(open(os.path.join(tmp, "rank0.json")))
    print(kind, dtype, "comm_nranks", r["comm_nranks"], "status", r["status"], "iterations", r["iterations"], "one", r["one"]["iterations"],
          "collectives", r["collectives"], "x equal", r["x"] == r["one"]["x"])
