import os, sys, time
os.environ["POGS_AMD_TRACE"]="1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import pogs_amd
from pogs_amd import synth
m,n=100000,10000
dev=torch.device("cuda:0")
g=torch.Generator(device=dev); g.manual_seed(1)
A=torch.randn((m,n),generator=g,device=dev,dtype=torch.float32)
torch.cuda.synchronize()
for i in range(3):
    print("---- create", i, flush=True)
    t0=time.time()
    s=pogs_amd.Solver(A.data_ptr(), dtype=np.float32, shape=(m,n), device_ptr=True, device=0)
    print("create s", time.time()-t0, flush=True)
    s.close()
