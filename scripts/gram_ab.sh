#!/bin/bash
# A/B of a Gram-kernel switch on one box, alternating: gram_ab.sh VAR   (VAR=1 against VAR unset)
V=${1:-POGS_AMD_GRAM_NOSYNC}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
  for mode in 0 1; do
    env $V=$mode python - <<'PY'
import os, numpy as np, torch, pogs_amd
m,n=100000,10000
A=torch.randn((m,n),device="cuda",dtype=torch.float32)
g=[]
for i in range(3):
    s=pogs_amd.Solver(A.data_ptr(),dtype=np.float32,shape=(m,n),device_ptr=True)
    g.append(s.stats()["gram_ms"]); s.close()
print({k:v for k,v in os.environ.items() if k.startswith("POGS_AMD_GRAM")}, "gram_ms", ["%.2f"%v for v in g])
PY
  done
done
