#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O; rm -f $O/gram_ablate.log
cd $R
for ab in ${ABLIST:-0 1 2 3}; do
  echo "== POGS_AMD_GRAM_ABLATE=$ab" >> $O/gram_ablate.log
  POGS_AMD_GRAM_ABLATE=$ab python - >> $O/gram_ablate.log 2>&1 <<'PY'
import numpy as np, torch, pogs_amd
m,n=100000,10000
A=torch.randn((m,n),device="cuda",dtype=torch.float32)
for i in range(3):
    s=pogs_amd.Solver(A.data_ptr(),dtype=np.float32,shape=(m,n),device_ptr=True)
    st=s.stats(); s.close()
    print("gram_ms %.2f"%st["gram_ms"])
PY
done
cat $O/gram_ablate.log
