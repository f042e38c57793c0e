#!/bin/bash
# round 5, GPU call 6: sparse suite on the reverted SpMV + uniform-coefficient prox; kernel trace with iteration gaps (c2, c4)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_sparse.py -q -x > gpurun_out/r05/tests6.log 2>&1; echo "sparse suite rc $?"; tail -3 gpurun_out/r05/tests6.log
for cfg in c2 c4; do
  timeout 900 bash scripts/profile_round.sh r05 $cfg --no-cpu > gpurun_out/r05/prof_$cfg.log 2>&1
  echo "== $cfg"; head -12 gpurun_out/prof_r05_$cfg/kernel_stats.csv | cut -c1-150; cat gpurun_out/prof_r05_$cfg/iter_gaps.txt | head -14
  python - <<PY
import json
d=json.loads(open("gpurun_out/prof_r05_$cfg/bench.json").read().strip().splitlines()[-1])
print("$cfg: it/s %.1f ms/step %.4f kernel ms %.4f frac %.3f iter frac %.3f ttc %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["iteration"]["frac"], d["time_to_converge_s"]))
PY
done
