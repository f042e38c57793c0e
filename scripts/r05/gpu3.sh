#!/bin/bash
# round 5, GPU call 3: the SpMV with producer wavefronts -- sparse suite, full-size C4 parity, C4 bench (twice), stamps
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_sparse.py -q -x > gpurun_out/r05/tests3a.log 2>&1; echo "sparse suite rc $?"; tail -3 gpurun_out/r05/tests3a.log
for i in 1 2; do
  POGS_AMD_SELL_STAMPS=$((i==1)) timeout 600 python bench.py --config c4 --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic > gpurun_out/r05/c4_prod.$i.json 2> gpurun_out/r05/c4_prod.$i.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05/c4_prod.$i.json").read().strip().splitlines()[-1])
    print("c4 run $i: it/s %.1f ms/step %.4f spmv ms %.4f frac %.3f iter frac %.3f iters %d parity %.2e ttc %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["iteration"]["frac"], d["solve_iterations"], d["parity_vs_reference"]["rel_x"], d["time_to_converge_s"]))
except Exception as e: print("c4 failed", e); print(open("gpurun_out/r05/c4_prod.$i.err").read()[-1500:])
PY
done
grep -h "stamps\]" gpurun_out/r05/c4_prod.1.err | head -4
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -k "c4 or wide_sparse or bitwise or sparse_solve_families" > gpurun_out/r05/tests3b.log 2>&1; echo "fullsize rc $?"; tail -4 gpurun_out/r05/tests3b.log
