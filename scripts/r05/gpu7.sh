#!/bin/bash
# round 5, GPU call 7: look-ahead Cholesky -- bit test, A/B of the setup at c2 / c3 / c2f64
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_dense.py -q -x -k "lookahead or projection_kkt or ill_conditioned or solve_families" > gpurun_out/r05/tests7.log 2>&1; echo "chol tests rc $?"; tail -3 gpurun_out/r05/tests7.log
for cfg in c2 c2f64 c3; do
for la in 1 0 1 0; do
  POGS_AMD_CHOL_AHEAD=$la timeout 600 python bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-live-traffic > gpurun_out/r05/chol_${cfg}_$la.json 2> gpurun_out/r05/chol_${cfg}_$la.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05/chol_${cfg}_$la.json").read().strip().splitlines()[-1])
    print("$cfg AHEAD=$la: chol_ms %.2f trtri %.2f gram %.2f init_s %.4f ttc %.4f (cycles max %.4f) it/s %.1f iters %d parity %.3e" % (d["setup_ms"]["chol_ms"], d["setup_ms"]["trtri_ms"], d["setup_ms"]["gram_ms"], d["init_s"], d["time_to_converge_s"], d["handle_cycles"]["max_time_to_converge_s"], d["value"], d["solve_iterations"], d["parity_vs_reference"]["rel_x"]))
except Exception as e: print("$cfg $la failed", e); print(open("gpurun_out/r05/chol_${cfg}_$la.err").read()[-1500:])
PY
done; done
