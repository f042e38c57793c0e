#!/bin/bash
# round 5, GPU call 14 (second session): the sparse suite on the two-slot storage, C4 profile on it, and the prox share of C3's pass
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_sparse.py -m gpu -q -x > gpurun_out/r05/t14.log 2>&1; echo "sparse suite rc $?"; tail -3 gpurun_out/r05/t14.log | cut -c1-300
timeout 1200 bash scripts/profile_round.sh r05 c4 --no-cpu > gpurun_out/r05/prof_c4.log 2>&1
show() {
python - <<PY
import json
try:
    d=json.loads(open("$1").read().strip().splitlines()[-1])
    print("$2: it/s %.1f ms/step %.4f kernel ms %.4f frac %.3f iter frac %.3f iters %s ttc %.4f traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["iteration"]["frac"], d.get("solve_iterations"), d["time_to_converge_s"], d["roofline"].get("traffic")))
except Exception as e: print("$2 failed", e)
PY
}
show gpurun_out/prof_r05_c4/bench.json c4
for rep in 1 2; do
  timeout 600 python bench.py --config c3 --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic --no-secondary > gpurun_out/r05/c3_logistic_$rep.json 2> gpurun_out/r05/c3_logistic_$rep.err
  show gpurun_out/r05/c3_logistic_$rep.json c3-logistic-$rep
  timeout 600 python bench.py --config c2 --m 200000 --n 5000 --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic --no-secondary > gpurun_out/r05/c3_lasso_$rep.json 2> gpurun_out/r05/c3_lasso_$rep.err
  show gpurun_out/r05/c3_lasso_$rep.json c3shape-lasso-$rep
done
