#!/bin/bash
# round 5, GPU call 26: the collection on the final sources of the second session -- full -m gpu suite, profile_round per workload, the default bench run
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r05/tests26.log 2>&1; echo "full suite rc $?"; tail -4 gpurun_out/r05/tests26.log | cut -c1-300
for cfg in c2 c3 c4 c2f64; do
  timeout 1200 bash scripts/profile_round.sh r05 $cfg > gpurun_out/r05/prof_$cfg.log 2>&1
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/prof_r05_$cfg/bench.json").read().strip().splitlines()[-1])
    cb=d.get("cpu_baseline") or {}
    print("$cfg: it/s %.1f ms/step %.4f kernel ms %.4f frac %.3f iter frac %.3f ttc %.4f traffic %s cpu %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["iteration"]["frac"], d["time_to_converge_s"], d["roofline"].get("traffic"), cb.get("value")))
except Exception as e: print("$cfg failed", e)
PY
done
timeout 1200 python bench.py > gpurun_out/r05/bench_final26.json 2> gpurun_out/r05/bench_final26.err; echo "bench rc $?"; tail -c 900 gpurun_out/r05/bench_final26.json
