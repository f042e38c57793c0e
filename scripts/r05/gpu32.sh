#!/bin/bash
# round 5, GPU call 32: C4's profile and the default bench run on the round's final sources (after the sparse setup change)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
timeout 1200 bash scripts/profile_round.sh r05 c4 > gpurun_out/r05/prof_c4.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/prof_r05_c4/bench.json").read().strip().splitlines()[-1])
cb=d.get("cpu_baseline") or {}
print("c4: it/s %.1f ms/step %.4f kernel ms %.4f frac %.3f iter frac %.3f ttc %.4f init %.4f traffic %s cpu %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["iteration"]["frac"], d["time_to_converge_s"], d["init_s"], d["roofline"].get("traffic"), cb.get("value")))
PY
SECONDS=0
timeout 1200 python bench.py > gpurun_out/r05/bench_final32.json 2> gpurun_out/r05/bench_final32.err; echo "default bench rc $? in $SECONDS s"; tail -c 700 gpurun_out/r05/bench_final32.json
SECONDS=0
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_driver_cmd32.json 2> gpurun_out/r05/bench_driver_cmd32.err; echo "driver command rc $? in $SECONDS s"
