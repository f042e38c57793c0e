#!/bin/bash
# round 5, GPU call 9: C3 with one row per step -- four workgroups per CU (128 VGPRs, a few spilled) and three -- against the shipped two rows / three workgroups
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
cp pogs_amd/libpogs_amd.so /tmp/orig.so
for v in c3R3 base c3R4 c3R3 base c3R4; do
  [ $v = base ] && cp /tmp/orig.so pogs_amd/libpogs_amd.so || cp pogs_amd/variants/libpogs_amd_$v.so pogs_amd/libpogs_amd.so
  timeout 600 python bench.py --config c3 --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic > gpurun_out/r05/c3_$v.json 2> gpurun_out/r05/c3_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05/c3_$v.json").read().strip().splitlines()[-1])
    print("c3 $v: it/s %.1f ms/step %.4f pass ms %.4f frac %.3f iters %d parity %.3e ttc %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["solve_iterations"], d["parity_vs_reference"]["rel_x"], d["time_to_converge_s"]))
except Exception as e: print("c3 $v failed", e); print(open("gpurun_out/r05/c3_$v.err").read()[-1200:])
PY
done
cp /tmp/orig.so pogs_amd/libpogs_amd.so
