#!/bin/bash
# round 5, GPU call 2: load-stream placement probe, the LDS-staged C3 pass (A/B + parity), Gram power telemetry
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
echo "== stream_pattern"; timeout 120 scripts/micro/bin/stream_pattern | tee gpurun_out/r05/stream_pattern.txt
for i in 1 2; do
for st in 1 0; do
  POGS_AMD_STAGE=$st timeout 600 python bench.py --config c3 --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic > gpurun_out/r05/c3_stage$st.$i.json 2> gpurun_out/r05/c3_stage$st.$i.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05/c3_stage$st.$i.json").read().strip().splitlines()[-1])
    print("c3 STAGE=$st run $i: it/s %.1f ms/step %.4f pass ms %.4f frac %.3f iters %d parity %.2e ttc %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["solve_iterations"], d["parity_vs_reference"]["rel_x"], d["time_to_converge_s"]))
except Exception as e: print("c3 stage $st failed", e); print(open("gpurun_out/r05/c3_stage$st.$i.err").read()[-1500:])
PY
done; done
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -x -k "c3 or every_streaming_shape" > gpurun_out/r05/tests2.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r05/tests2.log
echo "== gram power"; timeout 300 bash scripts/gram_power.sh r05 60 2>&1 | tail -60
