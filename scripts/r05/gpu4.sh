#!/bin/bash
# round 5, GPU call 4: what a tile boundary's x slice costs the load stream (micro), tile-shape variants of the SpMV, tests of the split dense solver
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
echo "== stream_pattern"; timeout 120 scripts/micro/bin/stream_pattern | tee gpurun_out/r05/stream_pattern2.txt
echo "== variants"; bash scripts/spmv_probe.sh base tall base tall 2>&1 | grep "^=="
cp pogs_amd/variants/libpogs_amd_tall.so /tmp/tall.so
for v in base tall; do
  cp pogs_amd/libpogs_amd.so /tmp/orig.so; cp pogs_amd/variants/libpogs_amd_$v.so pogs_amd/libpogs_amd.so
  timeout 600 python bench.py --config c4 --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic > gpurun_out/r05/c4_$v.json 2> gpurun_out/r05/c4_$v.err
  cp /tmp/orig.so pogs_amd/libpogs_amd.so
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05/c4_$v.json").read().strip().splitlines()[-1])
    print("c4 $v: it/s %.1f ms/step %.4f spmv ms %.4f frac %.3f iter frac %.3f iters %d parity %.2e ttc %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["iteration"]["frac"], d["solve_iterations"], d["parity_vs_reference"]["rel_x"], d["time_to_converge_s"]))
except Exception as e: print("c4 $v failed", e); print(open("gpurun_out/r05/c4_$v.err").read()[-1500:])
PY
done
timeout 1500 python -m pytest tests/test_gpu_dense.py tests/test_gpu_boundary.py tests/test_gpu_pool.py -q -x > gpurun_out/r05/tests4.log 2>&1; echo "dense suites rc $?"; tail -4 gpurun_out/r05/tests4.log
