#!/bin/bash
# round 5, GPU call 5: equal-width column blocks -- sparse suite, C4 bench twice, stamps
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
echo skip
for v in eq base eq; do
  cp pogs_amd/libpogs_amd.so /tmp/orig.so
  [ $v = base ] && cp pogs_amd/variants/libpogs_amd_base.so pogs_amd/libpogs_amd.so
  POGS_AMD_SELL_STAMPS=1 timeout 600 python bench.py --config c4 --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic > gpurun_out/r05/c4_$v.json 2> gpurun_out/r05/c4_$v.err
  cp /tmp/orig.so pogs_amd/libpogs_amd.so
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05/c4_$v.json").read().strip().splitlines()[-1])
    print("c4 $v: it/s %.1f ms/step %.4f spmv ms %.4f frac %.3f iter frac %.3f iters %d parity %.2e ttc %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["iteration"]["frac"], d["solve_iterations"], d["parity_vs_reference"]["rel_x"], d["time_to_converge_s"]))
except Exception as e: print("c4 $v failed", e); print(open("gpurun_out/r05/c4_$v.err").read()[-1500:])
PY
  grep -h "stamps\]" gpurun_out/r05/c4_$v.err | head -2
done
timeout 900 python -m pytest tests/test_gpu_sparse.py -q -x > gpurun_out/r05/tests5c.log 2>&1; echo "sparse rc $?"; tail -3 gpurun_out/r05/tests5c.log
