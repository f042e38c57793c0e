#!/bin/bash
# round 5, GPU call 15: wavefront sums in the ALU + unconditional loads in the row-streaming kernels -- the dense suite, then A / B
# against the previous library (pogs_amd/variants/libpogs_amd_base.so) on c2 / c3 / c2f64, alternating on this box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so pogs_amd/variants/libpogs_amd_base.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
timeout 600 python -m pytest tests/test_gpu_dense.py -m gpu -q -x -k "wavefront_sum" 2>&1 | tail -3
show() {
python - <<PY
import json
try:
    d=json.loads(open("$1").read().strip().splitlines()[-1])
    print("$2: it/s %.1f ms/step %.4f kernel ms %.4f frac %.3f iter frac %.3f iters %s relx %.3e ttc %.4f init %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["iteration"]["frac"], d.get("solve_iterations"), d["parity_vs_reference"]["rel_x"], d["time_to_converge_s"], d.get("init_s", -1)))
except Exception as e: print("$2 failed", e); print(open("$1".replace(".json",".err")).read()[-800:])
PY
}
cp pogs_amd/libpogs_amd.so /tmp/new.so
for rep in 1 2; do
 for v in base new; do
  [ $v = base ] && cp pogs_amd/variants/libpogs_amd_base.so pogs_amd/libpogs_amd.so || cp /tmp/new.so pogs_amd/libpogs_amd.so
  for cfg in c2 c3 c2f64; do
    timeout 600 python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic --no-secondary > gpurun_out/r05/ab15_${cfg}_${v}_$rep.json 2> gpurun_out/r05/ab15_${cfg}_${v}_$rep.err
    show gpurun_out/r05/ab15_${cfg}_${v}_$rep.json $cfg-$v-$rep
  done
 done
done
cp /tmp/new.so pogs_amd/libpogs_amd.so
timeout 1800 python -m pytest tests/test_gpu_dense.py -m gpu -q -x > gpurun_out/r05/t15.log 2>&1; echo "dense suite rc $?"; tail -3 gpurun_out/r05/t15.log | cut -c1-300
