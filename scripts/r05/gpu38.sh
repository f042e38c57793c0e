#!/bin/bash
# round 5, GPU call 38: ring depth of the two-slot SpMV (batches in flight per wavefront): 6 / 8 (shipped) / 10 / 12 at C4
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so pogs_amd/variants/*.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
cp pogs_amd/libpogs_amd.so /tmp/new.so
for rep in 1 2; do
 for v in nb6 new nb10 nb12; do
  [ $v = new ] && cp /tmp/new.so pogs_amd/libpogs_amd.so || cp pogs_amd/variants/libpogs_amd_$v.so pogs_amd/libpogs_amd.so
  timeout 300 python bench.py --config c4 --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic --no-secondary > gpurun_out/r05/ab38_c4_${v}_$rep.json 2> gpurun_out/r05/ab38_c4_${v}_$rep.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05/ab38_c4_${v}_$rep.json").read().strip().splitlines()[-1])
    print("c4-$v-$rep: it/s %.1f kernel ms %.4f iters %s relx %.3e" % (d["value"], d["roofline"]["avg_launch_ms"], d.get("solve_iterations"), d["parity_vs_reference"]["rel_x"]))
except Exception as e: print("c4-$v-$rep failed", e)
PY
 done
done
cp /tmp/new.so pogs_amd/libpogs_amd.so
