#!/bin/bash
# round 5, GPU call 39: tile shape of the two-slot SpMV at C4: 24576 columns x 12288 rows (wide), 15360 x 24576 (tall), 18432 x 16384 (shipped)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so pogs_amd/variants/*.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
cp pogs_amd/libpogs_amd.so /tmp/new.so
for rep in 1 2; do
 for v in wide new tall; do
  [ $v = new ] && cp /tmp/new.so pogs_amd/libpogs_amd.so || cp pogs_amd/variants/libpogs_amd_$v.so pogs_amd/libpogs_amd.so
  timeout 300 python bench.py --config c4 --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic --no-secondary > gpurun_out/r05/ab39_c4_${v}_$rep.json 2> gpurun_out/r05/ab39_c4_${v}_$rep.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05/ab39_c4_${v}_$rep.json").read().strip().splitlines()[-1])
    print("c4-$v-$rep: it/s %.1f kernel ms %.4f iters %s relx %.3e" % (d["value"], d["roofline"]["avg_launch_ms"], d.get("solve_iterations"), d["parity_vs_reference"]["rel_x"]))
except Exception as e: print("c4-$v-$rep failed", e)
PY
 done
done
cp /tmp/new.so pogs_amd/libpogs_amd.so
