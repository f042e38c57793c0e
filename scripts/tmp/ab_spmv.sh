#!/bin/bash
# on the GPU box: A/B of SpMV kernel variants
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/ab
cp pogs_amd/libpogs_amd.so /tmp/orig.so
for tag in "$@"; do
  cp pogs_amd/variants/libpogs_amd_$tag.so pogs_amd/libpogs_amd.so
  n=$(ls gpurun_out/ab | grep -c "^${tag}_bench")
  if [ "$n" = "0" ] && [ "$tag" != "base" ]; then
    timeout 600 python -m pytest tests/test_gpu_sparse.py -x -q > gpurun_out/ab/${tag}_tests.log 2>&1
    echo "$tag tests: $(tail -1 gpurun_out/ab/${tag}_tests.log)"
  fi
  timeout 300 python bench.py --config c4 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/ab/${tag}_bench$n.json 2> gpurun_out/ab/${tag}_bench$n.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab/${tag}_bench$n.json").read().strip().splitlines()[-1])
    print("$tag", "it/s %.1f"%d["value"], "spmv_ms %.4f"%d["roofline"]["avg_launch_ms"], "iters", d.get("solve_iterations"), "rel_x", d["parity_vs_reference"]["rel_x"])
except Exception as e:
    print("$tag FAILED", e)
PY
done
cp /tmp/orig.so pogs_amd/libpogs_amd.so
