#!/bin/bash
# round 5, GPU call 20: (1) which library fails test_wide_matrix_with_many_columns[float32-shape2]; (2) timing-only diagnostics of
# C3's pass: row functor without its stores / without its loads (wrong results, the kernel time is what is read), and the
# rows dealt to the workgroups in contiguous ranges instead of round-robin (valid results)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so pogs_amd/variants/*.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
cp pogs_amd/libpogs_amd.so /tmp/new.so
for v in base new lean3; do
  [ $v = new ] && cp /tmp/new.so pogs_amd/libpogs_amd.so || cp pogs_amd/variants/libpogs_amd_$v.so pogs_amd/libpogs_amd.so
  timeout 600 python -m pytest tests/test_gpu_dense.py -m gpu -q -x -k "test_wide_matrix_with_many_columns" > gpurun_out/r05/t20_$v.log 2>&1; echo "wide test on $v: rc $?"; grep "AssertionError: assert\|passed\|failed" gpurun_out/r05/t20_$v.log | tail -2
done
show() {
python - <<PY
import json
try:
    d=json.loads(open("$1").read().strip().splitlines()[-1])
    print("$2: it/s %.1f ms/step %.4f kernel ms %.4f frac %.3f iters %s relx %.3e" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d.get("solve_iterations"), d["parity_vs_reference"]["rel_x"]))
except Exception as e: print("$2 failed", e); print(open("$1".replace(".json",".err")).read()[-600:])
PY
}
for rep in 1 2; do
 for v in new blocked nostore noload; do
  [ $v = new ] && cp /tmp/new.so pogs_amd/libpogs_amd.so || cp pogs_amd/variants/libpogs_amd_$v.so pogs_amd/libpogs_amd.so
  for cfg in c3 c2; do
    timeout 600 python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic --no-secondary > gpurun_out/r05/ab20_${cfg}_${v}_$rep.json 2> gpurun_out/r05/ab20_${cfg}_${v}_$rep.err
    show gpurun_out/r05/ab20_${cfg}_${v}_$rep.json $cfg-$v-$rep
  done
 done
done
cp /tmp/new.so pogs_amd/libpogs_amd.so
