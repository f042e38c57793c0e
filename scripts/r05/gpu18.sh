#!/bin/bash
# round 5, GPU call 18: C3 with the lean pass (one dot, one accumulator until the approximate bounds ask for the exact
# residuals) in fp32 at 256 x 5, two rows per step, three workgroups per CU -- against the tree (ALU sums, C3 exception)
# and the library before this session's kernel changes; then the dense suite on the tree
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so pogs_amd/variants/*.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
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
 for v in base new lean3; do
  [ $v = new ] && cp /tmp/new.so pogs_amd/libpogs_amd.so || cp pogs_amd/variants/libpogs_amd_$v.so pogs_amd/libpogs_amd.so
  for cfg in c3; do
    timeout 600 python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic --no-secondary > gpurun_out/r05/ab18_${cfg}_${v}_$rep.json 2> gpurun_out/r05/ab18_${cfg}_${v}_$rep.err
    show gpurun_out/r05/ab18_${cfg}_${v}_$rep.json $cfg-$v-$rep
  done
 done
done
cp pogs_amd/variants/libpogs_amd_lean3.so pogs_amd/libpogs_amd.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt18 -o bench -- python $R/bench.py --config c3 --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic --no-secondary > /tmp/kt18.log 2>&1
python $R/scripts/rocpd_summary.py $(find /tmp/kt18 -name "*.db" | head -1) $R/gpurun_out/r05/c3_lean3_kernel_stats.csv; head -6 $R/gpurun_out/r05/c3_lean3_kernel_stats.csv | cut -c1-200
cd $R
cp /tmp/new.so pogs_amd/libpogs_amd.so
timeout 1800 python -m pytest tests/test_gpu_dense.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/r05/t18.log 2>&1; echo "dense + fullsize suites rc $?"; tail -3 gpurun_out/r05/t18.log | cut -c1-300
