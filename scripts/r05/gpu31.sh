#!/bin/bash
# round 5, GPU call 31: sell_fill_kernel as a wavefront per row (no cursor): the sparse suite (bitwise storage tests, unsorted /
# duplicate inputs), C4's setup and time to converge against the library before this session (base), kernel summary of the setup
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so pogs_amd/variants/*.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
timeout 1800 python -m pytest tests/test_gpu_sparse.py -m gpu -q > gpurun_out/r05/t31.log 2>&1; echo "sparse suite rc $?"; grep "passed\|failed" gpurun_out/r05/t31.log | tail -2
show() {
python - <<PY
import json
try:
    d=json.loads(open("$1").read().strip().splitlines()[-1])
    print("$2: it/s %.1f kernel ms %.4f iters %s relx %.3e ttc %.4f init %.4f cycles max ttc %.4f" % (d["value"], d["roofline"]["avg_launch_ms"], d.get("solve_iterations"), d["parity_vs_reference"]["rel_x"], d["time_to_converge_s"], d["init_s"], d["handle_cycles"]["max_time_to_converge_s"]))
except Exception as e: print("$2 failed", e); print(open("$1".replace(".json",".err")).read()[-600:])
PY
}
cp pogs_amd/libpogs_amd.so /tmp/new.so
for rep in 1 2; do
 for v in base new; do
  [ $v = new ] && cp /tmp/new.so pogs_amd/libpogs_amd.so || cp pogs_amd/variants/libpogs_amd_$v.so pogs_amd/libpogs_amd.so
  timeout 600 python bench.py --config c4 --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic --no-secondary > gpurun_out/r05/ab31_c4_${v}_$rep.json 2> gpurun_out/r05/ab31_c4_${v}_$rep.err
  show gpurun_out/r05/ab31_c4_${v}_$rep.json c4-$v-$rep
 done
done
cp /tmp/new.so pogs_amd/libpogs_amd.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt31 -o bench -- python $R/bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --no-secondary > /tmp/kt31.log 2>&1
python $R/scripts/rocpd_summary.py $(find /tmp/kt31 -name "*.db" | head -1) $R/gpurun_out/r05/c4_setup_kernel_stats.csv > /dev/null; grep "sell_fill\|fill_transpose\|sort_segments\|sell_count\|count_cols\|sell_plan\|refill\|scale_csr\|fillBuffer" $R/gpurun_out/r05/c4_setup_kernel_stats.csv | cut -c1-60,200-
