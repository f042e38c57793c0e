#!/bin/bash
# round 5, GPU call 34: segment sort of the transposed copy by rank for short segments: sparse suite, C4 setup, then the full suite + smoke
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_sparse.py -m gpu -q -x > gpurun_out/r05/t34.log 2>&1; echo "sparse suite rc $?"; grep "passed\|failed" gpurun_out/r05/t34.log | tail -2
for rep in 1 2; do
timeout 600 python bench.py --config c4 --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic --no-secondary > gpurun_out/r05/c4_34_$rep.json 2> gpurun_out/r05/c4_34_$rep.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r05/c4_34_$rep.json").read().strip().splitlines()[-1])
print("c4: it/s %.1f iters %s relx %.3e ttc %.4f init %.4f" % (d["value"], d.get("solve_iterations"), d["parity_vs_reference"]["rel_x"], d["time_to_converge_s"], d["init_s"]))
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt34 -o bench -- python $R/bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --no-secondary > /tmp/kt34.log 2>&1
python $R/scripts/rocpd_summary.py $(find /tmp/kt34 -name "*.db" | head -1) $R/gpurun_out/r05/c4_setup34_kernel_stats.csv > /dev/null; grep "sort_segments\|fill_transpose\|count_cols" $R/gpurun_out/r05/c4_setup34_kernel_stats.csv | awk -F'",' '{print substr($1,1,40), $2}'
cd $R
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05/tests34.log 2>&1; echo "full suite rc $?"; grep "passed\|failed" gpurun_out/r05/tests34.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
