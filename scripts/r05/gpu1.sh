#!/bin/bash
# round 5, GPU call 1: full -m gpu suite on the new sources, per-XCD stamps of the C4 SpMVs (uniform and auto-weighted), default bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
( POGS_AMD_SELL_STAMPS=1 timeout 600 python bench.py --config c4 --steps 100 --warmup 10 --no-cpu-baseline --no-live-traffic > gpurun_out/r05/c4_uniform.json 2> gpurun_out/r05/c4_uniform.err )
( POGS_AMD_SELL_STAMPS=1 POGS_AMD_XCD_WEIGHTS=auto POGS_AMD_TRACE=1 timeout 600 python bench.py --config c4 --steps 100 --warmup 10 --no-cpu-baseline --no-live-traffic > gpurun_out/r05/c4_auto.json 2> gpurun_out/r05/c4_auto.err )
grep -h "stamps\]\|plan_units" gpurun_out/r05/c4_uniform.err | head -8
grep -h "stamps\]\|plan_units" gpurun_out/r05/c4_auto.err | head -16
python - <<'PY'
import json
for t in ("uniform","auto"):
    try:
        d=json.loads(open("gpurun_out/r05/c4_%s.json"%t).read().strip().splitlines()[-1])
        print(t, "it/s %.1f ms/step %.4f spmv avg ms %.4f frac %.3f iters %d parity %.2e" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["solve_iterations"], d["parity_vs_reference"]["rel_x"]))
    except Exception as e: print(t, "failed", e)
PY
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r05/tests1.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r05/tests1.log
timeout 900 python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err; echo "bench rc $?"; tail -c 1500 gpurun_out/r05/bench_default.json
