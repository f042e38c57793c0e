#!/bin/bash
# round 5, GPU call 10: the full -m gpu suite and the default bench run on the final sources (after the FunctionVector change)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05
cat pogs_amd/libpogs_amd.so > /dev/null
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r05/tests10.log 2>&1; echo "full suite rc $?"; grep -E "passed|failed" gpurun_out/r05/tests10.log | tail -2
( time timeout 1200 python bench.py > gpurun_out/r05/bench_final2.json 2> gpurun_out/r05/bench_final2.err ) 2>&1 | grep real; echo "bench rc $?"; tail -c 700 gpurun_out/r05/bench_final2.json
python __graft_entry__.py --smoke 2>&1 | tail -2
