#!/bin/bash
# c4 through the RCCL code path with a one-rank communicator against the plain run, same box, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
  for fd in 0 1; do
    POGS_AMD_FORCE_DIST=$fd python bench.py --config c4 --steps 100 --warmup 10 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('BENCH_DETAIL ')][-1][13:])
print('force_dist=$fd', 'it/s %.1f'%d['value'], 'ms %.4f'%d['ms_per_step'], 'rccl_nranks', d['config']['rccl_nranks'], 'spmv/it %.2f'%d['roofline']['iteration']['spmv_per_iteration'], 'iters', d['solve_iterations'])"
  done
done
