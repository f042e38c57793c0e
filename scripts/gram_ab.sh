#!/bin/bash
# A/B of the staggered 256-tile Gram kernel against the lockstep one, same box (gpurun_out/gram_ab.log)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for rep in 1 2; do
for mode in 0 1; do
  echo "== POGS_AMD_GRAM_LOCKSTEP=$mode rep $rep" >> $O/gram_ab.log
  POGS_AMD_GRAM_LOCKSTEP=$mode python bench.py --config c2 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('gram_ms', d['setup_ms']['gram_ms'], 'tflops', d['gram_tflops'], 'iters', d['solve_iterations'], 'init', d['init_s'], 'ttc', d['time_to_converge_s'], 'parity', d['parity_vs_reference']['rel_x'], 'cycles_init', d['handle_cycles']['init_s'])" >> $O/gram_ab.log 2>&1
done
done
cat $O/gram_ab.log
