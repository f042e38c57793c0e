#!/bin/bash
# Development aid: A / B of ONE library under environment settings, alternating on one box.
#   ab_env.sh "<bench.py arguments>" <reps> "<ENV=a ...>" "<ENV=b ...>" ...      ("-" = no extra environment)
R=${GRAFT_REPO_ROOT:-/root/repo}
args=$1; reps=$2; shift 2
cd $R
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
for rep in $(seq 1 $reps); do
  for e in "$@"; do
    envs=$([ "$e" = "-" ] && echo "" || echo $e)
    env $envs python bench.py $args --no-cpu-baseline --no-live-traffic 2>/tmp/ab_err.log | python -c "
import json,sys
ls=[l for l in sys.stdin.read().splitlines() if l.startswith('BENCH_DETAIL ')]
if not ls: print('$e rep $rep: FAILED'); print(open('/tmp/ab_err.log').read()[-1500:]); sys.exit(0)
d=json.loads(ls[-1][13:]); rf=d['roofline']; par=d.get('parity_vs_reference') or {}
print('%-24s rep $rep: %8.1f it/s  ms/step %.4f  pass %.4f ms  iteration_frac %.3f  iters %s  status %s  rel_x %s  ttc %.4f s' % ('$e', d['value'], d['ms_per_step'], rf['avg_launch_ms'], rf['iteration_frac'], d['solve_iterations'], d['solve_status'], par.get('rel_x'), d['time_to_converge_s']))"
  done
done
