#!/bin/bash
# Development aid: A / B of library builds on ONE box, variants alternating.
#   ab_variants.sh "<bench.py arguments>" <reps> <tag> [<tag> ...]
# a tag names pogs_amd/variants/libpogs_amd_<tag>.so (scripts/build_variant.py); "base" is the regular build.
# Prints per run: it/s, the dominant kernel's mean launch time, iterations of the solve, rel_x against the fixture.
R=${GRAFT_REPO_ROOT:-/root/repo}
args=$1; reps=$2; shift 2
cd $R
cp pogs_amd/libpogs_amd.so /tmp/base.so
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
for rep in $(seq 1 $reps); do
  for tag in "$@"; do
    if [ "$tag" = base ]; then cp /tmp/base.so pogs_amd/libpogs_amd.so; else cp pogs_amd/variants/libpogs_amd_$tag.so pogs_amd/libpogs_amd.so; fi
    python bench.py $args --no-cpu-baseline --no-live-traffic 2>/tmp/ab_err.log | python -c "
import json,sys
ls=[l for l in sys.stdin.read().splitlines() if l.startswith('BENCH_DETAIL ')]
if not ls: print('$tag rep $rep: FAILED'); print(open('/tmp/ab_err.log').read()[-1500:]); sys.exit(0)
d=json.loads(ls[-1][13:]); rf=d['roofline']; par=d.get('parity_vs_reference') or {}
print('%-10s rep $rep: %8.1f it/s  pass %.4f ms (frac %.3f)  iteration_frac %.3f  iters %s  status %s  rel_x %s  ttc %.4f s  setup_ms %s' % ('$tag', d['value'], rf['avg_launch_ms'], rf['frac'], rf['iteration_frac'], d['solve_iterations'], d['solve_status'], par.get('rel_x'), d['time_to_converge_s'], {k[:-3]: round(v, 2) for k, v in d.get('setup_ms', {}).items()}))"
  done
done
cp /tmp/base.so pogs_amd/libpogs_amd.so
