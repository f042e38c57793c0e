#!/bin/bash
# tuning aid: Gram phase time of a bench configuration under env assignments.  usage: gram_env_probe.sh <config> "VAR=val VAR2=val" ...
cfg=$1; shift
python -c 'import torch; torch.zeros(1,device="cuda")'
for v in "$@"; do
  env $v timeout 300 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/l.json
  python -c "import json; d=json.load(open('/tmp/l.json')); print('$v', d['setup_ms']['gram_ms'], d['solve_iterations'], d['time_to_converge_s'])"
done
