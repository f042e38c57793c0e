#!/bin/bash
# tuning aid: Gram phase time against the rows per unit (POGS_AMD_GRAM_UROWS), C2 bench
python -c 'import torch; torch.zeros(1,device="cuda")'
for u in "$@"; do
  POGS_AMD_GRAM_UROWS=$u timeout 300 python bench.py --config c2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/l.json
  python -c "import json; d=json.load(open('/tmp/l.json')); print($u, d['setup_ms']['gram_ms'], d['solve_iterations'], d['time_to_converge_s'])"
done
