#!/bin/bash
# tuning aid: C4-style SpMV time under forced (rows per range, column groups).  usage: sell_cfg_probe.sh <rows m> "ENV=.. ENV=.." ...
m=$1; shift
python -c 'import torch; torch.zeros(1,device="cuda")'
for v in "$@"; do
  env $v timeout 300 python bench.py --config c4 --m $m --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/l.json
  python -c "import json; d=json.load(open('/tmp/l.json')); r=d['roofline']; print('$v', round(d['value'],1), round(r['avg_launch_ms'],4), round(r['achieved']))"
done
