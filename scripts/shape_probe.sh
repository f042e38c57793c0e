#!/bin/bash
# tuning aid: it/s and pass time of the dense lasso bench at other column counts.  usage: shape_probe.sh "m n" ...
python -c 'import torch; torch.zeros(1,device="cuda")'
for mn in "$@"; do
  set -- $mn
  timeout 300 python bench.py --config c2 --m $1 --n $2 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/l.json
  python -c "import json; d=json.load(open('/tmp/l.json')); r=d['roofline']; print('$mn', round(d['value'],1), round(d['ms_per_step'],4), round(r['avg_launch_ms'],4), round(r['bytes_per_launch']/r['avg_launch_ms']/1e6), 'GB/s')"
done
