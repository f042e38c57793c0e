#!/bin/bash
# Development aid: per-kernel totals of one short bench run (rocprofv3 --kernel-trace --stats).
# usage: gpu_kstats.sh [script.py args...]   (path relative to the repo root; default: bench.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/kstats
rm -rf $O; mkdir -p $O
if [ $# -gt 0 ]; then CMD="python $R/$*"; else CMD="python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline"; fi
rocprofv3 --kernel-trace --stats -d $O/kt -o b -- $CMD > $O/log 2>&1
db=$(find $O/kt -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py $db $O/kernel_stats.csv > /dev/null
head -14 $O/kernel_stats.csv | cut -c1-150
rm -rf $O/kt
