#!/bin/bash
# Development aid: per-kernel totals of one short bench run (rocprofv3 --kernel-trace --stats).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/kstats
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o b -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $O/log 2>&1
db=$(find $O/kt -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py $db $O/kernel_stats.csv > /dev/null
head -14 $O/kernel_stats.csv | cut -c1-150
rm -rf $O/kt
