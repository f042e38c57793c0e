#!/bin/bash
# Development aid: per-kernel totals of the C4 solve (rocprofv3 --kernel-trace --stats).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/kstats_c4
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o b -- python $R/scripts/quick_configs.py c4 > $O/log 2>&1
db=$(find $O/kt -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py $db $O/kernel_stats.csv > /dev/null
head -12 $O/kernel_stats.csv | cut -c1-200
rm -rf $O/kt
