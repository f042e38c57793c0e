#!/bin/bash
# Development aid: kernel totals of scripts/quick_c2.py under env assignments.  usage: gpu_kstats_env.sh tag VAR=val ...
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/kse_$tag
rm -rf $O; mkdir -p $O
env "$@" rocprofv3 --kernel-trace --stats -d $O/kt -o b -- python $R/scripts/quick_c2.py > $O/log 2>&1
db=$(find $O/kt -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py $db $O/kernel_stats.csv > /dev/null
grep -E "gram|split_f16|sum_slabs" $O/kernel_stats.csv | cut -c1-60,100-200
rm -rf $O/kt
