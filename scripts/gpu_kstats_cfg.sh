#!/bin/bash
# Development aid: per-kernel totals of `bench.py --config <cfg>` (rocprofv3 --kernel-trace --stats), GPU legs only.
# usage: gpu_kstats_cfg.sh <cfg> [tag]     -> gpurun_out/kstats_<cfg>[_tag]/kernel_stats.csv
cfg=${1:-c4}; tag=${2:+_$2}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/kstats_${cfg}${tag}
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o b -- python $R/bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline > $O/log 2>&1
db=$(find $O/kt -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py $db $O/kernel_stats.csv > /dev/null
head -${3:-16} $O/kernel_stats.csv | cut -c1-160
tail -1 $O/log | cut -c1-300
rm -rf $O/kt
