#!/bin/bash
# Development aid: kernel trace of one C2 solve -> per-kernel stats and idle gaps.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/trace_c2
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_c2 -- python $R/scripts/quick_c2.py "$@" > $R/gpurun_out/trace_c2.log 2>&1
f=$(find $R/gpurun_out/trace_c2 -name "*kernel_trace.csv" | head -1)
python $R/scripts/trace_gaps.py $f
rm -rf $R/gpurun_out/trace_c2
