#!/bin/bash
# Development aid: kernel trace of a python script -> per-kernel stats and idle gaps.
# usage: gpu_trace.sh <tag> <script> [args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
rm -rf $R/gpurun_out/trace_$tag
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$tag -- python $R/"$@" > $R/gpurun_out/trace_$tag.log 2>&1
f=$(find $R/gpurun_out/trace_$tag -name "*kernel_trace.csv" | head -1)
python $R/scripts/trace_gaps.py $f
rm -rf $R/gpurun_out/trace_$tag
