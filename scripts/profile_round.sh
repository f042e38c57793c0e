#!/bin/bash
# Round profile of the bench command on the GPU box.  Writes into gpurun_out/prof_<tag>_<config>/:
#   bench.json                plain run (no profiler), with the CPU baseline
#   kernel_stats.csv          rocprofv3 --kernel-trace --stats summary of the same command
#   pmc_traffic.json          FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes, corrected)
# usage: profile_round.sh <tag> [c2|c3|c4|c2f64]       (then copy the three files into profiles/)
tag=${1:-r06}
extra=${4:-}   # extra bench.py arguments (e.g. "--projector cgls"); the output directory gets a suffix
cfg=${2:-c2}
R=${GRAFT_REPO_ROOT:-/root/repo}
sfx=$(echo $extra | tr -cd 'a-z0-9')
O=$R/gpurun_out/prof_${tag}_${cfg}${sfx:+_$sfx}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat $R/pogs_amd/libpogs_amd.so > /dev/null   # fresh box: page cache cold, the first process would pay the disk reads
python -c 'import torch; torch.zeros(1, device="cuda")' > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt -o bench -- python $R/bench.py --config $cfg $extra --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic > $O/kt.log 2>&1
db=$(find $O/kt -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py $db $O/kernel_stats.csv
python $R/scripts/iter_gaps.py $db $O/iter_gaps.json > $O/iter_gaps.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o g -- python $R/bench.py --config $cfg $extra --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic > $O/pmc_$c.log 2>&1
done
python $R/scripts/pmc_summary.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
# the plain run comes last so that its line quotes the counters collected above (same sources, same box)
[ -z "$extra" ] && cp $O/pmc_traffic.json $R/profiles/pmc_traffic_${cfg}.json
# (third argument "--no-cpu": skip the CPU leg of the plain run -- minutes at c2 / c3 / c4)
# (stdout: the long record on a BENCH_DETAIL line, then the short contract line; bench.json keeps the long one)
python $R/bench.py --config $cfg $extra --steps 200 --warmup 20 $([ "$3" = "--no-cpu" ] && echo --no-cpu-baseline) > $O/bench.out 2> $O/bench.err
tail -n 1 $O/bench.out > $O/bench_line.json
grep '^BENCH_DETAIL ' $O/bench.out | tail -n 1 | cut -c14- > $O/bench.json
cat $O/bench_line.json
