#!/bin/bash
# Development aid: kernel times of the C4 SpMVs (rocprofv3 --kernel-trace --stats over scripts/spmv_probe.py) for one or more
# builds of the library: spmv_probe.sh <tag>[:ENV=value...] ...   (a tag names pogs_amd/variants/libpogs_amd_<tag>.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/probe
cp $R/pogs_amd/libpogs_amd.so /tmp/orig.so
for tagenv in "$@"; do
  tag=${tagenv%%:*}; envs=${tagenv#*:}; [ "$envs" = "$tagenv" ] && envs=""
  envs=$(echo $envs | tr ":" " ")
  if [ "$tag" = base ]; then cp /tmp/orig.so $R/pogs_amd/libpogs_amd.so; else cp $R/pogs_amd/variants/libpogs_amd_$tag.so $R/pogs_amd/libpogs_amd.so; fi
  rm -rf /tmp/kt_$tag
  env $envs timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o p -- python $R/scripts/spmv_probe.py > $R/gpurun_out/probe/$tag.log 2>&1
  db=$(find /tmp/kt_$tag -name "*.db" | head -1)
  python $R/scripts/rocpd_summary.py $db $R/gpurun_out/probe/$tag.csv > /dev/null 2>&1
  python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$R/gpurun_out/probe/$tag.csv")) if "spmv_sell" in r["kernel"] or "reduce_parts" in r["kernel"]]
print("== $tagenv:", "; ".join("%s x%s avg %.1f us" % (r["kernel"].split("(")[0][-40:], r["calls"], float(r["avg_us"])) for r in rows))
PY
done
cp /tmp/orig.so $R/pogs_amd/libpogs_amd.so
