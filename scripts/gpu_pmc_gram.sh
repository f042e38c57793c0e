#!/bin/bash
# Development aid: L2 hit / fabric fetch counters of the Gram kernels of one C2 solve.
# usage: gpu_pmc_gram.sh <tag> [env assignments...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_gram_$tag
rm -rf $O; mkdir -p $O
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  n=$(echo $c | tr " " "_")
  env "$@" rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -o g -- python $R/scripts/quick_c2.py > $O/$n.log 2>&1
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gram" in k or "split_f16" in k:
                k = k.split("(")[0][-40:]
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, v in agg.items():
            print(k, {c: "%.4g" % x for c, x in v.items()}, "launches", max(n[(k, c)] for c in v))
PY
