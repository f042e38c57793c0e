#!/bin/bash
# Development aid: SQ counters of the SpMV kernel over one C4 solve (which unit is the kernel waiting for?).
# usage: gpu_pmc_spmv.sh <tag> [env assignments...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_spmv_$tag
rm -rf $O; mkdir -p $O
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" \
         "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr " " "_" | cut -c1-60)
  env "$@" timeout 280 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -o g -- python $R/bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic > $O/$n.log 2>&1
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "spmv_sell_kernel" in k and "SpAxpbyNormOp" in k:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        print({c: "%.4g" % (x / max(n[c], 1)) for c, x in agg.items()}, "launches", max(n.values()) if n else 0)
PY
# the raw per-launch counter tables are large (gpurun copies at most 64 MiB back): keep the logs only
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
