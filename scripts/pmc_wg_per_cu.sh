#!/bin/bash
# Development aid: memory-path counters of ONE streaming kernel at two and at three workgroups per CU (C3's shape;
# scripts/micro/c3_bisect table 5), separate rocprofv3 --pmc passes, summed per launch and grouped by grid size.
#   pmc_wg_per_cu.sh [table]    5 (default): one kernel at 512 / 768 workgroups, grouped by grid size;
#                                9: the plain and the prefetching one-pass skeleton at 512 workgroups, grouped by kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
TABLE=${1:-5}
O=$R/gpurun_out/pmc_wg_per_cu
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
         "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES" "TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_BUSY" \
         "TCC_REQ TCC_HIT TCC_MISS TCC_SRC_FIFO_FULL" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_GATE_EN1"; do
  n=$(echo $c | tr " " "_" | cut -c1-50)
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -o g -- $R/scripts/micro/bin/c3_bisect $TABLE 5 > $O/$n.log 2>&1
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
        for r in csv.DictReader(open(f)):
            if "stream_rows" not in r["Kernel_Name"]: continue
            g = (r.get("Grid_Size") or r.get("Grid_Size_X")) if "$TABLE" == "5" else r["Kernel_Name"].split("<")[0].split("::")[-1]
            agg[g][r["Counter_Name"]] += float(r["Counter_Value"]); n[g][r["Counter_Name"]] += 1
        for g in sorted(agg):
            print("grid", g, {c: "%.5g" % (x / max(n[g][c], 1)) for c, x in agg[g].items()}, "launches", max(n[g].values()))
PY
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
