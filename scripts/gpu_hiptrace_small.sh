#!/bin/bash
# Kernel trace of the small one-shot solves: where is the tens-of-ms gap?
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/hiptrace
rm -rf $OUT; mkdir -p $OUT
POGS_AMD_FETCH=memcpy python $GRAFT_REPO_ROOT/scripts/quick_small3.py host 2>&1 | grep "^host" | sed 's/^/memcpy-fetch /'
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $GRAFT_REPO_ROOT/scripts/quick_small3.py host > $OUT/run.log 2>&1
grep "^host" $OUT/run.log
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for a, b in zip(rows, rows[1:]):
    gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e6
    dur = (int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1e6
    if gap > 20 or dur > 20:
        print("gap %.1f ms, dur %.1f ms at %.1f ms: after %s -> %s" % (gap, dur, (int(b["Start_Timestamp"]) - t0) / 1e6, a["Kernel_Name"][:70], b["Kernel_Name"][:70]))
PY
find $OUT -name "*.csv" -size +2M -delete
