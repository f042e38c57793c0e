"""Turns a rocprofv3 rocpd database (--kernel-trace --stats) into a per-kernel CSV summary."""
import csv
import re
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, tot, avg, pct in rows:
        name = re.sub(r"pogs_amd::|\(anonymous namespace\)::", "", str(name))
        w.writerow([name, calls, "%.3f" % tot, "%.3f" % avg, "%.3f" % pct])
print("wrote", out, len(rows), "kernels")
