"""Development aid: host API calls against kernel executions, from a rocprofv3 --hip-trace --kernel-trace rocpd database
(where does the host spend the time between the publish it polls for and the launches of the next iteration?).
    python scripts/host_timeline.py <db> [schema]"""
import re
import sqlite3
import sys

db = sys.argv[1]
con = sqlite3.connect(db)
names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view') order by name")]
if len(sys.argv) > 2:
    for t in names:
        cols = [r[1] for r in con.execute("pragma table_info(%s)" % t)]
        print(t, cols)
    for v in ("regions", "kernels"):
        if v in names:
            for r in con.execute("select * from %s limit 3" % v):
                print(v, r)
    sys.exit(0)
short = lambda n: re.sub(r"pogs_amd::|\(anonymous namespace\)::|void ", "", str(n)).split("(")[0].split("<")[0]
kern = [(short(n), int(s), int(e)) for n, s, e in con.execute("select name, start, end from kernels order by start")]
api = [(str(n), int(s), int(e)) for n, s, e in con.execute("select name, start, end from regions order by start")]
# iterations of the dense one-pass loop: pre_cols -> tri -> reduce_cols -> pass -> sum_publish
import bisect
api_starts = [a[1] for a in api]
pub_ends = [(i, e) for i, (n, s, e) in enumerate(kern) if n.startswith("sum_publish")]
rows = []
for i, pub_end in pub_ends:
    nxt = kern[i + 1:i + 6]
    if len(nxt) < 5 or not nxt[0][0].startswith("pre_cols"):
        continue
    # host calls that START within [pub_end - 30 us, start of the pass]
    lo = bisect.bisect_left(api_starts, pub_end - 30000)
    hi = bisect.bisect_left(api_starts, nxt[3][1])
    calls = [(n, (s - pub_end) / 1e3, (e - s) / 1e3) for n, s, e in api[lo:hi]]
    rows.append((pub_end, [(n, (s - pub_end) / 1e3, (e - s) / 1e3) for n, s, e in nxt], calls))
print("iterations found:", len(rows))
for pub_end, ks, calls in rows[60:64]:
    print("--- publish ends at 0; kernels (start us, duration us):", ", ".join("%s %.1f %.1f" % k for k in ks))
    for c in calls:
        print("      host %-28s starts %7.1f us  lasts %6.1f us" % c)
# medians over all iterations: start of each of the five kernels, and start / duration of the k-th hipLaunchKernel after the publish
import statistics as st
for j in range(5):
    v = [r[1][j][1] for r in rows]
    print("kernel %d (%s): starts %.1f us after the publish ends (median), lasts %.1f" % (j, rows[0][1][j][0], st.median(v), st.median([r[1][j][2] for r in rows])))
for k in range(6):
    v = []
    for r in rows:
        ls = [c for c in r[2] if "Launch" in c[0] and c[1] > 0]
        if len(ls) > k:
            v.append(ls[k])
    if v:
        print("launch call %d after the publish: starts %.1f us (median), lasts %.1f us" % (k, st.median([c[1] for c in v]), st.median([c[2] for c in v])))
