"""Idle time between the kernels of the ADMM loop, from a rocprofv3 --kernel-trace rocpd database: how long the GPU waits
between the launch that publishes iteration k's scalars and the first kernel of iteration k + 1 (the host's poll ->
decide -> launch bubble, VERDICT r04 item 5 ii), and between the other kernels of an iteration.
    python scripts/iter_gaps.py <db> <out.json>"""
import json
import re
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
tables = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
rows = None
for q in ("select name, start, end from kernels order by start",
          "select kernel_name, start, end from kernels order by start"):
    try:
        rows = list(con.execute(q))
        break
    except sqlite3.Error:
        continue
if rows is None:   # discover: a dispatch table with start / end and a symbol table with the names
    disp = [t for t in tables if "kernel_dispatch" in t][0]
    sym = [t for t in tables if "kernel_symbol" in t][0]
    cols = [r[1] for r in con.execute("pragma table_info(%s)" % disp)]
    scol = [r[1] for r in con.execute("pragma table_info(%s)" % sym)]
    name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else "name")
    rows = list(con.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, disp, sym)))
short = lambda n: re.sub(r"pogs_amd::|\(anonymous namespace\)::|void ", "", str(n)).split("(")[0].split("<")[0]
ev = [(short(n), int(s), int(e)) for n, s, e in rows]
gaps = {}
for (n0, s0, e0), (n1, s1, e1) in zip(ev, ev[1:]):
    g = (s1 - e0) / 1e3   # ns -> us
    if g < 0 or g > 2000:   # overlapping launches / set-up pauses
        continue
    gaps.setdefault((n0, n1), []).append(g)
res = {}
for (a, b), v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) < 20:
        continue
    v.sort()
    res["%s -> %s" % (a, b)] = {"n": len(v), "median_us": v[len(v) // 2], "p10_us": v[len(v) // 10], "p90_us": v[9 * len(v) // 10],
                                 "mean_us": sum(v) / len(v)}
json.dump(res, open(out, "w"), indent=1)
for k, d in list(res.items())[:14]:
    print("%-70s n %5d  median %6.2f us  p10 %6.2f  p90 %6.2f" % (k, d["n"], d["median_us"], d["p10_us"], d["p90_us"]))
