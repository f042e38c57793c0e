"""Summarise idle gaps on the GPU timeline from a rocprofv3 --kernel-trace CSV.

usage: trace_gaps.py <kernel_trace.csv> [min_gap_us]
Prints, per (previous kernel -> next kernel) pair, the summed idle time between them,
and the busy/wall split -- to find host syncs and launch bubbles (development aid).
"""
import csv
import collections
import re
import sys

path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"pogs_amd::|\(anonymous namespace\)::", "", name)
    m = re.match(r"([A-Za-z_0-9:]+)(<.*)?", name)
    base = m.group(1)
    ops = re.findall(r"(\w+Op)<", name)
    flags = re.findall(r"<(?:float|double), (\d+, \d+, \d+(?:, \w+)*)", name)
    return base + ("[" + ",".join(ops) + "]" if ops else "") + ("{" + flags[0] + "}" if flags else "")


busy = sum(e - s for s, e, _ in rows)
wall = rows[-1][1] - rows[0][0]
print("kernels %d  busy %.3f ms  wall %.3f ms" % (len(rows), busy / 1e6, wall / 1e6))
gaps = collections.defaultdict(lambda: [0, 0])
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g > 0:
        k = (short(n0), short(n1))
        gaps[k][0] += g
        gaps[k][1] += 1
print("idle by transition (top 25):")
for k, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print("  %9.3f ms  x%-5d avg %8.1f us   %s -> %s" % (g / 1e6, c, g / c / 1e3, k[0], k[1]))

stats = collections.defaultdict(lambda: [0, 0])
for s0, e0, n0 in rows:
    k = short(n0)
    stats[k][0] += e0 - s0
    stats[k][1] += 1
print("per kernel (top 25 by total):")
for k, (tot, c) in sorted(stats.items(), key=lambda kv: -kv[1][0])[:25]:
    print("  %9.3f ms  x%-5d avg %9.1f us   %s" % (tot / 1e6, c, tot / c / 1e3, k))
