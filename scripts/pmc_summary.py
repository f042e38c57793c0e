"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, csv output) per kernel.

FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (guides/MI355X_MICROARCH.md, HBM section):
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so the
read side is doubled; WRITE_SIZE is taken as is (uncalibrated, and <1 % of the traffic here).
usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import re
import sys


def load(fn, counter):
    agg = collections.defaultdict(list)
    with open(fn) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            name = re.sub(r"pogs_amd::|\(anonymous namespace\)::", "", row["Kernel_Name"])
            if name.startswith("void at::") or "rocclr" in name:
                continue
            agg[name].append(float(row["Counter_Value"]))
    return agg


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
out = {}
for name in sorted(set(fetch) | set(write)):
    f = fetch.get(name, [0.0])
    w = write.get(name, [0.0])
    # guarded launches of the device-resident CG loop that found the loop ended return at once
    # (cg_fused.h): they move nothing and are not launches of the kernel in the roofline's sense
    noop = 0
    if len(f) == len(w) and max(f) > 0:
        keep = [i for i in range(len(f)) if f[i] >= 0.02 * max(f)]
        noop = len(f) - len(keep)
        f, w = [f[i] for i in keep], [w[i] for i in keep]
    fe, wr = sum(f) / len(f) * 1024.0, sum(w) / len(w) * 1024.0
    out[name] = {"launches": len(f), "noop_launches_excluded": noop, "fetch_size_bytes_raw": fe, "write_size_bytes": wr,
                 "hbm_bytes_per_launch_corrected": 2.0 * fe + wr}
# which kernel sources the counters belong to: bench.py compares this with the sources it runs on
import hashlib
import os

csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pogs_amd", "csrc")
hh = hashlib.sha256()
for fn in sorted(os.listdir(csrc)):
    if fn.endswith((".h", ".hip")):
        hh.update(fn.encode())
        hh.update(open(os.path.join(csrc, fn), "rb").read())
out["_meta"] = {"csrc_sha16": hh.hexdigest()[:16]}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(out), "kernels")
