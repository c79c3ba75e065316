#!/usr/bin/env python3
"""FETCH_SIZE per dispatch of tools/pmc_calib/calib against the bytes each kernel is known to touch:
python tools/pmc_calib/report.py <counter_collection.csv>"""
import csv
import sys
from collections import defaultdict

M, TABLE = 4 << 20, 1 << 30
per = defaultdict(list)
by_dispatch = defaultdict(float)
names = {}
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    by_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
    names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
for d, v in by_dispatch.items():
    per[names[d]].append(v * 1024.0)  # the counter is in KB
known = {
    "k_stream": ("coalesced stream of the table", TABLE, TABLE),
    "void k_gather<1>": ("16-byte gathers, one per 64-byte line", M * 16 + M * 4, M * 64 + M * 4),
    "void k_gather<2>": ("32-byte gathers, one per 64-byte line", M * 32 + M * 4, M * 64 + M * 4),
}
print(f"{'kernel':20s} {'FETCH_SIZE B':>14s} {'algorithmic B':>14s} {'lines x 64 B':>14s} {'FETCH/alg':>10s} {'FETCH/lines':>12s}")
for k, vals in per.items():
    if k not in known:
        continue
    what, alg, lines = known[k]
    v = sorted(vals)[len(vals) // 2]
    print(f"{k:20s} {v:14.0f} {alg:14d} {lines:14d} {v / alg:10.3f} {v / lines:12.3f}   ({what})")
