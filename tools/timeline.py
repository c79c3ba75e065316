#!/usr/bin/env python3
"""Prints one frame of a rocprofv3 kernel trace (k_project to k_project) with inter-kernel gaps."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_project" in r["Kernel_Name"]]
a, b = idx[which], idx[which + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-28:]
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f} gap {(s - prev) / 1e3:6.1f} {name}")
    prev = e
    busy += e - s
span = int(rows[b]["Start_Timestamp"]) - t0
print(f"frame span {span / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us")
