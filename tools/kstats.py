#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: average duration per (kernel, grid size)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # leading dispatches to ignore (warm-up)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[skip:]
agg = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-38:]
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))
    agg[(name, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':40s} {'grid':>10s} {'calls':>6s} {'avg_us':>9s} {'total%':>7s}")
for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name:40s} {grid:10d} {len(v):6d} {sum(v) / len(v) / 1e3:9.2f} {100 * sum(v) / tot:7.2f}")
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"sum of kernel time {tot / 1e3:.1f} us over a span of {span / 1e3:.1f} us ({100 * tot / span:.1f}% busy)")
