#!/usr/bin/env python3
"""Merge two rocprofv3 --pmc counter CSVs (FETCH_SIZE pass, WRITE_SIZE pass) into per-kernel
averages per launch: python tools/pmc_merge.py fetch_counter_collection.csv write_counter_collection.csv"""
import csv
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0]
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1].add(r["Dispatch_Id"])
    return {k: v[0] / max(len(v[1]), 1) for k, v in acc.items()}


f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
print("kernel,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch")
for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, 0) + w.get(k, 0))):
    print(f'"{k}",{f.get(k, 0):.1f},{w.get(k, 0):.1f}')
