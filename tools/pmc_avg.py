#!/usr/bin/env python3
"""Per kernel, per counter: the average value per launch from rocprofv3 --pmc counter_collection CSVs
(python tools/pmc_avg.py a_counter_collection.csv [b_counter_collection.csv ...]); with --valu also the busy
fraction of the vector ALUs, SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8) (GRBM_GUI_ACTIVE is summed
over the 8 XCDs; SQ_ACTIVE_INST_VALU counts quad-cycles over every SIMD of the device)."""
import csv
import sys
from collections import defaultdict

paths = [a for a in sys.argv[1:] if not a.startswith("--")]
acc = defaultdict(lambda: [0.0, set()])
for path in paths:
    for r in csv.DictReader(open(path)):
        key = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
        acc[key][0] += float(r["Counter_Value"])
        acc[key][1].add(r["Dispatch_Id"])
avg = {k: v[0] / max(len(v[1]), 1) for k, v in acc.items()}
print("kernel,counter,value_per_launch")
for (k, c), v in sorted(avg.items()):
    print(f'"{k}",{c},{v:.1f}')
if "--valu" in sys.argv:
    print("kernel,valu_busy_frac")
    for k in sorted({k for k, _ in avg}):
        a, g = avg.get((k, "SQ_ACTIVE_INST_VALU")), avg.get((k, "GRBM_GUI_ACTIVE"))
        if a and g:
            print(f'"{k}",{a * 4 / (1024 * g / 8):.3f}')
