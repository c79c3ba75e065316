#!/usr/bin/env python3
"""profiles/traffic.json from one tools/collect_evidence.sh run that has been copied into profiles/ under a prefix:
python tools/make_traffic_json.py profiles/r02_b_   (reads <prefix>C2_pmc_fetch_write.csv, <prefix>C2_sq_counters.csv,
<prefix>pmc_composite_C0_C1_C3.csv, <prefix>bench_C*.json).  What bench.py reports as roofline.traffic / valu_frac is
read from the file this writes — PMC figures are collected under rocprofv3 in separate passes, never inside a bench run."""
import csv
import json
import os
import sys

prefix = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# front-to-back, early-out on, isotropic, lit records: the lane-efficient kernel where the screen has >= 2048 tiles (the
# instantiation bench.py's timed frames run: no consumed-entry counting; the counting one if only that was traced), round 2's
# kernel below that (C0)
KERNELS = ("void k_composite_px<true, true, false>", "void k_composite_px<true, true, true>", "void k_composite<0, true, false, true>")
KERNELS_EARLY_OUT_OFF = ("void k_composite_px<false, true, true>", "void k_composite_px<false, true, false>", "void k_composite<0, false, false, true>")


def fetch_write(path, kernels=KERNELS):
    rows = {r["kernel"]: r for r in csv.DictReader(open(path))}
    for kernel in kernels:
        if kernel in rows and float(rows[kernel]["FETCH_SIZE_KB_per_launch"]) > 0:
            return kernel, float(rows[kernel]["FETCH_SIZE_KB_per_launch"]), float(rows[kernel]["WRITE_SIZE_KB_per_launch"])
    raise SystemExit(f"{path}: no row for any of {kernels}")


def valu(path, kernels=KERNELS):
    rows, on = {}, False
    for line in open(path):
        if line.startswith("kernel,valu_busy_frac"):
            on = True
            continue
        if on and "," in line:
            k, v = line.rsplit(",", 1)
            rows[k.strip('"')] = float(v)
    for kernel in kernels:
        if kernel in rows:
            return rows[kernel]
    return None


out = {}
for cfg in ("C2", "C0", "C1", "C3"):
    src = f"{prefix}{cfg}_pmc_fetch_write.csv"
    if not os.path.exists(src):
        continue
    KERNEL, f, w = fetch_write(src)
    bench = json.load(open(f"{prefix}bench_{cfg}.json"))
    staged = bench["roofline"]["pairs_staged"]
    # FETCH_SIZE counts a coalesced stream at 1/2 on gfx950 and a random 16-/32-byte gather at one whole 64-byte line
    # (profiles/r01_e_pmc_fetch_size_calibration.txt): the composite's reads are those gathers plus the coalesced 4-byte
    # index stream, so traffic = FETCH + 1/2 * 4 B * staged entries + WRITE
    traffic = f * 1024 + 0.5 * 4 * staged + w * 1024
    entry = {"k_composite_hbm_bytes_per_launch": round(traffic), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
             "source": f"{os.path.relpath(src, root)} (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of "
                       f"`python bench.py --config {cfg} --no-cpu-baseline --no-parity --steps 5`, KB*1024 per launch of {KERNEL}; + half of "
                       "the 4-byte index stream, which FETCH_SIZE counts at 1/2: profiles/r01_e_pmc_fetch_size_calibration.txt); "
                       "committed file, not measured in the bench run",
             "configuration": "reference layouts (interleaved properties + normals), lit composite records"}
    sq = f"{prefix}{cfg}_sq_counters.csv"
    if os.path.exists(sq) and valu(sq) is not None:
        entry["valu_busy_frac"] = valu(sq)
        entry["valu_busy_frac_early_out_off"] = valu(sq, KERNELS_EARLY_OUT_OFF)
        entry["valu_source"] = (f"{os.path.relpath(sq, root)}: SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8), separate "
                                "--pmc passes; committed file, not measured in the bench run")
    out[cfg] = entry
old = os.path.join(root, "profiles", "traffic.json")
prev = json.load(open(old)) if os.path.exists(old) else {}
for k, v in prev.items():  # keep what this run does not cover (disc footprint, ProjectedSplat-record configurations)
    if k not in out and k not in ("C0", "C1", "C2", "C3"):
        out[k] = v
    elif k in ("C0", "C1", "C2", "C3") and k + "_projected_records_prelit_planes" not in prev:
        out[k + "_projected_records_prelit_planes"] = v  # round 1's measurement of that configuration
json.dump(out, open(old, "w"), indent=1)
print("wrote", old, list(out))
