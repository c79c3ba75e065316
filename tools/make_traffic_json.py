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
# (name prefixes: k_composite_px<EARLY_OUT, LIT32, COUNT, AHEAD, DISC>, k_composite<MODE, EARLY_OUT, DISC, LIT32>)
KERNELS = ("void k_composite_px<true, true, false,", "void k_composite_px<true, true, true,", "void k_composite<0, true, false, true>")
KERNELS_EARLY_OUT_OFF = ("void k_composite_px<false, true, true,", "void k_composite_px<false, true, false,", "void k_composite<0, false, false, true>")


def pick(rows, kernels):
    """The first row whose kernel name starts with one of `kernels` (in their order of preference)."""
    for kernel in kernels:
        for name in rows:
            if name.startswith(kernel):
                return name
    return None


def fetch_write(path, kernels=KERNELS):
    rows = {r["kernel"]: r for r in csv.DictReader(open(path))}
    rows = {k: r for k, r in rows.items() if float(r["FETCH_SIZE_KB_per_launch"]) > 0}
    kernel = pick(rows, kernels)
    if kernel is None:
        raise SystemExit(f"{path}: no row for any of {kernels}")
    return kernel, float(rows[kernel]["FETCH_SIZE_KB_per_launch"]), float(rows[kernel]["WRITE_SIZE_KB_per_launch"])


def tile_sort_lds(path):
    """k_tile_sort's own roof: LDS-array cycles (SQ_LDS_IDX_ACTIVE, all CUs) over the cycles the LDS pipes had
    (256 CUs x the launches' duration: GRBM_GUI_ACTIVE is summed over the 8 XCDs), and the share of them spent on bank
    conflicts, summed over the kernel's launches of a frame (one per size class)."""
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["kernel"].startswith("void k_tile_sort<"):
            acc[r["counter"]] = acc.get(r["counter"], 0.0) + float(r["value_per_launch"])
    if not acc.get("GRBM_GUI_ACTIVE"):
        return None
    cycles = acc["GRBM_GUI_ACTIVE"] / 8.0
    return {"lds_array_cycles_per_frame": round(acc["SQ_LDS_IDX_ACTIVE"]), "bank_conflict_cycles_per_frame": round(acc["SQ_LDS_BANK_CONFLICT"]),
            "lds_instructions_per_frame": round(acc["SQ_INSTS_LDS"]), "kernel_cycles": round(cycles),
            "lds_pipe_frac": round(acc["SQ_LDS_IDX_ACTIVE"] / (256.0 * cycles), 4),
            "conflict_share_of_array_cycles": round(acc["SQ_LDS_BANK_CONFLICT"] / acc["SQ_LDS_IDX_ACTIVE"], 4),
            "lds_issue_stall_cycles_per_frame": round(acc.get("SQ_WAIT_INST_LDS", 0.0)),
            "source": f"{os.path.relpath(path, root)} (rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ..., per launch, both size "
                      "classes added; frac = SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE / 8))"}


def valu(path, kernels=KERNELS):
    rows, on = {}, False
    for line in open(path):
        if line.startswith("kernel,valu_busy_frac"):
            on = True
            continue
        if on and "," in line:
            k, v = line.rsplit(",", 1)
            rows[k.strip('"')] = float(v)
    kernel = pick(rows, kernels)
    return rows[kernel] if kernel else None


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
    lds = f"{prefix}{cfg}_lds_counters.csv"
    if os.path.exists(lds):
        ts = tile_sort_lds(lds)
        if ts:
            entry["k_tile_sort_lds"] = ts
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
