#!/usr/bin/env python3
"""The rows of DESIGN.md §5's measurement table from one tools/collect_evidence.sh collection copied under a prefix:
python tools/design_table.py profiles/r03_m_"""
import json
import sys

prefix = sys.argv[1]
names = {"C0": "C0 10k @256²", "C1": "C1 1M @1080p", "C2": "**C2 5M @1080p**", "C3": "C3 10M @4K"}
def n(x):
    return f"{x:,.0f}".replace(",", " ")


for c in ("C0", "C1", "C2", "C3"):
    d = json.loads(open(f"{prefix}bench_{c}.json").read().strip().splitlines()[-1])
    r, e, st = d["roofline"], d.get("extra", {}), d["stage_ms"]
    eo, fl, par, cpu = e.get("composite_early_out_off"), e.get("two_frames_in_flight"), d.get("parity_vs_cpu_frame"), d.get("cpu_baseline")
    b = "**" if c == "C2" else ""
    print(f"| {names[c]} | {b}{d['ms_per_step']:.3f} ms{b} | {b}{n(d['frames_per_s'])}{b} | {b}{n(d['value'])}{b} | {n(d['config']['pairs_P'])} | "
          f"{n(r['pairs_consumed'])} / {n(r['pairs_staged'])} | {st['project']:.3f} / {st['bin_scatter']:.3f} / {st['bin_second_pass']:.3f} / "
          f"{st['bin_tile_sort']:.3f} / {st['composite']:.3f} | {b}{r['avg_launch_ms'] * 1e3:.1f}, {r['frac']:.3f}{b}"
          + (f"; VALU {r['valu_frac']:.2f} busy" if r.get("valu_frac") else "") +
          f" | {eo['avg_launch_ms']:.3f} ms, {eo['frac']:.3f} | {fl['ms_per_step']:.3f} ms ({n(fl['value'])} Msplats/s) | "
          f"≤{par['max_abs_lsb']} LSB | {cpu['seconds']:.2g} s / {cpu['all_cores']['seconds']:.2g} s |")
d = json.loads(open(f"{prefix}bench_C2.json").read().strip().splitlines()[-1])
print()
for k, v in d["roofline_per_kernel"].items():
    print(f"{k}: {v['bytes_per_frame'] / 1e6:.0f} MB in {v['ms'] * 1e3:.1f} us = {v['achieved_GBps'] / 1e3:.2f} TB/s = {v['frac']:.2f}")
fr = d["frame_roofline"]
print(f"frame: {fr['algorithmic_bytes_per_frame'] / 1e6:.0f} MB in {d['ms_per_step']:.4f} ms = {fr['frac']:.2f} ({fr['frac_of_measured_copy']:.2f} of the measured copy rate {d['roofline']['measured_copy_GBps'] / 1e3:.2f} TB/s)")
