#!/usr/bin/env python3
"""Diagnostic: the sequence of tests/test_gpu_stages.py::test_tile_lists_equal_the_reference_own_code (a tile-first Renderer, then a
sort-first Renderer on the same scene, each rendering ONE frame on a fresh binner) repeated; counts, offsets and lists of every frame
against the reference-executed fixture.  One sighting of wrong sort-first counts in 18 suite runs (profiles/r04_s_*) started this.
    python tools/repro_sortfirst_small.py [rounds=300] [fixture=ragged1000]"""
import os
import sys

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import splat_renderer_amd as sr

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
name = sys.argv[2] if len(sys.argv) > 2 else "ragged1000"
here = os.path.join(root, "tests", "golden")
g = np.load(os.path.join(here, f"ref_binsorted_{name}.npz"))
f = np.load(os.path.join(here, name + ".npz"))
w, h, tile = (int(x) for x in g["dims"])
n = g["projected"].shape[0]
dev = sr.Device(0)
props, nbuf = dev.createBufferFrom(f["props"]), dev.createBufferFrom(f["normals"])
bad = 0
r = None
poison = os.environ.get("REPRO_POISON")  # fill freed device memory with a pattern first: an uninitialised read shows
rng = np.random.default_rng(5)
for k in range(rounds):
    if poison:
        junk = []
        for size in (1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24):
            for _ in range(3):
                a = (rng.integers(0, 2**32, size // 4, dtype=np.uint32) if poison == "random" else np.full(size // 4, int(poison, 0), np.uint32))
                junk.append(dev.createBufferFrom(a))
        for b_ in junk:
            b_.destroy()
    for order in ("tileFirst", "sortFirst"):
        r = sr.Renderer(dev, None, "rgba8unorm", n, frameOrder=order)
        r.render(f["uniforms"], props, nbuf, None, w, h)
        total = r.finish()
        c = r.binner.getTileCountsBuffer().read(np.uint32)
        o = r.binner.getTileOffsetsBuffer().read(np.uint32)
        idx = r.binner.getTileIndicesBuffer().read(np.uint32, g["indices"].shape[0])
        ok = total == g["indices"].shape[0] and np.array_equal(c, g["counts"]) and np.array_equal(o, g["offsets"]) and np.array_equal(idx, g["indices"])
        if not ok:
            bad += 1
            print(f"round {k} {order}: MISMATCH total {total} counts_ok {np.array_equal(c, g['counts'])} offsets_ok {np.array_equal(o, g['offsets'])} "
                  f"lists_ok {np.array_equal(idx, g['indices'])}; counts {c.tolist()[:12]} ...", flush=True)
print(f"{rounds} rounds x 2 orders on {name}: {bad} frames differ from the fixture; ranking {dev.rankStatus()}")
