#!/usr/bin/env python3
"""Soak of what tests/test_gpu_stages.py::test_tile_lists_equal_the_reference_own_code does with one fixture: the staged
binner on the fixture's records and sorted order, and whole frames in both orders of work, `reps` times each, every
result against the reference's own lists (tests/golden/ref_binsorted_*.npz); prints what differed, if anything did.
    python tools/soak_fixture.py [small300] [reps=1500]"""
import os
import sys

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import splat_renderer_amd as sr

name = sys.argv[1] if len(sys.argv) > 1 else "small300"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
here = os.path.join(root, "tests", "golden")
g = np.load(os.path.join(here, f"ref_binsorted_{name}.npz"))
f = np.load(os.path.join(here, name + ".npz"))
w, h, tile = (int(x) for x in g["dims"])
n = g["projected"].shape[0]
total = g["indices"].shape[0]
dev = sr.Device(0)
bad = 0


def differ(what, rep, got, want):
    global bad
    if not np.array_equal(got, want):
        bad += 1
        k = np.nonzero(got != want)[0] if got.shape == want.shape else []
        print(f"MISMATCH {what} rep {rep}: {len(k)} entries differ, first at {k[:8]}: got {got[k[:8]]} want {want[k[:8]]}", flush=True)


for rep in range(reps):
    pbuf, sbuf = dev.createBufferFrom(g["projected"]), dev.createBufferFrom(g["sorted"])
    b = sr.GPUTileBinner(dev, tile)
    b.binSplats(None, pbuf, sbuf, n, w, h, numSorted=g["sorted"].shape[0])
    differ("staged counts", rep, b.getTileCountsBuffer().read(np.uint32), g["counts"])
    differ("staged offsets", rep, b.getTileOffsetsBuffer().read(np.uint32), g["offsets"])
    differ("staged indices", rep, b.getTileIndicesBuffer().read(np.uint32, total), g["indices"])
    for o in (b, pbuf, sbuf):
        o.destroy()
    props, nbuf = dev.createBufferFrom(f["props"]), dev.createBufferFrom(f["normals"])
    for order in ("tileFirst", "sortFirst"):
        r = sr.Renderer(dev, None, "rgba8unorm", n, frameOrder=order)
        r.render(f["uniforms"], props, nbuf, None, w, h)
        t = r.finish()
        if t != total:
            bad += 1
            print(f"MISMATCH {order} total rep {rep}: {t} != {total}", flush=True)
        differ(order + " counts", rep, r.binner.getTileCountsBuffer().read(np.uint32), g["counts"])
        differ(order + " indices", rep, r.binner.getTileIndicesBuffer().read(np.uint32, total), g["indices"])
        r.destroy()
    props.destroy()
    nbuf.destroy()
    if rep % 250 == 249:
        print(f"{rep + 1} reps, {bad} mismatches", flush=True)
print(f"{name}: {reps} reps ({w}x{h}, tile {tile}, {n} splats, {total} pairs): {bad} mismatches")
