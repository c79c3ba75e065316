#!/usr/bin/env python3
"""What would overlapping the per-tile sort with the composite be worth?  (measured before anything is built.)

The two kernels of one frame are a serial 72 + 48 us at C2 although a tile's composite needs only that tile's sort.  Any scheme
that overlaps them (two streams; one fused kernel) is bounded by what the two kernels take SIDE BY SIDE on the device.  This
probe measures exactly that with what exists: a C2 (or C3) frame is rendered; then the frame's per-tile sort is run again and
again on one context's stream (splat_debug_rerun_tile_sort: same pairs, same lists) while a second context's stream runs the
composite over a copy of the frame's lists — first each alone, then both at once.

    python tools/overlap_probe.py [C2] [iters]
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# (splat_debug_rerun_tile_sort is an experiment hook: the test build of the library carries it, the shipped one does not)
os.environ.setdefault("SPLAT_LIB_PATH", os.path.join(ROOT, "splat_renderer_amd", "libsplat_hip_hooks.so"))
import numpy as np
import splat_renderer_amd as sr
from splat_renderer_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n, w, h = sr.scene.CONFIGS[name]
tile = sr.scene.TILE
ntx = -(-w // tile)
props, normals = sr.scene.make_scene(n)
cam = sr.Camera()
cam.setAspect(w / h)
u = cam.uniforms(w, h)
dev = sr.Device(0)
lib = dev.lib
rerun = lib.splat_debug_rerun_tile_sort
pm = sr.SplatPropertyManager(dev, n)
pm.setFromArrays(props)
nbuf = dev.createBufferFrom(normals)
pbuf = pm.getPropertyBuffer()
r = sr.Renderer(dev, None, "rgba8unorm", n, records="lit")
for _ in range(30):
    r.render(u, pbuf, nbuf, None, w, h)
pairs = r.finish()
dev.sync()
b = r.binner
# the composite's inputs, out of the sort's way: a copy of the index lists
idx = dev.createBuffer(pairs * 4 + 64)
_lib.check(lib.splat_buf_copy(dev.ctx, idx.ptr, b.getTileIndicesBuffer().ptr, pairs * 4), dev.ctx)
dev.sync()
dev2 = sr.Device(0)  # a second context = a second stream of the same device
csr = sr.ComputeShaderRenderer(dev2, None, "rgba8unorm", earlyOut=True, recordFormat=_lib.RECORDS_LIT32)
records = r.projector.getRecordsBuffer()
cargs = (u, records, idx, records, records, b.getTileCountsBuffer(), b.getTileOffsetsBuffer(), tile, ntx, w, h)


def sort_once():
    _lib.check(rerun(dev.ctx, b._b), dev.ctx)


def comp_once():
    csr.render(*cargs)


def timed(fns, k):
    dev.sync(); dev2.sync()
    t0 = time.perf_counter()
    for _ in range(k):
        for f in fns:
            f()
    dev.sync(); dev2.sync()
    return (time.perf_counter() - t0) / k * 1e6


for f in (sort_once, comp_once):
    for _ in range(10):
        f()
timed([sort_once, comp_once], 50)
res = {}
for rep in range(3):
    res.setdefault("sort alone", []).append(timed([sort_once], iters))
    res.setdefault("composite alone", []).append(timed([comp_once], iters))
    res.setdefault("both, two streams", []).append(timed([sort_once, comp_once], iters))
print(f"{name}: pairs {pairs}; us per iteration (3 repeats of {iters})")
for k, v in res.items():
    print(f"  {k:22s} " + "  ".join(f"{x:7.1f}" for x in v))
s, c, both = (min(res[k]) for k in ("sort alone", "composite alone", "both, two streams"))
print(f"  serial sum {s + c:.1f} us; side by side {both:.1f} us; an overlap scheme can gain at most {s + c - both:.1f} us per frame")
