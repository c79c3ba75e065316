#!/usr/bin/env python3
"""One-off check far above the bench sizes: 40M splats @3840x2160 (72.9M tile-splat pairs) through both frame orders —
counts/offsets consistent, every list depth-ordered with index ties ascending, identical lists and images from the two
orders — and the frame time.  Size-independent properties only (no oracle)."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splat_renderer_amd as sr
n, w, h = 40_000_000, 3840, 2160
t0=time.time()
props, normals = sr.scene.make_scene(n)
print("scene", round(time.time()-t0,1), "s", flush=True)
cam = sr.Camera(); cam.setAspect(w/h); u = cam.uniforms(w, h)
dev = sr.Device(0)
pbuf, nbuf = dev.createBufferFrom(props), dev.createBufferFrom(normals)
for order in ("tileFirst", "sortFirst"):
    r = sr.Renderer(dev, None, "rgba8unorm", n, frameOrder=order)
    r.render(u, pbuf, nbuf, None, w, h)
    total = r.finish()
    counts = r.binner.getTileCountsBuffer().read(np.uint32)
    offsets = r.binner.getTileOffsetsBuffer().read(np.uint32)
    assert int(counts.sum(dtype=np.uint64)) == total
    assert np.array_equal(offsets, np.concatenate([[0], np.cumsum(counts, dtype=np.uint64)[:-1]]).astype(np.uint32))
    idx = r.binner.getTileIndicesBuffer().read(np.uint32, total)
    proj = r.projector.getRecordsBuffer().read(np.float32).reshape(n, 8)
    # (the frame leaves lit composite records {centre.xy, radius, depth | lit rgb, opacity} by default: depth is float 3;
    # ProjectedSplat records keep it in float 4)
    depth = proj[idx, 3 if r.recordFormat == sr._lib.RECORDS_LIT32 else 4]
    dd = np.diff(depth); starts = offsets[counts > 0][1:]
    bad = np.nonzero((dd < 0) | ((dd == 0) & (np.diff(idx.astype(np.int64)) <= 0)))[0] + 1
    assert np.isin(bad, starts).all()
    img = r.readPixels()
    if order == "tileFirst": first = (idx.copy(), img.copy())
    else: assert np.array_equal(idx, first[0]) and np.array_equal(img, first[1])
    dev.sync(); t0=time.perf_counter()
    for _ in range(5): r.render(u, pbuf, nbuf, None, w, h)
    dev.sync(); print(order, "pairs", total, "ms/frame", round((time.perf_counter()-t0)/5*1e3,3), flush=True)
    r.destroy()
print("ok")
