#!/usr/bin/env python3
"""Per-rank device time of a band frame WITHOUT an exchange: every rank projects all N splats itself and renders its
band of tile rows (splat_render_frame_planes with tile_row0/1).  One GPU, virtual ranks:
python tools/replicated_bench.py [C2] [G ...] [disc]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splat_renderer_amd as sr
from splat_renderer_amd import dist

footprint = "disc" if "disc" in sys.argv[1:] else "isotropic"
argv = [a for a in sys.argv[1:] if a != "disc"]
name = argv[0] if argv else "C2"
worlds = [int(a) for a in argv[1:]] or [1, 2, 4, 8]
n, w, h = sr.scene.CONFIGS[name]
props, normals = sr.scene.make_scene(n)
cam = sr.Camera()
cam.setAspect(w / h)
u = cam.uniforms(w, h)
dev = sr.Device(0)
pm = sr.SplatPropertyManager(dev, n)
pm.setFromArrays(props)
nbuf = dev.createBufferFrom(normals)
planes = pm.getLitPlanes(nbuf)
r = sr.Renderer(dev, None, "rgba8unorm", n, footprint=footprint)
r.render(u, planes, nbuf, None, w, h)
r.finish()
nty = -(-h // 16)
counts = r.binner.getTileCountsBuffer().read(np.uint32).reshape(nty, -1).sum(axis=1)
for world in worlds:
    bands = dist.balanced_rows(counts, world)
    out = []
    for (r0, r1) in bands:
        for _ in range(4):
            r.render(u, planes, nbuf, None, w, h, tileRows=(r0, r1))
        r.finish()
        dev.sync()
        t0 = time.perf_counter()
        K = 20
        for _ in range(K):
            r.render(u, planes, nbuf, None, w, h, tileRows=(r0, r1))
        dev.sync()
        out.append(((time.perf_counter() - t0) / K * 1e3, r1 - r0))
        r.finish()
    print(f"{name} {footprint} replicated projection G={world}: per-rank ms (rows): " + "  ".join(f"{t:.3f} ({rr})" for t, rr in out)
          + f"   max {max(t for t, _ in out):.3f} ms  [no exchange]")
