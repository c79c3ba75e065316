#!/usr/bin/env python3
"""Frame time with the reference's interleaved property records vs the two-plane layout vs planes with the colour
plane pre-lit, alternating on
one renderer (same box, same clocks): python tools/layout_ab.py [C2]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splat_renderer_amd as sr

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n, w, h = sr.scene.CONFIGS[name]
props, normals = sr.scene.make_scene(n)
cam = sr.Camera()
cam.setAspect(w / h)
u = cam.uniforms(w, h)
dev = sr.Device(0)
pm = sr.SplatPropertyManager(dev, n)
pm.setFromArrays(props)
nbuf = dev.createBufferFrom(normals)
r = sr.Renderer(dev, None, "rgba8unorm", n)
for label, pb in (("interleaved", pm.getPropertyBuffer()), ("planes", pm.getPropertyPlanes()), ("planes, colour pre-lit", pm.getLitPlanes(nbuf))) * 2:
    for _ in range(5):
        r.render(u, pb, nbuf, None, w, h)
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(100):
        r.render(u, pb, nbuf, None, w, h)
    dev.sync()
    print(name, label, round((time.perf_counter() - t0) / 100 * 1e3, 4), "ms/frame", flush=True)
