#!/usr/bin/env python3
"""Frame time of the property layouts (the reference's interleaved records | two planes | planes with the colour plane
pre-lit) x the records the frame's projector writes (lit composite records | ProjectedSplat), alternating on one box
in one process (same clocks): python tools/layout_ab.py [C2] [frames=100]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splat_renderer_amd as sr

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n, w, h = sr.scene.CONFIGS[name]
props, normals = sr.scene.make_scene(n)
cam = sr.Camera()
cam.setAspect(w / h)
u = cam.uniforms(w, h)
dev = sr.Device(0)
pm = sr.SplatPropertyManager(dev, n)
pm.setFromArrays(props)
nbuf = dev.createBufferFrom(normals)
renderers = {rec: sr.Renderer(dev, None, "rgba8unorm", n, records=rec) for rec in ("lit", "projected")}
layouts = (("interleaved", pm.getPropertyBuffer()), ("planes", pm.getPropertyPlanes()), ("planes, colour pre-lit", pm.getLitPlanes(nbuf)))
for _ in range(2):
    for label, pb in layouts:
        for rec, r in renderers.items():
            for _ in range(5):
                r.render(u, pb, nbuf, None, w, h)
            dev.sync()
            t0 = time.perf_counter()
            for _ in range(frames):
                r.render(u, pb, nbuf, None, w, h)
            dev.sync()
            print(f"{name} {label:24s} records={rec:9s} {(time.perf_counter() - t0) / frames * 1e3:.4f} ms/frame", flush=True)
