#!/usr/bin/env python3
"""Renders K frames of a config with NO timing events (for kernel-trace profiles of the undisturbed
frame): python tools/frames.py [C2] [K] [records=lit|projected] [layout=interleaved|planes|prelit] [footprint]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splat_renderer_amd as sr

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 30
records = sys.argv[3] if len(sys.argv) > 3 else "lit"
layout = sys.argv[4] if len(sys.argv) > 4 else "interleaved"
footprint = sys.argv[5] if len(sys.argv) > 5 else "isotropic"
n, w, h = sr.scene.CONFIGS[name]
props, normals = sr.scene.make_scene(n)
cam = sr.Camera()
cam.setAspect(w / h)
u = cam.uniforms(w, h)
dev = sr.Device(0)
pm = sr.SplatPropertyManager(dev, n)
pm.setFromArrays(props)
nbuf = dev.createBufferFrom(normals)
pbuf = {"interleaved": pm.getPropertyBuffer, "planes": pm.getPropertyPlanes, "prelit": lambda: pm.getLitPlanes(nbuf)}[layout]()
r = sr.Renderer(dev, None, "rgba8unorm", n, records=records, footprint=footprint, writeProjected=footprint != "disc")
for _ in range(5):
    r.render(u, pbuf, nbuf, None, w, h)
dev.sync()
t0 = time.perf_counter()
for _ in range(k):
    r.render(u, pbuf, nbuf, None, w, h)
dev.sync()
dt = (time.perf_counter() - t0) / k * 1e3
print(f"{name} records={records} layout={layout} footprint={footprint}: {dt:.4f} ms/frame, pairs {r.finish()}")
