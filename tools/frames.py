#!/usr/bin/env python3
"""Renders K frames of a config with NO timing events (for kernel-trace profiles of the undisturbed
frame): python tools/frames.py [C2] [K]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splat_renderer_amd as sr

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n, w, h = sr.scene.CONFIGS[name]
props, normals = sr.scene.make_scene(n)
cam = sr.Camera()
cam.setAspect(w / h)
u = cam.uniforms(w, h)
dev = sr.Device(0)
pbuf, nbuf = dev.createBufferFrom(props), dev.createBufferFrom(normals)
r = sr.Renderer(dev, None, "rgba8unorm", n)
for _ in range(k):
    r.render(u, pbuf, nbuf, None, w, h)
dev.sync()
print("pairs", r.finish())
