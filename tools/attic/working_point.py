#!/usr/bin/env python3
"""The reference's own working point: its demo scene (src/main.ts:55-81: two spheres and a box under smooth unions), the
point count its PointManager derives from it, and its frame (main.ts:146-190: fresh points, five projection steps onto
the surface, curvature, splat properties, one render) — per frame on one MI355X, generation and render apart.
The reference publishes no measurement; its planning document estimates 5.8 ms per frame for the render alone at about
this size (GPU_PIPELINE_PLAN.md:533-544, hardware unspecified): context, not a baseline.
    python tools/working_point.py [width=1920] [height=1080] [frames=200]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splat_renderer_amd as sr
from splat_renderer_amd import sdf

w = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 200
scene = sdf.SDFScene()
s1 = sdf.Sphere(id="sphere1", position=(0, 0, 0), radius=0.5)
b1 = sdf.Box(id="box1", position=(0.6, 0, 0), size=(0.3, 0.3, 0.3))
s2 = sdf.Sphere(id="sphere2", position=(0, 0.6, 0), radius=0.25)
scene.setRoot(sdf.smoothUnion(0.1, sdf.smoothUnion(0.15, s1, b1), s2))
dev = sr.Device(0)
src = sr.SdfSplatSource(dev, scene, seed=1)
n = src.numPoints
loop = sr.FrameLoop(dev, n, w, h)


def timed(fn, k):
    for _ in range(5):
        fn()
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    dev.sync()
    return (time.perf_counter() - t0) / k * 1e3


props, curv = src.step()
both = timed(lambda: loop.render(*src.step()), frames)
gen = timed(lambda: src.step(), frames)
ren = timed(lambda: loop.render(props, curv), frames)
loop.render(props, curv)
pairs = loop.renderer.finish()
print(f"reference demo scene, {n} splats @{w}x{h} ({pairs} tile-splat pairs): generation + render {both:.3f} ms per frame "
      f"({1e3 / both:.0f} frames/s); generation alone (fresh points, 5 surface steps, curvature, properties) {gen:.3f} ms, "
      f"render alone {ren:.3f} ms")
