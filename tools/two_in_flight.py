#!/usr/bin/env python3
"""Experiment: frames/s of one renderer on one stream against two renderers on two streams used alternately
(two frames in flight: the hardware overlaps one frame's latency-bound kernels with the other's)."""
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
K = 60
devs = [sr.Device(0) for _ in range(3)]
sets = []
for d in devs:
    sets.append((d, d.createBufferFrom(props), d.createBufferFrom(normals), sr.Renderer(d, None, "rgba8unorm", n)))


def loop(active, frames):
    for d, p, nb, r in active:
        for _ in range(3):
            r.render(u, p, nb, None, w, h)
        d.sync()
    t0 = time.perf_counter()
    for k in range(frames):
        d, p, nb, r = active[k % len(active)]
        r.render(u, p, nb, None, w, h)
    for d, _, _, _ in active:
        d.sync()
    return (time.perf_counter() - t0) / frames * 1e3


for m in (1, 2, 3, 1, 2, 3):
    print(f"{name}: {m} frame(s) in flight: {loop(sets[:m], K):.4f} ms/frame")
