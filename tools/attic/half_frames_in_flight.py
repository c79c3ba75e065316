#!/usr/bin/env python3
"""Experiment behind DESIGN.md §8.1: what would a frame gain if its splats went through project / scatter / tile-id sort in two
halves on two streams?  An upper bound with what exists: the two HALVES of the C2 scene as two whole frames (each with its own
per-tile sort and composite, which the real thing would do once, jointly) on one stream one after the other, and on two streams
side by side, against the one whole frame.  python tools/half_frames_in_flight.py [C2]"""
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
half = n // 2
K = 60


def make(dev, lo, hi):
    return dev, dev.createBufferFrom(props[lo:hi]), dev.createBufferFrom(normals[lo:hi]), sr.Renderer(dev, None, "rgba8unorm", hi - lo)


d0, d1 = sr.Device(0), sr.Device(0)
whole = make(d0, 0, n)
a_same, b_same = make(d0, 0, half), make(d0, half, n)  # both halves on ONE stream
b_other = make(d1, half, n)                            # the second half on another stream


def loop(sets, frames):
    for d, p, nb, r in sets:
        for _ in range(3):
            r.render(u, p, nb, None, w, h)
    for d in (d0, d1):
        d.sync()
    t0 = time.perf_counter()
    for _ in range(frames):
        for d, p, nb, r in sets:
            r.render(u, p, nb, None, w, h)
    for d in (d0, d1):
        d.sync()
    return (time.perf_counter() - t0) / frames * 1e3


for _ in range(2):
    print(f"{name}: the whole frame {loop([whole], K):.4f} ms | its two halves as two frames, one stream {loop([a_same, b_same], K):.4f} ms | "
          f"two streams {loop([a_same, b_other], K):.4f} ms   (each half frame has its own tile sort and composite)")
