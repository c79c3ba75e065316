#!/usr/bin/env python3
"""What bench.py's own instrumentation costs a C2 frame: 100 frames with timing off against 100 with the composite's
event pair and the per-tile consumed counters on (the timed region's setting), alternating: python tools/timing_overhead.py"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import splat_renderer_amd as sr
from splat_renderer_amd import _lib
n, w, h = sr.scene.CONFIGS["C2"]
props, normals = sr.scene.make_scene(n)
cam = sr.Camera(); cam.setAspect(w / h); u = cam.uniforms(w, h)
dev = sr.Device(0); lib, ctx = dev.lib, dev.ctx
pbuf, nbuf = dev.createBufferFrom(props), dev.createBufferFrom(normals)
r = sr.Renderer(dev, None, "rgba8unorm", n)
def loop(k):
    dev.sync(); t0 = time.perf_counter()
    for _ in range(k): r.render(u, pbuf, nbuf, None, w, h)
    dev.sync(); return (time.perf_counter() - t0) / k * 1e3
for _ in range(10): r.render(u, pbuf, nbuf, None, w, h)
for rep in range(3):
    dev.setTiming(False)
    a = loop(100)
    _lib.check(lib.splat_set_timing_stages(ctx, 1 << _lib.STAGE_COMPOSITE), ctx)
    dev.setTiming(True)
    b = loop(100)
    dev.setTiming(False)
    print(f"timing off {a:.4f} ms/frame, composite events + consumed counters on {b:.4f}", flush=True)
