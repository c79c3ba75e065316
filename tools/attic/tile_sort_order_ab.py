#!/usr/bin/env python3
"""Does the per-tile sort gain from taking its tiles longest-first?  Times the frame's bin_tile_sort stage (HIP events, all stages on)
with its workgroups in row-major order and in descending order of the tiles' list lengths (splat_debug_set_tile_sort_order; the order
comes from the frame's own counts — the best a history of the previous frame could give).  python tools/tile_sort_order_ab.py [C2]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splat_renderer_amd as sr
from splat_renderer_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n, w, h = sr.scene.CONFIGS[name]
props, normals = sr.scene.make_scene(n)
cam = sr.Camera()
cam.setAspect(w / h)
u = cam.uniforms(w, h)
dev = sr.Device(0)
pbuf, nbuf = dev.createBufferFrom(props), dev.createBufferFrom(normals)
r = sr.Renderer(dev, None, "rgba8unorm", n)
r.render(u, pbuf, nbuf, None, w, h)
r.finish()
ref8 = r.readPixels().copy()
counts = r.binner.getTileCountsBuffer().read(np.uint32)
orders = {"row-major": None, "descending list length": np.argsort(-counts.astype(np.int64), kind="stable").astype(np.uint32),
          "ascending list length": np.argsort(counts.astype(np.int64), kind="stable").astype(np.uint32)}
for label, order in orders.items():
    buf = dev.createBufferFrom(order) if order is not None else None
    _lib.check(dev.lib.splat_debug_set_tile_sort_order(dev.ctx, buf.ptr if buf else None), dev.ctx)
    for _ in range(5):
        r.render(u, pbuf, nbuf, None, w, h)
    _lib.check(dev.lib.splat_set_timing_stages(dev.ctx, 0xFFFFFFFF), dev.ctx)
    dev.setTiming(True)
    for _ in range(20):
        r.render(u, pbuf, nbuf, None, w, h)
    dev.sync()
    import ctypes as C
    cnt, tot = C.c_uint32(), C.c_double()
    _lib.check(dev.lib.splat_stage_time_stats(dev.ctx, _lib.STAGE_NAMES.index("bin_tile_sort"), C.byref(cnt), C.byref(tot)), dev.ctx)
    dev.setTiming(False)
    same = bool(np.array_equal(r.readPixels(), ref8))
    print(f"{name} tile sort order {label:24s}: bin_tile_sort {tot.value / max(cnt.value, 1) * 1e3:6.1f} us  same_image={same}")
    _lib.check(dev.lib.splat_debug_set_tile_sort_order(dev.ctx, None), dev.ctx)
