#!/usr/bin/env python3
"""Times k_composite alone on a built frame: early-out on/off, both modes (GPU box only)."""
import ctypes as C
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
r = sr.Renderer(dev, None, "rgba8unorm", n, records="projected")
r.render(u, pbuf, nbuf, None, w, h)
r.finish()
b = r.binner
counts = b.getTileCountsBuffer().read(np.uint32)
print(f"{name}: P={b.getTotalIndices()} tiles={counts.size} list len mean {counts.mean():.0f} max {counts.max()} p99 {np.percentile(counts, 99):.0f}")
cons = dev.createBuffer(counts.size * 16)  # two u64 per tile: {staged, consumed}
for mode in (0, 1):
    for eo in (True, False):
        csr = sr.ComputeShaderRenderer(dev, None, "rgba8unorm", mode=mode, earlyOut=eo)
        csr.consumedBuffer = cons
        args = (u, pbuf, b.getTileIndicesBuffer(), nbuf, r.projector.getProjectedBuffer(), b.getTileCountsBuffer(),
                b.getTileOffsetsBuffer(), 16, -(-w // 16), w, h)
        for _ in range(3):
            csr.render(*args)
        cons.zero()
        dev.setTiming(True)
        K = 10
        for _ in range(K):
            csr.render(*args)
        dev.sync()
        cnt, tot = C.c_uint32(), C.c_double()
        _lib.check(dev.lib.splat_stage_time_stats(dev.ctx, _lib.STAGE_COMPOSITE, C.byref(cnt), C.byref(tot)), dev.ctx)
        dev.setTiming(False)
        staged, used = (int(v) // K for v in cons.read(np.uint64).reshape(-1, 2).sum(axis=0))
        print(f"  mode {mode} early_out {eo}: {tot.value / cnt.value * 1e3:8.1f} us   entries staged {staged} consumed {used}")
        csr.consumedBuffer = None
        csr.destroy()
