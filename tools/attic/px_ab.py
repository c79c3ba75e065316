#!/usr/bin/env python3
"""Times the composite alone on one frame's LIT records and lists (the frame's default inputs): early-out on / off, with and
without the consumed-entry counters (k_composite_px has a counting and a non-counting instantiation).
    python tools/px_ab.py [C2] [launches]
The kernel is chosen by SPLAT_COMPOSITE (pixel | quadrant), the build by SPLAT_LIB_PATH (tools/build_variant.sh)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splat_renderer_amd as sr
from splat_renderer_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n, w, h = sr.scene.CONFIGS[name]
props, normals = sr.scene.make_scene(n)
cam = sr.Camera()
cam.setAspect(w / h)
u = cam.uniforms(w, h)
dev = sr.Device(0)
pbuf, nbuf = dev.createBufferFrom(props), dev.createBufferFrom(normals)
r = sr.Renderer(dev, None, "rgba8unorm", n, records="lit")
r.render(u, pbuf, nbuf, None, w, h)
r.finish()
ref8 = r.readPixels().copy()
b = r.binner
ntx, nty = -(-w // 16), -(-h // 16)
records = r.projector.getRecordsBuffer()
args = (u, pbuf, b.getTileIndicesBuffer(), nbuf, records, b.getTileCountsBuffer(), b.getTileOffsetsBuffer(), 16, ntx, w, h)
cons = dev.createBuffer(ntx * nty * 16)
out = []
order_buf = None
if os.environ.get("PX_ORDER"):  # experiment: tiles in descending order of the entries they consume (from a counting run)
    csr = sr.ComputeShaderRenderer(dev, None, "rgba8unorm", earlyOut=True, recordFormat=_lib.RECORDS_LIT32)
    csr.consumedBuffer = cons
    cons.zero()
    csr.render(*args)
    dev.sync()
    used = cons.read(np.uint64).reshape(-1, 2)[:, 1].astype(np.int64)
    csr.consumedBuffer = None
    csr.destroy()
    mode = os.environ["PX_ORDER"]
    if mode == "desc":
        order = np.argsort(-used, kind="stable")
    elif mode == "classes":  # what a two-counter partition would give: long tiles first, each class in arrival (row-major) order
        order = np.concatenate([np.nonzero(used > 400)[0], np.nonzero((used <= 400) & (used > 250))[0], np.nonzero(used <= 250)[0]])
    else:
        order = np.argsort(used, kind="stable")
    order_buf = dev.createBufferFrom(order.astype(np.uint32))
    _lib.check(dev.lib.splat_debug_set_tile_order(dev.ctx, order_buf.ptr), dev.ctx)
for eo in (True, False):
    for counting in (False, True):
        csr = sr.ComputeShaderRenderer(dev, None, "rgba8unorm", earlyOut=eo, recordFormat=_lib.RECORDS_LIT32)
        csr.consumedBuffer = cons if counting else None
        for _ in range(3):
            csr.render(*args)
        cons.zero()
        _lib.check(dev.lib.splat_set_timing_stages(dev.ctx, 1 << _lib.STAGE_COMPOSITE), dev.ctx)
        dev.setTiming(True)
        for _ in range(launches):
            csr.render(*args)
        dev.sync()
        cnt, tot = C.c_uint32(), C.c_double()
        _lib.check(dev.lib.splat_stage_time_stats(dev.ctx, _lib.STAGE_COMPOSITE, C.byref(cnt), C.byref(tot)), dev.ctx)
        dev.setTiming(False)
        same = bool(np.array_equal(csr.readPixels(), ref8)) if eo else None
        st, used = (int(v) // launches for v in cons.read(np.uint64).reshape(-1, 2).sum(axis=0)) if counting else (0, 0)
        out.append(f"eo={int(eo)} count={int(counting)}: {tot.value / cnt.value * 1e3:7.1f} us" + (f" (staged {st} consumed {used})" if counting else "")
                   + (f" same_image={same}" if eo else ""))
        csr.consumedBuffer = None
        csr.destroy()
print(f"{name} order={os.environ.get('PX_ORDER', 'row-major')} kernel={os.environ.get('SPLAT_COMPOSITE', 'default')} lib={os.path.basename(os.environ.get('SPLAT_LIB_PATH', 'default'))}: " + " | ".join(out))
