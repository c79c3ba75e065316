#!/usr/bin/env python3
"""Diagnostic (a -DPX_PROFILE build of the library, tools/build_variant.sh prof "-DPX_PROFILE"): per tile, how long the
consumer and builder waves of k_composite_px live and how long each waits at the chunk barriers (s_memtime ticks = shader
cycles).  python tools/px_profile.py [C2] [early_out=1]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splat_renderer_amd as sr
from splat_renderer_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
eo = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
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
b = r.binner
ntx, nty = -(-w // 16), -(-h // 16)
counts = b.getTileCountsBuffer().read(np.uint32)
args = (u, pbuf, b.getTileIndicesBuffer(), nbuf, r.projector.getRecordsBuffer(), b.getTileCountsBuffer(), b.getTileOffsetsBuffer(), 16, ntx, w, h)
cons = dev.createBuffer(ntx * nty * 16)
csr = sr.ComputeShaderRenderer(dev, None, "rgba8unorm", earlyOut=eo, recordFormat=_lib.RECORDS_LIT32)
csr.consumedBuffer = cons
for _ in range(3):
    csr.render(*args)
cons.zero()
csr.render(*args)
dev.sync()
v = cons.read(np.uint64).reshape(-1, 2)
f = lambda x, k: ((x >> np.uint64(16 * k)) & np.uint64(0xffff)).astype(np.float64)
nz = counts > 0
b_life, b_wait, b_fetch, b_chunks = f(v[:, 0], 0) * 16, f(v[:, 0], 1) * 16, f(v[:, 0], 2) * 16, f(v[:, 0], 3)
c_life, c_wait, c_trips, c_ntrips = f(v[:, 1], 0) * 16, f(v[:, 1], 1) * 16, f(v[:, 1], 2) * 16, f(v[:, 1], 3)
if len(sys.argv) > 3:  # per-tile arrays for offline analysis
    np.savez_compressed(sys.argv[3], counts=counts, b_life=b_life, b_wait=b_wait, b_chunks=b_chunks, c_life=c_life, c_wait=c_wait,
                        c_trips=c_trips, c_ntrips=c_ntrips)
print(f"{name} early_out={eo}: tiles with entries {nz.sum()} (cycles; a life beyond 1.05M cycles wraps)")
top = np.argsort(-c_life)[:12]
print("  the longest-lived tiles: tile count chunks | consumer life, barrier wait, in trips, trips | builder life, barrier wait")
for t in top:
    print(f"    {t:5d} {counts[t]:5d} {b_chunks[t]:3.0f} | {c_life[t]:7.0f} {c_wait[t]:7.0f} {c_trips[t]:7.0f} {c_ntrips[t]:4.0f} | {b_life[t]:7.0f} {b_wait[t]:7.0f}")
for lab, a in (("consumer life", c_life), ("consumer barrier wait", c_wait), ("consumer in trip loops", c_trips), ("consumer trips", c_ntrips),
               ("builder life", b_life), ("builder barrier wait", b_wait), ("builder record wait", b_fetch), ("builder chunks built", b_chunks)):
    x = a[nz]
    print(f"  {lab:24s} mean {x.mean():9.0f}  p50 {np.percentile(x, 50):9.0f}  p90 {np.percentile(x, 90):9.0f}  p99 {np.percentile(x, 99):9.0f}  max {x.max():9.0f}  sum {x.sum():.3e}")
print(f"  cycles per trip {c_trips[nz].sum() / max(c_ntrips[nz].sum(), 1):.0f}; record wait per chunk built {b_fetch[nz].sum() / max(b_chunks[nz].sum(), 1):.0f}; "
      f"builder busy-but-not-waiting per chunk {(b_life[nz].sum() - b_wait[nz].sum() - b_fetch[nz].sum()) / max(b_chunks[nz].sum(), 1):.0f}")
# by length: do the long tiles — the kernel's duration — run faster than the rest?
for lo_, hi_ in ((1, 8), (9, 15), (16, 23), (24, 1000)):
    m = nz & (b_chunks >= lo_) & (b_chunks <= hi_)
    if m.any():
        print(f"  tiles of {lo_}..{hi_} chunks: {m.sum():5d}  consumer life mean {c_life[m].mean():8.0f} max {c_life[m].max():8.0f}  cycles per trip "
              f"{c_trips[m].sum() / max(c_ntrips[m].sum(), 1):5.0f}  life per chunk {c_life[m].sum() / b_chunks[m].sum():6.0f}  barrier wait share {c_wait[m].sum() / c_life[m].sum():.2f}")
