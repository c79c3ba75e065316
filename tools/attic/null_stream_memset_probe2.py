#!/usr/bin/env python3
"""Probe, second form: a KERNEL of the library (splat_scan_u32, on a context's non-blocking stream) launched right after a plain
hipMemset (null stream) of the buffer it writes.  If the scan's output reads back as the fill pattern, the fill ran after the kernel.
    python tools/null_stream_memset_probe2.py [trials=200] [MiB=256] [contexts=6]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splat_renderer_amd as sr

hip = C.CDLL("libamdhip64.so")
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nctx = int(sys.argv[3]) if len(sys.argv) > 3 else 6
devs = [sr.Device(0) for _ in range(nctx)]  # (several contexts = several non-blocking streams: the runtime spreads them over its hardware queues)
n = 1024
src = np.ones(n, np.uint32)
lost, per_ctx = 0, [0] * nctx
for t in range(trials):
    d = devs[t % nctx]
    a = d.createBufferFrom(src)
    out = d.createBuffer(mib << 20)
    d.sync()
    rc = hip.hipMemset(C.c_void_p(out.ptr), 0xee, C.c_size_t(mib << 20))  # null stream; returns at once
    assert rc == 0
    sr._lib.check(d.lib.splat_scan_u32(d.ctx, C.c_void_p(a.ptr), C.c_void_p(out.ptr), n, None), d.ctx)  # the context's stream: writes out[0 .. n)
    d.sync()
    hip.hipDeviceSynchronize()
    got = out.read(np.uint32, n)
    if not np.array_equal(got, np.arange(n, dtype=np.uint32)):
        lost += 1
        per_ctx[t % nctx] += 1
    a.destroy()
    out.destroy()
print(f"{lost} of {trials} trials: the scan kernel's output (launched on a context's non-blocking stream right after a plain hipMemset of {mib} MiB "
      f"over its output buffer) was overwritten by the fill; per context {per_ctx}")
