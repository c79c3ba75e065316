#!/usr/bin/env python3
"""What a CU's LDS delivers in returning atomics — the per-tile sort's own roof.

splat_debug_lds_rate (test build of the library): W four-wave workgroups per CU, every wave iters x 4 LDS instructions at
pseudo-random counters of its own 256-entry table — returning atomic adds (what k_tile_sort, k_tf_scatter and k_tf_downsweep2 rank
with), plain reads, non-returning atomic adds.  Prints lanes per cycle and CU at the device's clock.

    python tools/lds_atomic_rate.py [iters=2000]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SPLAT_LIB_PATH", os.path.join(ROOT, "splat_renderer_amd", "libsplat_hip_hooks.so"))
import splat_renderer_amd as sr
from splat_renderer_amd import _lib

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = sr.Device(0)
clock_ghz = float(os.environ.get("SPLAT_CLOCK_GHZ", "2.4"))  # MI355X peak engine clock
print(f"iters {iters} x 4 instructions per wave; lanes per cycle and CU at {clock_ghz} GHz")
for kind, name in ((0, "returning atomic add"), (1, "plain read"), (2, "non-returning atomic add")):
    row = []
    for wgs in (1, 2, 3, 4, 6, 8):
        ms = C.c_float()
        _lib.check(dev.lib.splat_debug_lds_rate(dev.ctx, kind, wgs, iters, C.byref(ms)), dev.ctx)
        lanes = wgs * 4 * 64 * iters * 4  # per CU
        row.append(f"{wgs} wg: {lanes / (ms.value * 1e-3 * clock_ghz * 1e9):6.2f} ({ms.value * 1e3:7.1f} us)")
    print(f"  {name:24s} " + "   ".join(row))
dev.destroy()
