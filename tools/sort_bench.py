#!/usr/bin/env python3
"""Times RadixSorter.sort() alone: python tools/sort_bench.py N [bits]  (GPU box only)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splat_renderer_amd as sr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = sr.Device(0)
rng = np.random.default_rng(1)
keys = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
s = sr.RadixSorter(dev, n)
kb = dev.createBufferFrom(keys)
pb = dev.createBufferFrom(np.arange(n, dtype=np.uint32))
import ctypes as C
lib = dev.lib
def reset():
    lib.splat_buf_upload  # noqa
    # device-to-device copy via hipMemcpy is not in the ABI; re-upload instead (outside the timed part)
    s.getKeysBuffer().write(keys)
    s.getPayloadBuffer().write(np.arange(n, dtype=np.uint32))
for it in range(3):
    reset()
    s.sort(n, 0, bits)
dev.sync()
ts = []
for it in range(10):
    reset()
    dev.sync()
    t0 = time.perf_counter()
    s.sort(n, 0, bits)
    dev.sync()
    ts.append(time.perf_counter() - t0)
order = s.getSortedIndicesBuffer().read(np.uint32, n)
ok = np.array_equal(order, np.argsort(keys & np.uint32((1 << bits) - 1 if bits < 32 else 0xFFFFFFFF), kind="stable"))
print(f"n={n} bits={bits} items={os.environ.get('SPLAT_RADIX_ITEMS','auto')} min {min(ts)*1e6:.1f} us  median {sorted(ts)[5]*1e6:.1f} us  correct={ok}")
