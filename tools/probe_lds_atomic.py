import sys, ctypes as C
sys.path.insert(0,'/root/repo')
import splat_renderer_amd as sr
from splat_renderer_amd import _lib
d = sr.Device(0)
m = C.c_uint64()
for i in range(3):
    _lib.check(d.lib.splat_probe_lds_atomic_order(d.ctx, C.byref(m)), d.ctx)
    print("mismatches", m.value)
