#!/usr/bin/env python3
"""Probe (HIP runtime through ctypes; no library code): is plain hipMemset of device memory ordered before work that is launched
next on a NON-BLOCKING stream?  A buffer is filled with zeros by hipMemset (null stream), then at once a small fill kernel
(hipMemsetAsync) on a non-blocking stream writes ones over its first words; after both have finished the words are read back.  Zeros =
the null-stream fill ran AFTER the other stream's: the two are not ordered (radix_sort.hip: sorter_reserve's workspace fill was such a hipMemset until
round 4).  python tools/null_stream_memset_probe.py [trials=200] [MiB=64]"""
import ctypes as C
import sys
import time

hip = C.CDLL("libamdhip64.so")
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nbytes = mib << 20


def ck(rc, what):
    if rc != 0:
        raise SystemExit(f"{what}: hip error {rc}")


ck(hip.hipSetDevice(0), "hipSetDevice")
streams = []  # (several: the runtime multiplexes streams onto a few hardware queues, and one that shares the null stream's is ordered by luck)
for _ in range(8):
    st_ = C.c_void_p()
    ck(hip.hipStreamCreateWithFlags(C.byref(st_), 1), "hipStreamCreateWithFlags(hipStreamNonBlocking)")
    streams.append(st_)
dev = C.c_void_p()
ck(hip.hipMalloc(C.byref(dev), C.c_size_t(nbytes)), "hipMalloc")
ones, back = C.c_void_p(), C.c_void_p()
ck(hip.hipHostMalloc(C.byref(ones), C.c_size_t(4096), 0), "hipHostMalloc")
ck(hip.hipHostMalloc(C.byref(back), C.c_size_t(4096), 0), "hipHostMalloc")
C.memset(ones, 0x01, 4096)
lost, call_us, per_stream = 0, [], [0] * 8
for t in range(trials):
    stream = streams[t % 8]
    ck(hip.hipMemset(dev, 0x7f, C.c_size_t(nbytes)), "hipMemset")  # (a different pattern first, so that a zero read back is this trial's)
    ck(hip.hipDeviceSynchronize(), "sync")
    t0 = time.perf_counter()
    ck(hip.hipMemset(dev, 0, C.c_size_t(nbytes)), "hipMemset")
    call_us.append((time.perf_counter() - t0) * 1e6)
    ck(hip.hipMemsetAsync(dev, 0x01, C.c_size_t(4096), stream), "hipMemsetAsync (a fill KERNEL) on the non-blocking stream")
    ck(hip.hipDeviceSynchronize(), "sync")
    ck(hip.hipMemcpy(back, dev, C.c_size_t(4096), 2), "hipMemcpy D2H")
    word = C.cast(back, C.POINTER(C.c_uint32))[0]
    if word != 0x01010101:
        lost += 1
        per_stream[t % 8] += 1
call_us.sort()
print(f"hipMemset of {mib} MiB returns in {call_us[len(call_us) // 2]:.0f} us (median; the fill itself needs >= {nbytes / 4.5e6:.0f} us at 4.5 TB/s); "
      f"{lost} of {trials} trials: the small fill launched after it on a non-blocking stream was overwritten by it; per stream {per_stream}")
