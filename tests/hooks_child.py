#!/usr/bin/env python3
"""Scenarios that need the TEST HOOKS of include/splat.h (#ifdef SPLAT_TEST_HOOKS), run by tests/test_gpu_stages.py as a child
process with SPLAT_LIB_PATH = libsplat_hip_hooks.so — the shipped library neither exports the hooks nor carries their kernel
parameters, and the test process itself stays on the shipped library.

    python tests/hooks_child.py order_check
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (its HIP runtime first, as in tests/conftest.py)
import splat_renderer_amd as sr  # noqa: E402
from tests.helpers import assert_same, make_case, oracle_pipeline  # noqa: E402


def order_check():
    """A list is left as an out-of-lane-order rank would leave it (two neighbours swapped): the frame's report must carry
    the flag, the facade must get SPLAT_ERR_RETRY and render the frame again, the context must rank with ballots from
    then on, and the lists and the image that come back must be the oracle's.  First (host-synchronised) frame, sync-free
    frame, both size classes and a list long enough for the global-memory passes."""
    # (position of the swapped pair in the victim's list: 0; 63 | 64 and 255 | 256 are the pairs the check reads across a
    # wave / a round of the workgroup; 4000 lies in the long class's in-LDS range)
    cases = [(3000, 128, 96, 71, 1.0, False, 0), (20000, 640, 360, 72, 1.0, True, 63), (30000, 48, 32, 73, 8.0, True, 255),
             (30000, 48, 32, 75, 8.0, False, 4000), (6000, 16, 16, 74, 30.0, False, 1000)]
    for n, w, h, seed, rs, sync_free, position in cases:
        dev = sr.Device(0)  # (a failed check switches its context to ballots for good: one context per case)
        try:
            assert dev.rankStatus() == {"policy": "checked", "atomicsOrdered": True, "orderFaults": 0}
            props, normals, u = make_case(n, w, h, seed, rs)
            ref = oracle_pipeline(props, normals, u, w, h)
            victim = int(np.argmax(ref["counts"]))  # the longest list
            assert ref["counts"][victim] >= position + 2, (ref["counts"][victim], position)
            pbuf, nbuf = dev.createBufferFrom(props), dev.createBufferFrom(normals)
            r = sr.Renderer(dev, None, "rgba8unorm", n, frameOrder="tileFirst")
            if sync_free:  # a good first frame, then the fault hits a frame whose report is only read at the next call
                r.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
                good = r.readPixelsFloat().copy()
            dev.injectOrderFault(victim, position)
            r.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
            total = r.finish()  # learns of the failed check, renders the frame again (with ballots)
            # the facade books it as a MISRANKED frame, apart from capacity events (one flag for both hid it)
            assert r.framesMisranked == 1 and not r.previousFrameOverflowed
            st = dev.rankStatus()
            assert st["policy"] == "ballot" and st["orderFaults"] == 1, st
            assert total == ref["indices"].shape[0]
            assert_same(r.binner.getTileCountsBuffer().read(np.uint32), ref["counts"], ("order check", n, w, h, "counts"))
            assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], ("order check", n, w, h, "lists"),
                        offsets=ref["offsets"], keys=ref["keys"])
            img = r.readPixelsFloat()
            if sync_free:
                assert_same(img.view(np.uint32), good.view(np.uint32), ("order check", n, w, h, "image"))
            # and it stays right, without further faults, on the ballot ranking
            r.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
            assert r.finish() == total and dev.rankStatus()["orderFaults"] == 1
            assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], ("order check", n, w, h, "lists after"),
                        offsets=ref["offsets"], keys=ref["keys"])
            for o in (r, pbuf, nbuf):
                o.destroy()
        finally:
            dev.destroy()
    print(f"order_check ok: {len(cases)} cases")


if __name__ == "__main__":
    from splat_renderer_amd import _lib
    assert _lib.load().has_hooks, f"{_lib.LIB_PATH} is not the test build (make -C splat_renderer_amd/csrc hooks)"
    {"order_check": order_check}[sys.argv[1]]()
