#!/usr/bin/env python3
"""Scenarios that need the TEST HOOKS of include/splat.h (#ifdef SPLAT_TEST_HOOKS), run by tests/test_gpu_stages.py as a child
process with SPLAT_LIB_PATH = libsplat_hip_hooks.so — the shipped library neither exports the hooks nor carries their kernel
parameters, and the test process itself stays on the shipped library.

    python tests/hooks_child.py order_check
    python tests/hooks_child.py long_class_skip
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


def long_class_skip():
    """A sync-free frame after one that had no tile beyond the per-tile sort's short class launches that class alone
    (tile_sort_launch: the long class's launch would find nothing to do).  A tile that outgrows the class in such a frame —
    here a few thousand splats moved onto one spot between two frames — is sorted by the short kernel's global-memory
    passes and counted, the frame after launches both classes again, and every frame's lists are the oracle's."""
    n, w, h = 60000, 1920, 1080  # (8160 tiles: beyond a band's single class)
    dev = sr.Device(0)
    try:
        flat, normals, u = make_case(n, w, h, 83, 0.4)
        ref_f = oracle_pipeline(flat, normals, u, w, h)
        # the splat in the middle of the fullest tile: 5000 others are moved onto it (a hair apart in depth)
        ntx = -(-w // 16)
        full = int(np.argmax(ref_f["counts"]))
        lst = ref_f["indices"][ref_f["offsets"][full]:ref_f["offsets"][full] + ref_f["counts"][full]]
        tx, ty = (full % ntx) * 16 + 8, (full // ntx) * 16 + 8
        c = ref_f["proj"][lst]
        anchor = int(lst[np.argmin((c[:, 0] - tx) ** 2 + (c[:, 1] - ty) ** 2)])
        piled = flat.copy()
        movers = np.setdiff1d(np.arange(n), [anchor])[:5000]
        rng = np.random.default_rng(5)
        piled[movers, 0:3] = flat[anchor, 0:3] + rng.uniform(-1e-4, 1e-4, (5000, 3)).astype(np.float32)
        ref_p = oracle_pipeline(piled, normals, u, w, h)
        assert ref_f["counts"].max() <= 2048 and ref_p["counts"].max() > 4096, (ref_f["counts"].max(), ref_p["counts"].max())
        pf, pp = int(ref_f["indices"].shape[0]), int(ref_p["indices"].shape[0])
        assert pp < pf + pf // 2, (pf, pp)  # (the piled frame stays within the sync-free frames' headroom: no overflow path)
        fbuf, pbuf, nbuf = dev.createBufferFrom(flat), dev.createBufferFrom(piled), dev.createBufferFrom(normals)
        r = sr.Renderer(dev, None, "rgba8unorm", n, frameOrder="tileFirst")

        def frame(buf, ref, launches, what):
            r.render(u, buf, nbuf, None, w, h, wantFloat=True)
            assert dev.tileSortLaunches() == launches, (what, dev.tileSortLaunches(), launches)
            img = r.readPixelsFloat().copy()
            assert not r.previousFrameOverflowed and r.framesMisranked == 0, what
            total = r.binner.getTotalIndices()
            assert total == ref["indices"].shape[0], (what, total)
            assert_same(r.binner.getTileCountsBuffer().read(np.uint32), ref["counts"], ("long class skip", what, "counts"))
            assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], ("long class skip", what, "lists"),
                        offsets=ref["offsets"], keys=ref["keys"])
            return img

        a0 = frame(fbuf, ref_f, 2, "first frame (host-synchronised: nothing known)")
        a1 = frame(fbuf, ref_f, 1, "second frame")
        assert_same(a1.view(np.uint32), a0.view(np.uint32), ("long class skip", "image 2"))
        b0 = frame(pbuf, ref_p, 1, "piled frame, short class alone")
        b1 = frame(pbuf, ref_p, 2, "piled frame again, both classes")
        assert_same(b1.view(np.uint32), b0.view(np.uint32), ("long class skip", "piled image"))
        a2 = frame(fbuf, ref_f, 2, "flat again (the frame before had long tiles)")
        a3 = frame(fbuf, ref_f, 1, "flat, settled")
        assert_same(a2.view(np.uint32), a0.view(np.uint32), ("long class skip", "image 5"))
        assert_same(a3.view(np.uint32), a0.view(np.uint32), ("long class skip", "image 6"))
        assert dev.rankStatus()["orderFaults"] == 0
        for o in (r, fbuf, pbuf, nbuf):
            o.destroy()
    finally:
        dev.destroy()
    print("long_class_skip ok: 6 frames")


if __name__ == "__main__":
    from splat_renderer_amd import _lib
    assert _lib.load().has_hooks, f"{_lib.LIB_PATH} is not the test build (make -C splat_renderer_amd/csrc hooks)"
    {"order_check": order_check, "long_class_skip": long_class_skip}[sys.argv[1]]()
