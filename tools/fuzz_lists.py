#!/usr/bin/env python3
"""Differential sweep of the frame's binner against the oracle far beyond the cases pytest carries: random splat counts,
screen sizes (1 x 1 to 4096 x 4096 pixels: 1 to 65536 tiles, i.e. every split of the tile-id bits and screens with and
without a second sort pass), splat scales and cameras; counts, offsets and index lists must be identical, and a second,
sync-free frame must reproduce them.  Test infrastructure (it loads oracle/):
    python tools/fuzz_lists.py [cases=300] [seed=1]"""
import os
import sys
import time

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import splat_renderer_amd as sr
from helpers import make_case, oracle_pipeline

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = sr.Device(0)
t0, pairs_total, with_second_pass, worst = time.time(), 0, 0, 0
for case in range(cases):
    # (every tenth case is a frame of up to 400k splats: from 64 blocks / partitions up the kernels deal their blocks per XCD)
    n = int(rng.integers(60000, 400000)) if case % 10 == 9 else int(rng.choice([rng.integers(1, 300), rng.integers(300, 8000), rng.integers(8000, 60000)]))
    side = lambda: int(rng.choice([rng.integers(1, 64), rng.integers(64, 700), rng.integers(700, 4097)]))
    w, h = side(), side()
    # keep the pair count in hand: splats scaled so that one covers a few tiles at most on this screen
    rs = float(rng.choice([0.02, 0.1, 0.5, 1.0, 3.0])) * min(1.0, 600.0 / max(w, h)) * (3.0 if n < 300 else 1.0)
    cam = dict(distance=float(rng.uniform(1.2, 6.0)), azimuth=float(rng.uniform(0, 6.28)), elevation=float(rng.uniform(-1.2, 1.2)))
    props, normals, u = make_case(n, w, h, 5000 + case, rs, camera=cam)
    ref = oracle_pipeline(props, normals, u, w, h)
    total = int(ref["indices"].shape[0])
    if total > 30_000_000:
        continue
    pbuf, nbuf = dev.createBufferFrom(props), dev.createBufferFrom(normals)
    order = "sortFirst" if case % 3 == 2 else "tileFirst"  # (the staged API's order of work: a third of the cases)
    r = sr.Renderer(dev, None, "rgba8unorm", n, frameOrder=order)
    tag = (case, n, w, h, rs, total, order)
    for rep in range(2):
        r.render(u, pbuf, nbuf, None, w, h)
        assert r.finish() == total, tag
        assert np.array_equal(r.binner.getTileCountsBuffer().read(np.uint32), ref["counts"]), tag
        assert np.array_equal(r.binner.getTileOffsetsBuffer().read(np.uint32), ref["offsets"]), tag
        if total:
            assert np.array_equal(r.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"]), tag
    tiles = -(-w // 16) * -(-h // 16)
    with_second_pass += int(tiles > 256)
    pairs_total += total
    worst = max(worst, total)
    for o in (r, pbuf, nbuf):
        o.destroy()
    if case % 50 == 49:
        print(f"{case + 1} cases, {pairs_total} pairs compared so far, {time.time() - t0:.0f} s", flush=True)
status = dev.rankStatus()
assert status["orderFaults"] == 0, f"a frame of this sweep was re-rendered after a failed order check: {status}"
print(f"ok: {cases} random frames (x2: first and sync-free), {with_second_pass} on screens of more than 256 tiles, {pairs_total} pairs in all, "
      f"largest frame {worst} pairs: counts, offsets and lists identical to the oracle's" + f"; ranking {status}")
