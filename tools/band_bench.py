#!/usr/bin/env python3
"""Per-rank device time of the multi-GPU frame, measured on ONE GPU with virtual ranks (the
all-gather is a pre-built concat and is NOT included): python tools/band_bench.py [C2] [G ...] [disc]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splat_renderer_amd as sr
from splat_renderer_amd import dist

footprint = "disc" if "disc" in sys.argv[1:] else "isotropic"
argv = [a for a in sys.argv[1:] if a != "disc"]
name = argv[0] if argv else "C2"
worlds = [int(a) for a in argv[1:]] or [1, 2, 4, 8]
n, w, h = sr.scene.CONFIGS[name]
props, normals = sr.scene.make_scene(n)
cam = sr.Camera()
cam.setAspect(w / h)
u = cam.uniforms(w, h)
pt, nt = torch.from_numpy(props).cuda(), torch.from_numpy(normals).cuda()
for world in worlds:
    per = dist.shard_size(n, world)
    st = dist.HipStages(torch, 0, per * world, w, h, footprint=footprint)
    if os.environ.get("BAND_LAYOUT", "planes") != "interleaved":  # (bench.py's --layout: interleaved is its default, planes = lit once per property update)
        st.set_lit(pt.data_ptr(), nt.data_ptr(), n)
    brs = [dist.BandRenderer(st, n, w, h, r, world, None) for r in range(world)]
    for br in brs:
        st.project_slice(u, pt.data_ptr(), br.first, br.count, br.shard, nt.data_ptr())
    gathered = torch.cat([br.shard for br in brs], dim=0).contiguous()
    # balance bands by pairs per row from one calibration pass over all rows
    full = dist.BandRenderer(st, n, w, h, 0, 1, None)
    st.band_frame(gathered, per * world, pt.data_ptr(), nt.data_ptr(), 0, full.nty, full.image, settle=True)
    rows = st.row_pairs()
    bands = dist.balanced_rows(rows, world)
    out = []
    for r, br in enumerate(brs):
        r0, r1 = bands[r]
        for _ in range(3):
            st.project_slice(u, pt.data_ptr(), br.first, br.count, br.shard, nt.data_ptr())
            st.band_frame(gathered, per * world, pt.data_ptr(), nt.data_ptr(), r0, r1, br.image, settle=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 20
        for _ in range(K):
            st.project_slice(u, pt.data_ptr(), br.first, br.count, br.shard, nt.data_ptr())
            st.band_frame(gathered, per * world, pt.data_ptr(), nt.data_ptr(), r0, r1, br.image)
        torch.cuda.synchronize()
        out.append(((time.perf_counter() - t0) / K * 1e3, r1 - r0, st.kept))
        if os.environ.get("BAND_STAGES"):  # per-stage HIP-event intervals of this rank's band frame (every stage's events on: ~5 us of idle each)
            from splat_renderer_amd import _lib
            st.set_timing(True)
            for _ in range(5):
                st.band_frame(gathered, per * world, pt.data_ptr(), nt.data_ptr(), r0, r1, br.image)
            torch.cuda.synchronize()
            print(f"   rank {r}: " + "  ".join(f"{nm} {st.stage_avg_ms(i) * 1e3:.1f}" for i, nm in enumerate(_lib.STAGE_NAMES) if st.stage_avg_ms(i) > 0) + " us")
            st.set_timing(False)
    print(f"{name} {footprint} G={world}: per-rank ms (rows, kept): " + "  ".join(f"{t:.3f} ({rr},{k})" for t, rr, k in out)
          + f"   max {max(t for t, _, _ in out):.3f} ms  [+ all-gather of {per * st.rec_floats * 4 / 1e6:.0f} MB shards]")
    st.destroy()
