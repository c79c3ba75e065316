#!/usr/bin/env python3
"""bench.py — Msplats/s + frames/s of the tile-raster hot path on synthetic random-Gaussian scenes.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one frame (project -> depth keys -> radix sort -> tile bin -> composite) over inputs
already resident in HBM.  N=1: workload C2 (5M Gaussians @1920x1080, the configuration
BASELINE.json's metric is quoted on).  N>1: the same frame sharded by tile-row bands (strong
scaling: total work fixed) — with one RCCL all-gather of projected splats (north_star's cut), or
with every rank projecting all splats for its own band and nothing exchanged; both are timed on
the node after warm-up and the faster runs the timed region (--exchange auto, the default; the
line says which, and both trial times).  Prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import splat_renderer_amd as sr  # noqa: E402
from splat_renderer_amd import _lib, dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec); ~6.3 TB/s achievable


def composite_alg_bytes(p_used, width, height):
    """SURVEY §8d, unchanged: per CONSUMED list entry 4 (idx) + 32 (ProjectedSplat) + 16 (colour vec4) +
    16 (normal vec4) = 68 B in the reference's own layouts, plus 4 B per rgba8 pixel written."""
    return 68 * p_used + 4 * width * height


def composite_traffic_model(p_staged, width, height, records, prelit, disc=False):
    """What the kernel as built is expected to move: per STAGED entry (k_composite_px: chunks of 32, fetched up to three
    chunks past the last one walked; k_composite: 256-entry batches) the 4-byte index and its gathers — one 32-byte lit
    composite record; or ProjectedSplat 32 + colour 16 (+ normal 16 unless pre-lit)."""
    per = 4 + (32 if (records == "lit" and not disc) else (48 if prelit else 64))
    return per * p_staged + 4 * width * height


def composite_model_floor(p_staged, alg_bytes, copy_gbs, ahead):
    """DESIGN.md §4 "the composite's floor": the time k_composite_px AS DESIGNED (a builder and a consumer wave per 16x16 tile,
    chunks of 32 entries, a trip = every lane of the consumer takes its next entry) cannot beat on this frame, from counted work
    and measured constants (profiles/r04_a_px_list_cap_experiment_C2.txt, r04_o_trip_cost_C2.txt):
      fixed      5.5 us  an empty launch of the grid (8161 workgroups at C2) incl. its drain
                 4.5 us  a tile's start is three DEPENDENT memory round trips (tile descriptor -> list indices -> records) before its
                         first build: ~1.5 us each under load, overlapped with nothing of the same tile
      issue      chunks x (150 builder + trips_per_chunk x 41 consumer) wave-instructions, at one instruction per SIMD and cycle
                 (1024 SIMDs x 2.4 GHz): every wave instruction of the kernel issued back to back with no stall at all
      hbm        the algorithmic bytes at the box's measured copy rate
    floor = max(fixed + issue, hbm).  The kernel's measured time follows 22 us + 0.58 us per 1000 chunks (early-out on): 2.3x the
    issue term — VALU busy 0.50, waves parked at barriers / LDS latency the rest — over a 22 us intercept (the fixed part, the tiles'
    first two chunks, and the ordering workgroup, which alone takes 22 us and runs beside the tiles)."""
    chunks = p_staged / 32.0
    trips = 8.8 if ahead == 1 else 6.4  # per chunk, measured at C2 (3.97e5 / 2.91e5 trips over 45k chunks)
    instr = chunks * (150.0 + trips * (41.0 if ahead == 1 else 45.0))
    issue_us = instr / (1024 * 2400.0)
    hbm_us = alg_bytes / (copy_gbs * 1e3)
    floor = max(10.0 + issue_us, hbm_us)
    return {"model_floor_us": round(floor, 1), "fixed_us": 10.0, "issue_us": round(issue_us, 1), "hbm_at_copy_rate_us": round(hbm_us, 1),
            "chunks": round(chunks), "frac_of_peak_at_floor": round(alg_bytes / (floor * 1e-6) / 1e9 / HBM_PEAK_GBS, 3),
            "fit_us": round(22.0 + 0.58 * chunks / 1000.0, 1) if ahead == 1 else None,
            "note": "DESIGN.md section 4: what this design cannot beat on this frame (every wave instruction issued back to back on every "
                    "SIMD, three dependent round trips per tile start, an empty launch); fit_us = the measured line 22 us + 0.58 us per "
                    "1000 chunks (early-out on)"}


def per_kernel_rooflines(stage_ms, n, pairs, p_used, width, height, lit, disc, pmc=None):
    """every kernel group of the tile-first frame against the 8 TB/s HBM roof, from the bytes each is
    built to move (the tile-first frame's own byte table, which replaces SURVEY §8d's sort / count / fill
    rows — the composite row IS SURVEY §8d's) and its HIP-event interval in the all-stages loop (each interval carries
    the ~5 us of idle its event pair costs, so these fractions read slightly low next to a rocprofv3 kernel trace)."""
    rows = [
        ("project", "k_project_hist (projector + depth keys + tile ranges + first-pass histogram" + (", lit composite records)" if lit else ")"),
         (88 if lit else 72 if disc else 56) * n, "B/splat: 16 pos/radius (+ 32 colour, normal) read; 32-byte record + 4 key + 4 range written"),
        ("bin_scatter", "k_tf_scatter (pair expansion fused with the first tile-id sort pass)", 8 * n + 9 * pairs,
         "8 B/splat read (key, range) + 9 B/pair written (1 B high tile digit, 8 B key+index)"),
        ("bin_second_pass", "k_tf_upsweep2 + k_radix_rowscan + k_tf_downsweep2 (incl. the tile offsets)", 18 * pairs, "1 + 9 B/pair read, 8 B/pair written"),
        ("bin_tile_sort", "k_tile_sort x2 (PerTileSorter; also checks every list's order)", 12 * pairs, "8 B/pair read, 4 B/pair (index list) written"),
        ("composite", "k_composite_px / k_composite", composite_alg_bytes(p_used, width, height), "SURVEY 8d: 68 B x consumed entry + 4 B x pixel"),
    ]
    out = {}
    for key, kernels, nbytes, what in rows:
        ms = stage_ms.get(key, 0.0)
        if ms <= 0:
            continue
        gbs = nbytes / (ms / 1e3) / 1e9
        out[key] = {"kernels": kernels, "bytes_per_frame": int(nbytes), "bytes_model": what, "ms": round(ms, 4), "achieved_GBps": round(gbs, 1),
                    "frac": round(gbs / HBM_PEAK_GBS, 4)}
    # The per-tile sort moves its bytes once and spends its time in LDS: its own roof is the LDS pipe.  From
    # committed rocprofv3 counters (profiles/traffic.json names the csv), not measured in this run: the fraction of the kernel's
    # duration during which the CUs' LDS arrays were busy, and how much of that was bank conflicts.
    if pmc and pmc.get("k_tile_sort_lds") and "bin_tile_sort" in out:
        t = pmc["k_tile_sort_lds"]
        out["bin_tile_sort"]["lds"] = {"bound": "lds", "frac": t["lds_pipe_frac"], "conflict_share_of_array_cycles": t["conflict_share_of_array_cycles"],
                                       "lds_array_cycles_per_frame": t["lds_array_cycles_per_frame"],
                                       "lds_instructions_per_frame": t["lds_instructions_per_frame"], "source": t["source"],
                                       "reading": "well below the pipe's capacity with half of the busy cycles lost to conflicts: the kernel waits on its "
                                                  "chain of barriers and LDS round trips (SQ_WAIT_ANY 0.53 of its wave cycles), not on LDS bandwidth"}
    return out


def frame_alg_bytes(n, n_sorted, tiles, pairs, p_used, width, height, disc=False):
    """SURVEY §8d whole-frame model: project+key 56N, sort 68Np, count 20N, scan 8T, fill 20N+4P,
    composite 68 P_used + 4WH.  The oriented-disc projector also reads the normal (16) and writes
    the disc record (32) and, as benched, leaves the ProjectedSplat (32) out: 72N."""
    return (72 if disc else 56) * n + 68 * n_sorted + 20 * n + 8 * tiles + 20 * n + 4 * pairs + composite_alg_bytes(p_used, width, height)


def cpu_baseline(name, props, normals, u, width, height):
    """The oracle's whole frame (model A, front-to-back, early-out on) timed on this box's host
    cores.  Sample = ONE full frame of the same workload (about 7 s single-threaded for C2), once
    on 1 thread (the reference's path is one JS thread) and once on all cores."""
    from oracle import oracle as O  # the checker, timed as the reported CPU baseline only
    n = props.shape[0]
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    r1 = O.frame(u, props, normals, width, height, threads=1, want_f32=False)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    rn = O.frame(u, props, normals, width, height, threads=cores, want_f32=False)
    tn = time.perf_counter() - t0
    # model B = the CPU restatement of SequentialRenderer.ts (one oriented quad per splat, back to front):
    # the renderer BASELINE.json names; it reuses model A's sort (reverse order), so only its raster is timed
    proj = O.project(u, props)
    keys, pay = O.extract_keys(proj)
    _, order = O.sort_pairs(keys, pay)
    t0 = time.perf_counter()
    _, seq8 = O.sequential(u, props, normals, order[::-1].copy(), width, height)
    tb = time.perf_counter() - t0
    return {"sequential_u8": seq8,
            "value": n / t1 / 1e6, "unit": "Msplats/s", "cores": 1, "kind": "port",
            "sample": f"1 full frame of {name} (N={n}, {width}x{height}), oracle/oracle.c model A, 1 thread",
            "seconds": round(t1, 3), "stage_ms": [round(x, 1) for x in r1["stage_ms"]],
            "all_cores": {"value": n / tn / 1e6, "cores": cores, "seconds": round(tn, 3),
                          "note": "projector + composite banded over pthreads; the oracle's sort and binSorted stay serial: "
                                  f"{round((rn['stage_ms'][2] + rn['stage_ms'][3]) / 1e3, 2)} s of these {round(tn, 2)} s "
                                  "(its own stage_ms[2] + stage_ms[3]) — the ratio to 1 thread says nothing about what the host could do",
                          "stage_ms": [round(x, 1) for x in rn["stage_ms"]]},
            "model_b_sequential_raster": {"seconds": round(tb, 3), "cores": 1,
                                          "note": "oracle restatement of SequentialRenderer.ts raster only (sorted order given)"},
            "frame_u8": rn["out_u8"]}


def measured_copy_ceiling(dev):
    """the box's own device-to-device copy rate (read + write bytes per second), as a second
    denominator next to the 8 TB/s specification."""
    nbytes = 1 << 30
    a, b = dev.createBuffer(nbytes), dev.createBuffer(nbytes)
    a.zero()
    for _ in range(20):  # (~8 ms: the device's clocks take a few ms of load to reach what they sustain, tools/warm_probe.py)
        _lib.check(dev.lib.splat_buf_copy(dev.ctx, b.ptr, a.ptr, nbytes), dev.ctx)
    dev.sync()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        _lib.check(dev.lib.splat_buf_copy(dev.ctx, b.ptr, a.ptr, nbytes), dev.ctx)
    dev.sync()
    dt = time.perf_counter() - t0
    a.destroy()
    b.destroy()
    return 2.0 * nbytes * reps / dt / 1e9


def load_traffic(config):
    """PMC-derived figures of k_composite for this configuration, collected under rocprofv3 in separate --pmc passes
    and committed (profiles/traffic.json names the csv each comes from): they are NOT measured in this run."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(config) or {}
    except Exception:
        return {}


def time_composite_alone(dev, r, u, pbuf, nbuf, width, height, tile, early_out, launches=5):
    """k_composite by itself on the frame's own records and lists (outside the timed region): HIP-event duration per
    launch and the {staged, consumed} entry counts.  early_out=False is SURVEY §8d's composite-only figure
    (P_used = P: every entry of every list, 68 P + 4WH algorithmic bytes)."""
    lib, ctx = dev.lib, dev.ctx
    ntx, nty = -(-width // tile), -(-height // tile)
    csr = sr.ComputeShaderRenderer(dev, None, "rgba8unorm", earlyOut=early_out, footprint=r.footprint, recordFormat=r.recordFormat)
    csr.consumedBuffer = dev.createBuffer(ntx * nty * 16)
    b = r.binner
    records = r.projector.getRecordsBuffer()  # (lit composite records or ProjectedSplat: r.recordFormat)
    if r.footprint == _lib.FOOTPRINT_DISC:
        raise SystemExit("time_composite_alone: isotropic frames only")
    props = pbuf if not isinstance(pbuf, sr.host.PropertyPlanes) else None
    if props is None and r.recordFormat != _lib.RECORDS_LIT32:
        return None
    args = (u, props if props is not None else records, b.getTileIndicesBuffer(), nbuf if nbuf is not None else records, records,
            b.getTileCountsBuffer(), b.getTileOffsetsBuffer(), tile, ntx, width, height)
    csr.render(*args)  # warm
    csr.consumedBuffer.zero()
    _lib.check(lib.splat_set_timing_stages(ctx, 1 << _lib.STAGE_COMPOSITE), ctx)
    dev.setTiming(True)
    for _ in range(launches):
        csr.render(*args)
    dev.sync()
    cnt, tot = C.c_uint32(), C.c_double()
    _lib.check(lib.splat_stage_time_stats(ctx, _lib.STAGE_COMPOSITE, C.byref(cnt), C.byref(tot)), ctx)
    dev.setTiming(False)
    cons = csr.consumedBuffer.read(np.uint64).reshape(-1, 2).sum(axis=0) / launches
    csr.destroy()
    return {"ms": tot.value / max(cnt.value, 1), "staged": float(cons[0]), "consumed": float(cons[1])}


def orbit_leg(dev, n, width, height, tile, args, pbuf, nbuf, static_ms, static_composite_ms, frames=60, degrees=0.5):
    """K frames of an orbit, the camera turned by `degrees` of azimuth between frames by OrbitCameraController (a drag of
    degrees / 0.005 rad-per-pixel pixels: OrbitCameraController.ts:12,47-49), through FrameLoop; ms/frame, the composite's own
    duration (events on every fourth frame, as in the timed region), frames that had to be rendered again.  Then the SAME camera
    path with the composite's look-ahead bound switched off — to say which mechanism a slowdown comes from — and the default once
    more (the legs run one after the other: the second default says how much they drift)."""
    lib, ctx = dev.lib, dev.ctx
    loop = sr.FrameLoop(dev, n, width, height, tile, records=args.records)
    ctrl = sr.OrbitCameraController(loop.camera)
    dx = np.radians(degrees) / ctrl.rotationSpeed
    az0, el0 = loop.camera.azimuth, loop.camera.elevation
    warm = 12
    # the camera path once: the host's matrix arithmetic is not what is measured, and every leg renders the same frames
    ctrl.onMouseDown(sr.MouseEvent(0.0, 0.0, 0))
    uniforms = []
    for k in range(warm + frames):
        ctrl.onMouseMove(sr.MouseEvent((k + 1) * dx, 0.0, 0))
        uniforms.append(loop.camera.uniforms(width, height, time=k / 60.0).copy())
    ctrl.onMouseUp()
    turned = float(np.degrees(loop.camera.azimuth - az0))
    loop.camera.azimuth, loop.camera.elevation = az0, el0

    def run(us):
        again = 0
        dev.sync()
        t0 = time.perf_counter()
        for uk in us:
            loop.renderer.previousFrameOverflowed = False
            loop.renderer.render(uk, pbuf, nbuf, None, width, height)
            again += int(loop.renderer.previousFrameOverflowed)
        dev.sync()
        return (time.perf_counter() - t0) / len(us) * 1e3, again

    def leg(**opts):
        dev.compositeOptions(**opts)  # (also forgets the composite's history: the warm frames re-learn it under the moving camera)
        run(uniforms[:warm])
        _lib.check(lib.splat_set_timing_stages(ctx, 1 << _lib.STAGE_COMPOSITE), ctx)
        _lib.check(lib.splat_set_timing_sampling(ctx, 4), ctx)
        dev.setTiming(True)
        ms, again = run(uniforms[warm:])
        cnt, tot = C.c_uint32(), C.c_double()
        _lib.check(lib.splat_stage_time_stats(ctx, _lib.STAGE_COMPOSITE, C.byref(cnt), C.byref(tot)), ctx)
        dev.setTiming(False)
        _lib.check(lib.splat_set_timing_sampling(ctx, 1), ctx)
        loop.renderer.finish()
        return {"ms_per_step": round(ms, 4), "composite_ms": round(tot.value / max(cnt.value, 1), 4), "frames_rendered_again": again}
    run(uniforms)  # (the whole path once, untimed: a new renderer's first frames size its buffers)
    out = leg()
    out["pairs_P_of_the_last_view"] = loop.renderer.finish()
    out.update({"frames": frames, "degrees_per_frame": degrees, "degrees_turned_incl_warmup": round(turned, 2),
                "value": n / out["ms_per_step"] / 1e3, "unit": "Msplats/s",
                "static_ms_per_step": round(static_ms, 4), "static_composite_ms": round(static_composite_ms, 4),
                "over_static": round(out["ms_per_step"] / static_ms, 4),
                "note": "FrameLoop + OrbitCameraController, frames enqueued back to back (no host sync between frames); the static figures are "
                        "the timed region's; the views of an orbit differ from the static view in their pair totals too"})
    # the same renderer standing still at the orbit's last view: what of the difference to the timed region is the VIEW (the pair
    # total and its distribution over the tiles change with the camera), what the motion
    dev.compositeOptions()
    run([uniforms[-1]] * 20)
    still_ms, _ = run([uniforms[-1]] * 40)
    out["standing_still_at_the_last_view_ms_per_step"] = round(still_ms, 4)
    out["over_standing_still_at_the_last_view"] = round(out["ms_per_step"] / still_ms, 4)
    out["without_lookahead_bound"] = leg(predict=False)
    out["default_again"] = leg()
    out["frames_misranked"] = loop.renderer.framesMisranked
    dev.compositeOptions()
    loop.destroy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2", choices=sorted(sr.scene.CONFIGS))
    ap.add_argument("--layout", default="interleaved", choices=["planes", "interleaved"],
                    help="splat properties as the reference's interleaved 32-byte records + vec4 normals (default: shading happens "
                         "inside the timed frame) or as two vec4 planes with the colour plane lit once, outside the frame")
    ap.add_argument("--records", default="lit", choices=["lit", "projected"],
                    help="what the frame's projector writes and the composite gathers per staged list entry: the 32-byte lit "
                         "composite record (one line) or the reference's ProjectedSplat + colour + normal (three lines)")
    ap.add_argument("--footprint", default="isotropic", choices=["isotropic", "disc"],
                    help="isotropic: ComputeShaderRenderer's screen-space Gaussian (SURVEY §8a contract 3, the headline); "
                         "disc: SequentialRenderer's oriented disc (parity vs the CPU rasteriser of SequentialRenderer.ts; "
                         "48-byte exchange records at N>1)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "allgather", "none"],
                    help="multi-GPU: allgather = every rank projects 1/N of the splats, ONE RCCL all-gather of the records, band work "
                         "(north_star's cut); none = every rank projects all splats itself and renders its band (no collective: every "
                         "rank holds the scene anyway); auto (default) = both cuts are set up and timed on THIS node after warm-up (8 frames "
                         "each, slowest rank decides, all ranks take the same one) and the faster runs the timed region — which one ran "
                         "is config.exchange_chosen, both trial times are config.frame_loop_trial_ms")
    ap.add_argument("--collective", default="abi", choices=["abi", "torch"],
                    help="multi-GPU: who issues the frame's all-gather — the C ABI's own RCCL communicator (splat_comm_init / "
                         "splat_allgather_records, default; falls back to torch if RCCL cannot be bound) or torch.distributed")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="multi-GPU: do not overlap the next frame's projection + all-gather with the current frame's band work")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra measurements after the timed region (early-out-off composite, two frames in flight): for kernel "
                         "traces of the timed frames alone — frames in flight on two streams overlap, which inflates every kernel's duration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--band-path", action="store_true", default=os.environ.get("SPLAT_BENCH_BAND_PATH", "0") == "1",
                    help="run the N > 1 code path (slice projection, the all-gather through the communicator, band frame, the exchange's "
                         "self-checks) whatever WORLD_SIZE is: with one rank the band is the whole screen and the all-gather a one-rank "
                         "collective.  A rehearsal of run_multi on one GPU (VERDICT r4 item 1), not the N = 1 headline")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    name = args.config
    n, width, height = sr.scene.CONFIGS[name]
    tile = sr.scene.TILE
    ntx, nty = -(-width // tile), -(-height // tile)
    props, normals = sr.scene.make_scene(n)
    cam = sr.Camera()
    cam.setAspect(width / height)
    u = cam.uniforms(width, height)
    workload = f"{name}: {n} synthetic Gaussians @{width}x{height}, {tile}x{tile} tiles"

    if world == 1 and not args.band_path:  # one GPU is one GPU, launched through torchrun or not (the band path is for N > 1)
        result = run_single(args, name, n, width, height, tile, ntx, nty, props, normals, u, workload)
    else:
        result = run_multi(args, name, n, width, height, tile, ntx, nty, props, normals, u, workload, rank, local_rank, world)
    if rank == 0:
        print(json.dumps(result))


def run_single(args, name, n, width, height, tile, ntx, nty, props, normals, u, workload):
    dev = sr.Device(0)
    lib, ctx = dev.lib, dev.ctx
    pm = sr.SplatPropertyManager(dev, n)
    pm.setFromArrays(props)
    nbuf = dev.createBufferFrom(normals)
    # default: the reference's layouts (32-byte interleaved property records + vec4 normals), shading inside the frame.
    # --layout planes: two vec4 planes (what updatePlanesFromCurvature writes), the colour plane lit once per property
    # update, outside the frame
    prelit = args.layout == "planes"
    pbuf = pm.getLitPlanes(nbuf) if prelit else pm.getPropertyBuffer()
    disc = args.footprint == "disc"
    # (a disc frame's composite reads the disc records: the ProjectedSplat by-product is left out)
    r = sr.Renderer(dev, None, "rgba8unorm", n, tile, footprint=args.footprint, writeProjected=not disc, records=args.records)

    def frame():
        r.render(u, pbuf, nbuf, None, width, height)

    def stage_avg(sid):
        cnt, tot = C.c_uint32(), C.c_double()
        _lib.check(lib.splat_stage_time_stats(ctx, sid, C.byref(cnt), C.byref(tot)), ctx)
        return tot.value / max(cnt.value, 1)

    # the box's copy rate first: a second denominator for the rooflines — and ~12 ms of load, after which the W warm-up frames
    # and the timed region run at the clocks the device sustains (after idle the first ~50 frames of C2 take 0.345 ms, every
    # later one 0.3335: tools/warm_probe.py, profiles/r03_l_clock_ramp.txt; host-side scene generation leaves the device idle for seconds)
    copy_gbs = measured_copy_ceiling(dev)
    # ... and ~40 ms more of the frame itself, untimed and before the W warm-up frames the contract asks for: the ramp is ~50
    # frames long, the driver's W is a handful, and its K is twenty — a timed region of 7 ms would otherwise sit on it
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.04:
        for _ in range(10):
            frame()
        dev.sync()
    for _ in range(args.warmup):
        frame()
    dev.sync()
    # timed region: exactly K frames; HIP events bracket ONLY the roofline kernel (k_composite) on
    # the ctx stream, because every recorded stage costs ~10 us of stream idle per frame
    # ... on every TIMED_EVERY-th frame of the timed region: a timed launch costs the stream ~5 us even with the event pair on the
    # launch itself (0.326 against 0.321 ms per frame at C2 with a pair on every frame, profiles/r04_f_event_cost_C2.txt)
    TIMED_EVERY = int(os.environ.get("SPLAT_BENCH_EVENT_STRIDE", "4"))
    _lib.check(lib.splat_set_timing_stages(ctx, 1 << _lib.STAGE_COMPOSITE), ctx)
    _lib.check(lib.splat_set_timing_sampling(ctx, max(TIMED_EVERY, 1)), ctx)
    dev.setTiming(os.environ.get("SPLAT_BENCH_EVENTS", "1") != "0")  # (=0: what the event pairs cost the frame — a measuring knob)
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    dev.sync()
    dt = time.perf_counter() - t0
    if os.environ.get("SPLAT_BENCH_EVENTS", "1") == "0":  # (the kernel's duration then comes from a few launches after the timed region)
        dev.setTiming(True)
        for _ in range(5):
            frame()
        dev.sync()
    composite_ms = stage_avg(_lib.STAGE_COMPOSITE)
    cnt_, tot_ = C.c_uint32(), C.c_double()
    _lib.check(lib.splat_stage_time_stats(ctx, _lib.STAGE_COMPOSITE, C.byref(cnt_), C.byref(tot_)), ctx)
    launches_timed = int(cnt_.value)
    _lib.check(lib.splat_set_timing_sampling(ctx, 1), ctx)
    r.finish()  # settles the last timed frame's report: a failed order check in ANY timed frame has been counted by now
    ranking = dict(dev.rankStatus(), framesMisranked=r.framesMisranked, framesOverflowed=int(r.previousFrameOverflowed))
    # per-stage breakdown from a separate short loop with every stage's events on (not part of `value`), and the entries
    # the composite staged / consumed per frame: a property of the input, counted here so that the timed region runs the
    # kernel instantiation every production frame runs (the counting one is ~3 us slower at C2)
    _lib.check(lib.splat_set_timing_stages(ctx, 0xFFFFFFFF), ctx)
    dev.setTiming(True)
    extra_frames = min(args.steps, 10)
    for _ in range(extra_frames):
        frame()
    dev.sync()
    staged, consumed = C.c_uint64(), C.c_uint64()
    _lib.check(lib.splat_timing_consumed(ctx, C.byref(staged), C.byref(consumed)), ctx)
    p_staged, p_used = staged.value / extra_frames, consumed.value / extra_frames
    stage_ms = {sname: stage_avg(sid) for sid, sname in enumerate(_lib.STAGE_NAMES) if sname != "exchange"}
    stage_ms["composite"] = composite_ms  # (the timed region's own figure)
    dev.setTiming(False)
    pairs = r.binner.getTotalIndices()

    comp_bytes = composite_alg_bytes(p_used, width, height)
    comp_s = stage_ms["composite"] / 1e3
    achieved = comp_bytes / comp_s / 1e9
    frame_bytes = frame_alg_bytes(n, n, ntx * nty, pairs, p_used, width, height, disc)
    # (profiles/traffic.json holds counters for: the default configuration, the disc footprint, and round 1's
    # ProjectedSplat records + pre-lit planes; any other combination reports no PMC figures)
    key = name + (("_disc" if prelit else "_unmeasured") if disc else "" if (args.records == "lit" and not prelit) else
                  "_projected_records_prelit_planes" if (args.records == "projected" and prelit) else "_unmeasured")
    pmc = load_traffic(key)
    px = ntx * nty >= 2048 and os.environ.get("SPLAT_COMPOSITE", "")[:1].lower() != "q" or os.environ.get("SPLAT_COMPOSITE", "")[:1].lower() == "p"
    roofline = {"kernel": "k_composite_px" if px else "k_composite", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": comp_bytes, "avg_launch_ms": stage_ms["composite"],
                "formula": "SURVEY 8d: 68 B x pairs_consumed + 4 B x W x H, / avg launch (HIP events on the ctx stream inside the timed region)",
                "launches_timed": launches_timed, "launches_in_timed_region": args.steps,
                "pairs_consumed": round(p_used), "pairs_staged": round(p_staged),
                "traffic_model": composite_traffic_model(p_staged, width, height, args.records, prelit, disc),
                "traffic_model_note": "bytes the kernel as built is expected to move: per STAGED entry (k_composite_px: chunks of 32, fetched ahead; "
                                      "k_composite: 256-entry batches) 4 B index + its gathered record(s), + 4 B per pixel",
                "traffic": pmc.get("k_composite_hbm_bytes_per_launch"),
                "traffic_source": pmc.get("source", "profiles/traffic.json (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; not measured in this run)")
                                  if pmc.get("k_composite_hbm_bytes_per_launch") else None,
                "valu_frac": pmc.get("valu_busy_frac"),
                "valu_frac_source": pmc.get("valu_source") if pmc.get("valu_busy_frac") else None,
                "measured_copy_GBps": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs}
    if px and not disc:
        m = composite_model_floor(p_staged, comp_bytes, copy_gbs, ahead=1)
        roofline["model_floor_us"] = m.pop("model_floor_us")
        roofline["model"] = m
    result = {
        "metric": "Msplats/sec", "value": n * args.steps / dt / 1e6, "unit": "Msplats/s",
        "frames_per_s": args.steps / dt, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "n_splats": n, "width": width, "height": height, "tile": tile,
                   "pairs_P": pairs, "pairs_consumed_P_used": round(p_used), "pairs_staged": round(p_staged), "parallelism": "1 GPU",
                   "property_layout": ("two vec4 planes (pos,radius | lit rgb,opacity): shading kd(normal) applied once per property update, outside the frame"
                                       if prelit else "interleaved 32-byte records + vec4 normals (the reference's layouts); shading inside the timed frame"),
                   "records": ("disc records" if disc else
                               "32-byte lit composite records written by the projector (centre, radius, depth | lit rgb): one gathered line per staged entry"
                               if args.records == "lit" else "ProjectedSplat records; colour (+ normal) gathered per staged entry"),
                   "frame_order": os.environ.get("SPLAT_FRAME_ORDER", "tile-first (bin, then depth-sort per tile; library default)"),
                   "footprint": ("oriented disc (SequentialRenderer.ts:91-142), inverse homography per pixel" if disc else
                                 "isotropic screen-space Gaussian (ComputeShaderRenderer.ts:123-147)"),
                   "composite": "front-to-back, early-out at alpha>=0.99",
                   # how the timed frames ranked equal digits (include/splat.h NOTE on ranking): 'checked' = returning LDS atomics +
                   # a complete order check of every tile list; orderFaults > 0 would mean frames were re-rendered with ballots and
                   # the rest of the run took the slower ballot path — asserted zero below
                   "ranking": ranking,
                   # what ran before and inside the timed region besides the K frames
                   "untimed_prewarm": "~12 ms of device-to-device copies (the box's copy rate) + ~40 ms of the frame itself before the W warm-up frames",
                   "composite_event_stride": TIMED_EVERY},
        "roofline": roofline,
        "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
        "roofline_per_kernel": per_kernel_rooflines(stage_ms, n, pairs, p_used, width, height, args.records == "lit" and not disc, disc, pmc),
        "frame_roofline": {"algorithmic_bytes_per_frame": frame_bytes,
                           "achieved_GBps": frame_bytes / (dt / args.steps) / 1e9,
                           "frac": frame_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                           "frac_of_measured_copy": frame_bytes / (dt / args.steps) / 1e9 / copy_gbs},
    }
    assert ranking["orderFaults"] == 0 and ranking["framesMisranked"] == 0, f"a timed frame failed the tile-list order check and was rendered again: {ranking}"
    if not disc and not args.no_extras:
        # SURVEY 8d's composite-only figure: early-out OFF, every entry of every list consumed (68 P + 4WH), the same
        # kernel on the same frame's records and lists, outside the timed region
        off = time_composite_alone(dev, r, u, pbuf, nbuf, width, height, tile, early_out=False)
        if off is not None:
            ob = composite_alg_bytes(off["consumed"], width, height)
            result["extra"] = {"composite_early_out_off": {
                "avg_launch_ms": off["ms"], "pairs_consumed": round(off["consumed"]), "algorithmic_bytes_per_launch": ob,
                "achieved_GBps": ob / (off["ms"] / 1e3) / 1e9, "frac": ob / (off["ms"] / 1e3) / 1e9 / HBM_PEAK_GBS,
                "traffic_model": composite_traffic_model(off["staged"], width, height, args.records, prelit),
                "model": composite_model_floor(off["staged"], ob, copy_gbs, ahead=2) if px else None,
                "note": "k_composite alone with the alpha>=0.99 break disabled on the timed frame's lists: SURVEY 8d composite-only figure"}}
    if not disc and not args.no_extras:
        # two frames in flight (PipelinedRenderer: frames alternate between two streams): the same frames, more of them
        # per second; NOT the timed region above (whose composite durations would be inflated by the overlap)
        try:
            pr = sr.PipelinedRenderer(0, 2, n, tile, records=args.records)
            for _ in range(6):
                pr.render(u, pbuf, nbuf, None, width, height)
            pr.finish()
            k2 = max(args.steps, 20)
            t0 = time.perf_counter()
            for _ in range(k2):
                pr.render(u, pbuf, nbuf, None, width, height)
            pr.finish()
            dt2 = time.perf_counter() - t0
            same = bool(np.array_equal(pr.readPixels(), r.readPixels()))
            pr.destroy()
            result.setdefault("extra", {})["two_frames_in_flight"] = {
                "ms_per_step": dt2 / k2 * 1e3, "value": n * k2 / dt2 / 1e6, "unit": "Msplats/s", "frames": k2,
                "image_identical_to_the_timed_frames": same,
                "note": "frames alternate between two streams of the one GPU (splat_renderer_amd.PipelinedRenderer); throughput only, "
                        "a frame's latency is unchanged"}
        except Exception as e:  # (an extra: never costs the run its headline)
            result.setdefault("extra", {})["two_frames_in_flight"] = {"error": repr(e)}
    if not disc and not args.no_extras:
        # The camera MOVES in the reference's loop (src/main.ts:110-193, OrbitCameraController.ts:42-58): the timed region
        # above renders one view K times, which is the best case of everything the frame learns from the frame before it
        # (the composite's tile order and per-tile look-ahead, the sync-free pair limit).  Here: a left-button drag of
        # 0.5 degrees of azimuth per frame through the controller, frames enqueued back to back, nothing read in between.
        try:
            orb = orbit_leg(dev, n, width, height, tile, args, pbuf, nbuf, dt / args.steps * 1e3, composite_ms)
            result.setdefault("extra", {})["orbit"] = orb
            # next to the headline, not under extra: `value` renders ONE view K times — the best case of what a frame
            # learns from the frame before it; this is the same renderer with the camera turning 0.5 degrees per frame
            result["moving_camera"] = {"value": orb["value"], "unit": "Msplats/s", "ms_per_step": orb["ms_per_step"],
                                       "composite_ms": orb["composite_ms"], "over_static": orb["over_static"],
                                       "over_standing_still_at_the_last_view": orb["over_standing_still_at_the_last_view"],
                                       "frames": orb["frames"], "degrees_per_frame": orb["degrees_per_frame"],
                                       "note": "FrameLoop + OrbitCameraController; detail under extra.orbit"}
        except Exception as e:  # (an extra: never costs the run its headline)
            result.setdefault("extra", {})["orbit"] = {"error": repr(e)}
    if not args.no_cpu_baseline:
        cb = cpu_baseline(name, props, normals, u, width, height)
        ref8 = cb.pop("frame_u8")
        seq8 = cb.pop("sequential_u8")
        if disc:
            # this footprint IS SequentialRenderer's: the baseline beside it is the oracle's restatement of that
            # renderer (its raster + the projector / key / sort stages it is fed by), the parity image its output
            t = (sum(cb["stage_ms"][:3]) / 1e3) + cb["model_b_sequential_raster"]["seconds"]
            cb.update(value=n / t / 1e6, seconds=round(t, 3),
                      sample=f"1 full frame of {name} (N={n}, {width}x{height}), oracle/oracle.c: project + keys + sort, then the "
                             "software rasteriser of SequentialRenderer.ts (one oriented quad per splat, back to front), 1 thread")
            cb.pop("all_cores")
            ref8 = seq8
        if not args.no_parity:
            frame()  # (the early-out-off launches above wrote their own output texture, not the renderer's)
            got8 = r.readPixels()
            diff = np.abs(got8.astype(np.int16) - ref8.astype(np.int16))
            result["parity_vs_cpu_frame"] = {"max_abs_lsb": int(diff.max()),
                                             "pixels_off_by_more_than_1": int((diff.max(axis=2) > 1).sum())}
            if disc:
                # SequentialRenderer.ts blends every splat: its image is the early-out-OFF frame (the timed frames stop a
                # pixel at alpha >= 0.99, worth up to 0.01 * 255 = 2.55 LSB; the rim of a disc is a 0.044 step = 11 LSB
                # for the few pixels within rounding distance of one — tests/test_gpu_disc.py states both)
                rp = sr.Renderer(dev, None, "rgba8unorm", n, tile, earlyOut=False, footprint="disc")
                rp.render(u, pbuf, nbuf, None, width, height)
                d2 = np.abs(rp.readPixels().astype(np.int16) - ref8.astype(np.int16)).max(axis=2)
                rp.destroy()
                result["parity_vs_cpu_frame"] = {"early_out_off": {"max_abs_lsb": int(d2.max()), "pixels_off_by_more_than_1": int((d2 > 1).sum())},
                                                 "timed_frames_early_out_on": result["parity_vs_cpu_frame"],
                                                 "reference": "oracle software rasteriser of SequentialRenderer.ts, back to front"}
        result["cpu_baseline"] = cb
    r.destroy()
    pm.destroy()
    nbuf.destroy()
    dev.destroy()
    return result


def run_multi(args, name, n, width, height, tile, ntx, nty, props, normals, u, workload, rank, local_rank, world):
    # stdout carries ONE JSON line: RCCL prints a version banner to it when a communicator is created, so the C-level
    # stdout points at stderr until the line is ready
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        return _run_multi(args, name, n, width, height, tile, ntx, nty, props, normals, u, workload, rank, local_rank, world)
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)


def _run_multi(args, name, n, width, height, tile, ntx, nty, props, normals, u, workload, rank, local_rank, world):
    import torch
    import torch.distributed as td
    if world == 1 and "MASTER_ADDR" not in os.environ:  # --band-path without a launcher: a one-rank group of our own
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(local_rank)
    if not td.is_initialized():
        td.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    per = dist.shard_size(n, world)
    stages = dist.HipStages(torch, local_rank, per * world, width, height, tile, footprint=args.footprint)
    pt = torch.from_numpy(props).cuda()
    nt = torch.from_numpy(normals).cuda()
    gather, collective = td.all_gather_into_tensor, "torch.distributed.all_gather_into_tensor (RCCL)"
    if args.collective == "abi":
        def bcast(a):  # rank 0's unique id to every rank
            t = torch.from_numpy(a).cuda()
            td.broadcast(t, 0)
            return t.cpu().numpy()
        try:
            gather = dist.AbiAllGather(torch, stages, rank, world, bcast)
            collective = "splat_allgather_records (the C ABI's own RCCL communicator)"
        except Exception as e:  # (RCCL could not be bound in this process: every rank fails alike, before any collective)
            print(f"[rank {rank}] C-ABI communicator unavailable ({e!r}); torch.distributed issues the all-gather", file=sys.stderr)
    br = dist.BandRenderer(stages, n, width, height, rank, world, gather, tile, always_gather=args.band_path)
    if args.layout == "planes":  # as at N=1: shading applied once per property update, not per staged list entry
        stages.set_lit(pt.data_ptr(), nt.data_ptr(), n)

    def frame():
        br.render(u, pt.data_ptr(), nt.data_ptr())

    # warm-up: the first frame also calibrates the bands (equal pairs per band instead of equal rows:
    # the centre rows of this scene are denser)
    frame()
    torch.cuda.synchronize()
    rows_hist = br.rebalance(lambda t: td.all_reduce(t, op=td.ReduceOp.SUM))
    for _ in range(max(args.warmup - 1, 1)):
        frame()
    br.render(u, pt.data_ptr(), nt.data_ptr(), settle=True)  # bands changed: let the sync-free bounds re-learn
    pipe = None
    if not args.no_pipeline:
        try:  # warm the second buffer pair, stream and ctx outside the timed region
            pipe = dist.FramePipeline(torch, br, local_rank)
            pipe.run(2, lambda k: u, pt.data_ptr(), nt.data_ptr())
            torch.cuda.synchronize()
            ok = torch.ones(1, device="cuda")
        except Exception as e:  # (every rank must take the same loop: agree below)
            print(f"[rank {rank}] frame pipeline unavailable ({e!r}); serial loop", file=sys.stderr)
            pipe, ok = None, torch.zeros(1, device="cuda")
        td.all_reduce(ok, op=td.ReduceOp.MIN)
        if ok.item() < 1:
            pipe = None
    loop_ms = {}

    # which loop is faster on THIS node and rank count is measured, not assumed (a few frames of each, slowest rank
    # decides, all ranks take the same one)
    def timed(fn):
        torch.cuda.synchronize()
        td.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        v = torch.tensor([time.perf_counter() - t], dtype=torch.float64, device="cuda")
        td.all_reduce(v, op=td.ReduceOp.MAX)
        return float(v.item())
    trial = 8
    if pipe is not None:
        # with a cheap exchange the second stream's events cost more than the overlap returns
        loop_ms["serial"] = timed(lambda: [frame() for _ in range(trial)]) / trial * 1e3
        loop_ms["two_frames_in_flight"] = timed(lambda: pipe.run(trial, lambda k: u, pt.data_ptr(), nt.data_ptr())) / trial * 1e3
        if loop_ms["serial"] <= loop_ms["two_frames_in_flight"]:
            pipe.destroy()
            pipe = None
    # The same bands with no exchange: every rank holds the splats anyway (the composite gathers their colours), so it
    # can project all of them itself — 75 us at 5M splats against the all-gather of their records.  Its own ctx, sorter
    # and binner; the same tile rows; bit-identical band images (tests/test_gpu_stages.py).
    local = None
    try:  # (the trial of the other cut is an extra key: it must never cost the run its headline)
        lst = dist.HipStages(torch, local_rank, n, width, height, tile, footprint=args.footprint)
        if stages.lit is not None:
            lst.set_lit(pt.data_ptr(), nt.data_ptr(), n)
        local = dist.LocalBandRenderer(lst, n, width, height, rank, world, tile)
        local.row0, local.row1 = br.row0, br.row1
        for _ in range(3):
            local.render(u, pt.data_ptr(), nt.data_ptr())
        local.render(u, pt.data_ptr(), nt.data_ptr(), settle=True)
        ok = torch.ones(1, device="cuda")
    except Exception as e:
        print(f"[rank {rank}] exchange-free band renderer unavailable ({e!r})", file=sys.stderr)
        local, ok = None, torch.zeros(1, device="cuda")
    td.all_reduce(ok, op=td.ReduceOp.MIN)
    if ok.item() < 1:
        if args.exchange == "none":
            raise SystemExit("--exchange none: the exchange-free band renderer could not be set up on every rank")
        local = None
    if local is not None:
        if "serial" not in loop_ms:
            loop_ms["serial"] = timed(lambda: [frame() for _ in range(trial)]) / trial * 1e3
        loop_ms["no_exchange_every_rank_projects_all"] = timed(lambda: [local.render(u, pt.data_ptr(), nt.data_ptr())
                                                                        for _ in range(trial)]) / trial * 1e3
        faster = loop_ms["no_exchange_every_rank_projects_all"] < min(v for k, v in loop_ms.items() if k != "no_exchange_every_rank_projects_all")
        if args.exchange == "allgather" or (args.exchange == "auto" and not faster):
            local.stages.destroy()
            local = None
    if local is not None:  # the all-gather side is not run: release it
        if pipe is not None:
            pipe.destroy()
            pipe = None
        tstages = local.stages
    else:
        tstages = stages
    tstages.overflows = 0
    stages.consumed = torch.zeros((ntx * nty, 2), dtype=torch.int64, device="cuda")  # per tile {staged, consumed} (no atomics in the kernel)
    tstages.set_timing(True, (1 << _lib.STAGE_COMPOSITE) | _lib.TIMING_COUNT_ENTRIES)  # (band frames count their consumed entries in the timed frames)
    torch.cuda.synchronize()
    td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if local is not None:
        for _ in range(args.steps):
            local.render(u, pt.data_ptr(), nt.data_ptr())
    elif pipe is not None:
        # two frames in flight: frame k+1's projection + all-gather (second stream) under frame k's band work
        pipe.run(args.steps, lambda k: u, pt.data_ptr(), nt.data_ptr())
    else:
        for _ in range(args.steps):
            frame()
    torch.cuda.synchronize()
    td.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    td.all_reduce(tmax, op=td.ReduceOp.MAX)
    dt = float(tmax.item())
    comp_ms = tstages.stage_avg_ms(_lib.STAGE_COMPOSITE)
    # (before the settling frame below adds its own entries)
    p_used = (tstages.timing_consumed()[1] if local is not None else int(stages.consumed[:, 1].sum().item())) / args.steps
    # outside the timed region: proves no sync-free frame overflowed
    (local if local is not None else br).render(u, pt.data_ptr(), nt.data_ptr(), settle=True)
    assert tstages.overflows == 0, "a sync-free frame overflowed its pair limit in a static scene"
    ranking = dict(tstages.rank_status(), framesMisranked=tstages.misranked)
    tstages.set_timing(False)
    # self-checks of the exchange for the record (outside the timed region): what the communicator says about itself, and
    # every other rank's gathered shard against this rank's own projection of that slice, bit for bit
    rccl_view = gather.rccl_view() if hasattr(gather, "rccl_view") else (td.get_world_size(), td.get_rank())
    br.render(u, pt.data_ptr(), nt.data_ptr(), settle=True)
    verified = br.verify_exchange(u, pt.data_ptr(), nt.data_ptr(), include_self=args.band_path)
    r0, r1 = br.pixel_rows()
    kept = n if local is not None else stages.kept  # (no band filter without an exchange: every rank bins from all n splats)
    info = torch.tensor([kept, br.row0, br.row1, int(p_used), int(comp_ms * 1e6), rccl_view[0], rccl_view[1], verified,
                         ("checked", "atomic", "ballot").index(ranking["policy"]), int(ranking["atomicsOrdered"]), ranking["orderFaults"]],
                        dtype=torch.int64, device="cuda")
    infos = [torch.zeros_like(info) for _ in range(world)]
    td.all_gather(infos, info)
    infos = [[int(v) for v in t.tolist()] for t in infos]
    # (every rank sees every rank's counter: all of them stop together, none is left waiting in a collective)
    assert all(i[10] == 0 for i in infos), f"a timed frame failed the tile-list order check and was rendered again: orderFaults per rank {[i[10] for i in infos]}"
    # roofline of the dominant kernel on the slowest rank's composite (per launch = per band)
    slow = max(range(world), key=lambda k: infos[k][4])
    rows_px = min(infos[slow][2] * tile, height) - infos[slow][1] * tile
    comp_bytes = composite_alg_bytes(infos[slow][3], width, rows_px)
    achieved = comp_bytes / (infos[slow][4] / 1e9) / 1e9 if infos[slow][4] else 0.0
    result = {
        "metric": "Msplats/sec", "value": n * args.steps / dt / 1e6, "unit": "Msplats/s",
        "frames_per_s": args.steps / dt, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "n_splats": n, "width": width, "height": height, "tile": tile,
                   "parallelism": (f"tile-row bands x{world} (balanced by pairs per row), no exchange: every rank projects all "
                                   f"{n} splats from its own copy (" + ("--exchange none" if args.exchange == "none" else
                                   "--exchange auto: faster on this node than the all-gather of their records, see frame_loop_trial_ms"
                                   ) + ")" if local is not None else
                                   f"tile-row bands x{world} (balanced by pairs per row) + 1 RCCL all-gather of {per * stages.rec_floats * 4} B "
                                   f"shards per frame" + ("" if pipe is None else "; 2 frames in flight: the next frame's projection + "
                                                           "all-gather run on a second stream under this frame's band work")),
                   "exchange_policy": args.exchange, "exchange_chosen": "none" if local is not None else "allgather",
                   "collective": None if local is not None else collective,
                   "frame_loop_trial_ms": {k: round(v, 4) for k, v in loop_ms.items()},
                   "footprint": ("oriented disc (SequentialRenderer.ts:91-142)" if stages.disc else
                                 "isotropic screen-space Gaussian (ComputeShaderRenderer.ts:123-147)") +
                                ("" if local is not None else f", {stages.rec_floats * 4}-byte exchange records"),
                   "per_rank": [{"splats_kept": i[0], "tile_rows": [i[1], i[2]], "pairs_consumed": i[3],
                                 "composite_ms": i[4] / 1e6} for i in infos],
                   "exchange": {"collective": collective, "record_bytes": stages.rec_floats * 4,
                                "bytes_contributed_per_rank": per * stages.rec_floats * 4,
                                "bytes_received_per_rank": (world - 1) * per * stages.rec_floats * 4,
                                "rccl_ranks_seen": [i[5] for i in infos], "rccl_rank_of_each_process": [i[6] for i in infos],
                                "shards_verified_per_rank": [i[7] for i in infos], "shards_expected_per_rank": world - 1 + int(args.band_path),
                                "verification": "after the timed region every rank re-projected every other rank's slice from its own copy of "
                                                "the splats and compared it bit for bit with the block the all-gather delivered",
                                "in_timed_region": local is None},
                   "composite": "front-to-back, early-out at alpha>=0.99",
                   "ranking": {"policy": [("checked", "atomic", "ballot")[i[8]] for i in infos], "atomicsOrdered": [bool(i[9]) for i in infos],
                               "orderFaults": [i[10] for i in infos]}},
        "roofline": {"kernel": "k_composite_px" if ntx * nty >= 2048 else "k_composite", "bound": "hbm", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": comp_bytes,
                     "avg_launch_ms": infos[slow][4] / 1e6, "rank": slow},
        "cpu_baseline": None,  # reported at N=1 only
    }
    if args.band_path:
        result["config"]["band_path"] = ("--band-path: run_multi's code path (slice projection, all-gather through the communicator, band "
                                         "frame, self-checks) with WORLD_SIZE " + str(world) + "; not the N = 1 headline (run_single)")
    td.barrier()
    if pipe is not None:
        pipe.destroy()
    if local is not None:
        local.stages.destroy()
    if hasattr(gather, "destroy"):
        torch.cuda.synchronize()
        gather.destroy()
    stages.destroy()
    td.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
