"""GPU parity tests for the oriented-disc footprint (SequentialRenderer's / TileRenderer's splat).

Bit-exact: disc records, ProjectedSplat records (the disc's screen extent), keys, sort order, tile lists.
Tolerance (stated below): composited pixels, against the oracle's per-pixel restatement AND against the oracle's
software rasteriser of SequentialRenderer.ts (one oriented quad per splat, back to front).
"""
import ctypes as C

import numpy as np
import pytest

import splat_renderer_amd as sr
from oracle import oracle as O
from splat_renderer_amd import _lib
from tests.helpers import assert_same, make_case

pytestmark = pytest.mark.gpu

# float RGBA in [0,1], early-out off.  The GPU evaluates (u,v) with v_rcp_f32 and contracted FMAs and the
# Gaussian with exp2: a few ulp per layer.
TOL = 3e-5
# vs the software rasteriser (f64 edge functions / barycentrics, back-to-front blend): measured 5e-5 at C0
TOL_RASTER = 1e-4
# The fragment shader discards at u^2+v^2 > 1 where the Gaussian is still exp(-3.125) = 0.0439: a pixel within
# rounding distance of a rim may be kept by one evaluation and discarded by another.  The oracle flags every pixel
# within 1e-3 (in u^2+v^2) of some disc's rim; on those, and only those, a difference up to the step is accepted.
TOL_RIM = 0.045
MAX_RIM_FLIPS = 16       # pixels per image that actually flip (they are rare events even among flagged pixels)
TOL_EARLY_OUT_BOUND = 0.0101  # early-out on: (1 - 0.99) * max colour, as for the isotropic footprint

CASES = [
    (10000, 256, 256, 1234, 1.0),  # C0
    (3000, 160, 120, 8, 2.5),
    (20000, 333, 200, 9, 1.0),     # ragged right/bottom tiles
    (500, 64, 64, 6, 12.0),        # discs larger than the screen
    (60000, 640, 360, 10, 0.7),
]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def disc_case(n, w, h, seed, rs):
    props, normals, u = make_case(n, w, h, seed, rs)
    normals[::7, :3] = np.float32([0.05, 0.99, 0.1])  # |n.y| > 0.9 picks the other "up" (SequentialRenderer.ts:69)
    if n > 20:
        normals[5] = 0                                 # zero normal -> NaN tangent -> culled
        props[9, :3] = [0.0, 0.0, 50.0]                # behind the camera -> culled
        normals[11, :3] = 0.5 * normals[11, :3]        # non-unit normal: the bitangent scales with it (:96)
    return props, normals, u


def oracle_disc(props, normals, u, w, h, early_out=False, n_padded=None):
    proj, discs = O.project_disc(u, props, normals)
    keys, pay = O.extract_keys(proj, n_padded)
    _, order = O.sort_pairs(keys, pay)
    counts, offsets, idx = O.bin_sorted(proj, order, w, h)
    img, img8, consumed, rim = O.composite_disc(early_out, props[:, 4:], normals, discs, idx, counts, offsets, w, h)
    return dict(proj=proj, discs=discs, keys=keys, payload=pay, order=order, counts=counts, offsets=offsets, indices=idx,
                img=img, img8=img8, rim=rim)


def check_image(got, ref, tol=TOL):
    d = np.abs(got - ref["img"]).max(axis=2)
    assert d[ref["rim"] == 0].max() <= tol
    assert d.max() <= TOL_RIM
    assert (d > tol).sum() <= MAX_RIM_FLIPS


@pytest.mark.parametrize("n,w,h,seed,rs", CASES + [(1, 64, 64, 1, 1.0), (7, 64, 48, 2, 1.0)])
def test_disc_projector_bit_exact(device, n, w, h, seed, rs):
    props, normals, u = disc_case(n, w, h, seed, rs)
    ref = oracle_disc(props, normals, u, w, h, n_padded=sr.scene.padded_size(n))
    pm = sr.SplatPropertyManager(device, n)
    pm.setFromArrays(props)
    nbuf = device.createBufferFrom(normals)
    proj = sr.SplatProjector(device, n, footprint="disc")
    sorter = sr.RadixSorter(device, n)
    with pytest.raises(sr.SplatError):
        proj.project(None, u, pm.getPropertyBuffer())  # the disc projector reads the normals
    proj.project(None, u, pm.getPropertyBuffer(), sorter.getKeysBuffer(), sorter.getPayloadBuffer(), sorter.paddedSize, normalsBuffer=nbuf)
    assert_same(bits(proj.getDiscBuffer().read(np.float32)).reshape(n, 8), bits(ref["discs"]), "disc L83")
    assert_same(bits(proj.getProjectedBuffer().read(np.float32)).reshape(n, 8), bits(ref["proj"]), "disc L84")
    assert_same(sorter.getKeysBuffer().read(np.uint32), ref["keys"], "disc L85")
    assert_same(sorter.getPayloadBuffer().read(np.uint32), ref["payload"], "disc L86")
    # from a position plane (stride 1) through the C ABI directly: same bits
    planes = pm.getPropertyPlanes()
    out_p, out_d = device.createBuffer(n * 32), device.createBuffer(n * 32)
    uu = np.ascontiguousarray(u, np.float32)
    _lib.check(device.lib.splat_project_disc(device.ctx, uu.ctypes.data_as(C.POINTER(C.c_float)), planes.posRadius.ptr, 1, nbuf.ptr, 1, n,
                                             out_p.ptr, out_d.ptr, None, None, 0), device.ctx)
    assert_same(bits(out_d.read(np.float32)).reshape(n, 8), bits(ref["discs"]), "disc L93")
    assert_same(bits(out_p.read(np.float32)).reshape(n, 8), bits(ref["proj"]), "disc L94")
    with pytest.raises(sr.SplatError):
        sr.SplatProjector(device, 4).getDiscBuffer()
    for o in (pm, nbuf, proj, sorter, out_p, out_d):
        o.destroy()


# Which kernel composites the disc frames (Device.compositeOptions): the library's choice on these screens of fewer than 2048
# tiles — k_composite, a wave per 8x8 quadrant evaluating every entry that touches it — or k_composite_px (per-lane entry
# queues, the footprint evaluated for the four pixels of every lane an entry's box touches) on its two schedules.
DISC_KERNELS = ["default", "px1", "px2"]


def force_kernel(device, kernel):
    if kernel != "default":
        device.compositeOptions("pixel", ahead=1 if kernel == "px1" else 2)


@pytest.mark.parametrize("n,w,h,seed,rs", CASES)
@pytest.mark.parametrize("early_out", [False, True])
@pytest.mark.parametrize("kernel", DISC_KERNELS)
def test_disc_staged_pipeline_vs_oracle(device, n, w, h, seed, rs, early_out, kernel):
    try:
        force_kernel(device, kernel)
        disc_staged_pipeline_vs_oracle(device, n, w, h, seed, rs, early_out)
    finally:
        device.compositeOptions()


def disc_staged_pipeline_vs_oracle(device, n, w, h, seed, rs, early_out):
    props, normals, u = disc_case(n, w, h, seed, rs)
    ref = oracle_disc(props, normals, u, w, h, early_out)
    pm = sr.SplatPropertyManager(device, n)
    pm.setFromArrays(props)
    nbuf = device.createBufferFrom(normals)
    proj = sr.SplatProjector(device, n, footprint="disc")
    sorter = sr.RadixSorter(device, n)
    binner = sr.GPUTileBinner(device, 16)
    proj.project(None, u, pm.getPropertyBuffer(), sorter.getKeysBuffer(), sorter.getPayloadBuffer(), sorter.paddedSize, normalsBuffer=nbuf)
    sorter.sort()
    binner.binSplats(None, proj.getProjectedBuffer(), sorter.getSortedIndicesBuffer(), n, w, h)
    total = ref["indices"].shape[0]
    assert_same(sorter.getSortedIndicesBuffer().read(np.uint32, n), ref["order"][:n], "disc L116")
    assert binner.getTotalIndices() == total
    assert_same(binner.getTileCountsBuffer().read(np.uint32), ref["counts"], "disc L118")
    assert_same(binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], "disc L119")
    # TileRenderer with its own (the reference's) footprint
    tr = sr.TileRenderer(device, None, "rgba8unorm", earlyOut=early_out, footprint="disc")
    tr.bindTileData(proj.getDiscBuffer(), binner.getTileCountsBuffer(), binner.getTileOffsetsBuffer())
    tr.render(u, pm.getPropertyBuffer(), binner.getTileIndicesBuffer(), nbuf, ref["counts"], -(-w // 16), -(-h // 16), 16, 4096, w, h,
              wantFloat=True)
    got, got8 = tr.readPixelsFloat(), tr.readPixels()
    if early_out:
        assert np.abs(got - ref["img"]).max() <= TOL_RIM
        d = np.abs(got - ref["img"]).max(axis=2)
        assert d[ref["rim"] == 0].max() <= TOL_EARLY_OUT_BOUND
    else:
        check_image(got, ref)
        assert np.abs(got8.astype(int) - ref["img8"].astype(int)).max(axis=2)[ref["rim"] == 0].max() <= 1
    assert (got8[..., 3] == 255).all()
    for o in (pm, nbuf, proj, sorter, binner, tr):
        o.destroy()


@pytest.mark.parametrize("n,w,h,seed,rs", [(10000, 256, 256, 1234, 1.0), (3000, 160, 120, 8, 2.5)])
def test_sequential_renderer_is_the_reference_rasteriser_image(device, n, w, h, seed, rs):
    """SequentialRenderer (HIP, tile lists + inverse homography per pixel) against the oracle's software rasteriser of
    SequentialRenderer.ts fed the reversed (back-to-front) order — north_star's parity statement."""
    props, normals, u = make_case(n, w, h, seed, rs)
    ref = oracle_disc(props, normals, u, w, h)
    raster, raster8 = O.sequential(u, props, normals, ref["order"][::-1].copy(), w, h)
    pm = sr.SplatPropertyManager(device, n)
    pm.setFromArrays(props)
    nbuf, obuf = device.createBufferFrom(normals), device.createBufferFrom(ref["order"][:n])
    r = sr.SequentialRenderer(device, None, "rgba8unorm", n, earlyOut=False)
    r.render(u[:20], pm.getPropertyBuffer(), obuf, nbuf, w, h, wantFloat=True)
    got, got8 = r.readPixelsFloat(), r.readPixels()
    d = np.abs(got - raster).max(axis=2)
    off_rim = ref["rim"] == 0
    assert d[off_rim].max() <= TOL_RASTER
    assert d.max() <= TOL_RIM and (d > TOL_RASTER).sum() <= MAX_RIM_FLIPS
    assert np.abs(got8.astype(int) - raster8.astype(int)).max(axis=2)[off_rim].max() <= 1
    mse = float(np.mean((got[..., :3] - raster[..., :3]) ** 2))
    assert -10.0 * np.log10(max(mse, 1e-30)) > 70.0  # (the isotropic footprint is 18 dB from this image)
    for o in (r, pm, nbuf, obuf):
        o.destroy()


@pytest.mark.parametrize("order", ["tileFirst", "sortFirst"])
@pytest.mark.parametrize("layout", ["interleaved", "planes", "lit"])
def test_disc_whole_frame(device, order, layout):
    n, w, h = 40000, 400, 240
    props, normals, u = disc_case(n, w, h, 23, 1.5)
    ref = oracle_disc(props, normals, u, w, h, early_out=False)
    pm = sr.SplatPropertyManager(device, n)
    pm.setFromArrays(props)
    nbuf = device.createBufferFrom(normals)
    r = sr.Renderer(device, None, "rgba8unorm", n, earlyOut=False, frameOrder=order, footprint="disc")
    src = {"interleaved": pm.getPropertyBuffer, "planes": pm.getPropertyPlanes, "lit": lambda: pm.getLitPlanes(nbuf)}[layout]()
    for _ in range(3):  # the later ones are sync-free frames
        r.render(u, src, nbuf, None, w, h, wantFloat=True)
    total = r.finish()
    assert total == ref["indices"].shape[0]
    assert_same(bits(r.projector.getProjectedBuffer().read(np.float32)).reshape(n, 8), bits(ref["proj"]), "disc L177")
    assert_same(r.binner.getTileCountsBuffer().read(np.uint32), ref["counts"], "disc L178")
    assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], "disc L179")
    check_image(r.readPixelsFloat(), ref)
    if layout == "lit":
        with pytest.raises(sr.SplatError):  # the disc projector needs the normals even when the colours are pre-lit
            r.render(u, src, None, None, w, h)
    # without the ProjectedSplat by-product (nothing in a disc frame reads it): same lists, same image bits
    # (and with plain 32-byte disc records — colour and normal gathered per staged entry, as the staged API does — instead of
    # the frame's default, the lit colour behind each disc record: the same arithmetic on the same numbers)
    # (what the FRAME composited from — lit disc records inside the binner — is not what getRecordsBuffer() holds: ProjectedSplat)
    assert r.frameRecordFormat == _lib.RECORDS_LIT32 and r.recordFormat == _lib.RECORDS_PROJECTED
    r2 = sr.Renderer(device, None, "rgba8unorm", n, earlyOut=False, frameOrder=order, footprint="disc", writeProjected=False,
                     records="projected")
    r2.projector.getProjectedBuffer().zero()
    r2.render(u, src, nbuf, None, w, h, wantFloat=True)
    assert r2.finish() == total and r2.recordFormat == _lib.RECORDS_PROJECTED and r2.frameRecordFormat == _lib.RECORDS_PROJECTED
    assert_same(r2.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], "disc L189")
    assert_same(r2.readPixelsFloat().view(np.uint32), r.readPixelsFloat().view(np.uint32), "disc L190")
    assert not r2.projector.getProjectedBuffer().read(np.uint32).any()
    r2.destroy()
    with pytest.raises(sr.SplatError):
        sr.Renderer(device, None, "rgba8unorm", n, writeProjected=False)  # the isotropic composite reads them
    for o in (r, pm, nbuf):
        o.destroy()


def test_disc_and_isotropic_frames_alternate_on_one_renderer_pair(device):
    """Two renderers sharing nothing but the device: a disc frame must not disturb an isotropic one and vice versa."""
    n, w, h = 8000, 200, 120
    props, normals, u = make_case(n, w, h, 31, 2.0)
    ref = oracle_disc(props, normals, u, w, h, early_out=True)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    a = sr.Renderer(device, None, "rgba8unorm", n, footprint="disc")
    b = sr.Renderer(device, None, "rgba8unorm", n)
    imgs = []
    for _ in range(2):
        a.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
        b.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
        imgs.append((a.readPixelsFloat().copy(), b.readPixelsFloat().copy()))
    assert np.array_equal(imgs[0][0], imgs[1][0]) and np.array_equal(imgs[0][1], imgs[1][1])
    d = np.abs(imgs[0][0] - ref["img"]).max(axis=2)
    assert d[ref["rim"] == 0].max() <= TOL_EARLY_OUT_BOUND and d.max() <= TOL_RIM
    assert np.abs(imgs[0][0] - imgs[0][1]).max() > 0.05  # they ARE different footprints
    for o in (a, b, pbuf, nbuf):
        o.destroy()


def test_disc_rejects_what_it_does_not_support(device):
    n, w, h = 256, 64, 64
    props, normals, u = make_case(n, w, h, 3, 2.0)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    lib, ctx = device.lib, device.ctx
    out = device.createBuffer(w * h * 4)
    dummy = device.createBuffer(4096 * 4)
    cfg = _lib.CompositeCfg(_lib.MODE_REFERENCE_LITERAL, 1, 16, 0, 0xFFFFFFFF, 0, 0, _lib.FOOTPRINT_DISC)
    args = (pbuf.ptr + 16, 2, nbuf.ptr, 1, dummy.ptr, dummy.ptr, dummy.ptr, dummy.ptr, w, h, out.ptr, None, None)
    assert lib.splat_composite(ctx, C.byref(cfg), *args) == -1  # the literal back-to-front loop is ComputeShaderRenderer's
    cfg = _lib.CompositeCfg(_lib.MODE_FRONT_TO_BACK, 1, 16, 0, 0xFFFFFFFF, _lib.RECORDS_COMPACT, 0, _lib.FOOTPRINT_DISC)
    assert lib.splat_composite(ctx, C.byref(cfg), *args) == -1
    cfg = _lib.CompositeCfg(_lib.MODE_FRONT_TO_BACK, 1, 16, 0, 0xFFFFFFFF, 0, 0, 7)
    assert lib.splat_composite(ctx, C.byref(cfg), *args) == -1
    # lit disc records (48 bytes) exist only inside a frame's binner: the public composite does not take the pair
    # (disc footprint, LIT32) at its word and read idx * 48 out of a caller's 32-byte records
    cfg = _lib.CompositeCfg(_lib.MODE_FRONT_TO_BACK, 1, 16, 0, 0xFFFFFFFF, _lib.RECORDS_LIT32, 0, _lib.FOOTPRINT_DISC)
    assert lib.splat_composite(ctx, C.byref(cfg), *args) == -1
    # band frames exchange the isotropic footprint's records
    sorter, binner = sr.RadixSorter(device, n), sr.GPUTileBinner(device, 16)
    cfg = _lib.CompositeCfg(_lib.MODE_FRONT_TO_BACK, 1, 16, 0, 0xFFFFFFFF, 0, 0, _lib.FOOTPRINT_DISC)
    assert lib.splat_band_frame(ctx, sorter._s, binner._b, C.byref(cfg), pbuf.ptr, nbuf.ptr, dummy.ptr, 16, w, h, out.ptr, None, None) == -1
    uu = np.ascontiguousarray(u, np.float32)
    up = uu.ctypes.data_as(C.POINTER(C.c_float))
    assert lib.splat_project_disc(ctx, up, pbuf.ptr, 2, None, 1, n, dummy.ptr, dummy.ptr, None, None, 0) == -1
    assert lib.splat_project_disc(ctx, up, pbuf.ptr, 2, nbuf.ptr + 4, 1, n, dummy.ptr, dummy.ptr, None, None, 0) == -1  # alignment
    assert lib.splat_project_disc(ctx, up, pbuf.ptr, 2, nbuf.ptr, 1, 0, None, None, None, None, 0) == 0
    with pytest.raises(sr.SplatError):
        sr.Renderer(device, None, "rgba8unorm", n, footprint="hexagon")
    for o in (pbuf, nbuf, out, dummy, sorter, binner):
        o.destroy()


@pytest.mark.parametrize("world", [2, 4])
def test_disc_virtual_ranks_band_frame_matches_single_gpu(device, world):
    """SURVEY §8e for the oriented-disc footprint, without a cluster: every rank's work in turn on the one GPU, the
    all-gather of the 48-byte exchange records is a concat; the stitched rgba8 image must be bit-identical to the
    single-GPU disc frame, and the records those of the oracle."""
    import os
    if os.environ.get("SPLAT_FRAME_ORDER") == "sortfirst":
        pytest.skip("splat_band_frame takes oriented-disc records in the tile-first order of work only (splat.h: an INVALID error otherwise)")
    import torch
    from splat_renderer_amd import dist
    n, w, h = 30001, 400, 232  # odd n: the last shard is padded with NaN records
    props, normals, u = disc_case(n, w, h, 41, 1.5)
    ref = oracle_disc(props, normals, u, w, h, early_out=True)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    full = sr.Renderer(device, None, "rgba8unorm", n, footprint="disc")
    full.render(u, pbuf, nbuf, None, w, h)
    want = full.readPixels().copy()
    device.sync()
    pt, nt = torch.from_numpy(props).cuda(), torch.from_numpy(normals).cuda()
    per = dist.shard_size(n, world)
    stages = dist.HipStages(torch, 0, per * world, w, h, footprint="disc")
    if world == 4:
        stages.set_lit(pt.data_ptr(), nt.data_ptr(), n)
    renderers = [dist.BandRenderer(stages, n, w, h, r, world, None) for r in range(world)]
    with pytest.raises(ValueError):
        stages.project_slice(u, pt.data_ptr(), 0, 1, renderers[0].shard)  # the disc projector needs the normals
    for br in renderers:
        stages.project_slice(u, pt.data_ptr(), br.first, br.count, br.shard, nt.data_ptr())
    gathered = torch.cat([br.shard for br in renderers], dim=0).contiguous()
    rec = gathered.cpu().numpy()[:n]
    assert_same(bits(rec[:, :8]), bits(ref["discs"]), "disc L276")
    assert np.array_equal(bits(rec[:, 8]), bits(ref["proj"][:, 4])) and not rec[:, 9:].any()
    got = np.zeros_like(want)
    for br in renderers:
        stages.band_frame(gathered, per * world, pt.data_ptr(), nt.data_ptr(), br.row0, br.row1, br.image, settle=True)
        torch.cuda.synchronize()
        r0, r1 = br.pixel_rows()
        got[r0:r1] = br.image.cpu().numpy()[r0:r1]
    assert_same(got, want, "disc L284")
    # and a sort-first band frame says what it cannot do instead of rendering something else
    lib = device.lib
    _lib.check(lib.splat_bin_set_frame_order(stages.binner, 0), stages.ctx)  # SPLAT_FRAME_ORDER_SORT_FIRST
    cfg = _lib.CompositeCfg(_lib.MODE_FRONT_TO_BACK, 1, 16, 0, 4, _lib.RECORDS_DISC48, 0, _lib.FOOTPRINT_DISC)
    br = renderers[0]
    rc = lib.splat_band_frame(stages.ctx, stages.sorter, stages.binner, C.byref(cfg), pt.data_ptr(), nt.data_ptr(), gathered.data_ptr(),
                              per * world, w, h, br.image.data_ptr(), None, None)
    assert rc == -1
    stages.destroy()
    for o in (full, pbuf, nbuf):
        o.destroy()


def test_disc_frame_pipeline_two_frames_in_flight(device):
    """FramePipeline with the disc footprint (world 1: the exchange is the projection alone): frames stay apart."""
    import os
    if os.environ.get("SPLAT_FRAME_ORDER") == "sortfirst":
        pytest.skip("splat_band_frame takes oriented-disc records in the tile-first order of work only (splat.h: an INVALID error otherwise)")
    import torch
    from splat_renderer_amd import dist
    from splat_renderer_amd.camera import Camera
    n, w, h = 20000, 320, 200
    props, normals, _ = make_case(n, w, h, 51, 1.5)
    cams = []
    for k in range(4):
        cam = Camera()
        cam.setAspect(w / h)
        cam.rotate(0.2 * k, 0.05 * k)
        cams.append(cam.uniforms(w, h))
    pt, nt = torch.from_numpy(props).cuda(), torch.from_numpy(normals).cuda()
    stages = dist.HipStages(torch, 0, n, w, h, footprint="disc")
    br = dist.BandRenderer(stages, n, w, h, 0, 1, None)
    want = []
    for k in range(4):
        want.append(br.render(cams[k], pt.data_ptr(), nt.data_ptr(), settle=True).cpu().numpy().copy())
    pipe = dist.FramePipeline(torch, br, 0)
    got = []
    pipe.exchange(0, cams[0], pt.data_ptr(), nt.data_ptr())
    for k in range(4):
        if k + 1 < 4:
            pipe.exchange(k + 1, cams[k + 1], pt.data_ptr(), nt.data_ptr())
        got.append(pipe.band(k, pt.data_ptr(), nt.data_ptr(), settle=True).cpu().numpy().copy())
    for k in range(4):
        assert_same(got[k], want[k], "disc L325")
    assert not np.array_equal(want[0], want[1])
    pipe.destroy()
    stages.destroy()


def test_disc_full_size_C2_properties(device):
    """BASELINE's headline size (5M @1080p) with the oriented-disc footprint — too big for the oracle's composite in a
    unit test, so size-independent properties: both frame orders give the same lists and image; the records and
    bounds are the oracle's (bit for bit, the projector is cheap on the CPU); every list is depth-ordered with index
    ties ascending; sampled pairs overlap their tile; sampled pixels equal an f64 evaluation of their own list."""
    n, w, h = sr.scene.CONFIGS["C2"]
    props, normals, u = make_case(n, w, h)
    proj_ref, discs_ref = O.project_disc(u, props, normals)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    a = sr.Renderer(device, None, "rgba8unorm", n, frameOrder="sortFirst", footprint="disc", earlyOut=False)
    b = sr.Renderer(device, None, "rgba8unorm", n, frameOrder="tileFirst", footprint="disc", earlyOut=False)
    a.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
    b.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
    total = a.finish()
    assert total == b.finish()
    assert_same(bits(b.projector.getProjectedBuffer().read(np.float32)).reshape(n, 8), bits(proj_ref), "disc L346")
    counts = a.binner.getTileCountsBuffer().read(np.uint32)
    offsets = a.binner.getTileOffsetsBuffer().read(np.uint32)
    idx = a.binner.getTileIndicesBuffer().read(np.uint32, total)
    assert_same(offsets, b.binner.getTileOffsetsBuffer().read(np.uint32), "disc L350")
    assert_same(idx, b.binner.getTileIndicesBuffer().read(np.uint32, total), "disc L351")
    img = a.readPixelsFloat()
    assert_same(img.view(np.uint32), b.readPixelsFloat().view(np.uint32), "disc L353")
    assert int(counts.sum(dtype=np.uint64)) == total
    # depth order inside every list, ties by ascending index
    depth = proj_ref[idx, 4]
    dd = np.diff(depth)
    starts = offsets[counts > 0][1:]
    bad = np.nonzero((dd < 0) | ((dd == 0) & (np.diff(idx.astype(np.int64)) <= 0)))[0] + 1
    assert np.isin(bad, starts).all()
    # every pair overlaps its tile (spot check)
    rng = np.random.default_rng(1)
    pick = rng.integers(0, total, 200000)
    tile_of = np.searchsorted(offsets, pick, side="right") - 1
    while True:
        emp = counts[tile_of] == 0
        if not emp.any():
            break
        tile_of[emp] += 1
    tx, ty = tile_of % 120, tile_of // 120
    bb = proj_ref[idx[pick]]
    assert (np.maximum(bb[:, 0], 0) < (tx + 1) * 16).all() and (np.minimum(bb[:, 2], w) >= tx * 16).all()
    assert (np.maximum(bb[:, 1], 0) < (ty + 1) * 16).all() and (np.minimum(bb[:, 3], h) >= ty * 16).all()
    # sampled pixels: the "over" composite of the pixel's own tile list, evaluated in f64 from the f32 records
    k = np.float64(1.0) / np.sqrt(np.float64(3.0))
    worst, rim_hits = 0.0, 0
    for _ in range(300):
        px, py = int(rng.integers(0, w)), int(rng.integers(0, h))
        t = (py // 16) * 120 + (px // 16)
        lst = idx[offsets[t]:offsets[t] + counts[t]]
        r = discs_ref[lst].astype(np.float64)
        dx, dy = (px + 0.5) - r[:, 0], (py + 0.5) - r[:, 1]
        den = 1.0 - (r[:, 6] * dx + r[:, 7] * dy)
        uu, vv = (r[:, 2] * dx + r[:, 3] * dy) / den, (r[:, 4] * dx + r[:, 5] * dy) / den
        d2 = uu * uu + vv * vv
        if (np.abs(d2 - 1.0) < 1e-3).any():
            rim_hits += 1
            continue
        bnd = proj_ref[lst, :4].astype(np.float64)
        inside = (d2 <= 1.0) & (px + 0.5 >= bnd[:, 0]) & (px + 0.5 <= bnd[:, 2]) & (py + 0.5 >= bnd[:, 1]) & (py + 0.5 <= bnd[:, 3])
        g = np.where(inside, np.exp(-0.5 * d2 / 0.16), 0.0)
        trans = np.concatenate([[1.0], np.cumprod(1.0 - g)])
        nr = normals[lst].astype(np.float64)
        kd = 0.85 + 0.15 * np.maximum((nr[:, 0] + nr[:, 1] + nr[:, 2]) * k, 0.0)
        col = props[lst, 4:7].astype(np.float64) * kd[:, None]
        want = (col * (g * trans[:-1])[:, None]).sum(axis=0) + np.array([0.05, 0.05, 0.1]) * trans[-1]
        worst = max(worst, float(np.abs(img[py, px, :3] - want).max()))
    assert worst <= 1e-4 and rim_hits < 100
    for o in (a, b, pbuf, nbuf):
        o.destroy()


@pytest.mark.parametrize("footprint", ["isotropic", "disc"])
@pytest.mark.parametrize("prelit", [False, True])
def test_exchange_free_bands_stitch_to_the_single_gpu_frame(device, footprint, prelit):
    """dist.LocalBandRenderer: every (virtual) rank projects all splats from its own copy and renders its band — no
    exchange.  The stitched rgba8 image must be bit-identical to the single-GPU frame, for both footprints."""
    import torch
    from splat_renderer_amd import dist
    world, n, w, h = 3, 30001, 400, 232
    props, normals, u = make_case(n, w, h, 41, 1.5)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    full = sr.Renderer(device, None, "rgba8unorm", n, footprint=footprint)
    full.render(u, pbuf, nbuf, None, w, h)
    want = full.readPixels().copy()
    device.sync()
    pt, nt = torch.from_numpy(props).cuda(), torch.from_numpy(normals).cuda()
    stages = dist.HipStages(torch, 0, n, w, h, footprint=footprint)
    if prelit:
        stages.set_lit(pt.data_ptr(), nt.data_ptr(), n)
        assert_same(stages.pos_plane.cpu().numpy(), props[:, :4], "disc L421")
    got = np.zeros_like(want)
    for rank in range(world):
        lr = dist.LocalBandRenderer(stages, n, w, h, rank, world)
        for _ in range(2):  # the second one is a sync-free frame
            lr.render(u, pt.data_ptr(), nt.data_ptr())
        lr.render(u, pt.data_ptr(), nt.data_ptr(), settle=True)
        torch.cuda.synchronize()
        r0, r1 = lr.pixel_rows()
        got[r0:r1] = lr.image.cpu().numpy()[r0:r1]
    assert_same(got, want, "disc L431")
    stages.destroy()
    for o in (full, pbuf, nbuf):
        o.destroy()


@pytest.mark.parametrize("footprint", ["isotropic", "disc"])
def test_full_size_C2_eight_exchange_free_bands_stitch_bit_identically(device, footprint):
    """5M @1080p cut into 8 bands of tile rows, each rendered from all splats with the projector's conservative
    reach test (k_project_hist_band): a single splat wrongly rejected would change its band's pixels."""
    from splat_renderer_amd import dist
    n, w, h = sr.scene.CONFIGS["C2"]
    props, normals, u = make_case(n, w, h)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    r = sr.Renderer(device, None, "rgba8unorm", n, footprint=footprint)
    r.render(u, pbuf, nbuf, None, w, h)
    want = r.readPixels().copy()
    total = r.finish()
    counts_full = r.binner.getTileCountsBuffer().read(np.uint32).copy()
    nty = -(-h // 16)
    bands = dist.balanced_rows(counts_full.reshape(nty, -1).sum(axis=1), 8)
    got = np.zeros_like(want)
    pairs = 0
    for (r0, r1) in bands:
        r.render(u, pbuf, nbuf, None, w, h, tileRows=(r0, r1))
        got[r0 * 16:min(r1 * 16, h)] = r.readPixels()[r0 * 16:min(r1 * 16, h)]
        c = r.binner.getTileCountsBuffer().read(np.uint32).reshape(nty, -1)
        assert np.array_equal(c[r0:r1], counts_full.reshape(nty, -1)[r0:r1])  # the band's lists are the full frame's
        pairs += int(c[r0:r1].sum(dtype=np.uint64))
    assert pairs == total
    assert_same(got, want, "disc L461")
    for o in (r, pbuf, nbuf):
        o.destroy()


@pytest.mark.parametrize("footprint", ["isotropic", "disc"])
def test_frame_of_no_splats_is_the_background(device, footprint):
    w, h = 96, 64
    _, _, u = make_case(4, w, h, 1)
    pbuf, nbuf = device.createBuffer(256), device.createBuffer(256)
    pbuf.zero()
    nbuf.zero()
    r = sr.Renderer(device, None, "rgba8unorm", 0, footprint=footprint)
    for _ in range(2):
        r.render(u, pbuf, nbuf, None, w, h)
    img = r.readPixels()
    assert (img == np.array([13, 13, 26, 255], np.uint8)).all()  # (0.05, 0.05, 0.1, 1) as rgba8unorm
    assert r.finish() == 0
    for o in (r, pbuf, nbuf):
        o.destroy()


@pytest.mark.parametrize("kernel", DISC_KERNELS)
def test_disc_random_scenes(device, kernel):
    try:
        force_kernel(device, kernel)
        disc_random_scenes(device)
    finally:
        device.compositeOptions()


def disc_random_scenes(device):
    """Seeded sweep over sizes, aspect ratios, splat scales, camera distances (inside the cube too: splats behind the
    eye and discs crossing w = 0 are culled) and normal distributions (unit, scaled, near the |n.y| = 0.9 switch, zero):
    records and lists against the oracle bit for bit, the image within the stated tolerance."""
    rng = np.random.default_rng(20261004)
    for case in range(24):
        n = int(rng.integers(1, 6000))
        w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
        rs = float(rng.choice([0.05, 0.3, 1.0, 2.5, 8.0]))
        cam = dict(distance=float(rng.uniform(0.5, 6.0)), azimuth=float(rng.uniform(0, 6.28)), elevation=float(rng.uniform(-1.2, 1.2)))
        props, normals, u = make_case(n, w, h, 2000 + case, rs, camera=cam)
        kind = case % 4
        if kind == 1:
            normals[:, :3] *= rng.uniform(0.2, 3.0, (n, 1)).astype(np.float32)
        elif kind == 2:
            normals[:, 1] = np.float32(0.9) + rng.uniform(-1e-3, 1e-3, n).astype(np.float32)
        elif kind == 3:
            normals[rng.random(n) < 0.2] = 0
        ref = oracle_disc(props, normals, u, w, h, early_out=False)
        pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
        r = sr.Renderer(device, None, "rgba8unorm", n, earlyOut=False, footprint="disc",
                        frameOrder="tileFirst" if case % 2 else "sortFirst")
        r.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
        tag = (case, n, w, h, rs, cam)
        total = ref["indices"].shape[0]
        assert r.finish() == total, tag
        assert_same(bits(r.projector.getProjectedBuffer().read(np.float32)).reshape(n, 8), bits(ref["proj"]), ("disc L509", tag))
        assert_same(r.binner.getTileCountsBuffer().read(np.uint32), ref["counts"], ("disc L510", tag))
        if total:
            assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], ("disc L512", tag))
        got = r.readPixelsFloat()
        d = np.abs(got - ref["img"]).max(axis=2)
        off = ref["rim"] == 0
        assert (not off.any() or d[off].max() <= TOL) and d.max() <= TOL_RIM, (tag, float(d.max()))
        for o in (r, pbuf, nbuf):
            o.destroy()
