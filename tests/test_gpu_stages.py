"""GPU parity tests: every stage through the C ABI (via the host classes) against the oracle.

Bit-exact: ProjectedSplat records, keys, payload, sort order, scan, tile counts/offsets/lists.
Tolerance (stated below): composited pixels.
"""
import ctypes as C
import os

import numpy as np
import pytest

import splat_renderer_amd as sr
from oracle import oracle as O
from splat_renderer_amd import _lib
from tests.helpers import assert_same, make_case, oracle_pipeline

pytestmark = pytest.mark.gpu

# ---- composite tolerance (float RGBA in [0,1]) -------------------------------------------------
# early-out OFF: the GPU evaluates exp2(d^2 * k) instead of exp(-0.5 nd^2 / 0.25) and contracts
# FMAs; both are a few ulp per layer.
TOL_NO_EARLY_OUT = 2e-5
# early-out ON: a pixel whose alpha lands within float noise of the 0.99 threshold may stop one
# entry earlier/later than the oracle; what it then gains/loses is bounded by (1-0.99) * max colour.
TOL_EARLY_OUT_BOUND = 0.0101
FRAC_ABOVE_TIGHT = 2e-3  # at most this fraction of pixels may exceed TOL_NO_EARLY_OUT with early-out on


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def lit_records(u, props, normals):
    """The frame's 32-byte lit composite records (SPLAT_RECORDS_LIT32) from the oracle's stages: {centre x, y,
    screen radius, depth} = the oracle's compact record, {lit r, g, b, opacity} = the reference's shading
    (ComputeShaderRenderer.ts:143-145) in one IEEE operation per operator."""
    k = np.float32(0.577350269189625764)
    ndl = (normals[:, 0] * k + normals[:, 1] * k) + normals[:, 2] * k
    kd = np.float32(0.85) + np.float32(0.15) * np.maximum(ndl, np.float32(0))
    rec = np.empty((props.shape[0], 8), np.float32)
    rec[:, :4] = O.project_compact(u, props)
    rec[:, 4:7] = props[:, 4:7] * kd[:, None]
    rec[:, 7] = props[:, 7]
    return rec


def tile_max(a, tile):
    """Per-tile maximum of a per-pixel array (ragged edges included)."""
    h, w = a.shape
    nty, ntx = -(-h // tile), -(-w // tile)
    p = np.zeros((nty * tile, ntx * tile), a.dtype)
    p[:h, :w] = a
    return p.reshape(nty, tile, ntx, tile).max(axis=(1, 3))


def check_image_against_oracle(got, got8, want, want8, near=None):
    """The composite's stated tolerance.  Every pixel within TOL_NO_EARLY_OUT / 1 LSB of the oracle — except, with
    early-out on, the pixels the oracle flags as `near`: their alpha came within 2e-5 of the 0.99 threshold at some
    entry, so a correct float evaluation may stop one entry earlier or later, which is worth at most
    (1 - 0.99) * max colour (TOL_EARLY_OUT_BOUND, 3 LSB)."""
    err = np.abs(got - want).max(axis=2)
    err8 = np.abs(got8.astype(int) - want8.astype(int)).max(axis=2)
    if near is None:
        assert err.max() <= TOL_NO_EARLY_OUT
        assert err8.max() <= 1
        return
    strict = near == 0
    assert err[strict].max(initial=0) <= TOL_NO_EARLY_OUT, f"{(err[strict] > TOL_NO_EARLY_OUT).sum()} pixels off the threshold differ"
    assert err8[strict].max(initial=0) <= 1
    assert err.max() <= TOL_EARLY_OUT_BOUND and err8.max() <= 3


@pytest.mark.parametrize("n,w,h,seed", [(1, 64, 64, 1), (7, 64, 48, 2), (1000, 256, 256, 3), (10000, 256, 256, 1234),
                                        (50000, 640, 360, 5)])
def test_project_and_keys_bit_exact(device, n, w, h, seed):
    props, normals, u = make_case(n, w, h, seed)
    ref = oracle_pipeline(props, normals, u, w, h, n_padded=sr.scene.padded_size(n))
    pm = sr.SplatPropertyManager(device, n)
    pm.setFromArrays(props)
    proj = sr.SplatProjector(device, n)
    sorter = sr.RadixSorter(device, n)
    ext = sr.DepthKeyExtractor(device)
    enc = device.createCommandEncoder()
    proj.project(enc, u, pm.getPropertyBuffer())
    ext.extract(enc, proj.getProjectedBuffer(), sorter.getKeysBuffer(), sorter.getPayloadBuffer(), n, sorter.paddedSize)
    got = proj.getProjectedBuffer().read(np.float32).reshape(n, 8)
    assert_same(bits(got), bits(ref["proj"]), "L86")
    assert sorter.paddedSize == sr.scene.padded_size(n)
    assert_same(sorter.getKeysBuffer().read(np.uint32), ref["keys"], "L88")
    assert_same(sorter.getPayloadBuffer().read(np.uint32), ref["payload"], "L89")
    # fused form writes the same keys
    sorter.getKeysBuffer().zero()
    proj.project(enc, u, pm.getPropertyBuffer(), sorter.getKeysBuffer(), sorter.getPayloadBuffer(), sorter.paddedSize)
    assert_same(sorter.getKeysBuffer().read(np.uint32), ref["keys"], "L93")
    assert_same(bits(proj.getProjectedBuffer().read(np.float32)), bits(ref["proj"]).reshape(-1), "L94")
    for o in (pm, proj, sorter):
        o.destroy()


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 255, 256, 1023, 4095, 4096, 4097, 8191, 12289, 100000, 1000003])
@pytest.mark.parametrize("kind", ["random", "few_values", "all_equal", "sorted_desc"])
@pytest.mark.parametrize("mode", [0, 2], ids=["policy", "ballot"])
def test_radix_sort_stable(device, n, kind, mode):
    rng = np.random.default_rng(n * 7 + len(kind))
    if kind == "random":
        keys = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    elif kind == "few_values":  # heavy duplicates -> stability matters, skewed digits
        keys = rng.choice(np.array([0, 1, 0x80000000, 0xFFFFFFFF, 0x3F800000], np.uint32), size=n)
    elif kind == "all_equal":
        keys = np.full(n, 0xDEADBEEF, np.uint32)
    else:
        keys = np.arange(n, 0, -1, dtype=np.uint32) * np.uint32(2654435761)
        keys = np.sort(keys)[::-1].copy()
    payload = np.arange(n, dtype=np.uint32)
    s = sr.RadixSorter(device, max(n, 1))
    s.setMode(mode)
    if n:
        s.getKeysBuffer().write(keys)
        s.getPayloadBuffer().write(payload)
    s.sort(n)
    order = np.argsort(keys, kind="stable").astype(np.uint32)
    assert_same(s.getSortedIndicesBuffer().read(np.uint32, n), order, "L122")
    assert_same(s.getSortedKeysBuffer().read(np.uint32, n), keys[order], "L123")
    s.destroy()


@pytest.mark.parametrize("bits_range", [(0, 8), (0, 13), (8, 16), (0, 15), (5, 6)])
def test_radix_sort_bit_ranges(device, bits_range):
    b0, b1 = bits_range
    n = 50000
    rng = np.random.default_rng(b0 * 100 + b1)
    keys = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    payload = rng.permutation(n).astype(np.uint32)
    s = sr.RadixSorter(device, n)
    s.getKeysBuffer().write(keys)
    s.getPayloadBuffer().write(payload)
    s.sort(n, b0, b1)
    sub = (keys >> np.uint32(b0)) & np.uint32((1 << (b1 - b0)) - 1)
    order = np.argsort(sub, kind="stable")
    assert_same(s.getSortedIndicesBuffer().read(np.uint32, n), payload[order], "L140")
    assert_same(s.getSortedKeysBuffer().read(np.uint32, n), keys[order], "L141")
    s.destroy()


def test_lds_atomic_order_probe(device):
    """The fast ranking path rests on a measured hardware property; the probe must report it."""
    import ctypes as C
    from splat_renderer_amd import _lib
    bad = C.c_uint64(1)
    _lib.check(device.lib.splat_probe_lds_atomic_order(device.ctx, C.byref(bad)), device.ctx)
    assert bad.value == 0


def test_sort_capacity_error(device):
    s = sr.RadixSorter(device, 100)
    with pytest.raises(sr.SplatError):
        s.sort(s.paddedSize + 1)
    s.destroy()


@pytest.mark.parametrize("n", [1, 5, 255, 256, 1024, 1025, 8160, 16384, 16385, 32400, 100000, 1 << 20, 3000001])
def test_scan_exclusive(device, n):
    scanner = sr.PrefixSumScanner(device)
    rng = np.random.default_rng(n)
    a = rng.integers(0, 2000, size=n, dtype=np.uint32)
    if n == 5:
        a = np.array([1, 2, 3, 4, 5], np.uint32)  # the reference's one stated known answer
    src = device.createBufferFrom(a)
    dst = device.createBuffer(n * 4)
    tot = device.createBuffer(16)
    scanner.scan(None, src, dst, n, tot)
    want = np.concatenate([[0], np.cumsum(a, dtype=np.uint64)[:-1]]).astype(np.uint32)
    assert_same(dst.read(np.uint32, n), want, "L173")
    assert int(tot.read(np.uint32, 1)[0]) == int(a.sum(dtype=np.uint64) & 0xFFFFFFFF)
    if n == 5:
        assert dst.read(np.uint32, 5).tolist() == [0, 1, 3, 6, 10]  # GPU_PIPELINE_PLAN.md:632-635
    scanner.scan(None, src, src, n)  # in place
    assert_same(src.read(np.uint32, n), want, "L178")
    for b in (src, dst, tot):
        b.destroy()


CASES = [(1, 64, 64, 1, 1.0), (7, 64, 48, 2, 1.0), (1000, 256, 256, 3, 1.0), (10000, 256, 256, 1234, 1.0),
         (20000, 250, 130, 9, 1.0),  # ragged: width/height not multiples of 16
         (3000, 96, 80, 7, 3.0)]


def run_gpu_pipeline(device, props, normals, u, n, w, h, tile=16):
    pm = sr.SplatPropertyManager(device, n)
    pm.setFromArrays(props)
    nbuf = device.createBufferFrom(normals)
    proj = sr.SplatProjector(device, n)
    sorter = sr.RadixSorter(device, n)
    ext = sr.DepthKeyExtractor(device)
    binner = sr.GPUTileBinner(device, tile)
    enc = device.createCommandEncoder()
    proj.project(enc, u, pm.getPropertyBuffer())
    ext.extract(enc, proj.getProjectedBuffer(), sorter.getKeysBuffer(), sorter.getPayloadBuffer(), n, sorter.paddedSize)
    sorter.sort()
    binner.binSplats(enc, proj.getProjectedBuffer(), sorter.getSortedIndicesBuffer(), n, w, h)
    return dict(pm=pm, nbuf=nbuf, proj=proj, sorter=sorter, binner=binner)


def destroy_all(g):
    for k in ("pm", "proj", "sorter", "binner"):
        g[k].destroy()
    g["nbuf"].destroy()


@pytest.mark.parametrize("n,w,h,seed,rs", CASES)
def test_sort_and_bin_bit_exact(device, n, w, h, seed, rs):
    props, normals, u = make_case(n, w, h, seed, rs)
    ref = oracle_pipeline(props, normals, u, w, h)
    g = run_gpu_pipeline(device, props, normals, u, n, w, h)
    assert_same(g["sorter"].getSortedIndicesBuffer().read(np.uint32, n), ref["order"][:n], "L215")
    b = g["binner"]
    assert b.getTotalIndices() == ref["indices"].shape[0]
    assert_same(b.getTileCountsBuffer().read(np.uint32), ref["counts"], "L218")
    assert_same(b.getTileOffsetsBuffer().read(np.uint32), ref["offsets"], "L219")
    assert_same(b.getTileIndicesBuffer().read(np.uint32, ref["indices"].shape[0]), ref["indices"], "L220")
    destroy_all(g)


def test_binner_getters_throw_before_run(device):
    b = sr.GPUTileBinner(device, 16)
    for getter in (b.getTileOffsetsBuffer, b.getTileIndicesBuffer, b.getTileCountsBuffer):
        with pytest.raises(sr.SplatError, match="not initialized"):  # GPUTileBinner.ts:340-359
            getter()
    assert b.getTileSize() == 16
    b.destroy()


def test_bin_offscreen_and_padding(device):
    """Splats fully off-screen are culled (TileBinner.ts:437); 0xFFFFFFFF padding bins nowhere."""
    n, w, h = 6, 64, 64
    rec = np.zeros((n, 8), np.float32)
    rec[0, :4] = [-50, 10, -20, 30]     # left of the screen
    rec[1, :4] = [10, 70, 30, 90]       # below the screen
    rec[2, :4] = [-5, -5, 5, 5]         # straddles the top-left corner -> tile 0
    rec[3, :4] = [60, 60, 100, 100]     # straddles the bottom-right corner -> tile 15
    rec[4, :4] = [16, 16, 32, 32]       # exactly on tile edges -> tiles (1..2, 1..2)
    rec[5, :4] = [np.nan, 0, 10, 10]    # NaN bins nowhere
    sorted_idx = np.array([3, 0xFFFFFFFF, 4, 2, 1, 0, 5, 0xFFFFFFFF], np.uint32)
    counts, offsets, idx = O.bin_sorted(rec, sorted_idx, w, h)
    assert counts.sum() == 1 + 1 + 4
    pbuf = device.createBufferFrom(rec)
    sbuf = device.createBufferFrom(sorted_idx)
    b = sr.GPUTileBinner(device, 16)
    b.binSplats(None, pbuf, sbuf, n, w, h, numSorted=sorted_idx.shape[0])
    assert_same(b.getTileCountsBuffer().read(np.uint32), counts, "L250")
    assert_same(b.getTileOffsetsBuffer().read(np.uint32), offsets, "L251")
    assert_same(b.getTileIndicesBuffer().read(np.uint32, idx.shape[0]), idx, "L252")
    b.destroy()
    pbuf.destroy()
    sbuf.destroy()


def test_bin_empty(device):
    rec = np.zeros((4, 8), np.float32)
    rec[:, :4] = [-10, -10, -5, -5]
    b = sr.GPUTileBinner(device, 16)
    pbuf = device.createBufferFrom(rec)
    sbuf = device.createBufferFrom(np.arange(4, dtype=np.uint32))
    b.binSplats(None, pbuf, sbuf, 4, 100, 50)
    assert b.getTotalIndices() == 0
    assert not b.getTileCountsBuffer().read(np.uint32).any()
    assert not b.getTileOffsetsBuffer().read(np.uint32).any()
    b.getTileIndicesBuffer()
    b.destroy()


# Which kernel composites, and for k_composite_px on which schedule (Device.compositeOptions; every combination must give the
# oracle's image and the oracle's per-tile stop positions):
#   default   the library's choice: k_composite on these screens (fewer than 2048 tiles)
#   px1       k_composite_px, builder one chunk ahead, no lane runs ahead, first launch (no history)
#   px2       k_composite_px as frames run it (builder two chunks ahead, lanes run ahead), first launch
#   px2_warm  the same after two launches over the same lists: every tile's look-ahead is what the launch before walked
#   px2_under / px1_under  the launch before saw EMPTY lists (costs 0): every tile that needs a second chunk is mispredicted and
#             takes the on-demand path (exposed gather, extra barrier)
COMPOSITE_KERNELS = ["default", "px1", "px2", "px2_warm", "px2_under", "px1_under"]


@pytest.mark.parametrize("n,w,h,seed,rs", CASES)
@pytest.mark.parametrize("mode", [sr.MODE_FRONT_TO_BACK, sr.MODE_REFERENCE_LITERAL])
@pytest.mark.parametrize("early_out", [False, True])
@pytest.mark.parametrize("kernel", COMPOSITE_KERNELS)
def test_composite_vs_oracle(device, n, w, h, seed, rs, mode, early_out, kernel):
    if kernel != "default" and mode != sr.MODE_FRONT_TO_BACK:
        pytest.skip("k_composite_px composites nearest-on-top only")
    try:
        if kernel != "default":
            device.compositeOptions("pixel", ahead=1 if kernel.startswith("px1") else 2, predict=True)
        composite_vs_oracle(device, n, w, h, seed, rs, mode, early_out, kernel)
    finally:
        device.compositeOptions()  # the library's defaults again (the context is shared)


def composite_vs_oracle(device, n, w, h, seed, rs, mode, early_out, kernel):
    props, normals, u = make_case(n, w, h, seed, rs)
    ref = oracle_pipeline(props, normals, u, w, h)
    want, want8, _, stop, near = O.composite(mode, early_out, props[:, 4:], normals, ref["proj"], ref["indices"], ref["counts"],
                                             ref["offsets"], w, h, want_stops=True)
    g = run_gpu_pipeline(device, props, normals, u, n, w, h)
    b = g["binner"]
    ntx, nty = -(-w // 16), -(-h // 16)
    for fmt in (_lib.RECORDS_PROJECTED, _lib.RECORDS_LIT32):
        r = sr.ComputeShaderRenderer(device, None, "rgba8unorm", mode=mode, earlyOut=early_out, recordFormat=fmt)
        r.consumedBuffer = device.createBuffer(ntx * nty * 16)
        r.consumedBuffer.zero()
        device.forgetCompositeHistory()  # (the shared context may hold another scene's per-tile costs for this screen size)
        records = g["proj"].getProjectedBuffer()
        if fmt == _lib.RECORDS_LIT32:  # the frame's lit composite records, built here from the oracle's stages
            records = device.createBufferFrom(lit_records(u, props, normals))
        render = lambda: r.render(u, g["pm"].getPropertyBuffer(), b.getTileIndicesBuffer(), g["nbuf"], records,
                                  b.getTileCountsBuffer(), b.getTileOffsetsBuffer(), 16, ntx, w, h, wantFloat=True)
        if kernel == "px2_warm":  # two launches leave costs and an order behind for the third
            render()
            render()
        elif kernel.endswith("_under"):  # a launch over empty lists leaves cost 0 for every tile
            cbuf = b.getTileCountsBuffer()
            saved = cbuf.read(np.uint32).copy()
            cbuf.write(np.zeros_like(saved))
            render()
            cbuf.write(saved)
        r.consumedBuffer.zero()
        render()
        got = r.readPixelsFloat()
        got8 = r.readPixels()
        check_image_against_oracle(got, got8, want, want8, near if early_out else None)
        assert (got8[..., 3] == 255).all()
        # per tile {entries staged, entries consumed}: consumed = the largest number of entries any pixel of the tile
        # visits (SURVEY §8d P_used), exact wherever no pixel of the tile sits on the threshold
        cons = r.consumedBuffer.read(np.uint64).reshape(nty * ntx, 2)
        tile_stop, tile_near = tile_max(stop, 16), tile_max(near, 16) > 0
        ok = ~tile_near.reshape(-1)
        assert_same(cons[ok, 1], tile_stop.reshape(-1)[ok].astype(np.uint64), "L301")
        # entries gathered: k_composite_px (nearest-on-top on screens of at least 2048 tiles, or SPLAT_COMPOSITE=pixel) works in
        # chunks of 32; its builder stays two chunks ahead of the chunk being walked and fetches two further ahead — unless
        # the previous launch over the same band left costs behind, in which case it gathers what THAT launch walked and
        # anything beyond only when it is needed (the shared test context may hold such a history, from another scene:
        # any bound gives the same image).  k_composite (smaller screens, the reference-literal blend,
        # SPLAT_COMPOSITE=quadrant) stages batches of 256
        counts64 = ref["counts"].astype(np.uint64)
        forced = "p" if kernel != "default" else os.environ.get("SPLAT_COMPOSITE", "")[:1].lower()
        px = mode == sr.MODE_FRONT_TO_BACK and (forced == "p" or (forced != "q" and ntx * nty >= 2048))
        if px and early_out:
            walked = (cons[:, 1] + np.uint64(31)) // np.uint64(32) * np.uint64(32)
            # at least every walked chunk, at most the look-ahead beyond them (2 built + 2 fetched); a tile is done when all
            # its pixels have stopped at a chunk's END, so one whose last pixel stops on a chunk's last entry walks no more
            assert np.all(cons[:, 0] >= np.minimum(counts64, cons[:, 1])), "entries staged per tile (k_composite_px): fewer than consumed"
            assert np.all(cons[:, 0] <= np.minimum(counts64, walked + np.uint64(128))), "entries staged per tile (k_composite_px): beyond the look-ahead"
            if kernel == "px2_warm":  # gathered: the chunks the tile touched in the launch before (the last walked, or the one after it) + one
                assert np.all(cons[:, 0] <= np.minimum(counts64, walked + np.uint64(64))), "a warm launch gathers what the one before needed and one chunk more"
            staged_want = cons[:, 0]
        elif px:
            staged_want = counts64
        else:
            staged_want = np.minimum(counts64, (cons[:, 1] + np.uint64(255)) // np.uint64(256) * np.uint64(256))
        assert_same(cons[:, 0], staged_want, "entries staged per tile")
        if not early_out:
            assert_same(cons[:, 1], ref["counts"].astype(np.uint64), "L305")
        if fmt == _lib.RECORDS_LIT32:
            records.destroy()
        r.destroy()
    destroy_all(g)


def test_tile_renderer_fronts_composite(device):
    """TileRenderer.render with EXACTLY the reference's eleven arguments (src/TileRenderer.ts:234-246: uniformData, splatPropertyBuffer,
    splatIndicesBuffer, curvatureBuffer, tileCountsData: Uint32Array, numTilesX, numTilesY, tileSize, maxSplatsPerTile, width, height) and
    no prior bind: the projected records and tile offsets are those of the projector and binner that last ran on the device — after
    the staged stages, and after a whole-frame Renderer (whose projector leaves lit composite records).  bindTileData overrides."""
    n, w, h = 2000, 128, 96
    props, normals, u = make_case(n, w, h, 11, 2.0)
    ref = oracle_pipeline(props, normals, u, w, h)
    want, _, _ = O.composite(O.MODE_FRONT_TO_BACK, True, props[:, 4:], normals, ref["proj"], ref["indices"], ref["counts"],
                             ref["offsets"], w, h)
    fresh = sr.Device(0)  # nothing has run on it: nothing to composite from
    with pytest.raises(sr.SplatError, match="no SplatProjector"):
        sr.TileRenderer(fresh, None, "rgba8unorm").render(u, None, None, None, ref["counts"], 8, 6, 16, 4096, w, h)
    fresh.destroy()
    g = run_gpu_pipeline(device, props, normals, u, n, w, h)  # (SplatProjector.project ... GPUTileBinner.binSplats on `device`)
    b = g["binner"]
    tr = sr.TileRenderer(device, None, "rgba8unorm")
    counts_host = b.getTileCountsBuffer().read(np.uint32)  # the reference's tileCountsData: a Uint32Array on the host
    args = (u, g["pm"].getPropertyBuffer(), b.getTileIndicesBuffer(), g["nbuf"], counts_host, 8, 6, 16, 4096, w, h)
    tr.render(*args)  # the reference's call, argument for argument
    first8 = tr.readPixels().copy()
    tr.render(*args, wantFloat=True)  # (+ the float image, to hold it to the oracle)
    assert_same(tr.readPixels(), first8, "tile renderer: the same call, the same bytes")
    got = tr.readPixelsFloat().copy()
    assert np.abs(got - want).max() <= TOL_EARLY_OUT_BOUND
    with pytest.raises(sr.SplatError):  # a count per tile
        tr.render(u, g["pm"].getPropertyBuffer(), b.getTileIndicesBuffer(), g["nbuf"], counts_host[:-1], 8, 6, 16, 4096, w, h)
    # the override: the same buffers bound by hand give the same bits
    tr2 = sr.TileRenderer(device, None, "rgba8unorm")
    tr2.bindTileData(g["proj"].getProjectedBuffer(), b.getTileCountsBuffer(), b.getTileOffsetsBuffer())
    tr2.render(*args, wantFloat=True)
    assert_same(tr2.readPixelsFloat().view(np.uint32), got.view(np.uint32), "tile renderer: bound = found")
    # after a whole frame: the Renderer's projector (lit composite records) and binner are the device's last
    r = sr.Renderer(device, None, "rgba8unorm", n)
    r.render(u, g["pm"].getPropertyBuffer(), g["nbuf"], None, w, h, wantFloat=True)
    frame = r.readPixelsFloat().copy()
    tr.render(u, g["pm"].getPropertyBuffer(), r.binner.getTileIndicesBuffer(), g["nbuf"], r.binner.getTileCountsBuffer().read(np.uint32),
              8, 6, 16, 4096, w, h, wantFloat=True)
    assert tr.recordFormat == _lib.RECORDS_LIT32
    assert_same(tr.readPixelsFloat().view(np.uint32), frame.view(np.uint32), "tile renderer after a frame = the frame")
    for o in (tr, tr2, r):
        o.destroy()
    destroy_all(g)


@pytest.mark.parametrize("bands", [2, 3, 8])
def test_band_rendering_stitches_bit_identically(device, bands):
    """rendering tile-row bands separately (what each rank of a multi-GPU frame does)
    and stitching gives the single-device image bit for bit."""
    n, w, h = 20000, 320, 200
    props, normals, u = make_case(n, w, h, 21, 1.5)
    pbuf = device.createBufferFrom(props)
    nbuf = device.createBufferFrom(normals)
    full = sr.Renderer(device, None, "rgba8unorm", n)
    full.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
    want = full.readPixelsFloat().copy()
    nty = -(-h // 16)
    stitched = np.zeros_like(want)
    r = sr.Renderer(device, None, "rgba8unorm", n)
    for k in range(bands):
        r0, r1 = nty * k // bands, nty * (k + 1) // bands
        r.output and r.output.zero()
        r.render(u, pbuf, nbuf, None, w, h, tileRows=(r0, r1), wantFloat=True)
        img = r.readPixelsFloat()
        stitched[r0 * 16:min(r1 * 16, h)] = img[r0 * 16:min(r1 * 16, h)]
    assert_same(stitched.view(np.uint32), want.view(np.uint32), "L351")
    full.destroy()
    r.destroy()
    pbuf.destroy()
    nbuf.destroy()


def test_update_props(device):
    n = 5000
    rng = np.random.default_rng(5)
    pos = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    cur = rng.standard_normal((n, 4)).astype(np.float32)
    want = O.update_props(pos, cur)
    pm = sr.SplatPropertyManager(device, n)
    d0 = pm.getPropertyBuffer().read(np.float32).reshape(n, 8)
    assert np.allclose(d0[:, 3], 0.04) and np.allclose(d0[:, 7], 0.7) and (d0[:, 4:7] == 1).all()  # :33-50
    pb, cb = device.createBufferFrom(pos), device.createBufferFrom(cur)
    pm.updateFromCurvature(None, pb, cb)
    got = pm.getPropertyBuffer().read(np.float32).reshape(n, 8)
    assert_same(bits(got), bits(want), "L370")
    for o in (pm, pb, cb):
        o.destroy()


@pytest.mark.parametrize("records", ["lit", "projected"])
@pytest.mark.parametrize("order", ["sortFirst", "tileFirst", "default"])
def test_full_frame_C0(device, order, records):
    """BASELINE configs[0]: 10k Gaussians @256x256, whole frame through splat_render_frame."""
    n, w, h = sr.scene.CONFIGS["C0"]
    props, normals, u = make_case(n, w, h)
    ref = oracle_pipeline(props, normals, u, w, h)
    assert ref["indices"].shape[0] == 137051  # SURVEY §8 dry-run statistic for this scene
    want, want8, _, _, near = O.composite(O.MODE_FRONT_TO_BACK, True, props[:, 4:], normals, ref["proj"], ref["indices"],
                                          ref["counts"], ref["offsets"], w, h, want_stops=True)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    r = sr.Renderer(device, None, "rgba8unorm", n, frameOrder=order, records=records)
    r.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
    if order == "sortFirst":  # (the tile-first order never sorts the splats globally)
        assert_same(r.sorter.getSortedIndicesBuffer().read(np.uint32, n), ref["order"], "L389")
    assert_same(r.binner.getTileIndicesBuffer().read(np.uint32), ref["indices"], "L390")
    check_image_against_oracle(r.readPixelsFloat(), r.readPixels(), want, want8, near)
    # what the frame's projector left behind: the reference's ProjectedSplat records, or the lit composite records
    rec = bits(r.projector.getRecordsBuffer().read(np.float32)).reshape(n, 8)
    assert_same(rec, bits(lit_records(u, props, normals) if records == "lit" else ref["proj"]), "L394")
    if records == "lit":  # code written against the reference's ProjectedSplat layout must not get these by accident
        with pytest.raises(sr.SplatError):
            r.projector.getProjectedBuffer()
    else:
        assert r.projector.getProjectedBuffer() is r.projector.getRecordsBuffer()
    r.destroy()
    pbuf.destroy()
    nbuf.destroy()


@pytest.mark.parametrize("order", ["tileFirst", "sortFirst"])
def test_lit_composite_records_give_the_same_frame_bit_for_bit(device, order):
    """A frame whose projector writes the 32-byte lit composite records (SPLAT_RECORDS_LIT32: one gathered line per
    staged list entry) against the same frame from the reference's ProjectedSplat records + colour + normal gathers:
    identical lists and float image, bit for bit, from every property layout (interleaved records, two planes, planes
    with a pre-lit colour plane), in both composite modes; the records themselves against the oracle."""
    n, w, h = 60000, 500, 300
    props, normals, u = make_case(n, w, h, 41, 1.5)
    want_rec = bits(lit_records(u, props, normals))
    pm = sr.SplatPropertyManager(device, n)
    pm.setFromArrays(props)
    nbuf = device.createBufferFrom(normals)
    for mode in (sr.MODE_FRONT_TO_BACK, sr.MODE_REFERENCE_LITERAL):
        a = sr.Renderer(device, None, "rgba8unorm", n, mode=mode, frameOrder=order, records="projected")
        a.render(u, pm.getPropertyBuffer(), nbuf, None, w, h, wantFloat=True)
        total = a.finish()
        img = a.readPixelsFloat().view(np.uint32)
        lists = a.binner.getTileIndicesBuffer().read(np.uint32, total)
        for layout in ("interleaved", "planes", "prelit"):
            b = sr.Renderer(device, None, "rgba8unorm", n, mode=mode, frameOrder=order)  # records="lit" is the default
            pbuf = {"interleaved": pm.getPropertyBuffer, "planes": pm.getPropertyPlanes, "prelit": lambda: pm.getLitPlanes(nbuf)}[layout]()
            for _ in range(2):  # the second frame is sync-free
                b.render(u, pbuf, None if layout == "prelit" else nbuf, None, w, h, wantFloat=True)
            assert b.finish() == total, layout
            assert b.recordFormat == _lib.RECORDS_LIT32
            assert_same(bits(b.projector.getRecordsBuffer().read(np.float32)).reshape(n, 8), want_rec, ("L425", layout))
            assert_same(b.binner.getTileIndicesBuffer().read(np.uint32, total), lists, ("L426", layout))
            assert_same(b.readPixelsFloat().view(np.uint32), img, ("L427", layout))
            b.destroy()
        a.destroy()
    # a band of tile rows (the exchange-free multi-GPU cut): records of splats that can reach the band, same pixels
    full = sr.Renderer(device, None, "rgba8unorm", n, frameOrder=order)
    full.render(u, pm.getPropertyBuffer(), nbuf, None, w, h)
    band = sr.Renderer(device, None, "rgba8unorm", n, frameOrder=order)
    band.render(u, pm.getPropertyBuffer(), nbuf, None, w, h, tileRows=(5, 11))
    assert_same(band.readPixels()[80:176], full.readPixels()[80:176], "L435")
    for o in (full, band, pm, nbuf):
        o.destroy()


@pytest.mark.parametrize("world", [2, 4])
def test_virtual_ranks_band_frame_matches_single_gpu(device, world):
    """SURVEY §8e without a cluster: every rank's device work (project_slice -> [gather] -> band_keys
    -> sort -> bin -> composite) runs in turn on the one GPU, the all-gather is a concat; the
    stitched rgba8 image must be bit-identical to the single-GPU frame."""
    import torch
    from splat_renderer_amd import dist
    n, w, h = 30001, 400, 232  # odd n: the last shard is padded with NaN records
    props, normals, u = make_case(n, w, h, 41, 1.5)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    full = sr.Renderer(device, None, "rgba8unorm", n)
    full.render(u, pbuf, nbuf, None, w, h)
    want = full.readPixels().copy()
    device.sync()
    pt, nt = torch.from_numpy(props).cuda(), torch.from_numpy(normals).cuda()
    per = dist.shard_size(n, world)
    stages = dist.HipStages(torch, 0, per * world, w, h)
    if world == 4:  # one of the two cases with the colour plane pre-lit (as bench.py runs the multi-GPU frame)
        stages.set_lit(pt.data_ptr(), nt.data_ptr(), n)
    renderers = [dist.BandRenderer(stages, n, w, h, r, world, None) for r in range(world)]
    for br in renderers:  # phase 1: every rank projects its slice
        stages.project_slice(u, pt.data_ptr(), br.first, br.count, br.shard)
    gathered = torch.cat([br.shard for br in renderers], dim=0).contiguous()
    got = np.zeros_like(want)
    kept = []
    for br in renderers:  # phase 2: every rank renders its band from the gathered records
        stages.band_frame(gathered, per * world, pt.data_ptr(), nt.data_ptr(), br.row0, br.row1, br.image, settle=True)
        torch.cuda.synchronize()
        r0, r1 = br.pixel_rows()
        got[r0:r1] = br.image.cpu().numpy()[r0:r1]
        kept.append(stages.kept)
    assert_same(got, want, "L471")
    assert all(0 < k < n for k in kept) and sum(kept) >= n * 0.9  # bands keep a subset; overlaps allowed
    stages.destroy()
    full.destroy()
    pbuf.destroy()
    nbuf.destroy()


@pytest.mark.parametrize("name", ["tiny7", "small300", "ragged1000"])
def test_golden_fixtures(device, name):
    """Committed vectors (tests/golden/*.npz, generated by the oracle): inputs in, every stage out."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    n, w, h, _ = (int(x) for x in g["dims"])
    gpu = run_gpu_pipeline(device, g["props"], g["normals"], g["uniforms"], n, w, h)
    assert_same(bits(gpu["proj"].getProjectedBuffer().read(np.float32)), bits(g["projected"]).reshape(-1), "L486")
    assert_same(gpu["sorter"].getSortedIndicesBuffer().read(np.uint32, n), g["order"][:n], "L487")
    b = gpu["binner"]
    assert_same(b.getTileCountsBuffer().read(np.uint32), g["counts"], "L489")
    assert_same(b.getTileOffsetsBuffer().read(np.uint32), g["offsets"], "L490")
    assert_same(b.getTileIndicesBuffer().read(np.uint32, g["indices"].shape[0]), g["indices"], "L491")
    for mode, key in ((sr.MODE_FRONT_TO_BACK, "image_front_to_back"), (sr.MODE_REFERENCE_LITERAL, "image_literal")):
        r = sr.ComputeShaderRenderer(device, None, "rgba8unorm", mode=mode, earlyOut=True)
        r.render(g["uniforms"], gpu["pm"].getPropertyBuffer(), b.getTileIndicesBuffer(), gpu["nbuf"],
                 gpu["proj"].getProjectedBuffer(), b.getTileCountsBuffer(), b.getTileOffsetsBuffer(), 16, -(-w // 16), w, h,
                 wantFloat=True)
        err = np.abs(r.readPixelsFloat() - g[key])
        assert err.max() <= TOL_EARLY_OUT_BOUND and (err.max(axis=2) > TOL_NO_EARLY_OUT).mean() <= FRAC_ABOVE_TIGHT
        assert np.abs(r.readPixels().astype(int) - g[key + "_u8"].astype(int)).max() <= 3
        r.destroy()
    destroy_all(gpu)


@pytest.mark.parametrize("name", ["tiny7", "small300", "ragged1000", "edges"])
def test_tile_lists_equal_the_reference_own_code(device, name):
    """ref_binsorted_*.npz: outputs of the reference's own binSorted loops (src/TileBinner.ts:426-495) run under Node by
    tests/golden/make_ref_fixtures.py.  The HIP binner on the same records and sorted order — and, for the scene
    fixtures, the whole frame in both orders of work from the scene's properties — must give exactly those lists."""
    import os
    here = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(here, f"ref_binsorted_{name}.npz"))
    w, h, tile = (int(x) for x in g["dims"])
    n = g["projected"].shape[0]
    pbuf, sbuf = device.createBufferFrom(g["projected"]), device.createBufferFrom(g["sorted"])
    # (what a mismatch report needs to tell a tie-order swap — the ranking — from anything else: the depth keys, the
    # offsets that name the tile, and which ranking / block size this process used)
    keys = O.extract_keys(g["projected"])[0]
    info = {"SPLAT_RANK": os.environ.get("SPLAT_RANK", "(library default)"), "n": n, "tiles": int(g["counts"].shape[0]),
            "first_pass_block": 256 if n <= (1 << 20) else 1024}
    b = sr.GPUTileBinner(device, tile)
    b.binSplats(None, pbuf, sbuf, n, w, h, numSorted=g["sorted"].shape[0])
    assert_same(b.getTileCountsBuffer().read(np.uint32), g["counts"], (name, "staged binner", "counts"), extra=info)
    assert_same(b.getTileOffsetsBuffer().read(np.uint32), g["offsets"], (name, "staged binner", "offsets"), extra=info)
    assert_same(b.getTileIndicesBuffer().read(np.uint32, g["indices"].shape[0]), g["indices"], (name, "staged binner", "lists"),
                offsets=g["offsets"], keys=keys, extra=info)
    for o in (b, pbuf, sbuf):
        o.destroy()
    if name == "edges":
        return
    f = np.load(os.path.join(here, name + ".npz"))
    props, nbuf = device.createBufferFrom(f["props"]), device.createBufferFrom(f["normals"])
    for order in ("tileFirst", "sortFirst"):
        r = sr.Renderer(device, None, "rgba8unorm", n, frameOrder=order)
        r.render(f["uniforms"], props, nbuf, None, w, h)
        total = r.finish()
        assert total == g["indices"].shape[0], (name, order, total)
        assert_same(r.binner.getTileCountsBuffer().read(np.uint32), g["counts"], (name, "frame", order, "counts"), extra=info)
        assert_same(r.binner.getTileOffsetsBuffer().read(np.uint32), g["offsets"], (name, "frame", order, "offsets"), extra=info)
        assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, g["indices"].shape[0]), g["indices"], (name, "frame", order, "lists"),
                    offsets=g["offsets"], keys=keys, extra=info)
        r.destroy()
    props.destroy()
    nbuf.destroy()


def test_fresh_renderers_first_frames(device):
    """The FIRST frame of a fresh Renderer (every buffer, sorter and workspace reserved inside that frame), in both orders of
    work, twenty times over: the fixture's lists every time.  (What the reference-fixture test above does once per run — where,
    once in three rounds, a workspace fill on the null stream landed after the sort that used the workspace.)"""
    import os
    here = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(here, "ref_binsorted_ragged1000.npz"))
    f = np.load(os.path.join(here, "ragged1000.npz"))
    w, h, tile = (int(x) for x in g["dims"])
    n = g["projected"].shape[0]
    props, nbuf = device.createBufferFrom(f["props"]), device.createBufferFrom(f["normals"])
    for k in range(20):
        for order in ("tileFirst", "sortFirst"):
            r = sr.Renderer(device, None, "rgba8unorm", n, frameOrder=order)
            r.render(f["uniforms"], props, nbuf, None, w, h)
            assert r.finish() == g["indices"].shape[0]
            assert_same(r.binner.getTileCountsBuffer().read(np.uint32), g["counts"], ("fresh renderer", k, order, "counts"))
            assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, g["indices"].shape[0]), g["indices"], ("fresh renderer", k, order, "lists"),
                        offsets=g["offsets"])
            r.destroy()
    for o in (props, nbuf):
        o.destroy()


def test_scan_equals_the_reference_own_code(device):
    """ref_scan.npz: the loop of PrefixSumScanner.scanCPU (src/PrefixSumScanner.ts:150-155) run under Node."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_scan.npz"))
    sc = sr.PrefixSumScanner(device)
    k = 0
    while f"in{k}" in g.files:
        a = g[f"in{k}"]
        ib, ob = device.createBufferFrom(a), device.createBuffer(max(a.nbytes, 16))
        sc.scan(None, ib, ob, a.shape[0])
        assert_same(ob.read(np.uint32, a.shape[0]), g[f"out{k}"], ("L547", k))
        ib.destroy()
        ob.destroy()
        k += 1
    assert k >= 5


@pytest.mark.parametrize("tile", [8, 10, 24, 32])
def test_bin_other_tile_sizes(device, tile):
    """Binning is tile-size generic (GPUTileBinner's ctor takes any tileSize); power-of-two sizes
    take the exact f32 path, the others the f64 path — both must equal binSorted."""
    n, w, h = 20000, 333, 211
    props, normals, u = make_case(n, w, h, 17, 1.5)
    ref = oracle_pipeline(props, normals, u, w, h, tile=tile)
    g = run_gpu_pipeline(device, props, normals, u, n, w, h, tile=tile)
    b = g["binner"]
    assert_same(b.getTileCountsBuffer().read(np.uint32), ref["counts"], "L563")
    assert_same(b.getTileOffsetsBuffer().read(np.uint32), ref["offsets"], "L564")
    assert_same(b.getTileIndicesBuffer().read(np.uint32, ref["indices"].shape[0]), ref["indices"], "L565")
    destroy_all(g)


def test_full_size_C2_properties(device):
    """BASELINE's headline size (5M @1080p): too big for the oracle's composite in a unit test, so
    the size-independent properties: the sort is a permutation in non-decreasing key order with
    index-ascending ties; per-tile lists are depth-ordered sub-sequences of it; counts sum to P;
    offsets are their exclusive scan; sampled pairs overlap their tile; the image is opaque."""
    n, w, h = sr.scene.CONFIGS["C2"]
    props, normals, u = make_case(n, w, h)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    r = sr.Renderer(device, None, "rgba8unorm", n, frameOrder="sortFirst", records="projected")
    r.render(u, pbuf, nbuf, None, w, h)
    order = r.sorter.getSortedIndicesBuffer().read(np.uint32, n)
    keys = r.sorter.getSortedKeysBuffer().read(np.uint32, n)
    assert_same(np.sort(order), np.arange(n, dtype=np.uint32), "L581")
    dk = np.diff(keys.astype(np.int64))
    assert (dk >= 0).all()
    ties = dk == 0
    assert (np.diff(order.astype(np.int64))[ties] > 0).all()  # stable: equal keys keep index order
    proj = r.projector.getProjectedBuffer().read(np.float32).reshape(n, 8)
    assert np.array_equal(keys, (proj[order, 4].view(np.uint32) ^ np.uint32(0x80000000)))  # all depths positive
    counts = r.binner.getTileCountsBuffer().read(np.uint32)
    offsets = r.binner.getTileOffsetsBuffer().read(np.uint32)
    total = r.binner.getTotalIndices()
    assert int(counts.sum(dtype=np.uint64)) == total
    assert_same(offsets, np.concatenate([[0], np.cumsum(counts, dtype=np.uint64)[:-1]]).astype(np.uint32), "L592")
    idx = r.binner.getTileIndicesBuffer().read(np.uint32, total)
    rank = np.empty(n, np.uint32)
    rank[order] = np.arange(n, dtype=np.uint32)
    ranks = rank[idx].astype(np.int64)
    d = np.diff(ranks)
    starts = offsets[counts > 0][1:]  # list boundaries: the only places a decrease is allowed
    bad = np.nonzero(d <= 0)[0] + 1
    assert np.isin(bad, starts).all()
    # every pair really overlaps its tile (spot check 200k pairs)
    rng = np.random.default_rng(0)
    pick = rng.integers(0, total, 200000)
    tile_of = np.searchsorted(offsets, pick, side="right") - 1
    # skip empty tiles that share an offset with the next one
    while True:
        emp = counts[tile_of] == 0
        if not emp.any():
            break
        tile_of[emp] += 1
    tx, ty = tile_of % 120, tile_of // 120
    b = proj[idx[pick]]
    assert (np.maximum(b[:, 0], 0) < (tx + 1) * 16).all() and (np.minimum(b[:, 2], w) >= tx * 16).all()
    assert (np.maximum(b[:, 1], 0) < (ty + 1) * 16).all() and (np.minimum(b[:, 3], h) >= ty * 16).all()
    img = r.readPixels()
    assert (img[..., 3] == 255).all()
    assert img[..., :3].std() > 10  # not a constant image
    r.destroy()
    pbuf.destroy()
    nbuf.destroy()


def test_sync_free_frames_repeat_bit_exactly(device):
    """After the first frame the binner stops waiting for its pair total (device-side count, async
    readback).  Frames 2..5 of a static scene must reproduce frame 1 bit for bit."""
    n, w, h = 40000, 480, 270
    props, normals, u = make_case(n, w, h, 51, 1.5)
    ref = oracle_pipeline(props, normals, u, w, h)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    r = sr.Renderer(device, None, "rgba8unorm", n)
    r.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
    first = r.readPixelsFloat().copy()
    for _ in range(4):
        r.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
    assert not r.previousFrameOverflowed
    assert_same(r.readPixelsFloat().view(np.uint32), first.view(np.uint32), "L636")
    assert r.binner.getTotalIndices() == ref["indices"].shape[0]
    assert_same(r.binner.getTileCountsBuffer().read(np.uint32), ref["counts"], "L638")
    assert_same(r.binner.getTileOffsetsBuffer().read(np.uint32), ref["offsets"], "L639")
    assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, ref["indices"].shape[0]), ref["indices"], "L640")
    r.destroy()
    pbuf.destroy()
    nbuf.destroy()


def test_sync_free_overflow_is_detected_and_recovered(device):
    """A frame whose pairs exceed 1.5x the previous frame's cannot fit its sync-free limit: the
    library reports it at the next call and the host facade renders it again; the final lists and
    image are the oracle's."""
    import os
    if os.environ.get("SPLAT_BIN_SYNC") == "1":
        pytest.skip("SPLAT_BIN_SYNC=1: every frame reads its pair total back before it sizes anything: no frame can overflow")
    n, w, h = 20000, 320, 200
    small, normals, u = make_case(n, w, h, 61, 0.5)
    big = small.copy()
    big[:, 3] *= 6.0  # ~20x the pairs
    ref = oracle_pipeline(big, normals, u, w, h)
    assert ref["indices"].shape[0] > 3 * oracle_pipeline(small, normals, u, w, h)["indices"].shape[0]
    want, _, _ = O.composite(O.MODE_FRONT_TO_BACK, True, big[:, 4:], normals, ref["proj"], ref["indices"], ref["counts"],
                             ref["offsets"], w, h)
    sbuf, bbuf, nbuf = device.createBufferFrom(small), device.createBufferFrom(big), device.createBufferFrom(normals)
    r = sr.Renderer(device, None, "rgba8unorm", n)
    r.render(u, sbuf, nbuf, None, w, h, wantFloat=True)   # sync frame: learns P_small
    r.render(u, sbuf, nbuf, None, w, h, wantFloat=True)   # sync-free
    r.render(u, bbuf, nbuf, None, w, h, wantFloat=True)   # sync-free, overflows its limit
    got = r.readPixelsFloat()                              # finish(): detects, re-renders
    assert r.previousFrameOverflowed
    assert r.binner.getTotalIndices() == ref["indices"].shape[0]
    assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, ref["indices"].shape[0]), ref["indices"], "L666")
    err = np.abs(got - want)
    assert err.max() <= TOL_EARLY_OUT_BOUND and (err.max(axis=2) > TOL_NO_EARLY_OUT).mean() <= FRAC_ABOVE_TIGHT
    # and the low-level contract: the raw ABI call after an overflowed frame returns CAPACITY once
    r2 = sr.Renderer(device, None, "rgba8unorm", n)
    r2.render(u, sbuf, nbuf, None, w, h)
    r2.render(u, sbuf, nbuf, None, w, h)
    r2.render(u, bbuf, nbuf, None, w, h)
    with pytest.raises(sr.SplatError) as ei:
        r2.binner.getTotalIndices()
    assert ei.value.code == -4 and "render that frame again" in str(ei.value)
    r2.render(u, bbuf, nbuf, None, w, h)
    assert r2.finish() == ref["indices"].shape[0]
    for o in (r, r2, sbuf, bbuf, nbuf):
        o.destroy()


def test_per_tile_sorter_validates_order(device):
    """PerTileSorter keeps its call site; as a validator it finds 0 out-of-order neighbours in the
    binner's lists and > 0 once two entries of a list are swapped."""
    n, w, h = 20000, 320, 200
    props, normals, u = make_case(n, w, h, 71, 1.5)
    g = run_gpu_pipeline(device, props, normals, u, n, w, h)
    b = g["binner"]
    total = b.getTotalIndices()
    pts = sr.PerTileSorter(device, validate=True)
    assert pts.sort(None, g["proj"].getProjectedBuffer(), b.getTileCountsBuffer(), b.getTileOffsetsBuffer(),
                    b.getTileIndicesBuffer(), 20 * 13, 4096, total) == 0
    counts = b.getTileCountsBuffer().read(np.uint32)
    offsets = b.getTileOffsetsBuffer().read(np.uint32)
    t = int(np.argmax(counts))
    idx = b.getTileIndicesBuffer().read(np.uint32, total)
    o = int(offsets[t])
    idx[o], idx[o + 5] = idx[o + 5], idx[o]
    b.getTileIndicesBuffer().write(idx)
    assert pts.sort(None, g["proj"].getProjectedBuffer(), b.getTileCountsBuffer(), b.getTileOffsetsBuffer(),
                    b.getTileIndicesBuffer(), 20 * 13, 4096, total) > 0
    assert sr.PerTileSorter(device).sort(None, None, None, None, None, 0, 0) is None  # reference call shape, no-op
    destroy_all(g)


def test_sequential_renderer_honours_given_order(device):
    """SequentialRenderer.render(uniforms, props, sortedIdx, curvature, W, H): the image is the
    composite of exactly the given order (here: a deliberately NON-depth order, index order); with
    ComputeShaderRenderer's footprint here — the class's own footprint is covered in test_gpu_disc.py."""
    n, w, h = 3000, 160, 96
    props, normals, u = make_case(n, w, h, 81, 2.0)
    order = np.arange(n, dtype=np.uint32)
    proj = O.project(u, props)
    counts, offsets, idx = O.bin_sorted(proj, order, w, h)
    want, _, _ = O.composite(O.MODE_FRONT_TO_BACK, True, props[:, 4:], normals, proj, idx, counts, offsets, w, h)
    pm = sr.SplatPropertyManager(device, n)
    pm.setFromArrays(props)
    nbuf, obuf = device.createBufferFrom(normals), device.createBufferFrom(order)
    r = sr.SequentialRenderer(device, None, "rgba8unorm", n, footprint="isotropic")
    r.render(u[:20], pm.getPropertyBuffer(), obuf, nbuf, w, h, wantFloat=True)  # 20-float block: W,H appended
    err = np.abs(r.readPixelsFloat() - want)
    assert err.max() <= TOL_EARLY_OUT_BOUND and (err.max(axis=2) > TOL_NO_EARLY_OUT).mean() <= FRAC_ABOVE_TIGHT
    for o in (r, pm, nbuf, obuf):
        o.destroy()


# ---- tile-first frame order (bin in index order, PerTileSorter-style depth sort per tile) -----------
TF_CASES = CASES + [
    (20000, 64, 64, 21, 6.0),     # 16 tiles, thousands of entries each: lists beyond the in-LDS capacity
    (30000, 48, 32, 22, 8.0),     # 6 tiles x ~30000 entries: the long-list (global memory) passes
    (5000, 1920, 1080, 23, 1.0),  # 8160 tiles, most of them empty or single-entry
]


def _tile_first_frame(device, props, normals, u, n, w, h, want_float=True):
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    r = sr.Renderer(device, None, "rgba8unorm", n, frameOrder="tileFirst")
    r.render(u, pbuf, nbuf, None, w, h, wantFloat=want_float)
    return r, pbuf, nbuf


@pytest.mark.parametrize("n,w,h,seed,rs", TF_CASES)
def test_tile_first_lists_and_image_match_sort_first(device, n, w, h, seed, rs):
    props, normals, u = make_case(n, w, h, seed, rs)
    ref = oracle_pipeline(props, normals, u, w, h)
    r, pbuf, nbuf = _tile_first_frame(device, props, normals, u, n, w, h)
    total = ref["indices"].shape[0]
    assert r.binner.getTotalIndices() == total
    assert_same(r.binner.getTileCountsBuffer().read(np.uint32), ref["counts"], "L750")
    assert_same(r.binner.getTileOffsetsBuffer().read(np.uint32), ref["offsets"], "L751")
    assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], "L752")
    got = r.readPixelsFloat()
    r2 = sr.Renderer(device, None, "rgba8unorm", n, frameOrder="sortFirst")
    r2.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
    assert np.array_equal(got.view(np.uint32), r2.readPixelsFloat().view(np.uint32))  # same lists -> same bits
    for o in (r, r2, pbuf, nbuf):
        o.destroy()


@pytest.mark.parametrize("kind", ["all_equal", "seven_points", "quadruplets"])
def test_tile_first_depth_ties_resolve_by_index(device, kind):
    """Depth is the distance to the camera, so splats at the same position share a key exactly: equal
    keys must come out in ascending index (the stable order); a tile whose keys are all equal takes no
    radix pass at all.  Radii still differ, so the duplicates cover different tile rectangles."""
    n, w, h = 6000, 128, 96
    props, normals, u = make_case(n, w, h, 31, 2.0)
    distinct = {"all_equal": 1, "seven_points": 7, "quadruplets": n // 4}[kind]
    props[:, :3] = props[np.arange(n) % distinct, :3] * np.float32(0.3 if distinct < 10 else 1.0)
    ref = oracle_pipeline(props, normals, u, w, h)
    assert np.unique(ref["keys"][:n]).size <= distinct  # the construction really produces ties
    r, pbuf, nbuf = _tile_first_frame(device, props, normals, u, n, w, h, want_float=False)
    total = ref["indices"].shape[0]
    assert r.binner.getTotalIndices() == total
    assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], "L775")
    for o in (r, pbuf, nbuf):
        o.destroy()


@pytest.mark.parametrize("n,rs,cls", [(3000, 1.0, "short"), (8000, 1.0, "long")])
def test_tile_first_wide_depth_range_inside_a_tile(device, n, rs, cls):
    """The per-tile sort runs one 8-bit pass per byte of (largest - smallest depth key) of the tile: three at the bench
    sizes.  A tile with splats both a hair's breadth from the eye and far away (depths 10^5 apart: keys 2^27 and more
    apart) takes four — in either size class, with the same lists.  (A variant of the kernel that packed the
    remaining key bits and a 13-bit list position into one LDS word had to send exactly these tiles down another
    path; it measured no faster and was dropped, the case stays.)"""
    w, h = 64, 64
    props, normals, u = make_case(n, w, h, 71, rs)
    eye = np.asarray(u[16:19], np.float32)
    toward = -eye / np.linalg.norm(eye)
    rng = np.random.default_rng(5)
    near = rng.choice(n, 40, replace=False)
    props[near, :3] = eye + toward * rng.uniform(2e-5, 6e-5, size=(40, 1)).astype(np.float32)  # the scene sits ~3 away
    props[near, 3] = np.float32(2e-7)
    ref = oracle_pipeline(props, normals, u, w, h)
    counts, offsets, idx, keys = ref["counts"], ref["offsets"], ref["indices"], ref["keys"]
    wide = 0
    for t in range(counts.size):
        c = int(counts[t])
        if c and ((c <= 2048) if cls == "short" else (2048 < c <= 6144)):
            k = keys[idx[offsets[t]:offsets[t] + c]].astype(np.int64)
            wide += int(k.max() - k.min() >= (1 << 27))
    assert wide > 0, "the construction must put a wide depth range into a tile of this size class"
    r, pbuf, nbuf = _tile_first_frame(device, props, normals, u, n, w, h, want_float=False)
    total = idx.shape[0]
    assert r.binner.getTotalIndices() == total
    assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, total), idx, "L807")
    for o in (r, pbuf, nbuf):
        o.destroy()


def test_tile_first_sixteen_bit_tile_ids(device):
    """The largest screen the fast path takes: 256 x 256 tiles, tile ids of 16 bits sorted as 8 + 8.  The bottom tile
    row is high digit 255 — the digit the second pass's padding slots borrow (they rank behind every real pair of a
    partition) — and with 65536 tiles for 390 000 pairs almost every partition of the second pass is a partial one."""
    n, w, h = 6000, 4096, 4096
    props, normals, u = make_case(n, w, h, 91, 0.1, camera=dict(distance=1.6))
    ref = oracle_pipeline(props, normals, u, w, h)
    counts = ref["counts"].reshape(256, 256)
    assert counts[255].sum() > 500 and counts[0].sum() > 500 and counts[:, 255].sum() > 500
    r, pbuf, nbuf = _tile_first_frame(device, props, normals, u, n, w, h, want_float=False)
    total = ref["indices"].shape[0]
    assert r.binner.getTotalIndices() == total
    assert_same(r.binner.getTileCountsBuffer().read(np.uint32), ref["counts"], "L824")
    assert_same(r.binner.getTileOffsetsBuffer().read(np.uint32), ref["offsets"], "L825")
    assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], "L826")
    r.render(u, pbuf, nbuf, None, w, h)  # and again, sync-free
    r.finish()
    assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], "L829")
    for o in (r, pbuf, nbuf):
        o.destroy()


def test_tile_first_with_ballot_ranking(monkeypatch):
    """SPLAT_RANK=ballot: the escape from the measured lane order of returning LDS atomics (splat.h).  A context created
    under it never asks the hardware and ranks with ballots in every kernel of the frame's binner — first-pass scatter,
    second-pass downsweep, per-tile sort, both size classes and the global-memory passes: same lists, same image."""
    monkeypatch.setenv("SPLAT_RANK", "ballot")
    dev = sr.Device(0)  # (the ranking is resolved once per context, at its first sort)
    try:
        for n, w, h, seed, rs in [(3000, 128, 96, 41, 1.0), (20000, 640, 360, 42, 1.0), (20000, 64, 64, 43, 6.0), (30000, 48, 32, 44, 8.0)]:
            props, normals, u = make_case(n, w, h, seed, rs)
            ref = oracle_pipeline(props, normals, u, w, h)
            pbuf, nbuf = dev.createBufferFrom(props), dev.createBufferFrom(normals)
            r = sr.Renderer(dev, None, "rgba8unorm", n, frameOrder="tileFirst")
            for rep in range(2):  # first and sync-free
                r.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
                total = r.finish()
                assert total == ref["indices"].shape[0], (n, w, h)
                assert_same(r.binner.getTileCountsBuffer().read(np.uint32), ref["counts"], "L850")
                assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], ("L851", n, w, h, rep), offsets=ref["offsets"])
            got = r.readPixelsFloat()
            r2 = sr.Renderer(dev, None, "rgba8unorm", n, frameOrder="sortFirst")
            r2.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
            assert_same(got.view(np.uint32), r2.readPixelsFloat().view(np.uint32), "L855")
            for o in (r, r2, pbuf, nbuf):
                o.destroy()
    finally:
        dev.destroy()


def test_timed_frames_count_entries_only_when_asked(device):
    """splat_set_timing_stages' bit 31 (SPLAT_TIMING_COUNT_ENTRIES, on by default): a timed whole-frame call counts the list
    entries its composite staged and consumed.  bench.py clears the bit for its timed region — the frames whose rate it
    reports run the kernel instantiation every production frame runs — and takes the counts from later frames: both
    settings must give the same image, and the counts must be the oracle's consumed entries."""
    import ctypes as C
    from splat_renderer_amd import _lib
    n, w, h = 20000, 640, 360
    props, normals, u = make_case(n, w, h, 58, 1.0)
    ref = oracle_pipeline(props, normals, u, w, h)
    _, _, consumed_ref = O.composite(0, True, props[:, 4:], normals, ref["proj"], ref["indices"], ref["counts"], ref["offsets"], w, h)[:3]
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    r = sr.Renderer(device, None, "rgba8unorm", n)
    lib, ctx = device.lib, device.ctx

    def counts_after(mask, frames=3):
        _lib.check(lib.splat_set_timing_stages(ctx, mask), ctx)
        device.setTiming(True)
        for _ in range(frames):
            r.render(u, pbuf, nbuf, None, w, h)
        r.finish()
        staged, consumed = C.c_uint64(), C.c_uint64()
        _lib.check(lib.splat_timing_consumed(ctx, C.byref(staged), C.byref(consumed)), ctx)
        img = r.readPixels().copy()
        ms = device.stageTimeMs(_lib.STAGE_COMPOSITE)
        device.setTiming(False)
        return staged.value, consumed.value, img, ms

    try:
        s0, c0, img0, ms0 = counts_after(1 << _lib.STAGE_COMPOSITE)  # events on the composite, no counting
        assert (s0, c0) == (0, 0) and ms0 > 0
        s1, c1, img1, _ = counts_after(0xFFFFFFFF)
        assert c1 % 3 == 0 and s1 >= c1 > 0
        # (per tile the kernel reports the entries visited before its LAST pixel stopped: the oracle's per-pixel visits, maximised
        # over the tile, is what test_composite_vs_oracle pins; here: the same frame gives the same number every time)
        s2, c2, _, _ = counts_after(0xFFFFFFFF, frames=1)
        assert (s2, c2) == (s1 // 3, c1 // 3)
        assert_same(img0.view(np.uint32), img1.view(np.uint32), "counting and non-counting composite: same image")
        assert consumed_ref > 0
    finally:
        _lib.check(lib.splat_set_timing_stages(ctx, 0xFFFFFFFF), ctx)
        for o in (r, pbuf, nbuf):
            o.destroy()


def test_order_check_catches_a_misranked_list_and_the_frame_is_rendered_again():
    """The default ranking of the tile-first frame rests on nothing unverified (include/splat.h, NOTE on ranking): the
    per-tile sort checks every finished list for strictly increasing (depth key, splat index) order.  tests/hooks_child.py
    leaves a list as an out-of-lane-order rank would (two neighbours swapped — a TEST HOOK, compiled only into
    libsplat_hip_hooks.so: the shipped kernels do not carry its parameters) and asserts the whole recovery; it runs as a
    child process on that build, this process keeps the shipped library."""
    import subprocess
    import sys
    assert not device_lib_has_hooks(), "the suite must run on the shipped library (SPLAT_LIB_PATH points at a hooks build?)"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SPLAT_LIB_PATH=_lib.HOOKS_LIB_PATH)
    env.pop("SPLAT_RANK", None)  # (the DEFAULT policy is what this is about, whatever the suite runs under)
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "hooks_child.py"), "order_check"], cwd=root, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0 and "order_check ok: 5 cases" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_long_class_launch_is_skipped_only_while_no_tile_needs_it():
    """tile_sort_launch launches the per-tile sort's short class alone in a sync-free frame whose predecessor had no tile
    beyond it; tests/hooks_child.py counts the launches (a TEST HOOK) through a sequence in which a tile outgrows the class
    between two frames, and holds every frame's lists against the oracle's."""
    import subprocess
    import sys
    if os.environ.get("SPLAT_BIN_SYNC") == "1":
        pytest.skip("SPLAT_BIN_SYNC=1: no frame is sync-free")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SPLAT_LIB_PATH=_lib.HOOKS_LIB_PATH)
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "hooks_child.py"), "long_class_skip"], cwd=root, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0 and "long_class_skip ok: 6 frames" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def device_lib_has_hooks():
    return bool(getattr(_lib.load(), "has_hooks", False))


def test_ranking_policies(monkeypatch):
    """SPLAT_RANK unset: 'checked' (atomics only on the checked frame path); =atomic: atomics wherever the probe passes —
    the staged RadixSorter / GPUTileBinner too; =ballot: nowhere.  Every policy gives the oracle's order and lists."""
    n, w, h = 20000, 320, 200
    props, normals, u = make_case(n, w, h, 91, 1.0)
    ref = oracle_pipeline(props, normals, u, w, h)
    for env, policy in ((None, "checked"), ("atomic", "atomic"), ("ballot", "ballot")):
        if env is None:
            monkeypatch.delenv("SPLAT_RANK", raising=False)
        else:
            monkeypatch.setenv("SPLAT_RANK", env)
        dev = sr.Device(0)
        try:
            st = dev.rankStatus()
            assert st["policy"] == policy and st["atomicsOrdered"] == (policy != "ballot"), st
            gpu = run_gpu_pipeline(dev, props, normals, u, n, w, h)  # staged API: projector, sorter, binner
            assert_same(gpu["sorter"].getSortedIndicesBuffer().read(np.uint32, n), ref["order"][:n], ("policy", policy, "order"))
            b = gpu["binner"]
            assert_same(b.getTileIndicesBuffer().read(np.uint32, ref["indices"].shape[0]), ref["indices"], ("policy", policy, "staged lists"),
                        offsets=ref["offsets"], keys=ref["keys"])
            destroy_all(gpu)
            pbuf, nbuf = dev.createBufferFrom(props), dev.createBufferFrom(normals)
            for order in ("tileFirst", "sortFirst"):
                r = sr.Renderer(dev, None, "rgba8unorm", n, frameOrder=order)
                for rep in range(2):
                    r.render(u, pbuf, nbuf, None, w, h)
                    total = r.finish()
                    assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], ("policy", policy, order, rep),
                                offsets=ref["offsets"], keys=ref["keys"])
                r.destroy()
            assert dev.rankStatus()["orderFaults"] == 0
            pbuf.destroy()
            nbuf.destroy()
        finally:
            dev.destroy()


def test_tile_first_sync_free_repeat_and_overflow(device):
    import os
    if os.environ.get("SPLAT_BIN_SYNC") == "1":
        pytest.skip("SPLAT_BIN_SYNC=1: every frame reads its pair total back before it sizes anything: no frame can overflow")
    n, w, h = 20000, 320, 200
    small, normals, u = make_case(n, w, h, 61, 0.5)
    big = small.copy()
    big[:, 3] *= 6.0
    ref_s, ref_b = oracle_pipeline(small, normals, u, w, h), oracle_pipeline(big, normals, u, w, h)
    sbuf, bbuf, nbuf = device.createBufferFrom(small), device.createBufferFrom(big), device.createBufferFrom(normals)
    r = sr.Renderer(device, None, "rgba8unorm", n, frameOrder="tileFirst")
    r.render(u, sbuf, nbuf, None, w, h, wantFloat=True)
    first = r.readPixelsFloat().copy()
    for _ in range(3):
        r.render(u, sbuf, nbuf, None, w, h, wantFloat=True)  # sync-free
    assert not r.previousFrameOverflowed
    assert_same(r.readPixelsFloat().view(np.uint32), first.view(np.uint32), "L875")
    assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, ref_s["indices"].shape[0]), ref_s["indices"], "L876")
    r.render(u, bbuf, nbuf, None, w, h, wantFloat=True)      # outgrows the sync-free limit
    r.readPixelsFloat()                                       # finish(): detects, renders again
    assert r.previousFrameOverflowed
    assert r.binner.getTotalIndices() == ref_b["indices"].shape[0]
    assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, ref_b["indices"].shape[0]), ref_b["indices"], "L881")
    for o in (r, sbuf, bbuf, nbuf):
        o.destroy()


# SURVEY §8 dry-run statistics of the bench scenes (seed 1234): tile-splat pairs P
FULL_SIZE_PAIRS = {"C1": 4632329, "C2": 11280103, "C3": 29483686}


@pytest.mark.parametrize("name", ["C1", "C2", "C3"])
def test_full_size_frame_lists_and_pixels(device, name):
    """BASELINE.json's GPU configurations at FULL size (C1 1M @1080p, C2 5M @1080p, C3 10M @4K) through the frame the
    bench times (tile-first order, lit composite records, sync-free after the first frame).  The oracle's composite
    of a whole frame takes too long for a unit test, so:
      * lists: counts sum to P, offsets are their exclusive scan, and on sampled tiles the list is EXACTLY
        TileBinner.binSorted's — the splats whose clamped tile range covers the tile (ranges recomputed here from the
        oracle's projector), in (depth key, index) order;
      * pixels: two 32-row bands (screen centre, top edge) against the oracle's composite of those rows run on the
        oracle's own records and colours with the GPU's lists (just shown equal to binSorted on the sampled tiles, and
        equal to the sort-first order's lists over the whole frame), at the composite's stated tolerance."""
    n, w, h = sr.scene.CONFIGS[name]
    tile = 16
    ntx, nty = -(-w // tile), -(-h // tile)
    props, normals, u = make_case(n, w, h)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    r = sr.Renderer(device, None, "rgba8unorm", n)
    for _ in range(3):  # the third frame is sync-free
        r.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
    total = r.finish()
    assert total == FULL_SIZE_PAIRS[name]
    assert r.recordFormat == _lib.RECORDS_LIT32
    counts = r.binner.getTileCountsBuffer().read(np.uint32)
    offsets = r.binner.getTileOffsetsBuffer().read(np.uint32)
    idx = r.binner.getTileIndicesBuffer().read(np.uint32, total)
    assert int(counts.sum(dtype=np.uint64)) == total
    assert_same(offsets, np.concatenate([[0], np.cumsum(counts, dtype=np.uint64)[:-1]]).astype(np.uint32), "L916")
    # the frame's records against the oracle's projector (every splat)
    proj = O.project(u, props)
    rec = r.projector.getRecordsBuffer().read(np.float32).reshape(n, 8)  # (lit composite records: asserted above)
    assert np.array_equal(bits(rec[:, :3]), bits(O.project_compact(u, props)[:, :3])) and np.array_equal(bits(rec[:, 3]), bits(proj[:, 4]))
    # clamped tile ranges as TileBinner.binSorted forms them (src/TileBinner.ts:432-442), vectorised
    mnx, mny = np.maximum(proj[:, 0], 0), np.maximum(proj[:, 1], 0)
    mxx, mxy = np.minimum(proj[:, 2], np.float32(w)), np.minimum(proj[:, 3], np.float32(h))
    on = (mnx < mxx) & (mny < mxy)
    tx0, ty0 = np.floor(mnx / tile).astype(np.int64), np.floor(mny / tile).astype(np.int64)
    tx1 = np.minimum(np.floor(mxx / tile), ntx - 1).astype(np.int64)
    ty1 = np.minimum(np.floor(mxy / tile), nty - 1).astype(np.int64)
    keys = proj[:, 4].view(np.uint32) ^ np.uint32(0x80000000)  # all depths are positive in this scene
    rng = np.random.default_rng(7)
    busiest = int(np.argmax(counts))
    sample = set(int(t) for t in rng.integers(0, ntx * nty, 24)) | {busiest, 0, ntx * nty - 1, (nty // 2) * ntx + ntx // 2}
    for t in sorted(sample):
        tx, ty = t % ntx, t // ntx
        members = np.nonzero(on & (tx0 <= tx) & (tx <= tx1) & (ty0 <= ty) & (ty <= ty1))[0]
        members = members[np.lexsort((members, keys[members]))]
        got = idx[offsets[t]:offsets[t] + counts[t]]
        assert_same(got, members.astype(np.uint32), ("L937", f"tile {t} ({tx},{ty}): list differs from binSorted"))
    # the sort-first order (global depth sort, then bin) gives the same lists over the whole frame
    a = sr.Renderer(device, None, "rgba8unorm", n, frameOrder="sortFirst")
    a.render(u, pbuf, nbuf, None, w, h)
    assert a.finish() == total
    assert_same(a.binner.getTileIndicesBuffer().read(np.uint32, total), idx, "L942")
    assert_same(a.readPixels(), r.readPixels(), "L943")
    a.destroy()
    # pixel parity on two bands of 32 rows
    img, img8 = r.readPixelsFloat(), r.readPixels()
    assert (img8[..., 3] == 255).all()
    mid = (nty // 2) * tile
    for r0 in (mid - 16, 0):
        want, want8, _, _, near = O.composite(O.MODE_FRONT_TO_BACK, True, props[:, 4:], normals, proj, idx, counts, offsets, w, h,
                                              rows=(r0, r0 + 32), want_stops=True)
        check_image_against_oracle(img[r0:r0 + 32], img8[r0:r0 + 32], want[r0:r0 + 32], want8[r0:r0 + 32], near[r0:r0 + 32])
    r.destroy()
    pbuf.destroy()
    nbuf.destroy()


def test_tile_first_full_size_C2(device):
    """5M @1080p: both frame orders must give identical tile lists (11.28M entries) and images."""
    n, w, h = sr.scene.CONFIGS["C2"]
    props, normals, u = make_case(n, w, h)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    a = sr.Renderer(device, None, "rgba8unorm", n, frameOrder="sortFirst")
    b = sr.Renderer(device, None, "rgba8unorm", n, frameOrder="tileFirst")
    a.render(u, pbuf, nbuf, None, w, h)
    b.render(u, pbuf, nbuf, None, w, h)
    total = a.binner.getTotalIndices()
    assert total == b.binner.getTotalIndices() == 11280103
    assert_same(a.binner.getTileOffsetsBuffer().read(np.uint32), b.binner.getTileOffsetsBuffer().read(np.uint32), "L969")
    assert_same(a.binner.getTileIndicesBuffer().read(np.uint32, total), b.binner.getTileIndicesBuffer().read(np.uint32, total), "L970")
    assert_same(a.readPixels(), b.readPixels(), "L971")
    for _ in range(3):  # sync-free frames
        b.render(u, pbuf, nbuf, None, w, h)
    assert_same(a.readPixels(), b.readPixels(), "L974")
    for o in (a, b, pbuf, nbuf):
        o.destroy()


@pytest.mark.parametrize("order", ["tileFirst", "sortFirst"])
def test_frame_with_nothing_on_screen_then_something(device, order):
    """Every splat behind/outside the view: no pairs at all (empty lists, background image), on a first
    frame and on a sync-free frame after a populated one; then the populated scene again."""
    n, w, h = 3000, 160, 96
    props, normals, u = make_case(n, w, h, 71, 1.0)
    away = props.copy()
    away[:, 0] += np.float32(500.0)  # far off to the side: every clamped box is empty
    ref = oracle_pipeline(props, normals, u, w, h)
    ref_away = oracle_pipeline(away, normals, u, w, h)
    assert ref_away["indices"].shape[0] == 0 and ref["indices"].shape[0] > 0
    want_away, _, _ = O.composite(O.MODE_FRONT_TO_BACK, True, away[:, 4:], normals, ref_away["proj"], ref_away["indices"],
                                  ref_away["counts"], ref_away["offsets"], w, h)
    pbuf, abuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(away), device.createBufferFrom(normals)
    r = sr.Renderer(device, None, "rgba8unorm", n, frameOrder=order)
    r.render(u, abuf, nbuf, None, w, h, wantFloat=True)            # first frame: empty
    assert r.binner.getTotalIndices() == 0
    assert not r.binner.getTileCountsBuffer().read(np.uint32).any()
    assert_same(r.readPixelsFloat(), want_away, "L997")
    for buf, rr in ((pbuf, ref), (pbuf, ref), (abuf, ref_away), (pbuf, ref)):
        r.render(u, buf, nbuf, None, w, h, wantFloat=True)
        total = rr["indices"].shape[0]
        assert r.finish() == total
        assert_same(r.binner.getTileCountsBuffer().read(np.uint32), rr["counts"], "L1002")
        if total:
            assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, total), rr["indices"], "L1004")
    for o in (r, pbuf, abuf, nbuf):
        o.destroy()


def test_compact_exchange_records_rebuild_projected_records_bit_exactly(device):
    """The multi-GPU exchange format (16 B: centre, radius, depth) against the oracle, and its expansion
    against the projector's own 32-byte records — bit for bit, including originalIndex."""
    n, w, h = 20000, 250, 130
    props, normals, u = make_case(n, w, h, 9, 1.0)
    want16 = O.project_compact(u, props)
    want32 = O.project(u, props)
    assert_same(O.expand_compact(want16).view(np.uint32), want32.view(np.uint32), "L1016")
    lib, ctx = device.lib, device.ctx
    pbuf = device.createBufferFrom(props)
    first, count = 777, 15000
    rec16 = device.createBuffer(count * 16)
    rec32 = device.createBuffer(count * 32)
    uf = np.ascontiguousarray(u, np.float32)
    _lib.check(lib.splat_project_slice_compact(ctx, uf.ctypes.data_as(C.POINTER(C.c_float)), pbuf.ptr, 2, first, count, rec16.ptr), ctx)
    got16 = rec16.read(np.float32).reshape(count, 4)
    assert_same(got16.view(np.uint32), want16[first:first + count].view(np.uint32), "L1025")
    _lib.check(lib.splat_expand_compact(ctx, rec16.ptr, count, first, rec32.ptr), ctx)
    got32 = rec32.read(np.float32).reshape(count, 8)
    assert_same(got32.view(np.uint32), want32[first:first + count].view(np.uint32), "L1028")
    for o in (pbuf, rec16, rec32):
        o.destroy()


@pytest.mark.parametrize("order", ["tileFirst", "sortFirst"])
def test_band_frame_from_compact_records_matches_frame_from_projected_records(device, order):
    """splat_band_frame fed the 16-byte records (what the ranks all-gather) and fed the 32-byte records
    must give the same lists and the same float image, in either order of work."""
    n, w, h = 30000, 320, 208
    props, normals, u = make_case(n, w, h, 13, 1.5)
    ref = oracle_pipeline(props, normals, u, w, h)
    lib, ctx = device.lib, device.ctx
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    rec16, rec32 = device.createBuffer(n * 16), device.createBuffer(n * 32)
    uf = np.ascontiguousarray(u, np.float32)
    up = uf.ctypes.data_as(C.POINTER(C.c_float))
    _lib.check(lib.splat_project_slice_compact(ctx, up, pbuf.ptr, 2, 0, n, rec16.ptr), ctx)
    _lib.check(lib.splat_project_slice(ctx, up, pbuf.ptr, 2, 0, n, rec32.ptr), ctx)
    images = []
    for fmt, rec in ((_lib.RECORDS_COMPACT, rec16), (_lib.RECORDS_PROJECTED, rec32)):
        sorter, binner = sr.RadixSorter(device, n), sr.GPUTileBinner(device, 16)
        binner.setFrameOrder(order)
        out = device.createBuffer(w * h * 16)
        cfg = _lib.CompositeCfg(_lib.MODE_FRONT_TO_BACK, 1, 16, 0, 0xFFFFFFFF, fmt)
        _lib.check(lib.splat_band_frame(ctx, sorter._s, binner._b, C.byref(cfg), pbuf.ptr, nbuf.ptr, rec.ptr, n, w, h, None, out.ptr,
                                        None), ctx)
        binner._tiles = -(-w // 16) * -(-h // 16)
        total = ref["indices"].shape[0]
        assert binner.getTotalIndices() == total
        assert_same(binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], "L1058")
        images.append(out.read(np.float32).reshape(h, w, 4).copy())
        for o in (sorter, binner, out):
            o.destroy()
    assert_same(images[0].view(np.uint32), images[1].view(np.uint32), "L1062")
    for o in (pbuf, nbuf, rec16, rec32):
        o.destroy()


def test_switching_frame_order_between_frames_on_one_renderer(device):
    """The two orders of work share the binner's buffers and its sync-free bookkeeping: alternating them
    frame by frame (and changing the resolution in between) must keep giving the oracle's lists."""
    n = 25000
    props, normals, _ = make_case(n, 320, 200, 17, 1.5)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    r = sr.Renderer(device, None, "rgba8unorm", n)
    sequence = [("tileFirst", 320, 200), ("sortFirst", 320, 200), ("tileFirst", 320, 200), ("tileFirst", 480, 272),
                ("sortFirst", 480, 272), ("sortFirst", 160, 96), ("tileFirst", 160, 96), ("tileFirst", 320, 200)]
    refs = {}
    for order, w, h in sequence:
        _, _, u = make_case(n, w, h, 17, 1.5)
        if (w, h) not in refs:
            refs[(w, h)] = oracle_pipeline(props, normals, u, w, h)
        ref = refs[(w, h)]
        r.binner.setFrameOrder(order)
        r.render(u, pbuf, nbuf, None, w, h)
        total = ref["indices"].shape[0]
        assert r.finish() == total, (order, w, h)
        assert_same(r.binner.getTileCountsBuffer().read(np.uint32), ref["counts"], ("L1086", order, w, h))
        assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], ("L1087", order, w, h), offsets=ref["offsets"])
    for o in (r, pbuf, nbuf):
        o.destroy()


def test_tile_first_random_scenes(device):
    """Seeded sweep over sizes, aspect ratios, splat scales and camera distances (screens down to a
    single tile, splats from sub-pixel to screen-filling): tile lists against the oracle, and the image
    of every third case against the sort-first order bit for bit."""
    rng = np.random.default_rng(20260301)
    for case in range(24):
        n = int(rng.integers(1, 6000))
        w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
        rs = float(rng.choice([0.05, 0.3, 1.0, 2.5, 8.0]))
        cam = dict(distance=float(rng.uniform(1.2, 6.0)), azimuth=float(rng.uniform(0, 6.28)), elevation=float(rng.uniform(-1.2, 1.2)))
        props, normals, u = make_case(n, w, h, 1000 + case, rs, camera=cam)
        ref = oracle_pipeline(props, normals, u, w, h)
        r, pbuf, nbuf = _tile_first_frame(device, props, normals, u, n, w, h)
        total = ref["indices"].shape[0]
        tag = (case, n, w, h, rs)
        assert r.binner.getTotalIndices() == total, tag
        assert_same(r.binner.getTileCountsBuffer().read(np.uint32), ref["counts"], ("L1108", tag))
        if total:
            assert_same(r.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], ("L1110", tag))
        if case % 3 == 0:
            got = r.readPixelsFloat()
            r2 = sr.Renderer(device, None, "rgba8unorm", n, frameOrder="sortFirst")
            r2.render(u, pbuf, nbuf, None, w, h, wantFloat=True)
            assert_same(got.view(np.uint32), r2.readPixelsFloat().view(np.uint32), ("L1115", tag))
            r2.destroy()
        for o in (r, pbuf, nbuf):
            o.destroy()


def test_property_planes_give_the_same_frame_as_interleaved_records(device):
    """SplatPropertyManager's native layout (two vec4 planes) through splat_render_frame_planes: same
    records, lists and image, bit for bit, as the reference's interleaved buffer; and
    updatePlanesFromCurvature writes what updateFromCurvature writes."""
    n, w, h = 40000, 400, 240
    props, normals, u = make_case(n, w, h, 23, 1.5)
    pm = sr.SplatPropertyManager(device, n)
    pm.setFromArrays(props)
    nbuf = device.createBufferFrom(normals)
    a = sr.Renderer(device, None, "rgba8unorm", n, records="projected")
    b = sr.Renderer(device, None, "rgba8unorm", n, records="projected")
    a.render(u, pm.getPropertyBuffer(), nbuf, None, w, h, wantFloat=True)
    planes = pm.getPropertyPlanes()
    assert_same(planes.posRadius.read(np.float32).reshape(n, 4), props[:, :4], "L1134")
    assert_same(planes.colorOpacity.read(np.float32).reshape(n, 4), props[:, 4:], "L1135")
    for _ in range(3):  # also as sync-free frames
        b.render(u, planes, nbuf, None, w, h, wantFloat=True)
    total = a.finish()
    assert b.finish() == total
    assert_same(a.projector.getRecordsBuffer().read(np.uint32), b.projector.getRecordsBuffer().read(np.uint32), "L1140")
    assert_same(a.binner.getTileIndicesBuffer().read(np.uint32, total), b.binner.getTileIndicesBuffer().read(np.uint32, total), "L1141")
    assert_same(a.readPixelsFloat().view(np.uint32), b.readPixelsFloat().view(np.uint32), "L1142")
    # K12 into planes == K12 into interleaved records
    rng = np.random.default_rng(5)
    pos = rng.standard_normal((n, 4)).astype(np.float32)
    cur = rng.standard_normal((n, 4)).astype(np.float32)
    posb, curb = device.createBufferFrom(pos), device.createBufferFrom(cur)
    pm.updateFromCurvature(None, posb, curb)
    want = pm.getPropertyBuffer().read(np.float32).reshape(n, 8).copy()
    assert_same(want, O.update_props(pos, cur), "L1150")
    pm.setFromArrays(props)  # scramble, then update the planes only
    planes = pm.updatePlanesFromCurvature(None, posb, curb)
    assert_same(planes.posRadius.read(np.float32).reshape(n, 4), want[:, :4], "L1153")
    assert_same(planes.colorOpacity.read(np.float32).reshape(n, 4), want[:, 4:], "L1154")
    for o in (a, b, pm, nbuf, posb, curb):
        o.destroy()


def test_frame_pipeline_keeps_frames_apart(device):
    """dist.FramePipeline (two frames in flight: the next frame's projection + exchange on a second stream
    under this frame's band work) with a camera that moves every frame: every band image must be the
    one the unpipelined BandRenderer gives for that frame's camera."""
    import torch
    from splat_renderer_amd import dist
    n, w, h = 60000, 480, 272
    props, normals, _ = make_case(n, w, h, 29, 1.5)
    cams = [make_case(n, w, h, 29, 1.5, camera=dict(azimuth=0.5 + 0.3 * k, elevation=0.5 - 0.1 * k))[2] for k in range(5)]
    pt, nt = torch.from_numpy(props).cuda(), torch.from_numpy(normals).cuda()
    stages = dist.HipStages(torch, 0, n, w, h)
    br = dist.BandRenderer(stages, n, w, h, 0, 1, None)
    want = []
    for u in cams:
        br.render(u, pt.data_ptr(), nt.data_ptr(), settle=True)
        torch.cuda.synchronize()
        want.append(br.image.cpu().numpy().copy())
    assert not np.array_equal(want[0], want[1])
    pipe = dist.FramePipeline(torch, br, 0)
    pipe.exchange(0, cams[0], pt.data_ptr())
    for k in range(len(cams)):
        if k + 1 < len(cams):
            pipe.exchange(k + 1, cams[k + 1], pt.data_ptr())
        img = pipe.band(k, pt.data_ptr(), nt.data_ptr(), settle=True)
        torch.cuda.synchronize()
        assert_same(img.cpu().numpy(), want[k], ("L1184", k))
    pipe.destroy()
    stages.destroy()


def test_new_entry_points_reject_bad_arguments(device):
    """Error behaviour of the frame-level entry points: invalid arguments come back as SPLAT_ERR_INVALID with a
    message, nothing is launched."""
    lib, ctx = device.lib, device.ctx
    n, w, h = 1000, 64, 64
    props, normals, u = make_case(n, w, h, 3, 1.0)
    pm = sr.SplatPropertyManager(device, n)
    pm.setFromArrays(props)
    nbuf = device.createBufferFrom(normals)
    r = sr.Renderer(device, None, "rgba8unorm", n)
    with pytest.raises(KeyError):
        r.binner.setFrameOrder("depthFirst")
    assert lib.splat_bin_set_frame_order(r.binner._b, 7) == -1
    assert b"argument check failed" in lib.splat_last_error(ctx)
    out = device.createBuffer(w * h * 4)
    uf = np.ascontiguousarray(u, np.float32).ctypes.data_as(C.POINTER(C.c_float))
    planes = pm.getPropertyPlanes()
    cfg = _lib.CompositeCfg(_lib.MODE_FRONT_TO_BACK, 1, 16, 0, 0xFFFFFFFF)
    proj = r.projector.getRecordsBuffer().ptr
    # a missing colour plane
    assert lib.splat_render_frame_planes(ctx, r.sorter._s, r.binner._b, C.byref(cfg), uf, planes.posRadius.ptr, None, nbuf.ptr, n, w, h,
                                         proj, out.ptr, None) == -1
    # the composite only implements 16-pixel tiles, and the cfg must agree with the binner
    bad = _lib.CompositeCfg(_lib.MODE_FRONT_TO_BACK, 1, 32, 0, 0xFFFFFFFF)
    assert lib.splat_render_frame_planes(ctx, r.sorter._s, r.binner._b, C.byref(bad), uf, planes.posRadius.ptr,
                                         planes.colorOpacity.ptr, nbuf.ptr, n, w, h, proj, out.ptr, None) == -1
    # more splats than the sorter was created for
    assert lib.splat_render_frame_planes(ctx, r.sorter._s, r.binner._b, C.byref(cfg), uf, planes.posRadius.ptr,
                                         planes.colorOpacity.ptr, nbuf.ptr, 10 * n + 100000, w, h, proj, out.ptr, None) == -4
    # an unknown record format for the band frame / the composite
    badfmt = _lib.CompositeCfg(_lib.MODE_FRONT_TO_BACK, 1, 16, 0, 0xFFFFFFFF, 9)
    assert lib.splat_band_frame(ctx, r.sorter._s, r.binner._b, C.byref(badfmt), pm.getPropertyBuffer().ptr, nbuf.ptr, proj, n, w, h,
                                out.ptr, None, None) == -1
    # and after all that a good frame still renders
    r.render(u, planes, nbuf, None, w, h)
    assert r.finish() == oracle_pipeline(props, normals, u, w, h)["indices"].shape[0]
    for o in (r, pm, nbuf, out):
        o.destroy()


def test_prelit_colour_plane_gives_the_same_image_bit_for_bit(device):
    """Shading applied once per splat (splat_lit_colors, SplatPropertyManager.getLitPlanes) instead of once
    per staged list entry: the same float image bit for bit (the lighting arithmetic is explicitly rounded
    in both places), in both composite modes, and the plane follows property and normal updates."""
    n, w, h = 40000, 400, 240
    props, normals, u = make_case(n, w, h, 37, 1.5)
    pm = sr.SplatPropertyManager(device, n)
    pm.setFromArrays(props)
    nbuf = device.createBufferFrom(normals)
    # the plane itself against the oracle's shading
    lit = pm.getLitPlanes(nbuf).colorOpacity.read(np.float32).reshape(n, 4)
    ndl = (normals[:, 0] * np.float32(0.577350269189625764) + normals[:, 1] * np.float32(0.577350269189625764)) \
        + normals[:, 2] * np.float32(0.577350269189625764)
    kd = np.float32(0.85) + np.float32(0.15) * np.maximum(ndl, np.float32(0))
    assert np.array_equal(lit[:, :3], props[:, 4:7] * kd[:, None]) and np.array_equal(lit[:, 3], props[:, 7])
    for mode in (sr.MODE_FRONT_TO_BACK, sr.MODE_REFERENCE_LITERAL):
        a = sr.Renderer(device, None, "rgba8unorm", n, mode=mode)
        b = sr.Renderer(device, None, "rgba8unorm", n, mode=mode)
        a.render(u, pm.getPropertyPlanes(), nbuf, None, w, h, wantFloat=True)
        b.render(u, pm.getLitPlanes(nbuf), None, None, w, h, wantFloat=True)  # no normals needed
        assert_same(a.readPixelsFloat().view(np.uint32), b.readPixelsFloat().view(np.uint32), "L1249")
        a.destroy()
        b.destroy()
    # new normals in a new buffer, then new properties: the cached plane follows
    normals2 = np.roll(normals, 1, axis=0).copy()
    nbuf2 = device.createBufferFrom(normals2)
    a = sr.Renderer(device, None, "rgba8unorm", n)
    b = sr.Renderer(device, None, "rgba8unorm", n)
    a.render(u, pm.getPropertyBuffer(), nbuf2, None, w, h, wantFloat=True)
    b.render(u, pm.getLitPlanes(nbuf2), None, None, w, h, wantFloat=True)
    assert_same(a.readPixelsFloat().view(np.uint32), b.readPixelsFloat().view(np.uint32), "L1259")
    props2 = props.copy()
    props2[:, 4:7] = props[::-1, 4:7]
    pm.setFromArrays(props2)
    a.render(u, pm.getPropertyBuffer(), nbuf2, None, w, h, wantFloat=True)
    b.render(u, pm.getLitPlanes(nbuf2), None, None, w, h, wantFloat=True)
    assert_same(a.readPixelsFloat().view(np.uint32), b.readPixelsFloat().view(np.uint32), "L1265")
    for o in (a, b, pm, nbuf, nbuf2):
        o.destroy()


def test_abi_communicator_one_rank_self_test(device):
    """splat_comm_* / splat_allgather_records on RCCL (the multi-GPU frame's one exchange behind the C ABI): a
    one-rank communicator on this device gathers a shard onto itself, out of place and in place, timed as the
    EXCHANGE stage; bad arguments come back as errors.  (More ranks need more GPUs: the driver's scaling run.)"""
    lib, ctx = device.lib, device.ctx
    ident = np.zeros(_lib.COMM_ID_BYTES, np.uint8)
    _lib.check(lib.splat_comm_unique_id(ident.ctypes.data))
    assert ident.any()
    comm = C.c_void_p()
    assert lib.splat_comm_init(ctx, 1, 1, ident.ctypes.data, C.byref(comm)) == -1  # rank out of range
    _lib.check(lib.splat_comm_init(ctx, 0, 1, ident.ctypes.data, C.byref(comm)), ctx)
    rank, world = C.c_int(-1), C.c_int(-1)
    _lib.check(lib.splat_comm_rank(comm, C.byref(rank), C.byref(world)))
    assert (rank.value, world.value) == (0, 1)
    rec = np.random.default_rng(3).standard_normal((5000, 4)).astype(np.float32)
    shard, out = device.createBufferFrom(rec), device.createBuffer(rec.nbytes)
    out.zero()
    device.setTiming(True)
    _lib.check(lib.splat_allgather_records(ctx, comm, shard.ptr, out.ptr, rec.nbytes), ctx)
    assert_same(out.read(np.float32).reshape(rec.shape), rec, "L1289")
    assert device.stageTimeMs(_lib.STAGE_EXCHANGE) >= 0.0
    device.setTiming(False)
    _lib.check(lib.splat_allgather_records(ctx, comm, shard.ptr, shard.ptr, rec.nbytes), ctx)  # in place
    assert_same(shard.read(np.float32).reshape(rec.shape), rec, "L1293")
    assert lib.splat_allgather_records(ctx, None, shard.ptr, out.ptr, rec.nbytes) == -1
    lib.splat_comm_destroy(comm)
    shard.destroy()
    out.destroy()


def test_abi_all_gather_serves_the_band_renderer(device):
    """dist.AbiAllGather — the C ABI's RCCL communicator as BandRenderer's all_gather — on a one-rank communicator: the
    collective runs on the ctx of the current torch stream (main stream and a second, registered stream, as
    FramePipeline uses it) and delivers the shard."""
    import torch
    from splat_renderer_amd import dist
    n, w, h = 20000, 320, 200
    stages = dist.HipStages(torch, 0, n, w, h)
    gather = dist.AbiAllGather(torch, stages, 0, 1, lambda ident: ident)
    shard = torch.randn((n, 4), device="cuda")
    out = torch.zeros_like(shard)
    gather(out, shard)
    torch.cuda.synchronize()
    assert torch.equal(out, shard)
    side = torch.cuda.Stream()
    proj = dist.ProjectStage(torch, 0, side)
    gather.register(side, proj.ctx)
    out.zero_()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        gather(out, shard)
    side.synchronize()
    assert torch.equal(out, shard)
    proj.destroy()
    gather.destroy()
    stages.destroy()


def test_frame_loop_dolly_in_and_out_sync_free(device):
    """Thirty frames enqueued without reading anything back in between, the camera dollying in and out so that the pair
    total swings by more than the sync-free headroom (12.5 %) frame over frame: frames whose pairs outgrow the limit
    sized from the frame before are caught at the next call and rendered again.  Every seventh frame's lists (four
    different camera distances) and the last checked frame's pixels against the oracle; the layout of the first sort pass (runs aligned to the second pass's
    partitions) moves with every frame's digit totals."""
    import os
    if os.environ.get("SPLAT_BIN_SYNC") == "1":
        pytest.skip("SPLAT_BIN_SYNC=1: every frame reads its pair total back before it sizes anything: no frame can overflow")
    n, w, h = 60000, 480, 272
    props, normals, _ = make_case(n, w, h, 77, 1.0)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    loop = sr.FrameLoop(device, n, w, h)
    totals, overflowed = [], 0
    for k in range(30):
        u = loop.camera.uniforms(w, h, time=k / 60.0).copy()
        loop.render(pbuf, nbuf)
        overflowed += int(loop.renderer.previousFrameOverflowed)
        loop.renderer.previousFrameOverflowed = False
        if k % 7 == 6:
            total = loop.renderer.finish()
            overflowed += int(loop.renderer.previousFrameOverflowed)
            loop.renderer.previousFrameOverflowed = False
            ref = oracle_pipeline(props, normals, u, w, h)
            assert total == ref["indices"].shape[0], k
            assert_same(loop.renderer.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"], ("L1350", k))
            totals.append(total)
            if k == 27:
                _, want8, _, _, near = O.composite(O.MODE_FRONT_TO_BACK, True, props[:, 4:], normals, ref["proj"], ref["indices"],
                                                   ref["counts"], ref["offsets"], w, h, want_stops=True)
                d8 = np.abs(loop.readPixels().astype(int) - want8.astype(int)).max(axis=2)
                assert d8[near == 0].max() <= 1 and d8.max() <= 3
        loop.camera.zoom(-0.45 if (k // 3) % 2 == 0 else 0.45)  # three steps in, three steps out
    assert overflowed >= 1, "the dolly must outgrow the sync-free limit at least once"
    assert max(totals) > 1.2 * min(totals)
    for o in (loop, pbuf, nbuf):
        o.destroy()


def test_headless_frame_loop_orbits_the_camera(device, tmp_path):
    """the frame loop of src/main.ts:110-193 without a browser.  Four frames of an orbit
    (Camera.rotate, then a pan, a zoom and a drag through OrbitCameraController), enqueued back to back — the second
    onwards sync-free, each with a different camera — and every frame's lists and pixels against the oracle run with
    that frame's uniforms; the frames written as PNG read back identical."""
    n, w, h = 12000, 320, 208
    props, normals, _ = make_case(n, w, h, 61, 1.5)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    loop = sr.FrameLoop(device, n, w, h)
    ctl = sr.OrbitCameraController(loop.camera)
    moves = [lambda: loop.camera.rotate(2 * np.pi / 4, 0.0), lambda: loop.camera.pan(0.2, -0.1),
             lambda: ctl.onWheel(sr.MouseEvent(deltaY=400.0)),
             lambda: (ctl.onMouseDown(sr.MouseEvent(10, 10, button=0)), ctl.onMouseMove(sr.MouseEvent(70, 40)), ctl.onMouseUp())]
    shots = []
    for k, move in enumerate(moves):
        u = loop.camera.uniforms(w, h, time=k / 60.0).copy()
        loop.render(pbuf, nbuf)
        shots.append((u, loop.readPixels().copy(), loop.renderer.binner.getTotalIndices(),
                      loop.renderer.binner.getTileIndicesBuffer().read(np.uint32, loop.renderer.binner.getTotalIndices())))
        move()
    assert len({s[1].tobytes() for s in shots}) == len(shots)  # the camera really moved
    for k, (u, got8, total, idx) in enumerate(shots):
        ref = oracle_pipeline(props, normals, u, w, h)
        assert total == ref["indices"].shape[0], k
        assert_same(idx, ref["indices"], ("L1387", k), offsets=ref["offsets"])
        _, want8, _, _, near = O.composite(O.MODE_FRONT_TO_BACK, True, props[:, 4:], normals, ref["proj"], ref["indices"], ref["counts"],
                                           ref["offsets"], w, h, want_stops=True)
        d8 = np.abs(got8.astype(int) - want8.astype(int)).max(axis=2)
        assert d8[near == 0].max() <= 1 and d8.max() <= 3, k
        path = tmp_path / f"frame_{k}.png"
        sr.write_png(path, got8)
        assert_same(sr.read_png(path), got8, "L1394")
    # the same frames without reading anything back in between (all sync-free): same last image
    cam2 = sr.Camera()
    loop2 = sr.FrameLoop(device, n, w, h, camera=cam2)
    ctl2 = sr.OrbitCameraController(cam2)
    for k in range(4):
        loop2.render(pbuf, nbuf)
        if k < 3:
            [lambda: cam2.rotate(2 * np.pi / 4, 0.0), lambda: cam2.pan(0.2, -0.1), lambda: ctl2.onWheel(sr.MouseEvent(deltaY=400.0))][k]()
    assert_same(loop2.readPixels(), shots[3][1], "L1403")
    for o in (loop, loop2, pbuf, nbuf):
        o.destroy()


def test_pipelined_renderer_keeps_frames_identical(device):
    """Two frames in flight (frames alternate between two streams): every frame — a different camera each — is the frame
    the one-stream Renderer gives, bit for bit, including the sync-free ones."""
    n, w, h = 30000, 400, 240
    props, normals, _ = make_case(n, w, h, 71, 1.5)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    cam = sr.Camera()
    cam.setAspect(w / h)
    uniforms = []
    for k in range(6):
        uniforms.append(cam.uniforms(w, h).copy())
        cam.rotate(0.3, 0.05)
    one = sr.Renderer(device, None, "rgba8unorm", n)
    want = []
    for u in uniforms:
        one.render(u, pbuf, nbuf, None, w, h)
        want.append(one.readPixels().copy())
    pr = sr.PipelinedRenderer(0, 2, n)
    got = []
    for k, u in enumerate(uniforms):
        pr.render(u, pbuf, nbuf, None, w, h)
        if k >= 3:  # the first three go unread: frames 1 and 2 are enqueued while frame 0 is in flight
            got.append((k, pr.readPixels().copy()))
    pr.finish()
    for k, img in got:
        assert_same(img, want[k], ("L1433", k))
    assert len({w_.tobytes() for w_ in want}) == len(want)
    for o in (pr, one, pbuf, nbuf):
        o.destroy()


def test_four_contexts_interleaved_keep_their_frames(device):
    """Four contexts (four non-blocking streams) fed alternately from one thread, nothing waited for in between — each its own
    scene, screen size and order of work, each rendering several sync-free frames: every image is the one the same context
    gives alone.  (Distinct contexts are independent by contract; this holds the library to it: no shared scratch, no
    process-wide state, nothing on the null stream.)"""
    cases = [(20000, 320, 200, 81, 1.0, "tileFirst"), (9000, 400, 240, 82, 2.0, "sortFirst"), (30000, 256, 256, 83, 0.7, "tileFirst"),
             (5000, 640, 352, 84, 3.0, "tileFirst")]
    devs, sets, want = [sr.Device(0) for _ in cases], [], []
    for d, (n, w, h, seed, rs, order) in zip(devs, cases):
        props, normals, u = make_case(n, w, h, seed, rs)
        r = sr.Renderer(d, None, "rgba8unorm", n, frameOrder=order)
        pb, nb = d.createBufferFrom(props), d.createBufferFrom(normals)
        r.render(u, pb, nb, None, w, h)
        want.append(r.readPixels().copy())  # alone (and the frame that sizes the sync-free ones)
        sets.append((r, u, pb, nb, w, h))
    for rounds in range(5):
        for r, u, pb, nb, w, h in sets:  # enqueue all four, twice over, before anything is read
            r.render(u, pb, nb, None, w, h)
        for r, u, pb, nb, w, h in reversed(sets):
            r.render(u, pb, nb, None, w, h)
        for k, (r, u, pb, nb, w, h) in enumerate(sets):
            assert_same(r.readPixels(), want[k], ("four contexts", rounds, k))
    assert len({w_.tobytes() for w_ in want}) == len(want)
    for (r, u, pb, nb, w, h), d in zip(sets, devs):
        assert r.framesMisranked == 0
        for o in (r, pb, nb, d):
            o.destroy()


def test_C4_workload_eight_virtual_ranks_through_the_all_gather_cut(device):
    """BASELINE.json configs[4] — 10M Gaussians @3840x2160 sharded by tile rows over 8 ranks with ONE all-gather of projected
    splats — as far as one GPU can run it: the eight ranks' device work runs in turn on the one device.  Every rank projects
    its slice of the splats into 16-byte exchange records (splat_project_slice_compact), every shard travels through the C
    ABI's own RCCL communicator (a one-rank communicator: the collective is real, its peers are not), and every rank renders
    its band — cut so that the bands carry equal pairs, as bench.py cuts them — from the gathered records (splat_band_frame:
    band filter + tile-first binning + lit composite records for the kept splats + composite).  The stitched rgba8 image
    must be the single-GPU C3 frame byte for byte, the bands' pair totals must add up to the frame's 29 483 686, and every
    rank's gathered blocks must be what a local projection of the other ranks' slices gives."""
    import torch
    from splat_renderer_amd import dist
    name, world = "C3", 8
    n, w, h = sr.scene.CONFIGS[name]
    props, normals, u = make_case(n, w, h)
    pbuf, nbuf = device.createBufferFrom(props), device.createBufferFrom(normals)
    full = sr.Renderer(device, None, "rgba8unorm", n)
    full.render(u, pbuf, nbuf, None, w, h)
    assert full.finish() == FULL_SIZE_PAIRS[name]
    want = full.readPixels().copy()
    for o in (full, pbuf, nbuf):
        o.destroy()
    pt, nt = torch.from_numpy(props).cuda(), torch.from_numpy(normals).cuda()
    per = dist.shard_size(n, world)
    stages = dist.HipStages(torch, 0, per * world, w, h)
    gather = dist.AbiAllGather(torch, stages, 0, 1, lambda ident: ident)
    gathered = stages.new_records(per * world)
    ranks = []
    for r in range(world):  # phase 1: every rank projects its slice; the communicator delivers it into its block
        br = dist.BandRenderer(stages, n, w, h, r, world, None, gathered=gathered)  # (the virtual ranks share the gathered records)
        stages.project_slice(u, pt.data_ptr(), br.first, br.count, br.shard)
        gather(gathered[r * per:(r + 1) * per], br.shard)
        ranks.append(br)
    torch.cuda.synchronize()
    # the bands: equal pairs per band from the rows' pair counts (one calibration frame over all rows, as bench.py's first frame)
    nty = -(-h // 16)
    stages.band_frame(gathered, per * world, pt.data_ptr(), nt.data_ptr(), 0, nty, ranks[0].image, settle=True)
    rows = stages.row_pairs()
    assert int(rows.sum()) == FULL_SIZE_PAIRS[name] == stages.pairs
    bands = dist.balanced_rows(rows, world)
    assert bands[0][0] == 0 and bands[-1][1] == nty and all(bands[r][1] == bands[r + 1][0] for r in range(world - 1))
    got = np.zeros_like(want)
    pairs, kept = [], []
    for r, br in enumerate(ranks):  # phase 2: every rank renders its band from the gathered records, twice (the second sync-free)
        br.row0, br.row1 = bands[r]
        for settle in (False, True):
            stages.band_frame(gathered, per * world, pt.data_ptr(), nt.data_ptr(), br.row0, br.row1, br.image, settle=settle)
        pairs.append(stages.pairs)
        kept.append(stages.kept)
        assert br.verify_exchange(u, pt.data_ptr()) == world - 1, f"rank {r}: a gathered block differs from a local projection of that slice"
        r0, r1 = br.pixel_rows()
        got[r0:r1] = br.image.cpu().numpy()[r0:r1]
    assert_same(got, want, "C4: the eight bands do not stitch into the single-GPU frame")
    assert sum(pairs) == FULL_SIZE_PAIRS[name], pairs
    assert max(pairs) <= 1.2 * min(pairs), f"bands are cut for equal pairs: {pairs}"
    assert all(0 < k < n for k in kept) and sum(kept) >= n * 0.9, kept
    assert stages.overflows == 0 and stages.misranked == 0 and stages.rank_status()["orderFaults"] == 0
    gather.destroy()
    stages.destroy()


@pytest.mark.parametrize("launcher", ["torchrun", "plain"])
def test_bench_band_path_runs_as_a_fresh_process(device, launcher):
    """bench.py's N > 1 code path (run_multi: slice projection, the all-gather through the C ABI's RCCL communicator,
    band frame, band rebalancing, the loop trial, the exchange-free trial, the exchange's self-checks and the JSON line)
    launched as a FRESH child process with one rank — under torch.distributed.run as the driver launches it, and
    plainly.  With one GPU nothing else reaches that function (WORLD_SIZE 1 takes run_single): VERDICT r4 item 1."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (torchrun: the default policy, --exchange auto — both cuts timed, the faster run; plain: north_star's all-gather cut forced)
    tail = [os.path.join(root, "bench.py"), "--gpus", "1", "--band-path", "--config", "C1", "--steps", "6", "--warmup", "2"] + \
           ([] if launcher == "torchrun" else ["--exchange", "allgather"])
    if launcher == "torchrun":
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
    else:
        cmd = [sys.executable] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    n, w, h = sr.scene.CONFIGS["C1"]
    cfg = line["config"]
    assert line["n_gpus"] == 1 and line["steps"] == 6 and line["unit"] == "Msplats/s" and "band_path" in cfg
    assert line["value"] > 0 and abs(line["value"] - n / line["ms_per_step"] / 1e3) <= 1e-6 * line["value"]
    ex = cfg["exchange"]
    assert ex["rccl_ranks_seen"] == [1] and ex["rccl_rank_of_each_process"] == [0]
    assert ex["shards_verified_per_rank"] == [1] and ex["shards_expected_per_rank"] == 1  # (the rank's own block, out of place)
    assert cfg["ranking"]["orderFaults"] == [0] and (os.environ.get("SPLAT_RANK") or cfg["ranking"]["policy"] == ["checked"])
    assert cfg["per_rank"][0]["tile_rows"] == [0, -(-h // 16)] and cfg["per_rank"][0]["pairs_consumed"] > 0
    assert {"serial", "no_exchange_every_rank_projects_all"} <= set(cfg["frame_loop_trial_ms"])
    assert cfg["exchange_policy"] == ("auto" if launcher == "torchrun" else "allgather")
    assert cfg["exchange_chosen"] in (("allgather", "none") if launcher == "torchrun" else ("allgather",))
    assert (cfg["collective"] is None) == (cfg["exchange_chosen"] == "none") and ex["in_timed_region"] == (cfg["exchange_chosen"] == "allgather")
    assert 0 < line["roofline"]["frac"] < 1 and line["roofline"]["kernel"] == "k_composite_px"
