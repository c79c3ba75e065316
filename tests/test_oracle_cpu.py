"""CPU tests of the oracle: known answers, C-vs-NumPy twin agreement, golden fixtures.

The reference holds no tests for this path; the two known answers it does state are
checked first, then the two independent restatements are held against each other and against the
committed fixtures.
"""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import np_oracle as NP
from oracle import oracle as O
from splat_renderer_amd import scene
from tests.helpers import make_case, oracle_pipeline

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_scan_known_answer():
    # /root/reference/GPU_PIPELINE_PLAN.md:632-635
    out, total = O.scan_exclusive(np.array([1, 2, 3, 4, 5], np.uint32))
    assert out.tolist() == [0, 1, 3, 6, 10] and total == 15
    out2, total2 = NP.scan_exclusive(np.array([1, 2, 3, 4, 5], np.uint32))
    assert out2.tolist() == [0, 1, 3, 6, 10] and total2 == 15
    assert O.scan_exclusive(np.zeros(7, np.uint32))[0].tolist() == [0] * 7
    big = np.arange(1000, dtype=np.uint32)
    assert np.array_equal(O.scan_exclusive(big)[0], NP.scan_exclusive(big)[0])


def test_depth_key_mapping_definition():
    # src/shaders/extract-depth-keys.wgsl:55-59: positive floats flip the sign bit, negative flip all
    depths = np.array([0.0, 1.0, 1.5, 2.0, 1e-30, 3.4e38, -0.0, -1.0, -2.0], np.float32)
    rec = np.zeros((depths.shape[0], 8), np.float32)
    rec[:, 4] = depths
    keys, payload = O.extract_keys(rec, 12)
    assert keys[0] == 0x80000000 and keys[1] == (0x3F800000 ^ 0x80000000)
    assert keys[6] == 0x7FFFFFFF and keys[7] == (0xBF800000 ^ 0xFFFFFFFF)
    assert (keys[9:] == 0xFFFFFFFF).all() and (payload[9:] == 0xFFFFFFFF).all()
    assert payload[:9].tolist() == list(range(9))
    # order-preserving on everything but NaN: key order == float order
    rng = np.random.default_rng(0)
    d = np.concatenate([rng.standard_normal(5000).astype(np.float32) * 10, depths])
    rec = np.zeros((d.shape[0], 8), np.float32)
    rec[:, 4] = d
    k, _ = O.extract_keys(rec)
    order_k = np.argsort(k, kind="stable")
    assert (np.diff(d[order_k]) >= 0).all()
    assert np.array_equal(k, NP.extract_keys(rec)[0])


@pytest.mark.parametrize("n,w,h,seed,rs", [(1, 64, 64, 1, 1.0), (7, 64, 48, 2, 1.0), (500, 96, 80, 7, 3.0),
                                           (3000, 250, 130, 9, 1.0), (10000, 256, 256, 1234, 1.0)])
def test_c_oracle_matches_numpy_twin(n, w, h, seed, rs):
    props, normals, u = make_case(n, w, h, seed, rs)
    a = oracle_pipeline(props, normals, u, w, h, n_padded=scene.padded_size(n))
    proj = NP.project(u, props[:, :4])
    assert np.array_equal(bits(a["proj"]), bits(proj))
    keys, pay = NP.extract_keys(proj, scene.padded_size(n))
    assert np.array_equal(a["keys"], keys) and np.array_equal(a["payload"], pay)
    sk, order = NP.sort_pairs(keys, pay)
    assert np.array_equal(a["order"], order) and np.array_equal(a["sorted_keys"], sk)
    if n <= 3000:
        counts, offsets, idx = NP.bin_sorted(proj, order, w, h)
        assert np.array_equal(a["counts"], counts) and np.array_equal(a["offsets"], offsets)
        assert np.array_equal(a["indices"], idx)
    if n <= 500:
        for mode in (0, 1):
            for eo in (False, True):
                img, img8, _ = O.composite(mode, eo, props[:, 4:], normals, a["proj"], a["indices"], a["counts"],
                                           a["offsets"], w, h)
                twin = NP.composite(mode, eo, props[:, 4:], normals, proj, a["indices"], a["counts"], a["offsets"], w, h)
                assert np.abs(img - twin).max() < 1e-6
                assert np.abs(img8.astype(int) - NP.unorm8(twin).astype(int)).max() <= 1


def test_sort_is_stable_and_pads_last():
    keys = np.array([5, 1, 5, 0xFFFFFFFF, 1, 0, 0xFFFFFFFF, 5], np.uint32)
    pay = np.arange(8, dtype=np.uint32)
    sk, sp = O.sort_pairs(keys, pay)
    assert sp.tolist() == [5, 1, 4, 0, 2, 7, 3, 6]
    assert sk.tolist() == sorted(keys.tolist())


def test_bin_sorted_semantics():
    w, h = 64, 64
    rec = np.zeros((6, 8), np.float32)
    rec[0, :4] = [-50, 10, -20, 30]   # fully left of the screen: culled (TileBinner.ts:437)
    rec[1, :4] = [10, 70, 30, 90]     # fully below
    rec[2, :4] = [-5, -5, 5, 5]       # clamps into tile 0
    rec[3, :4] = [60, 60, 100, 100]   # clamps into the last tile
    rec[4, :4] = [16, 16, 32, 32]     # exactly on tile edges: tiles 1..2 x 1..2 (floor(32/16) = 2)
    rec[5, :4] = [np.nan, 0, 10, 10]  # NaN bins nowhere
    order = np.array([3, 0xFFFFFFFF, 4, 2, 1, 0, 5], np.uint32)
    counts, offsets, idx = O.bin_sorted(rec, order, w, h)
    c2, o2, i2 = NP.bin_sorted(rec, order, w, h)
    assert np.array_equal(counts, c2) and np.array_equal(offsets, o2) and np.array_equal(idx, i2)
    grid = counts.reshape(4, 4)
    assert grid[0, 0] == 1 and grid[3, 3] == 1 and grid[1:3, 1:3].tolist() == [[1, 1], [1, 1]] and counts.sum() == 6
    assert idx.tolist() == [2, 4, 4, 4, 4, 3]  # per tile, in `order` order


def test_bin_sorted_equals_gpu_shader_range_on_screen():
    """count-tile-hits.wgsl's range equals binSorted's for every splat binSorted keeps."""
    n, w, h = 5000, 320, 200
    props, normals, u = make_case(n, w, h, 3, 2.0)
    a = oracle_pipeline(props, normals, u, w, h)
    ntx, nty = 20, 13
    hits = 0
    for s in range(n):
        rec = a["proj"][s]
        mnx, mny, mxx, mxy = max(rec[0], 0), max(rec[1], 0), min(rec[2], w), min(rec[3], h)
        if mnx >= mxx or mny >= mxy:
            continue
        r = O.gpu_tile_range(rec, 16, ntx, nty)
        hits += int((r[2] - r[0] + 1) * (r[3] - r[1] + 1))
    assert hits == a["indices"].shape[0]


def test_composite_modes_consume_the_same_entries():
    """Front-to-back and the literal loop stop at the same list position (alpha is order-free);
    with a single-entry list they are the same image."""
    n, w, h = 400, 96, 80
    props, normals, u = make_case(n, w, h, 11, 3.0)
    a = oracle_pipeline(props, normals, u, w, h)
    args = (props[:, 4:], normals, a["proj"], a["indices"], a["counts"], a["offsets"], w, h)
    _, _, c0 = O.composite(0, True, *args)
    _, _, c1 = O.composite(1, True, *args)
    assert c0 == c1
    one = oracle_pipeline(props[:1], normals[:1], u, w, h)
    i0, _, _ = O.composite(0, True, props[:1, 4:], normals[:1], one["proj"], one["indices"], one["counts"], one["offsets"], w, h)
    i1, _, _ = O.composite(1, True, props[:1, 4:], normals[:1], one["proj"], one["indices"], one["counts"], one["offsets"], w, h)
    assert np.abs(i0 - i1).max() < 1e-6


def test_background_and_alpha():
    # no splats: every pixel is the background (0.05, 0.05, 0.1, 1) — ComputeShaderRenderer.ts:193-197
    w, h = 32, 32
    img, img8, _ = O.composite(0, True, np.zeros((1, 4), np.float32), np.zeros((1, 4), np.float32),
                               np.zeros((1, 8), np.float32), np.zeros(0, np.uint32), np.zeros(4, np.uint32),
                               np.zeros(4, np.uint32), w, h)
    assert np.allclose(img[..., :3], [0.05, 0.05, 0.1]) and (img[..., 3] == 1).all()
    assert (img8[..., :3] == [13, 13, 26]).all() and (img8[..., 3] == 255).all()


def test_model_b_single_splat_facing_camera():
    """SequentialRenderer restatement: one splat at the origin, normal towards the camera, draws a
    disc of alpha exp(-d2/0.32) over the clear colour; its centre pixel is almost the lit colour."""
    w = h = 65
    vp, eye = O.camera(aspect=1.0)
    u = O.uniforms(vp, eye, w, h)
    nrm = (eye / np.linalg.norm(eye)).astype(np.float32)
    props = np.array([[0, 0, 0, 0.3, 1.0, 0.5, 0.25, 1.0]], np.float32)
    normals = np.array([[nrm[0], nrm[1], nrm[2], 1.0]], np.float32)
    img, img8 = O.sequential(u, props, normals, np.array([0], np.uint32), w, h)
    c = img[h // 2, w // 2]
    kd = 0.85 + 0.15 * max(float(nrm.sum() / np.sqrt(3.0)), 0.0)
    assert np.allclose(c[:3], np.array([1.0, 0.5, 0.25]) * kd, atol=0.02)
    assert np.allclose(img[0, 0, :3], [0.05, 0.05, 0.1])  # untouched corner = clear colour
    assert (img[..., 3] > 0.999).all()                      # alpha stays 1 over an opaque clear


def test_update_props_matches_twin():
    rng = np.random.default_rng(2)
    pos = rng.uniform(-1, 1, (100, 4)).astype(np.float32)
    cur = rng.standard_normal((100, 4)).astype(np.float32)
    assert np.array_equal(bits(O.update_props(pos, cur)), bits(NP.update_props(pos, cur)))


REF_BINS = ("tiny7", "small300", "ragged1000", "edges")


@pytest.mark.parametrize("name", REF_BINS)
def test_oracle_tile_lists_equal_the_reference_own_code(name):
    """ref_binsorted_*.npz hold what the REFERENCE's binSorted loops (src/TileBinner.ts:426-495, executed under Node
    by tests/golden/make_ref_fixtures.py) produce for these records and sorted orders — including off-screen,
    straddling, degenerate, NaN and infinite bounds and padding indices: both restatements must give exactly that."""
    g = np.load(os.path.join(GOLDEN, f"ref_binsorted_{name}.npz"))
    w, h, tile = (int(x) for x in g["dims"])
    for impl in (O, NP):
        counts, offsets, idx = impl.bin_sorted(g["projected"], g["sorted"], w, h, tile)
        assert np.array_equal(counts, g["counts"]) and np.array_equal(offsets, g["offsets"]), impl.__name__
        assert np.array_equal(idx, g["indices"]), impl.__name__
    if name != "edges":  # the same inputs as the oracle-made fixture of that name: one more link between the two sets
        f = np.load(os.path.join(GOLDEN, name + ".npz"))
        assert np.array_equal(f["indices"], g["indices"]) and np.array_equal(bits(f["projected"]), bits(g["projected"]))


def test_oracle_scan_equals_the_reference_own_code():
    """ref_scan.npz: the loop of PrefixSumScanner.scanCPU (src/PrefixSumScanner.ts:150-155) executed under Node."""
    g = np.load(os.path.join(GOLDEN, "ref_scan.npz"))
    k = 0
    while f"in{k}" in g.files:
        assert np.array_equal(O.scan_exclusive(g[f"in{k}"])[0], g[f"out{k}"]), k
        assert np.array_equal(NP.scan_exclusive(g[f"in{k}"])[0], g[f"out{k}"]), k
        k += 1
    assert k >= 5 and g["out0"].tolist() == [0, 1, 3, 6, 10]  # the reference's stated known answer


@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if not os.path.basename(p).startswith("ref_")))
def test_oracle_reproduces_golden(path):
    g = np.load(path)
    n, w, h, seed = (int(x) for x in g["dims"])
    props, normals = scene.make_scene(n, seed=seed)
    props[:, 3] *= g["radius_scale"]
    assert np.array_equal(bits(props), bits(g["props"])) and np.array_equal(bits(normals), bits(g["normals"]))
    vp, eye = O.camera(aspect=w / h)
    u = O.uniforms(vp, eye, w, h)
    assert np.array_equal(bits(u), bits(g["uniforms"]))
    a = oracle_pipeline(props, normals, u, w, h, n_padded=scene.padded_size(n))
    assert np.array_equal(bits(a["proj"]), bits(g["projected"]))
    for k, gk in (("keys", "keys"), ("payload", "payload"), ("order", "order"), ("counts", "counts"),
                  ("offsets", "offsets"), ("indices", "indices")):
        assert np.array_equal(a[k], g[gk]), k
    img, img8, _ = O.composite(0, True, props[:, 4:], normals, a["proj"], a["indices"], a["counts"], a["offsets"], w, h)
    assert np.abs(img - g["image_front_to_back"]).max() < 1e-6 and np.array_equal(img8, g["image_front_to_back_u8"])
    lit, lit8, _ = O.composite(1, True, props[:, 4:], normals, a["proj"], a["indices"], a["counts"], a["offsets"], w, h)
    assert np.abs(lit - g["image_literal"]).max() < 1e-6
    mb, mb8 = O.sequential(u, props, normals, a["order"][:n][::-1].copy(), w, h)
    assert np.abs(mb - g["image_model_b"]).max() < 1e-6


def test_C0_digest():
    """BASELINE configs[0] (10k @256x256, plumbing config): integer results by digest."""
    with open(os.path.join(GOLDEN, "C0_digest.json")) as f:
        d = json.load(f)
    n, w, h = scene.CONFIGS["C0"]
    props, normals, u = make_case(n, w, h)
    a = oracle_pipeline(props, normals, u, w, h, n_padded=scene.padded_size(n))
    assert a["indices"].shape[0] == d["pairs"] == 137051  # also SURVEY §8's dry-run figure
    sha = lambda x: hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()  # noqa: E731
    for k, ak in (("projected", "proj"), ("keys", "keys"), ("order", "order"), ("counts", "counts"),
                  ("offsets", "offsets"), ("indices", "indices")):
        assert sha(a[ak]) == d["sha256"][k], k


def test_frame_entry_point_matches_staged_calls():
    n, w, h = 2000, 128, 96
    props, normals, u = make_case(n, w, h, 4, 2.0)
    a = oracle_pipeline(props, normals, u, w, h)
    want, want8, _ = O.composite(0, True, props[:, 4:], normals, a["proj"], a["indices"], a["counts"], a["offsets"], w, h)
    for threads in (1, 3):
        r = O.frame(u, props, normals, w, h, threads=threads)
        assert r["total_pairs"] == a["indices"].shape[0]
        assert np.array_equal(r["out_u8"], want8) and np.abs(r["out_f32"] - want).max() == 0


# ---- oriented-disc footprint (SequentialRenderer's splat evaluated per pixel) -------------------------
DISC_TOL = 1e-4      # |disc composite - software rasteriser| per channel, float image, no early-out
DISC_RIM_TOL = 0.045  # on pixels within 1e-3 (in u^2+v^2) of a disc's rim: the discard is a step of exp(-3.125) = 0.0439


def _disc_case(n, w, h, seed, rs):
    from splat_renderer_amd import scene
    from splat_renderer_amd.camera import Camera
    props, normals = scene.make_scene(n, seed=seed)
    props[:, 3] *= rs
    cam = Camera()
    cam.setAspect(w / h)
    return props, normals, cam.uniforms(w, h)


@pytest.mark.parametrize("n,w,h,seed,rs", [(2000, 128, 96, 5, 3.0), (10000, 256, 256, 1234, 1.0), (400, 64, 64, 6, 12.0)])
def test_disc_projector_matches_numpy_twin(n, w, h, seed, rs):
    props, normals, u = _disc_case(n, w, h, seed, rs)
    normals[::7, :3] = [0.05, 0.99, 0.1]   # |n.y| > 0.9: the other "up" (:69)
    normals[5] = 0                          # zero normal: NaN tangent -> culled
    props[9, :3] = [0.0, 0.0, 50.0]         # behind the camera
    proj, discs = O.project_disc(u, props, normals)
    proj2, discs2 = NP.project_disc(u, props[:, :4], normals)
    assert np.array_equal(discs.view(np.uint32), discs2.view(np.uint32))
    assert np.array_equal(proj.view(np.uint32), proj2.view(np.uint32))
    assert not discs[5].any() and not proj[5, :4].any() and not discs[9].any()
    # the sort key is the isotropic projector's (distance to the eye)
    assert np.array_equal(proj[:, 4], O.project(u, props)[:, 4])


def test_disc_bounds_are_the_extent_of_the_unit_circle():
    props, normals, u = _disc_case(300, 256, 256, 7, 6.0)
    proj, discs = O.project_disc(u, props, normals)
    th = np.linspace(0, 2 * np.pi, 20001)
    worst = 0.0
    for i in range(0, 300, 7):
        if not discs[i].any():
            continue
        # forward map of the unit circle: solve (u,v) = B d / (1 - q.d) for d by rasterising its inverse densely
        r = discs[i].astype(np.float64)
        B = np.array([[r[2], r[3]], [r[4], r[5]]])
        A = np.linalg.inv(B)            # A / w_c
        g = A.T @ r[6:8]                # g / w_c
        uv = np.stack([np.cos(th), np.sin(th)])
        w_ = 1.0 + g @ uv
        d = (A @ uv) / w_
        lo, hi = d.min(axis=1) + r[:2], d.max(axis=1) + r[:2]
        size = max(hi[0] - lo[0], hi[1] - lo[1])
        worst = max(worst, np.abs(np.concatenate([lo, hi]) - proj[i, :4]).max() / size)
    assert worst < 1e-4  # (the polyline under-samples the extremes by ~1e-8 of the size; f32 records by ~1e-6)


@pytest.mark.parametrize("n,w,h,seed,rs", [(10000, 256, 256, 1234, 1.0), (3000, 160, 120, 8, 2.5)])
def test_disc_composite_is_the_sequential_renderer_image(n, w, h, seed, rs):
    """Tile lists + per-pixel inverse homography == one oriented quad per splat through a rasteriser, back to front."""
    props, normals, u = _disc_case(n, w, h, seed, rs)
    proj, discs = O.project_disc(u, props, normals)
    keys, pay = O.extract_keys(proj)
    _, order = O.sort_pairs(keys, pay)
    counts, offsets, idx = O.bin_sorted(proj, order, w, h)
    img, img8, _, rim = O.composite_disc(False, props[:, 4:], normals, discs, idx, counts, offsets, w, h)
    ref, ref8 = O.sequential(u, props, normals, order[::-1].copy(), w, h)
    d = np.abs(img - ref).max(axis=2)
    assert d[rim == 0].max() <= DISC_TOL
    assert d.max() <= DISC_RIM_TOL
    assert (d > DISC_TOL).sum() <= 8   # rim flips are rare events even among rim pixels
    d8 = np.abs(img8.astype(int) - ref8.astype(int)).max(axis=2)
    assert d8[rim == 0].max() <= 1
    # pre-lit colours (normals=None) give the same bits as shading in the loop
    lit = props[:, 4:].copy()
    k = np.float32(1.0) / np.sqrt(np.float32(3.0))
    ndl = (normals[:, 0] * k + normals[:, 1] * k) + normals[:, 2] * k
    lit[:, :3] *= (np.float32(0.85) + np.float32(0.15) * np.maximum(ndl, np.float32(0)))[:, None]
    img2, _, _, _ = O.composite_disc(False, lit, None, discs, idx, counts, offsets, w, h)
    assert np.array_equal(img, img2)
