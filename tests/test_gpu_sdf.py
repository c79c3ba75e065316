"""SDF splat generation on the GPU against the oracle: gradients, projection steps, scale factors and
the vec4(normal, scale) buffer are bit-exact (both sides do one IEEE operation per operator, in the same order); a frame
rendered from generated splats equals the oracle's frame from the oracle's splats."""
import ctypes as C

import numpy as np
import pytest

import splat_renderer_amd as sr
from oracle import oracle as O
from splat_renderer_amd import _lib, sdf
from tests.helpers import oracle_pipeline
from tests.test_sdf_cpu import main_ts_scene

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same(a, b):
    """Bit-equal, except that a NaN matches any NaN (0/0 has a different sign bit on x86 and on the GPU)."""
    a, b = np.ascontiguousarray(a, np.float32).reshape(-1), np.ascontiguousarray(b, np.float32).reshape(-1)
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a[~na].view(np.uint32), b[~nb].view(np.uint32))


def scenes():
    out = {"main_ts": main_ts_scene()}
    for name, prim in (("sphere", sdf.Sphere(position=(0.1, -0.2, 0.05), radius=0.45)), ("box", sdf.Box(position=(0, 0.1, 0), size=(0.4, 0.25, 0.3))),
                       ("torus", sdf.Torus(position=(0, 0, 0.1), majorRadius=0.5, minorRadius=0.15)),
                       ("capsule", sdf.Capsule(position=(-0.1, 0, 0), height=0.8, radius=0.2))):
        s = sr.SDFScene()
        s.setRoot(prim)
        out[name] = s
    s = sr.SDFScene()  # every operation, three levels deep
    s.setRoot(sdf.subtraction(sdf.union(sdf.Sphere(radius=0.5), sdf.intersection(sdf.Box(size=(0.6, 0.2, 0.6)), sdf.Torus(majorRadius=0.45, minorRadius=0.2))),
                              sdf.smoothUnion(0.08, sdf.Capsule(position=(0.2, 0, 0), height=1.2, radius=0.12), sdf.Sphere(position=(0, 0.4, 0), radius=0.2))))
    out["all_ops"] = s
    return out


@pytest.mark.parametrize("name", ["main_ts", "sphere", "box", "torus", "capsule", "all_ops"])
def test_sdf_stages_bit_exact(device, name):
    scene = scenes()[name]
    prog = scene.program()
    n = 20011
    rng = np.random.default_rng(5)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-1.2, 1.2, (n, 3))
    pos[:64, :3] = 0.0  # degenerate points: centres, axes (zero gradients, sign(0))
    pos[64:128, 1:3] = 0.0
    gs, cs, pu = sr.GradientSampler(device, scene, n), sr.CurvatureSampler(device, scene, n), sr.PositionUpdater(device, None, n)
    a, b = device.createBufferFrom(pos), device.createBuffer(n * 16)
    want = pos
    for _ in range(5):  # src/main.ts:149-172
        gs.evaluateGradients(None, None, a)
        wg = O.sdf_gradients(prog, want)
        assert same(gs.getGradientBuffer().read(np.float32), wg)
        pu.updatePositions(None, None, a, gs.getGradientBuffer(), b)
        want = O.sdf_update_positions(want, wg)
        assert same(b.read(np.float32), want)
        a, b = b, a
    cs.computeScaleFactors(None, a)
    wsf = O.sdf_scale_factors(prog, want)
    got_sf = cs.getScaleFactorsBuffer().read(np.float32)
    assert same(got_sf, wsf) and np.isnan(wsf).sum() < n // 50  # (NaN only where a normal is undefined: a zero gradient)
    cur = cs.getCurvatureBuffer(gs.getGradientBuffer()).read(np.float32).reshape(n, 4)
    assert same(cur, O.sdf_curvature(wg, wsf))
    for o in (gs, cs, a, b):
        o.destroy()


def test_sdf_program_errors_and_animation(device):
    lib, ctx = device.lib, device.ctx
    n = 256
    pos = device.createBufferFrom(np.zeros((n, 4), np.float32))
    out = device.createBuffer(n * 16)

    def run(program):
        arr = (_lib.SdfInstr * max(len(program), 1))()
        for k, (op, a) in enumerate(program):
            arr[k].op = op
            for j, v in enumerate(a):
                arr[k].a[j] = v
        return lib.splat_sdf_gradients(ctx, C.cast(arr, C.c_void_p), len(program), pos.ptr, n, out.ptr)
    assert run([(16, [])]) == -1 and b"without two operands" in lib.splat_last_error(ctx)
    assert run([(0, [0, 0, 0, 1]), (0, [0, 0, 0, 1])]) == -1 and b"does not reduce" in lib.splat_last_error(ctx)
    assert run([(7, [])]) == -1 and b"unknown opcode" in lib.splat_last_error(ctx)
    assert run([(0, [0, 0, 0, 1])] * 9 + [(16, [])] * 8) == -1 and b"deeper" in lib.splat_last_error(ctx)
    assert run([(0, [0, 0, 0, 1])] * 17 + [(16, [])] * 16) == -1  # more than SPLAT_SDF_MAX_INSTR
    assert run([]) == 0  # the empty scene: distance 1000, gradient (0, 1, 0) (CodeGenerator.ts:282-286)
    assert np.array_equal(out.read(np.float32).reshape(n, 4), np.tile(np.float32([1000, 0, 1, 0]), (n, 1)))
    # animating a primitive (src/main.ts:114-120): new parameters reach the kernel with updateSceneParameters()
    scene = main_ts_scene()
    gs = sr.GradientSampler(device, scene, n)
    gs.evaluateGradients(None, None, pos)
    before = gs.getGradientBuffer().read(np.float32).copy()
    scene.get("sphere1").position[0] = 0.3
    gs.evaluateGradients(None, None, pos)
    assert np.array_equal(gs.getGradientBuffer().read(np.float32), before)  # not yet
    gs.updateSceneParameters()
    gs.evaluateGradients(None, None, pos)
    assert np.array_equal(bits(gs.getGradientBuffer().read(np.float32)).reshape(n, 4), bits(O.sdf_gradients(scene.program(), np.zeros((n, 4), np.float32))))
    assert not np.array_equal(gs.getGradientBuffer().read(np.float32), before)
    # a structural change needs no rebuild here, only a new program
    scene.setRoot(sdf.union(scene.get("sphere1"), sdf.Torus(id="ring")))
    gs.rebuildIfNeeded()
    gs.evaluateGradients(None, None, pos)
    assert np.array_equal(bits(gs.getGradientBuffer().read(np.float32)).reshape(n, 4), bits(O.sdf_gradients(scene.program(), np.zeros((n, 4), np.float32))))
    for o in (gs, pos, out):
        o.destroy()
    with pytest.raises(sr.SplatError):
        sr.PointManager(device, sr.SDFScene())  # "Scene must have at least one primitive" (PointManager.ts:47-49)


@pytest.mark.parametrize("seeding", ["device", "host"])
def test_point_manager_seeds_on_the_scene_box(device, seeding):
    """PointManager.generateRandomPositions (src/PointManager.ts:96-189): points on the faces of the scene's scaled box, a face by
    area.  Drawn on the device (the default: splat_sdf_seed_positions, bit for bit the oracle's restatement of the same
    counter hash) or on the host (NumPy) — fresh at every reinitialize()."""
    scene = main_ts_scene()
    pm = sr.PointManager(device, scene, seed=5, seeding=seeding)
    n = pm.getNumPoints()
    mn, mx = sdf.seeding_box(scene)
    clouds = []
    for k in range(3):
        got = pm.getCurrentPositionBuffer().read(np.float32).reshape(n, 4)
        want = O.sdf_seed_positions(mn, mx, n, 5 + k) if seeding == "device" else sdf.seed_positions(scene, n, seed=5 + k)
        assert np.array_equal(bits(got), bits(want)), k
        on_face = np.isclose(got[:, :3], mn).any(axis=1) | np.isclose(got[:, :3], mx).any(axis=1)
        assert on_face.all() and (got[:, :3] >= mn - 1e-6).all() and (got[:, :3] <= mx + 1e-6).all() and (got[:, 3] == 0).all()
        clouds.append(got.copy())
        pm.reinitialize()
    assert not np.array_equal(clouds[0], clouds[1]) and not np.array_equal(clouds[1], clouds[2])
    # faces in proportion to their areas (six faces, 124k points: a percent is many sigmas)
    d = (mx - mn).astype(np.float64)
    areas = np.array([d[1] * d[2], d[1] * d[2], d[0] * d[2], d[0] * d[2], d[0] * d[1], d[0] * d[1]])
    g = clouds[0]
    counts = np.array([(g[:, 0] == mn[0]).sum(), (g[:, 0] == mx[0]).sum(), (g[:, 1] == mn[1]).sum(), (g[:, 1] == mx[1]).sum(),
                       (g[:, 2] == mn[2]).sum(), (g[:, 2] == mx[2]).sum()])
    assert np.abs(counts / n - areas / areas.sum()).max() < 0.01
    pm.destroy()


@pytest.mark.parametrize("seeding", ["device", "host"])
def test_fused_generator_equals_the_staged_calls(device, seeding):
    """splat_sdf_generate (the producer half of main.ts's frame in one launch) against the thirteen stage calls through the
    reference's classes: positions, the gradient of the last evaluation, vec4(normal, scale) and the property records, bit for
    bit, over three frames (fresh clouds, then one frame that keeps its points) with the scene animated in between."""
    scene_a, scene_b = main_ts_scene(), main_ts_scene()
    a = sr.SdfSplatSource(device, scene_a, seed=3, seeding=seeding)
    b = sr.SdfSplatSource(device, scene_b, seed=3, seeding=seeding)
    n = a.numPoints
    for frame, reinit in enumerate((True, True, False)):
        for sc in (scene_a, scene_b):
            sc.get("sphere1").position[0] = np.float32(0.2 * frame)
        pa, ca = a.step(reinitialize=reinit, fused=True)
        pb, cb = b.step(reinitialize=reinit, fused=False)
        for name, x, y, width in (("props", pa, pb, 8), ("curvature", ca, cb, 4),
                                  ("positions", a.pointManager.getCurrentPositionBuffer(), b.pointManager.getCurrentPositionBuffer(), 4),
                                  ("gradients", a.gradientSampler.getGradientBuffer(), b.gradientSampler.getGradientBuffer(), 4)):
            assert np.array_equal(bits(x.read(np.float32)).reshape(n, width), bits(y.read(np.float32)).reshape(n, width)), (frame, name)
    a.destroy()
    b.destroy()


def test_frame_from_generated_splats(device):
    """The reference's whole frame (src/main.ts:110-193) with the tile-raster path as its renderer: seeded points on the
    scene's box, five projection steps, curvature, SplatPropertyManager.updateFromCurvature, Renderer.render — against the
    oracle doing every one of those steps itself."""
    scene = main_ts_scene()
    src = sr.SdfSplatSource(device, scene, seed=11)
    n, w, h = src.numPoints, 480, 320
    assert n == 123849
    cam = sr.Camera()
    cam.setAspect(w / h)
    u = cam.uniforms(w, h)
    r = sr.Renderer(device, None, "rgba8unorm", n)
    for frame in range(2):  # the second frame: fresh points (seed 12), the first sphere moved (main.ts:114)
        scene.get("sphere1").position[0] = np.float32(0.3 * frame)
        props_buf, cur_buf = src.step()
        r.render(u, props_buf, cur_buf, None, w, h, wantFloat=True)
        prog = scene.program()
        # (the cloud is drawn on the device: point i a pure function of (seed, i); PointManager's constructor drew seed 11)
        pos = O.sdf_seed_positions(*sdf.seeding_box(scene), n, 11 + frame + 1)
        for _ in range(5):
            grad = O.sdf_gradients(prog, pos)
            pos = O.sdf_update_positions(pos, grad)
        cur = O.sdf_curvature(grad, O.sdf_scale_factors(prog, pos))
        props = O.update_props(pos, cur)
        assert np.array_equal(bits(props_buf.read(np.float32)).reshape(n, 8), bits(props))
        assert np.array_equal(bits(cur_buf.read(np.float32)).reshape(n, 4), bits(cur))
        ref = oracle_pipeline(props, cur, u, w, h)
        total = r.finish()
        assert total == ref["indices"].shape[0]
        assert np.array_equal(r.binner.getTileIndicesBuffer().read(np.uint32, total), ref["indices"])
        want, want8, _, _, near = O.composite(O.MODE_FRONT_TO_BACK, True, props[:, 4:], cur, ref["proj"], ref["indices"], ref["counts"],
                                              ref["offsets"], w, h, want_stops=True)
        d8 = np.abs(r.readPixels().astype(int) - want8.astype(int)).max(axis=2)
        assert d8[near == 0].max() <= 1 and d8.max() <= 3
        assert (want8[..., :3].astype(int).std() > 5)  # a lit surface, not a flat background
    # most points sit on the surface after five steps
    assert np.median(np.abs(O.sdf_gradients(prog, pos)[:, 0])) < 1e-3
    r.destroy()
    src.destroy()
