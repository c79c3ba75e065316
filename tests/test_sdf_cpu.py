"""SDF splat generation, CPU side: the scene graph classes mirror src/sdf/*.ts, the oracle's
restatement of the generated sceneSDF gives the analytic answers, and five projection steps land points on the surface."""
import math

import numpy as np

import splat_renderer_amd as sr
from oracle import oracle as O
from splat_renderer_amd import sdf


def main_ts_scene():
    """The scene src/main.ts:59-85 builds."""
    s = sr.SDFScene()
    a = sdf.Sphere(id="sphere1", position=(0, 0, 0), radius=0.5)
    b = sdf.Box(id="box1", position=(0.6, 0, 0), size=(0.3, 0.3, 0.3))
    c = sdf.Sphere(id="sphere2", position=(0, 0.6, 0), radius=0.25)
    s.setRoot(sdf.smoothUnion(0.1, sdf.smoothUnion(0.15, a, b), c))
    return s


def test_scene_graph_mirrors_the_reference_classes():
    s = main_ts_scene()
    assert s.getStructureHash() == "O:smooth_union:(O:smooth_union:(P:sphere:sphere1,P:box:box1),P:sphere:sphere2)"  # Scene.ts:139-150
    assert [p.id for p in s.getPrimitives()] == ["sphere1", "box1", "sphere2"]
    assert [o.k for o in s.getOperations()] == [0.1, 0.15]  # pre-order (Scene.ts:121-133)
    assert s.get("box1").getParamNames() == ["box1_center", "box1_size"]
    assert s.get("box1").getParamValues() == [np.float32(0.6), 0.0, 0.0, 0, np.float32(0.3), np.float32(0.3), np.float32(0.3), 0]
    # postfix: children first, then the operation (CodeGenerator.ts:291-346 emits result_0..result_4 in this order)
    assert [op for op, _ in s.program()] == [0, 1, 19, 0, 19]
    assert sdf.Sphere().radius == 0.5 and sdf.Torus().majorRadius == 0.5 and sdf.Torus().minorRadius == 0.2
    assert sdf.Capsule().height == 1.0 and sdf.Capsule().radius == 0.3 and np.array_equal(sdf.Box().size, [0.5, 0.5, 0.5])
    assert sdf.Sphere().id.startswith("prim_") and sdf.SmoothUnion().id.startswith("smin_")
    assert sdf.SmoothUnion(0.2).getParamNames()[0].endswith("_k")
    # PointManager.calculatePointCount (:22-39): floor(30000 sqrt(area)) per primitive, clamped to [10000, 200000]
    want = sum(math.floor(30000 * math.sqrt(a)) for a in (4 * math.pi * 0.25, 2 * 3 * 0.36, 4 * math.pi * 0.0625))
    assert sdf.point_count(s) == want == 123849
    assert sdf.point_count(sr.SDFScene()) == 50000
    one = sr.SDFScene()
    one.setRoot(sdf.Sphere(radius=0.01))
    assert sdf.point_count(one) == 10000
    assert math.isclose(sdf.Torus().getSurfaceArea(), 4 * math.pi ** 2 * 0.1) and math.isclose(
        sdf.Capsule().getSurfaceArea(), 2 * math.pi * 0.3 + 4 * math.pi * 0.09)


def test_seeding_covers_the_faces_of_the_scaled_global_box():
    s = main_ts_scene()
    p = sdf.seed_positions(s, 20000, seed=3)
    assert p.shape == (20000, 4) and (p[:, 3] == 0).all()
    assert np.array_equal(p, sdf.seed_positions(s, 20000, seed=3)) and not np.array_equal(p, sdf.seed_positions(s, 20000, seed=4))
    # global AABB of the three primitives = [-0.5, 0.9] x [-0.5, 0.85] x [-0.5, 0.5], scaled 1.5x about min + max / 2
    # (Primitive.ts:283-290 as written)
    mn, mx = sdf.scaleAABB((np.array([-0.5, -0.5, -0.5]), np.array([0.9, 0.85, 0.5])), 1.5)
    on_face = np.zeros(20000, bool)
    for ax in range(3):
        on_face |= np.isclose(p[:, ax], mn[ax]) | np.isclose(p[:, ax], mx[ax])
        assert (p[:, ax] >= mn[ax] - 1e-6).all() and (p[:, ax] <= mx[ax] + 1e-6).all()
    assert on_face.all()
    # every face gets points, the larger ones more
    assert all((np.isclose(p[:, ax], v)).sum() > 1000 for ax in range(3) for v in (mn[ax], mx[ax]))


def test_oracle_primitives_have_the_analytic_distances_and_gradients():
    pts = np.array([[1, 0, 0, 0], [0, 0, 0.2, 0], [0.7, 0.1, 0, 0], [0.1, 0.05, 0.02, 0]], np.float32)
    g = O.sdf_gradients([(0, [0, 0, 0, 0.5])], pts)  # sphere
    assert np.allclose(g[0], [0.5, 1, 0, 0]) and np.allclose(g[1], [-0.3, 0, 0, 1])
    g = O.sdf_gradients([(1, [0, 0, 0, 0.3, 0.3, 0.3])], pts)  # box: outside along x; inside nearest face x
    assert np.allclose(g[0], [0.7, 1, 0, 0]) and np.allclose(g[3], [-0.2, 1, 0, 0])
    g = O.sdf_gradients([(2, [0, 0, 0, 0.5, 0.2])], pts)  # torus (major 0.5, minor 0.2)
    assert np.allclose(g[0], [0.3, 1, 0, 0]) and np.isclose(g[2, 0], math.hypot(0.2, 0.1) - 0.2)
    g = O.sdf_gradients([(3, [0, 0, 0, 1.0, 0.3])], np.array([[0.5, 0.2, 0, 0], [0, 1.0, 0, 0]], np.float32))  # capsule
    assert np.allclose(g[0], [0.2, 1, 0, 0]) and np.allclose(g[1], [0.2, 0, 1, 0])
    a, b = (0, [0, 0, 0, 0.5]), (0, [0.6, 0, 0, 0.5])
    q = np.array([[0.3, 0.0, 0, 0], [-0.2, 0.1, 0, 0]], np.float32)
    ga, gb = O.sdf_gradients([a], q), O.sdf_gradients([b], q)
    assert np.array_equal(O.sdf_gradients([a, b, (16, [])], q), np.where((ga[:, :1] < gb[:, :1]), ga, gb))  # union
    assert np.array_equal(O.sdf_gradients([a, b, (17, [])], q), np.where((ga[:, :1] > gb[:, :1]), ga, gb))  # intersection
    assert np.array_equal(O.sdf_gradients([a, b, (18, [])], q), np.where((ga[:, :1] > -gb[:, :1]), ga, -gb))  # subtraction
    sm = O.sdf_gradients([a, b, (19, [0.1])], q)
    assert (sm[:, 0] <= np.minimum(ga[:, 0], gb[:, 0]) + 1e-7).all()  # a smooth union never lies outside the union
    assert np.array_equal(O.sdf_gradients([], q), np.tile(np.float32([1000, 0, 1, 0]), (2, 1)))  # empty scene


def test_oracle_points_settle_on_the_surface_and_curvature_marks_edges():
    s = main_ts_scene()
    prog = s.program()
    p = sdf.seed_positions(s, 4000, seed=1)
    for _ in range(5):  # src/main.ts:149-172
        p = O.sdf_update_positions(p, O.sdf_gradients(prog, p))
    d = O.sdf_gradients(prog, p)[:, 0]
    assert np.abs(d).max() < 0.05 and np.median(np.abs(d)) < 1e-3
    sf = O.sdf_scale_factors(prog, p)
    assert sf.min() >= 0.01 - 1e-7 and sf.max() <= 1.0 + 1e-7
    box = sr.SDFScene()
    box.setRoot(sdf.Box(position=(0, 0, 0), size=(0.4, 0.4, 0.4)))
    face = np.array([[0.4, 0.0, 0.0, 0], [0.4, 0.39, 0.39, 0]], np.float32)  # the middle of a face; next to a corner
    sfb = O.sdf_scale_factors(box.program(), face)
    assert sfb[0] > 0.999 and sfb[1] < 0.9
    cur = O.sdf_curvature(O.sdf_gradients(box.program(), face), sfb)
    assert np.allclose(cur[0], [1, 0, 0, sfb[0]])


def test_oracle_seeding_is_a_pure_function_of_seed_and_index():
    """orc_sdf_seed_positions (the restatement of splat_sdf_seed_positions): point i of cloud s does not depend on n, clouds
    differ, every point lies on a face of the box and the uniforms fill the faces."""
    mn, mx = np.array([-1.0, -0.5, -2.0], np.float32), np.array([1.5, 0.75, 0.25], np.float32)
    a, b = O.sdf_seed_positions(mn, mx, 5000, 7), O.sdf_seed_positions(mn, mx, 20000, 7)
    assert np.array_equal(a, b[:5000]) and not np.array_equal(a, O.sdf_seed_positions(mn, mx, 5000, 8))
    on_face = (b[:, :3] == mn).any(axis=1) | (b[:, :3] == mx).any(axis=1)
    assert on_face.all() and (b[:, :3] >= mn).all() and (b[:, :3] <= mx).all() and (b[:, 3] == 0).all()
    free = b[(b[:, 0] == mn[0])][:, 1:3]  # the -x face: y and z uniform over the box
    lo, hi = np.array([mn[1], mn[2]]), np.array([mx[1], mx[2]])
    assert np.abs(free.mean(axis=0) - (lo + hi) / 2).max() < 0.05 * (hi - lo).max()


# ---- pinned to an execution of the reference's own statements (tests/golden/ref_host.json, made by
# tests/golden/make_ref_host_fixtures.py: getSurfaceArea, calculatePointCount, scaleAABB, initializeDefaults, the scene walk
# of WGSLCodeGenerator.generateSceneSDF — extracted as text and run under Node) -------------------------------------------
def _ref_host():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "ref_host.json")) as f:
        return json.load(f)


def _prim_from(d):
    kw = {k: v for k, v in d.items() if k not in ("prim",)}
    return {"sphere": sdf.Sphere, "box": sdf.Box, "torus": sdf.Torus, "capsule": sdf.Capsule}[d["prim"]](**kw)


def _node_from(d):
    if "prim" in d:
        return _prim_from(d)
    a, b = (_node_from(c) for c in d["children"])
    if d["op"] == "smooth_union":
        return sdf.smoothUnion(d["k"], a, b)
    return {"union": sdf.union, "intersection": sdf.intersection, "subtraction": sdf.subtraction}[d["op"]](a, b)


def _emitted(wgsl):
    """[(kind, id or None)] of the `let result_k = ...` lines, in emission order."""
    import re
    out = []
    for ln in wgsl:
        m = re.match(r"\s*let result_\d+ = (sdg|op)(\w+)\((.*)\);", ln)
        if not m:
            continue
        kind = {"Sphere": "sphere", "Box": "box", "Torus": "torus", "Capsule": "capsule", "Union": "union", "Intersection": "intersection",
                "Subtraction": "subtraction", "SmoothUnion": "smooth_union"}[m.group(2)]
        pid = re.search(r"sceneParams\.(\w+)_center", m.group(3))
        out.append((kind, pid.group(1) if (pid and m.group(1) == "sdg") else None))
    return out


def test_host_side_formulas_equal_the_reference_own_code():
    ref = _ref_host()
    inp, out = ref["inputs"], ref["outputs"]
    # the four getSurfaceArea() bodies (src/sdf/Primitive.ts), bit for bit (doubles)
    for d, want in zip(inp["prims"], out["areas"]):
        assert _prim_from(d).getSurfaceArea() == want, d
    # scaleAABB as written (centre = min + max / 2)
    for c, want in zip(inp["boxes"], out["scaleAABB"]):
        mn, mx = sdf.scaleAABB((np.array(c["min"], np.float64), np.array(c["max"], np.float64)), c["scale"])
        assert mn.tolist() == want["min"] and mx.tolist() == want["max"], c
    # PointManager.calculatePointCount incl. both clamps and the empty scene
    for name, d in inp["scenes"].items():
        sc = sr.SDFScene()
        sc.setRoot(_node_from(d))
        assert sdf.point_count(sc) == out["scenes"][name]["pointCount"], name
    assert sdf.point_count(sr.SDFScene()) == out["pointCountEmptyScene"] == 50000
    # SplatPropertyManager.initializeDefaults
    from splat_renderer_amd.host import default_properties
    assert np.array_equal(default_properties(3).reshape(-1), np.array(out["defaults"], np.float32))


def test_scene_program_order_equals_the_reference_code_generator():
    """The postfix program the kernels evaluate lists the nodes in the order WGSLCodeGenerator.generateSceneSDF's traverse()
    emits its `let result_k` lines (src/sdf/CodeGenerator.ts:289-351, executed under Node for the six test scenes)."""
    from splat_renderer_amd import _lib
    ref = _ref_host()
    code = {"sphere": _lib.SDF_SPHERE, "box": _lib.SDF_BOX, "torus": _lib.SDF_TORUS, "capsule": _lib.SDF_CAPSULE, "union": _lib.SDF_UNION,
            "intersection": _lib.SDF_INTERSECTION, "subtraction": _lib.SDF_SUBTRACTION, "smooth_union": _lib.SDF_SMOOTH_UNION}
    for name, d in ref["inputs"]["scenes"].items():
        sc = sr.SDFScene()
        sc.setRoot(_node_from(d))
        want = _emitted(ref["outputs"]["scenes"][name]["wgsl"])
        assert [op for op, _ in sc.program()] == [code[k] for k, _ in want], name
        # the primitives appear in the walk in the order getPrimitives() lists them (Scene.ts: insertion order of the same walk)
        assert [p.id for p in sc.getPrimitives()] == [i for _, i in want if i is not None], name
        assert len(want) >= 1 and ref["outputs"]["scenes"][name]["wgsl"][-2].strip() == f"return result_{len(want) - 1};"
