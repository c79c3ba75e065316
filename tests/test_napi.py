"""The reference's own host language: JS classes over the N-API shim (splat_renderer_amd/napi).

CPU part: the addon loads under Node, exports one function per ABI verb it wraps, throws a JS
Error (never falls back) without a GPU, and its Camera matches the oracle bit for bit.
GPU part: a whole frame driven stage by stage from JS equals the oracle's.
"""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import make_case, oracle_pipeline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAPI = os.path.join(ROOT, "splat_renderer_amd", "napi")
NODE = shutil.which("node")

pytestmark = pytest.mark.skipif(NODE is None or not os.path.exists("/usr/include/node/node_api.h"),
                                reason="node / N-API headers not present")


def ensure_built():
    if not os.path.exists(os.path.join(NAPI, "splat_napi.node")):
        import __graft_entry__ as g
        g.build()


def node(script, *args):
    return subprocess.run([NODE, "-e", script, *args], cwd=NAPI, capture_output=True, text=True, timeout=120)


def test_addon_loads_and_exports():
    ensure_built()
    r = node("const a=require('./splat_napi.node');console.log(JSON.stringify({abi:a.abi_version(),names:Object.keys(a)}))")
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert d["abi"] == 3
    for name in ("ctx_create", "project", "extract_keys", "sort_run", "scan_u32", "bin_run", "composite", "render_frame",
                 "update_props", "buf_upload", "buf_download", "project_slice_compact", "band_frame", "band_settle",
                 "comm_unique_id", "comm_init", "comm_destroy", "allgather_records"):
        assert name in d["names"]
    r = node("const sr=require('./index.js');console.log(JSON.stringify(Object.keys(sr)))")
    assert r.returncode == 0, r.stderr
    for cls in ("Camera", "PointManager", "SplatPropertyManager", "Renderer", "Comm", "BandRenderer"):  # north_star's API surface
        assert cls in json.loads(r.stdout)


def test_js_camera_matches_oracle():
    ensure_built()
    r = node("const sr=require('./index.js');const c=new sr.Camera();c.setAspect(16/9);"
             "console.log(JSON.stringify(Array.from(c.uniforms(1920,1080))))")
    assert r.returncode == 0, r.stderr
    u = np.array(json.loads(r.stdout), np.float32)
    vp, eye = O.camera(aspect=16 / 9)
    assert np.array_equal(u[:16].view(np.uint32), vp.view(np.uint32)) and np.array_equal(u[16:19], eye)
    assert u[20] == 1920 and u[21] == 1080


def test_js_camera_pan_rotate_zoom_match_python_camera():
    """Camera.pan (src/Camera.ts:61-83) and the other verbs from JS against splat_renderer_amd/camera.py, bit for bit:
    both restate gl-matrix's Float32Array-store semantics."""
    ensure_built()
    import splat_renderer_amd as sr
    steps = [("pan", 0.3, -0.2), ("rotate", 0.7, -0.4), ("pan", -1.25, 0.5), ("zoom", 1.5, 0), ("pan", 0.01, 2.0), ("rotate", 2.0, 1.9)]
    js = "const sr=require('./index.js');const c=new sr.Camera();c.setAspect(1.5);const out=[];" + "".join(
        f"c.{v}({a}{'' if v == 'zoom' else ',' + str(b)});out.push(Array.from(c.uniforms(300,200)).concat(Array.from(c.target)));" for v, a, b in steps) + \
        "console.log(JSON.stringify(out))"
    r = node(js)
    assert r.returncode == 0, r.stderr
    got = np.array(json.loads(r.stdout), np.float32)
    cam = sr.Camera()
    cam.setAspect(1.5)
    for k, (v, a, b) in enumerate(steps):
        getattr(cam, v)(*((a,) if v == "zoom" else (a, b)))
        want = np.concatenate([cam.uniforms(300, 200), cam.target]).astype(np.float32)
        assert np.array_equal(got[k].view(np.uint32), want.view(np.uint32)), (k, v)


def test_js_sdf_scene_encodes_like_the_python_host():
    """The SDF scene graph from JS (src/sdf/Scene.ts builders): same structure hash and the same postfix program as
    splat_renderer_amd/sdf.py, which the GPU tests hold against the oracle."""
    ensure_built()
    from tests.test_sdf_cpu import main_ts_scene
    r = node("const sr=require('./index.js');const s=new sr.SDFScene();"
             "const a=new sr.Sphere({id:'sphere1',radius:0.5}),b=new sr.Box({id:'box1',position:[0.6,0,0],size:[0.3,0.3,0.3]}),"
             "c=new sr.Sphere({id:'sphere2',position:[0,0.6,0],radius:0.25});"
             "s.setRoot(sr.smoothUnion(0.1,sr.smoothUnion(0.15,a,b),c));"
             "console.log(JSON.stringify({hash:s.getStructureHash(),prog:Array.from(s.program()),ops:s.getOperations().map(o=>o.k),"
             "names:s.get('box1').getParamNames(),area:new sr.Capsule().getSurfaceArea()}))")
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    scene = main_ts_scene()
    assert d["hash"] == scene.getStructureHash() and d["ops"] == [0.1, 0.15] and d["names"] == ["box1_center", "box1_size"]
    want = np.zeros((5, 8), np.float32)
    for k, (op, a) in enumerate(scene.program()):
        want[k, 0] = op
        want[k, 1:1 + len(a)] = a
    assert np.array_equal(np.array(d["prog"], np.float32).reshape(5, 8), want)
    assert abs(d["area"] - (2 * np.pi * 0.3 + 4 * np.pi * 0.09)) < 1e-12


def test_js_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ensure_built()
    r = node("const sr=require('./index.js');try{new sr.Device(0);console.log('NO THROW')}catch(e){console.log('threw '+e.message)}")
    assert r.returncode == 0 and r.stdout.startswith("threw libsplat_hip -6"), r.stdout + r.stderr


@pytest.mark.gpu
def test_js_frame_matches_oracle(tmp_path):
    ensure_built()
    n, w, h = 4000, 208, 120
    props, normals, u = make_case(n, w, h, 23, 1.5)
    ref = oracle_pipeline(props, normals, u, w, h)
    _, want8, _ = O.composite(O.MODE_FRONT_TO_BACK, True, props[:, 4:], normals, ref["proj"], ref["indices"], ref["counts"],
                              ref["offsets"], w, h)
    props.tofile(tmp_path / "props.f32")
    normals.tofile(tmp_path / "normals.f32")
    r = subprocess.run([NODE, "render_frame.js", str(tmp_path / "props.f32"), str(tmp_path / "normals.f32"), str(n), str(w),
                        str(h), str(tmp_path / "out.rgba8"), str(tmp_path / "order.u32"), str(tmp_path / "counts.u32"),
                        str(tmp_path / "indices.u32"), str(tmp_path / "frame.rgba8"), str(tmp_path / "seq.rgba8"),
                        str(tmp_path / "discframe.rgba8")], cwd=NAPI, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert np.array_equal(np.array(info["uniforms"], np.float32).view(np.uint32), u.view(np.uint32))
    assert info["pairs"] == ref["indices"].shape[0]
    assert np.array_equal(np.fromfile(tmp_path / "order.u32", np.uint32), ref["order"])
    assert np.array_equal(np.fromfile(tmp_path / "counts.u32", np.uint32), ref["counts"])
    assert np.array_equal(np.fromfile(tmp_path / "indices.u32", np.uint32), ref["indices"])
    got8 = np.fromfile(tmp_path / "out.rgba8", np.uint8).reshape(h, w, 4)
    assert np.abs(got8.astype(int) - want8.astype(int)).max() <= 3
    assert (np.abs(got8.astype(int) - want8.astype(int)).max(axis=2) > 1).mean() <= 2e-3
    # TileRenderer.render from JS with the reference's eleven arguments and no bindTileData (src/TileRenderer.ts:234-246)
    assert info["tileRendererEqualsStaged"] is True
    # the whole-frame facade from JS: Renderer.render on the two-plane property layout, tile-first order
    assert info["framePairs"] == ref["indices"].shape[0]
    frame8 = np.fromfile(tmp_path / "frame.rgba8", np.uint8).reshape(h, w, 4)
    assert np.array_equal(frame8, got8)  # same lists, same composite: same bytes as the staged path
    assert info["recordFormat"] == 3  # the facade's projector wrote lit composite records (SPLAT_RECORDS_LIT32)
    # Device.compositeOptions / forgetCompositeHistory / setTiming / stageTimeStats / timingConsumed from JS: another schedule
    # of k_composite_px, the same bytes; the other kernel within 1 LSB; one composite launch timed; its consumed entries
    # within its staged ones within the lists
    assert info["schedulesKeepTheBytes"] is True and info["kernelsWithinOneLsb"] is True
    t = info["timed"]
    assert t["samples"] == 1 and 0.0 < t["totalMs"] < 50.0
    assert 0 < t["consumed"] <= t["staged"] <= ref["indices"].shape[0] + 32 * ref["counts"].shape[0]
    # north_star's multi-GPU frame from JS (one-rank RCCL communicator behind the C ABI) and PointManager
    assert info["bandEqualsFrame"] is True and info["bandPairs"] == ref["indices"].shape[0]
    assert info["pointManagerOk"] is True
    # Device.rankStatus() from JS: no frame of this script was re-rendered after a failed order check (splat_rank_status)
    assert info["ranking"] == {"policy": "checked", "atomicsOrdered": True, "orderFaults": 0} or os.environ.get("SPLAT_RANK")
    # SequentialRenderer from JS draws its own footprint (the oriented disc): against the oracle's per-pixel restatement
    # (early-out on, as the class renders) and, off the rims, against the software rasteriser of SequentialRenderer.ts
    dproj, discs = O.project_disc(u, props, normals)
    dkeys, dpay = O.extract_keys(dproj)
    _, dorder = O.sort_pairs(dkeys, dpay)
    dcounts, doffsets, didx = O.bin_sorted(dproj, dorder, w, h)
    _, dwant8, _, rim = O.composite_disc(True, props[:, 4:], normals, discs, didx, dcounts, doffsets, w, h)
    _, raster8 = O.sequential(u, props, normals, dorder[::-1].copy(), w, h)
    assert info["seqPairs"] == didx.shape[0] and info["discFramePairs"] == didx.shape[0]
    seq8 = np.fromfile(tmp_path / "seq.rgba8", np.uint8).reshape(h, w, 4)
    d = np.abs(seq8.astype(int) - dwant8.astype(int)).max(axis=2)
    assert d[rim == 0].max() <= 3 and d.max() <= 12 and (d > 1).mean() <= 2e-3
    assert np.abs(seq8.astype(int) - raster8.astype(int)).max(axis=2)[rim == 0].max() <= 3  # early-out: (1 - 0.99) * 255
    dframe8 = np.fromfile(tmp_path / "discframe.rgba8", np.uint8).reshape(h, w, 4)
    assert np.array_equal(dframe8, seq8)  # same lists, same composite


@pytest.mark.gpu
def test_js_sdf_generation_matches_oracle(tmp_path):
    """GradientSampler / PositionUpdater / CurvatureSampler / PointManager driven from JS (src/main.ts:146-180): bit-exact
    against the oracle's five projection steps, scale factors and vec4(normal, scale)."""
    ensure_built()
    from splat_renderer_amd import sdf
    from tests.test_sdf_cpu import main_ts_scene
    scene = main_ts_scene()
    n = 5000
    pos = sdf.seed_positions(scene, n, seed=21)
    pos.tofile(tmp_path / "pos.f32")
    r = subprocess.run([NODE, "sdf_generate.js", str(tmp_path / "pos.f32"), str(n), str(tmp_path / "out.f32"), str(tmp_path / "seeded.f32")],
                       cwd=NAPI, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["hash"] == scene.getStructureHash()
    assert info["fusedEqual"] is True  # native.sdf_generate from JS: the staged calls' bits
    # new PointManager(device, scene, seed) in JS: the reference's point count, clouds drawn on the device = the oracle's
    ns = sdf.point_count(scene)
    assert info["seededCount"] == ns
    clouds = np.fromfile(tmp_path / "seeded.f32", np.float32).reshape(2, ns, 4)
    for k in range(2):
        assert np.array_equal(clouds[k].view(np.uint32), O.sdf_seed_positions(*sdf.seeding_box(scene), ns, 33 + k).view(np.uint32)), k
    got = np.fromfile(tmp_path / "out.f32", np.float32).reshape(3, n, 4)
    scene.get("sphere1").position[0] = np.float32(0.1)
    prog = scene.program()
    for _ in range(5):
        grad = O.sdf_gradients(prog, pos)
        pos = O.sdf_update_positions(pos, grad)
    cur = O.sdf_curvature(grad, O.sdf_scale_factors(prog, pos))
    for k, want in enumerate((pos, grad, cur)):
        assert np.array_equal(got[k].view(np.uint32), want.view(np.uint32)), k


@pytest.mark.gpu
def test_js_frame_loop_matches_the_python_one(tmp_path):
    """FrameLoop + OrbitCameraController in the reference's own language (napi/index.js; the event
    mapping of src/OrbitCameraController.ts:18-70, the loop of src/main.ts:110-193).  Four frames of an orbit driven from
    Node — rotate, pan, wheel, left-button drag — against the same moves through splat_renderer_amd/frameloop.py: the
    same uniform blocks bit for bit, the same pair totals, the same images byte for byte."""
    ensure_built()
    import splat_renderer_amd as sr
    n, w, h = 12000, 320, 208
    props, normals, _ = make_case(n, w, h, 61, 1.5)
    props.tofile(tmp_path / "props.f32")
    normals.tofile(tmp_path / "normals.f32")
    r = subprocess.run([NODE, "frame_loop.js", str(tmp_path / "props.f32"), str(tmp_path / "normals.f32"), str(n), str(w), str(h),
                        str(tmp_path / "js_")], cwd=NAPI, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["refused"] is True and info["recordFormat"] == 3
    dev = sr.Device(0)
    try:
        pbuf, nbuf = dev.createBufferFrom(props), dev.createBufferFrom(normals)
        loop = sr.FrameLoop(dev, n, w, h)
        ctl = sr.OrbitCameraController(loop.camera)
        moves = [lambda: loop.camera.rotate(2 * np.pi / 4, 0.0), lambda: loop.camera.pan(0.2, -0.1),
                 lambda: ctl.onWheel(sr.MouseEvent(deltaY=400.0)),
                 lambda: (ctl.onMouseDown(sr.MouseEvent(10, 10, button=0)), ctl.onMouseMove(sr.MouseEvent(70, 40)), ctl.onMouseUp())]
        last = None
        for k, move in enumerate(moves):
            u = loop.camera.uniforms(w, h, time=k / 60.0)
            assert np.array_equal(np.array(info["uniforms"][k], np.float32).view(np.uint32), u.view(np.uint32)), k
            loop.render(pbuf, nbuf)
            want = loop.readPixels()
            assert loop.renderer.finish() == info["pairs"][k], k
            got = np.fromfile(tmp_path / f"js_{k}.rgba8", np.uint8).reshape(h, w, 4)
            assert np.array_equal(got, want), k
            last = want
            move()
        assert np.array_equal(np.fromfile(tmp_path / "js_sync_free_last.rgba8", np.uint8).reshape(h, w, 4), last)
        for o in (loop, pbuf, nbuf):
            o.destroy()
    finally:
        dev.destroy()


def test_js_host_formulas_equal_the_reference_own_code():
    """napi/index.js against tests/golden/ref_host.json (the reference's own statements executed under Node): surface areas,
    scaleAABB, PointManager's point count, SplatPropertyManager's defaults, and the order of SDFScene.program()."""
    ensure_built()
    here = os.path.join(ROOT, "tests", "golden", "ref_host.json")
    script = r"""
const sr = require('./index.js');
const ref = JSON.parse(require('fs').readFileSync(process.argv[1], 'utf8'));
const inp = ref.inputs, out = ref.outputs;
const prim = (d) => new ({ sphere: sr.Sphere, box: sr.Box, torus: sr.Torus, capsule: sr.Capsule }[d.prim])(d);
const node = (d) => {
  if (d.prim) return prim(d);
  const a = node(d.children[0]), b = node(d.children[1]);
  if (d.op === 'smooth_union') return sr.smoothUnion(d.k, a, b);
  return { union: sr.union, intersection: sr.intersection, subtraction: sr.subtraction }[d.op](a, b);
};
const bad = [];
inp.prims.forEach((d, i) => { if (prim(d).getSurfaceArea() !== out.areas[i]) bad.push(['area', i]); });
inp.boxes.forEach((c, i) => {
  const r = sr.scaleAABB({ min: c.min, max: c.max }, c.scale);
  if (JSON.stringify([r.min, r.max]) !== JSON.stringify([out.scaleAABB[i].min, out.scaleAABB[i].max])) bad.push(['scaleAABB', i]);
});
const ops = {};
for (const name of Object.keys(inp.scenes)) {
  const sc = new sr.SDFScene();
  sc.setRoot(node(inp.scenes[name]));
  if (sr.PointManager.calculatePointCount(sc) !== out.scenes[name].pointCount) bad.push(['pointCount', name]);
  const prog = sc.program();
  ops[name] = [];
  for (let k = 0; k < prog.length; k += 8) ops[name].push(prog[k]);
}
if (sr.PointManager.calculatePointCount(new sr.SDFScene()) !== out.pointCountEmptyScene) bad.push(['pointCount', 'empty']);
if (JSON.stringify(Array.from(sr.SplatPropertyManager.defaultProperties(3))) !== JSON.stringify(Array.from(Float32Array.from(out.defaults)))) bad.push(['defaults']);
console.log(JSON.stringify({ bad, ops }));
"""
    r = subprocess.run([NODE, "-e", script, here], cwd=NAPI, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["bad"] == [], d["bad"]
    from tests.test_sdf_cpu import _emitted, _ref_host
    code = {"sphere": 0, "box": 1, "torus": 2, "capsule": 3, "union": 16, "intersection": 17, "subtraction": 18, "smooth_union": 19}
    ref = _ref_host()
    for name, ops in d["ops"].items():
        assert ops == [code[k] for k, _ in _emitted(ref["outputs"]["scenes"][name]["wgsl"])], name


def test_index_js_is_the_stripped_twin_of_index_ts():
    """The host is authored as TypeScript with erasable syntax only (napi/index.ts: `declare` fields, `type` / `interface`
    statements, annotations in function and member headers — SURVEY §7 step 2); the file Node loads is its mechanical twin.
    The committed index.js must be exactly `node strip_types.js index.ts`, and the stripped text must be free of the
    syntax the stripper claims to erase."""
    r = subprocess.run([NODE, "strip_types.js", "index.ts"], cwd=NAPI, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    with open(os.path.join(NAPI, "index.js")) as f:
        committed = f.read()
    assert r.stdout == committed, "index.js is stale: run `node strip_types.js index.ts > index.js` in splat_renderer_amd/napi"
    ts = open(os.path.join(NAPI, "index.ts")).read()
    assert ts.count("\n  declare ") > 100 and "interface PropertyPlanes" in ts and "): Buffer_ {" in ts  # (it IS annotated)
    for line in committed.split("\n"):
        assert not line.strip().startswith(("declare ", "interface ", "type ")), line
    # every annotated header of index.ts appears in index.js with the same name and arity
    import re
    heads_ts = re.findall(r"^  (?:static |get |set )?(\w+)(?:<[^>]*>)?\(", ts, flags=re.M)
    heads_js = re.findall(r"^  (?:static |get |set )?(\w+)\(", committed, flags=re.M)
    assert heads_ts == heads_js


def test_index_d_ts_declares_what_index_ts_defines():
    """napi/index.d.ts (what a TypeScript caller compiles against) is written by hand beside index.ts (no tsc in this image to emit
    it): every exported class of index.ts is declared there, every method the declaration promises exists in index.ts with the
    same number of parameters (optional ones included), and TileRenderer.render carries the reference's eleven arguments."""
    import re
    ts = open(os.path.join(NAPI, "index.ts")).read()
    dts = open(os.path.join(NAPI, "index.d.ts")).read()

    def split_top(text):
        out, depth, cur = [], 0, ""
        for k, ch in enumerate(text):
            if ch in "([{<":
                depth += 1
            elif ch in ")]}" or (ch == ">" and text[k - 1] != "="):
                depth -= 1
            if ch == "," and depth == 0:
                out.append(cur)
                cur = ""
            else:
                cur += ch
        return [x for x in out + [cur] if x.strip()]

    def classes(text, decl):
        found = {}
        for m in re.finditer(r"^(?:export )?(?:declare )?class (\w+)(?: extends \w+)? \{[^\n]*\n(.*?)^\}", text, flags=re.M | re.S):
            methods = {}
            for line in m.group(2).split("\n"):
                h = re.match(r"^  (?:static |async )*(\w+)(?:<[^>]*>)?\((.*)\)(?::[^{;]*)?[;{]", line) if decl else \
                    re.match(r"^  (?:static |async )*(\w+)(?:<[^>]*>)?\((.*?)\)(?:: [^{]*)? \{", line)
                if h and h.group(1) not in ("if", "for", "while", "switch", "return"):
                    methods[h.group(1)] = len(split_top(h.group(2)))
            found[m.group(1)] = methods
        return found

    one_liners = r"^export (?:declare )?class \w+(?: extends \w+)? \{.*\}[ \t]*$"  # (a whole declaration on one line: fields only)
    defined, declared = classes(ts, False), classes(re.sub(one_liners, "", dts, flags=re.M), True)
    exported = set(re.search(r"module\.exports = \{(.*?)\};", ts, flags=re.S).group(1).replace("\n", " ").replace(" ", "").split(","))
    exported_classes = {c for c in defined if c in exported or c.rstrip("_") in exported}
    declared_names = set(re.findall(r"export (?:declare )?class (\w+)", dts))  # (one-line declarations too)
    assert {c.rstrip("_") for c in exported_classes} <= declared_names, sorted({c.rstrip("_") for c in exported_classes} - declared_names)
    checked = 0
    for cls, methods in declared.items():
        impl = defined.get(cls) or defined.get(cls + "_")
        assert impl is not None, f"index.d.ts declares class {cls}, index.ts has none"
        for name, arity in methods.items():
            owner = impl
            if name not in owner:  # inherited (TileRenderer extends ComputeShaderRenderer)
                base = re.search(rf"class {cls}_? extends (\w+)", ts)
                owner = defined.get(base.group(1), {}) if base else {}
            assert name in owner, f"index.d.ts: {cls}.{name} is not defined in index.ts"
            assert owner[name] == arity, f"{cls}.{name}: {arity} parameters declared, {owner[name]} defined"
            checked += 1
    assert checked > 80
    assert declared["TileRenderer"]["render"] == 11 and defined["TileRenderer"]["render"] == 11  # src/TileRenderer.ts:234-246


def test_index_ts_field_annotations_agree_with_what_is_assigned():
    """No TypeScript compiler is in this image (`declare bound: boolean` held an array, `last: number` a tuple).  A
    check of the kind of error that slips through without one: for every `declare field: T;` of a class of napi/index.ts, every
    `this.field = <literal>` in that class — an array literal, true / false, null, a number, a string, `new X(...)` — must be
    admissible for T; and every field a class assigns is declared (by it or a base class)."""
    import re
    ts = open(os.path.join(NAPI, "index.ts")).read()
    bodies = {m.group(1): (m.group(2), m.group(3)) for m in
              re.finditer(r"^class (\w+)(?: extends (\w+))? \{[^\n]*\n(.*?)^\}", ts, flags=re.M | re.S)}
    problems, checked = [], 0
    for cls, (base, body) in bodies.items():
        fields = dict(re.findall(r"^  declare (\w+): (.*?);(?: //.*)?$", body, flags=re.M))
        for name, t in fields.items():
            if re.search(r"\b(any|unknown)\b", t):
                continue
            for a in re.finditer(r"this\.%s = ([^;]+);" % re.escape(name), body):
                rhs, ok = a.group(1).strip(), None
                if rhs.startswith("["):
                    ok = "[]" in t or "Array" in t or t.lstrip().startswith("[")
                elif rhs in ("true", "false"):
                    ok = "boolean" in t
                elif rhs == "null":
                    ok = "null" in t
                elif re.fullmatch(r"-?\d+(\.\d+)?(e-?\d+)?", rhs):
                    ok = "number" in t
                elif re.fullmatch(r"'[^']*'|\"[^\"]*\"", rhs):
                    ok = "string" in t or rhs.strip("'\"") in t
                elif rhs.startswith("new "):
                    c = re.match(r"new (\w+)", rhs).group(1)
                    ok = c in t or c.rstrip("_") in t
                if ok is not None:
                    checked += 1
                    if not ok:
                        problems.append(f"{cls}.{name}: declared `{t}`, assigned `{rhs[:50]}`")
        declared, b = set(fields), base
        while b in bodies:
            declared |= set(re.findall(r"^  declare (\w+):", bodies[b][1], flags=re.M))
            b = bodies[b][0]
        for u in sorted(set(re.findall(r"this\.(\w+) = ", body)) - declared):
            problems.append(f"{cls}.{u}: assigned but not declared")
    assert not problems, "\n".join(problems)
    assert checked > 60
