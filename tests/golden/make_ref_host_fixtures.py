#!/usr/bin/env python3
"""Generates tests/golden/ref_host.json — OUTPUTS OF THE REFERENCE'S OWN HOST-SIDE CODE.

The method of make_ref_fixtures.py, applied to the remaining pure-CPU statements of the reference: they are read from
/root/reference as TEXT, TypeScript-only syntax is stripped IN MEMORY, and they are executed under the container's Node
(the program goes to `node -e`, the inputs on stdin; nothing of either is written to disk).  Only resulting numbers and
strings are stored, next to the inputs that produced them:

  * the four `getSurfaceArea()` bodies                         src/sdf/Primitive.ts:106-108, 159-164, 217-219, 272-278
  * `PointManager.calculatePointCount`                         src/PointManager.ts:22-39 (with those bodies)
  * `scaleAABB` as written                                     src/sdf/Primitive.ts:281-290
  * `SplatPropertyManager.initializeDefaults`' fill loop       src/SplatPropertyManager.ts:35-50
  * the emission order of `WGSLCodeGenerator.generateSceneSDF` src/sdf/CodeGenerator.ts:289-351, for the six scenes of
    tests/test_gpu_sdf.py (rebuilt here from plain descriptions that are stored with the outputs)

gl-matrix is not in the container: scaleAABB's two calls into it, vec3.scaleAndAdd(out, a, b, s) =
a + b * s and vec3.sub(out, a, b) = a - b, get those documented definitions — stated here because that part is then a
restatement, not an execution.  (scaleAABB passes plain arrays as `out`, so its arithmetic is in doubles.)

tests/test_sdf_cpu.py holds splat_renderer_amd/sdf.py, the C oracle and napi/index.js to these numbers.
Run from the repo root:  python tests/golden/make_ref_host_fixtures.py
"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"


def strip_types(js):
    js = re.sub(r"(\b(?:let|const|var)\s+\w+)\s*:\s*[\w<>\[\]| ]+(?=\s*=)", r"\1", js)  # let x: T = ...
    js = re.sub(r"\((\w+)\s*:\s*[\w<>\[\]| ]+\)\s*(?::\s*[\w<>\[\]| ]+)?\s*=>", r"(\1) =>", js)  # (node: SceneNode): string =>
    js = re.sub(r"(\w|\))!(?=[.\[,;)\s])", r"\1", js)
    js = re.sub(r"\s+as\s+\w+(\[\])?", "", js)
    return js


def between(text, first, last, include_last=False, start=0):
    a = text.index(first, start)
    b = text.index(last, a + len(first))
    return text[a:b + (len(last) if include_last else 0)]


def method_body(text, class_name, signature):
    """the statements between the braces of `signature` inside `class <class_name>`"""
    c = text.index(f"class {class_name} ")
    a = text.index(signature, c)
    a = text.index("{", a) + 1
    depth, i = 1, a
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[a:i - 1]


def reference_statements():
    prim = open(os.path.join(REF, "sdf", "Primitive.ts")).read()
    areas = {k: strip_types(method_body(prim, k, "getSurfaceArea(): number")) for k in ("Sphere", "Box", "Torus", "Capsule")}
    scale = strip_types(between(prim, "const center = vec3.scaleAndAdd(", "};", include_last=True, start=prim.index("export function scaleAABB")))
    pm = open(os.path.join(REF, "PointManager.ts")).read()
    count = strip_types(between(pm, "const primitives = scene.getPrimitives();", "// Clamp to reasonable range", start=pm.index("calculatePointCount")))
    spm = open(os.path.join(REF, "SplatPropertyManager.ts")).read()
    defaults = strip_types(between(spm, "const data = new Float32Array(this.numSplats * 8);", "this.device.queue.writeBuffer("))
    cg = open(os.path.join(REF, "sdf", "CodeGenerator.ts")).read()
    walk = strip_types(between(cg, "let varCounter = 0;", 'return lines.join("\\n");', include_last=True, start=cg.index("generateSceneSDF")))
    types = between(prim, "export const PrimitiveType = {", "} as const;", include_last=True).replace("export ", "").replace(" as const", "")
    op = open(os.path.join(REF, "sdf", "Operation.ts")).read()
    types += "\n" + between(op, "export const OperationType = {", "} as const;", include_last=True).replace("export ", "").replace(" as const", "")
    return areas, scale, count, defaults, walk, types


PROGRAM = """
const input = JSON.parse(require('fs').readFileSync(0, 'utf8'));
%(TYPES)s
// gl-matrix 3.4.4's documented definitions of the two functions scaleAABB calls (the library is absent: see the header)
const vec3 = {
  scaleAndAdd(out, a, b, s) { out[0] = a[0] + b[0] * s; out[1] = a[1] + b[1] * s; out[2] = a[2] + b[2] * s; return out; },
  sub(out, a, b) { out[0] = a[0] - b[0]; out[1] = a[1] - b[1]; out[2] = a[2] - b[2]; return out; },
};
const AREA = {
  sphere: function () { %(A_SPHERE)s },
  box: function () { %(A_BOX)s },
  torus: function () { %(A_TORUS)s },
  capsule: function () { %(A_CAPSULE)s },
};
const _center = [], _currentScale = [];
function scaleAABB(aabb, scale) {
  %(SCALE)s
}
function calculatePointCount(scene) {
  %(COUNT)s
}
function initializeDefaults() {
  %(DEFAULTS)s
  return Array.from(data);
}
function generateSceneSDF(root) {
  %(WALK)s
}
// plain descriptions -> the objects those statements read (fields and getters of src/sdf/*.ts)
let nextSmin = 0;
function makePrim(d) {
  // (position and size are gl-matrix vec3 in the reference — vec3.clone(params.x): Float32Array storage; the other parameters numbers)
  const p = Object.assign({}, d, { id: d.id, position: Float32Array.from(d.position || [0, 0, 0]) });
  if (d.prim === 'box') p.size = Float32Array.from(d.size);
  p.getType = () => d.prim;
  p.getSurfaceArea = AREA[d.prim].bind(p);
  return p;
}
function makeNode(d, prims) {
  if (d.prim) { const p = makePrim(d); prims.push(p); return { type: 'primitive', primitive: p }; }
  const id = d.op === 'smooth_union' ? 'smin_' + (nextSmin++) : null;
  const operation = { getType: () => d.op, getParamNames: () => (id ? [id + '_k'] : []) };
  return { type: 'operation', operation, children: d.children.map((c) => makeNode(c, prims)) };
}
const out = { areas: [], scaleAABB: [], scenes: {} };
for (const d of input.prims) out.areas.push(AREA[d.prim].call(makePrim(d)));
for (const c of input.boxes) { const r = scaleAABB({ min: c.min, max: c.max }, c.scale); out.scaleAABB.push({ min: Array.from(r.min), max: Array.from(r.max) }); }
for (const name of Object.keys(input.scenes)) {
  const prims = [];
  nextSmin = 0;
  const root = makeNode(input.scenes[name], prims);
  out.scenes[name] = { pointCount: calculatePointCount({ getPrimitives: () => prims }), wgsl: generateSceneSDF(root).split('\\n') };
}
out.pointCountEmptyScene = calculatePointCount({ getPrimitives: () => [] });
out.defaults = initializeDefaults.call({ numSplats: 3 });
process.stdout.write(JSON.stringify(out));
"""


def scenes():
    """tests/test_gpu_sdf.py::scenes() as plain descriptions (ids given, so that the generated names are fixed)."""
    s = lambda i, pos, r: {"prim": "sphere", "id": i, "position": list(pos), "radius": r}
    main_ts = {"op": "smooth_union", "k": 0.1, "children": [
        {"op": "smooth_union", "k": 0.15, "children": [s("sphere1", (0, 0, 0), 0.5), {"prim": "box", "id": "box1", "position": [0.6, 0, 0], "size": [0.3, 0.3, 0.3]}]},
        s("sphere2", (0, 0.6, 0), 0.25)]}
    all_ops = {"op": "subtraction", "children": [
        {"op": "union", "children": [s("a", (0, 0, 0), 0.5), {"op": "intersection", "children": [
            {"prim": "box", "id": "b", "position": [0, 0, 0], "size": [0.6, 0.2, 0.6]},
            {"prim": "torus", "id": "c", "position": [0, 0, 0], "majorRadius": 0.45, "minorRadius": 0.2}]}]},
        {"op": "smooth_union", "k": 0.08, "children": [
            {"prim": "capsule", "id": "d", "position": [0.2, 0, 0], "height": 1.2, "radius": 0.12}, s("e", (0, 0.4, 0), 0.2)]}]}
    return {"main_ts": main_ts, "sphere": s("p", (0.1, -0.2, 0.05), 0.45),
            "box": {"prim": "box", "id": "p", "position": [0, 0.1, 0], "size": [0.4, 0.25, 0.3]},
            "torus": {"prim": "torus", "id": "p", "position": [0, 0, 0.1], "majorRadius": 0.5, "minorRadius": 0.15},
            "capsule": {"prim": "capsule", "id": "p", "position": [-0.1, 0, 0], "height": 0.8, "radius": 0.2}, "all_ops": all_ops,
            # the clamps of calculatePointCount: a speck (-> 10000) and a hall (-> 200000)
            "tiny": s("p", (0, 0, 0), 0.01), "huge": {"prim": "box", "id": "p", "position": [0, 0, 0], "size": [9, 9, 9]}}


def main():
    areas, scale, count, defaults, walk, types = reference_statements()
    program = PROGRAM % {"TYPES": types, "A_SPHERE": areas["Sphere"], "A_BOX": areas["Box"], "A_TORUS": areas["Torus"],
                         "A_CAPSULE": areas["Capsule"], "SCALE": scale, "COUNT": count, "DEFAULTS": defaults, "WALK": walk}
    prims = [{"prim": "sphere", "id": "p", "radius": r} for r in (0.5, 0.25, 0.01, 1.7)] + \
            [{"prim": "box", "id": "p", "size": sz} for sz in ([0.5, 0.5, 0.5], [0.3, 0.3, 0.3], [0.6, 0.2, 0.6], [1.25, 0.1, 2.0])] + \
            [{"prim": "torus", "id": "p", "majorRadius": a, "minorRadius": b} for a, b in ((0.5, 0.2), (0.45, 0.2), (0.5, 0.15), (2.0, 0.01))] + \
            [{"prim": "capsule", "id": "p", "height": hgt, "radius": r} for hgt, r in ((1.0, 0.3), (1.2, 0.12), (0.8, 0.2), (0.0, 0.5))]
    boxes = [{"min": [-0.5, -0.5, -0.5], "max": [0.9, 0.85, 0.5], "scale": 1.5}, {"min": [-1, -1, -1], "max": [1, 1, 1], "scale": 1.5},
             {"min": [0.1, 0.2, 0.3], "max": [0.4, 0.9, 1.6], "scale": 1.5}, {"min": [-2, 0, 1], "max": [-1, 3, 1.5], "scale": 0.5},
             {"min": [0, 0, 0], "max": [0, 0, 0], "scale": 1.5}]
    inp = {"prims": prims, "boxes": boxes, "scenes": scenes()}
    r = subprocess.run(["node", "-e", program], input=json.dumps(inp), capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit("node failed:\n" + r.stderr[-3000:])
    out = json.loads(r.stdout)
    ver = subprocess.run(["node", "--version"], capture_output=True, text=True).stdout.strip()
    with open(os.path.join(HERE, "ref_host.json"), "w") as f:
        json.dump({"generated_by": "tests/golden/make_ref_host_fixtures.py (reference statements executed under node " + ver + ")",
                   "inputs": inp, "outputs": out}, f, indent=1)
    print("ref_host.json:", len(out["areas"]), "areas,", len(out["scaleAABB"]), "scaled boxes,", len(out["scenes"]), "scenes; point counts",
          {k: v["pointCount"] for k, v in out["scenes"].items()}, "empty scene", out["pointCountEmptyScene"])
    for k in ("main_ts", "all_ops"):
        print(k, [ln.strip() for ln in out["scenes"][k]["wgsl"]][1:-2])


if __name__ == "__main__":
    main()
