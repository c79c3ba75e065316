"""SDF splat generation: scene graph + GradientSampler / PositionUpdater / CurvatureSampler.

Mirrors /root/reference/src/sdf/{Primitive,Operation,Scene}.ts (class names, fields, defaults, builder
functions, getStructureHash) and the three samplers' verbs (src/GradientSampler.ts, src/PositionUpdater.ts,
src/CurvatureSampler.ts).  The reference turns a scene graph into generated WGSL (sdf/CodeGenerator.ts) and a
packed uniform block (sdf/ParameterEncoder.ts); here the graph becomes a postfix program of
`splat_sdf_instr` records (include/splat.h) evaluated by a stack machine in csrc/sdf.hip, so
`updateSceneParameters()` after animating a primitive is just a re-encode and a structural change needs
no recompilation.  Nothing here evaluates an SDF on the CPU and nothing imports oracle/.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import SdfInstr, SplatError, check


# ---- src/sdf/Primitive.ts ---------------------------------------------------------------------------------------
class Primitive:
    _next_id = 0

    def __init__(self, id=None, position=None):
        if id is None:
            id = f"prim_{Primitive._next_id}"
            Primitive._next_id += 1
        self.id = id
        self.position = np.array(position if position is not None else (0, 0, 0), np.float32)  # vec3 (Float32Array)


class Sphere(Primitive):  # :61-112
    TYPE = "sphere"

    def __init__(self, id=None, position=None, radius=0.5):
        super().__init__(id, position)
        self.radius = radius

    def getParamNames(self):
        return [f"{self.id}_center", f"{self.id}_radius"]

    def getParamValues(self):
        return [*map(float, self.position), self.radius]

    def getAABB(self):
        p, r = self.position.astype(np.float64), self.radius
        return (p - r).astype(np.float32), (p + r).astype(np.float32)

    def getSurfaceArea(self):
        return 4 * math.pi * self.radius * self.radius

    def _instr(self):
        return _lib.SDF_SPHERE, [*map(float, self.position), self.radius]


class Box(Primitive):  # :114-165
    TYPE = "box"

    def __init__(self, id=None, position=None, size=None):
        super().__init__(id, position)
        self.size = np.array(size if size is not None else (0.5, 0.5, 0.5), np.float32)

    def getParamNames(self):
        return [f"{self.id}_center", f"{self.id}_size"]

    def getParamValues(self):
        return [*map(float, self.position), 0, *map(float, self.size), 0]

    def getAABB(self):
        p, s = self.position.astype(np.float64), self.size.astype(np.float64)
        return (p - s).astype(np.float32), (p + s).astype(np.float32)

    def getSurfaceArea(self):
        w, h, d = (float(v) * 2 for v in self.size)
        return 2 * (w * h + w * d + h * d)

    def _instr(self):
        return _lib.SDF_BOX, [*map(float, self.position), *map(float, self.size)]


class Torus(Primitive):  # :167-222
    TYPE = "torus"

    def __init__(self, id=None, position=None, majorRadius=0.5, minorRadius=0.2):
        super().__init__(id, position)
        self.majorRadius, self.minorRadius = majorRadius, minorRadius

    def getParamNames(self):
        return [f"{self.id}_center", f"{self.id}_radii"]

    def getParamValues(self):
        return [*map(float, self.position), 0, self.majorRadius, self.minorRadius, 0, 0]

    def getAABB(self):
        p = self.position.astype(np.float64)
        e = np.array([self.majorRadius + self.minorRadius, self.minorRadius, self.majorRadius + self.minorRadius])
        return (p - e).astype(np.float32), (p + e).astype(np.float32)

    def getSurfaceArea(self):
        return 4 * math.pi * math.pi * self.majorRadius * self.minorRadius

    def _instr(self):
        return _lib.SDF_TORUS, [*map(float, self.position), self.majorRadius, self.minorRadius]


class Capsule(Primitive):  # :224-281
    TYPE = "capsule"

    def __init__(self, id=None, position=None, height=1.0, radius=0.3):
        super().__init__(id, position)
        self.height, self.radius = height, radius

    def getParamNames(self):
        return [f"{self.id}_center", f"{self.id}_params"]

    def getParamValues(self):
        return [*map(float, self.position), 0, self.height, self.radius, 0, 0]

    def getAABB(self):
        p = self.position.astype(np.float64)
        e = np.array([self.radius, self.height / 2 + self.radius, self.radius])
        return (p - e).astype(np.float32), (p + e).astype(np.float32)

    def getSurfaceArea(self):
        return 2 * math.pi * self.radius * self.height + 4 * math.pi * self.radius * self.radius

    def _instr(self):
        return _lib.SDF_CAPSULE, [*map(float, self.position), self.height, self.radius]


def scaleAABB(aabb, scale):
    """src/sdf/Primitive.ts:283-290 as written: centre = min + max / 2 (vec3.scaleAndAdd(_, min, max, 1/2)), not the
    midpoint — kept, so that seeding boxes equal the reference's."""
    mn, mx = (np.asarray(a, np.float64) for a in aabb)
    center = mn + mx * 0.5
    extent = mx - mn
    return center + extent * (-scale / 2), center + extent * (scale / 2)


# ---- src/sdf/Operation.ts ---------------------------------------------------------------------------------------
class Operation:
    def getParamNames(self):
        return []

    def getParamValues(self):
        return []


class Union(Operation):
    TYPE, OP = "union", _lib.SDF_UNION


class Intersection(Operation):
    TYPE, OP = "intersection", _lib.SDF_INTERSECTION


class Subtraction(Operation):
    TYPE, OP = "subtraction", _lib.SDF_SUBTRACTION


class SmoothUnion(Operation):
    TYPE, OP = "smooth_union", _lib.SDF_SMOOTH_UNION
    _next_id = 0

    def __init__(self, k=0.1):
        self.k = k
        self.id = f"smin_{SmoothUnion._next_id}"
        SmoothUnion._next_id += 1

    def getParamNames(self):
        return [f"{self.id}_k"]

    def getParamValues(self):
        return [self.k]


# ---- src/sdf/Scene.ts -------------------------------------------------------------------------------------------
def primitive(prim):  # :20-25
    return prim if isinstance(prim, dict) else {"type": "primitive", "primitive": prim}


def _binary(op, a, b):
    return {"type": "operation", "operation": op, "children": [primitive(a), primitive(b)]}


def union(a, b):  # :30-36
    return _binary(Union(), a, b)


def intersection(a, b):  # :41-47
    return _binary(Intersection(), a, b)


def subtraction(a, b):  # :52-58
    return _binary(Subtraction(), a, b)


def smoothUnion(k, a, b):  # :64-70
    return _binary(SmoothUnion(k), a, b)


class SDFScene:  # :72-152
    def __init__(self):
        self.root = None
        self.primitiveMap = {}

    def setRoot(self, node):
        self.root = primitive(node)
        self.primitiveMap = {}
        self._collect(self.root)

    def _collect(self, node):
        if node["type"] == "primitive":
            self.primitiveMap[node["primitive"].id] = node["primitive"]
        else:
            for c in node["children"]:
                self._collect(c)

    def get(self, id):
        return self.primitiveMap.get(id)

    def getPrimitives(self):
        return list(self.primitiveMap.values())

    def getRoot(self):
        return self.root

    def getOperations(self):
        ops = []

        def walk(node):
            if node["type"] == "operation":
                ops.append(node["operation"])
                for c in node["children"]:
                    walk(c)
        if self.root:
            walk(self.root)
        return ops

    def getStructureHash(self):
        def walk(node):
            if node["type"] == "primitive":
                return f"P:{node['primitive'].TYPE}:{node['primitive'].id}"
            return f"O:{node['operation'].TYPE}:({','.join(walk(c) for c in node['children'])})"
        return walk(self.root) if self.root else ""

    def program(self):
        """The scene graph as the postfix program the kernels evaluate: children first, then their operation — the
        order WGSLCodeGenerator.generateSceneSDF's traverse() emits its `let result_k = ...` lines
        (src/sdf/CodeGenerator.ts:291-346).  Returns [(op, [params]), ...] with the primitives' and operations'
        CURRENT parameter values."""
        out = []

        def walk(node):
            if node["type"] == "primitive":
                out.append(node["primitive"]._instr())
            else:
                for c in node["children"]:
                    walk(c)
                op = node["operation"]
                out.append((op.OP, op.getParamValues()))
        if self.root:
            walk(self.root)
        if len(out) > _lib.SDF_MAX_INSTR:
            raise SplatError(-1, f"scene graph has {len(out)} nodes; the evaluator takes {_lib.SDF_MAX_INSTR}")
        return out


def _encode(program):
    arr = (SdfInstr * max(len(program), 1))()
    for k, (op, a) in enumerate(program):
        arr[k].op = op
        for j, v in enumerate(a):
            arr[k].a[j] = v
    return arr, len(program)


def seeding_box(scene):
    """The box PointManager seeds on (src/PointManager.ts:96-107): every primitive's AABB, merged, scaled 1.5x with the
    reference's scaleAABB as written; (-1,-1,-1)..(1,1,1) for a scene without primitives.  Two (3,) f32 arrays."""
    prims = scene.getPrimitives()
    if not prims:
        return np.full(3, -1.0, np.float32), np.full(3, 1.0, np.float32)
    boxes = [p.getAABB() for p in prims]
    mn = np.min([b[0] for b in boxes], axis=0).astype(np.float64)
    mx = np.max([b[1] for b in boxes], axis=0).astype(np.float64)
    mn, mx = scaleAABB((mn, mx), 1.5)
    return mn.astype(np.float32), mx.astype(np.float32)


def seed_positions(scene, numPoints, seed=0):
    """PointManager.generateRandomPositions (src/PointManager.ts:96-189): points on the faces of the scene's global AABB
    (every primitive's box, scaled 1.5x), a face chosen with probability proportional to its area — with a SEEDED
    generator (the reference draws from Math.random, so its clouds cannot be reproduced; the distribution is the same)."""
    prims = scene.getPrimitives()
    if not prims:
        mn, mx = np.full(3, -1.0), np.full(3, 1.0)
    else:
        boxes = [p.getAABB() for p in prims]
        mn = np.min([b[0] for b in boxes], axis=0).astype(np.float64)
        mx = np.max([b[1] for b in boxes], axis=0).astype(np.float64)
        mn, mx = scaleAABB((mn, mx), 1.5)
    d = mx - mn
    areas = np.array([d[1] * d[2], d[1] * d[2], d[0] * d[2], d[0] * d[2], d[0] * d[1], d[0] * d[1]])
    rng = np.random.default_rng(seed)
    face = np.searchsorted(np.cumsum(areas), rng.random(numPoints) * areas.sum(), side="right").clip(0, 5)
    uvw = mn + rng.random((numPoints, 3)) * d
    axis, hi = face // 2, face % 2
    uvw[np.arange(numPoints), axis] = np.where(hi == 1, mx[axis], mn[axis])
    pos = np.zeros((numPoints, 4), np.float32)
    pos[:, :3] = uvw
    return pos


def point_count(scene):
    """PointManager.calculatePointCount (src/PointManager.ts:22-39)."""
    prims = scene.getPrimitives()
    if not prims:
        return 50000
    total = sum(math.floor(30000 * math.sqrt(p.getSurfaceArea())) for p in prims)
    return max(10000, min(total, 200000))


# ---- the samplers ------------------------------------------------------------------------------------------------
class _SceneStage:
    def __init__(self, device, scene, numPoints):
        self.device, self.scene, self.numPoints = device, scene, numPoints
        self.currentStructureHash = scene.getStructureHash()
        self.updateSceneParameters()

    def updateSceneParameters(self):
        """Re-encode the scene's current parameter values (call after animating a primitive: src/main.ts:114-120)."""
        self._program, self._count = _encode(self.scene.program())

    def rebuildIfNeeded(self):
        """The reference regenerates and recompiles its shader when the graph's structure changes
        (src/GradientSampler.ts:96-125); here a structural change is just another program."""
        h = self.scene.getStructureHash()
        if h != self.currentStructureHash:
            self.currentStructureHash = h
            self.updateSceneParameters()

    def getScene(self):
        return self.scene


class GradientSampler(_SceneStage):
    """src/GradientSampler.ts:6-172: gradients[i] = sceneSDF(positions[i]) = vec4(distance, gradient)."""

    def __init__(self, device, scene, numPoints):
        super().__init__(device, scene, numPoints)
        self.gradientBuffer = device.createBuffer(numPoints * 16)

    def evaluateGradients(self, commandEncoder, uniformBuffer, positionBuffer):  # :134-158
        d = self.device
        check(d.lib.splat_sdf_gradients(d.ctx, C.cast(self._program, C.c_void_p), self._count, positionBuffer.ptr, self.numPoints,
                                        self.gradientBuffer.ptr), d.ctx)

    def getGradientBuffer(self):  # :160-162
        return self.gradientBuffer

    def destroy(self):  # :168-171
        self.gradientBuffer.destroy()


class PositionUpdater:
    """src/PositionUpdater.ts:1-85 + src/shaders/update-positions.wgsl: one projection step along the gradient."""

    def __init__(self, device, shaderCode=None, numPoints=0):
        self.device, self.numPoints = device, numPoints  # (shaderCode: the reference passes its WGSL text; unused here)

    def updatePositions(self, commandEncoder, uniformBuffer, currentPositionBuffer, gradientBuffer, nextPositionBuffer):  # :59-84
        d = self.device
        check(d.lib.splat_sdf_update_positions(d.ctx, currentPositionBuffer.ptr, gradientBuffer.ptr, self.numPoints,
                                               nextPositionBuffer.ptr), d.ctx)


class CurvatureSampler(_SceneStage):
    """src/CurvatureSampler.ts:5-234: one scale factor per point from the variation of the normal around it."""

    def __init__(self, device, scene, numPoints):
        super().__init__(device, scene, numPoints)
        self.scaleFactorsBuffer = device.createBuffer(numPoints * 4)
        self.curvatureBuffer = None

    def computeScaleFactors(self, commandEncoder, positionBuffer):  # :196-222
        d = self.device
        check(d.lib.splat_sdf_scale_factors(d.ctx, C.cast(self._program, C.c_void_p), self._count, positionBuffer.ptr, self.numPoints,
                                            self.scaleFactorsBuffer.ptr), d.ctx)

    def getScaleFactorsBuffer(self):  # :224-226
        return self.scaleFactorsBuffer

    def getCurvatureBuffer(self, gradientBuffer):
        """vec4(normal, scaleFactor) per point — the buffer SplatPropertyManager.updateFromCurvature and the composite
        bind as curvatureData (the reference's samplers write its two halves to separate buffers: SURVEY I4)."""
        d = self.device
        if self.curvatureBuffer is None:
            self.curvatureBuffer = d.createBuffer(self.numPoints * 16)
        check(d.lib.splat_sdf_curvature(d.ctx, gradientBuffer.ptr, self.scaleFactorsBuffer.ptr, self.numPoints,
                                        self.curvatureBuffer.ptr), d.ctx)
        return self.curvatureBuffer

    def destroy(self):  # :228-233
        self.scaleFactorsBuffer.destroy()
        if self.curvatureBuffer is not None:
            self.curvatureBuffer.destroy()
