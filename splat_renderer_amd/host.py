"""Host-side mirror of the reference's stage classes over the C ABI (include/splat.h).

Same class names, verbs, argument order and error behaviour as the TypeScript classes under
/root/reference/src, so that tests read like tests of the reference would:

    device = Device()                                   # GPUDevice + queue
    props = SplatPropertyManager(device, n)
    projector = SplatProjector(device, n)
    sorter = RadixSorter(device, n)
    extractor = DepthKeyExtractor(device)
    binner = GPUTileBinner(device, 16)
    renderer = ComputeShaderRenderer(device, None, "rgba8unorm")
    enc = device.createCommandEncoder()
    projector.project(enc, uniforms, props.getPropertyBuffer())
    extractor.extract(enc, projector.getProjectedBuffer(), sorter.getKeysBuffer(), sorter.getPayloadBuffer(), n, n_padded)
    sorter.sort()
    binner.binSplats(enc, projector.getProjectedBuffer(), sorter.getSortedIndicesBuffer(), n, W, H)
    renderer.render(uniforms, props.getPropertyBuffer(), binner.getTileIndicesBuffer(), normals, ...)

A GPUCommandEncoder maps to "the ctx stream": recording is immediate enqueue, `enc` arguments are
accepted for signature compatibility and ignored.  A GPUBuffer maps to `Buffer` (device pointer +
size).  Nothing here computes on the CPU and nothing imports oracle/.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import CompositeCfg, SplatError, check

U32_MAX = 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------
class Buffer:
    """GPUBuffer equivalent: an owned or borrowed device allocation with a byte size."""

    def __init__(self, device, ptr, size, owned=True):
        self.device, self.ptr, self.size, self._owned = device, ptr, int(size), owned
        self._host_shadow = None  # small uniform buffers keep a host copy (the ABI takes host uniforms)

    def destroy(self):
        if self._owned and self.ptr:
            check(self.device.lib.splat_buf_free(self.device.ctx, self.ptr), self.device.ctx)
        self.ptr = None

    def write(self, array, offset=0):
        a = np.ascontiguousarray(array)
        if offset + a.nbytes > self.size:
            raise ValueError("write past the end of the buffer")
        check(self.device.lib.splat_buf_upload(self.device.ctx, self.ptr + offset, a.ctypes.data, a.nbytes), self.device.ctx)
        if self.size <= 256:
            if self._host_shadow is None:
                self._host_shadow = np.zeros(self.size, np.uint8)
            self._host_shadow[offset:offset + a.nbytes] = a.view(np.uint8).reshape(-1)
        return self

    def read(self, dtype=np.uint8, count=None, offset=0):
        dtype = np.dtype(dtype)
        if count is None:
            count = (self.size - offset) // dtype.itemsize
        out = np.empty(count, dtype)
        if out.nbytes:
            check(self.device.lib.splat_buf_download(self.device.ctx, out.ctypes.data, self.ptr + offset, out.nbytes),
                  self.device.ctx)
        return out

    def zero(self):
        check(self.device.lib.splat_buf_zero(self.device.ctx, self.ptr, self.size), self.device.ctx)


class _Queue:
    def __init__(self, device):
        self._d = device

    def writeBuffer(self, buffer, offset, data):
        buffer.write(data, offset)

    def submit(self, _command_buffers=None):
        pass  # work is enqueued on the stream as it is recorded

    def onSubmittedWorkDone(self):
        self._d.sync()


class CommandEncoder:
    """Accepted wherever the reference takes a GPUCommandEncoder; recording is immediate."""

    def finish(self):
        return None


class Device:
    """GPUDevice + queue equivalent: one HIP device + one stream (splat_ctx)."""

    def __init__(self, ordinal=0, stream=None):
        self.lib = _lib.load()
        p = C.c_void_p()
        if stream is None:
            check(self.lib.splat_ctx_create(ordinal, C.byref(p)))
        else:
            check(self.lib.splat_ctx_create_on_stream(ordinal, C.c_void_p(stream), C.byref(p)))
        self.ctx = p
        self.ordinal = ordinal
        self.queue = _Queue(self)
        # the SplatProjector and GPUTileBinner that last ran on this device (their project() / binSplats(), or a Renderer's
        # frame): what TileRenderer.render — whose reference signature names neither — composites from unless told otherwise
        self.lastProjector = None
        self.lastBinner = None

    def createBuffer(self, size):
        p = C.c_void_p()
        check(self.lib.splat_buf_alloc(self.ctx, int(size), C.byref(p)), self.ctx)
        return Buffer(self, p.value, size)

    def createBufferFrom(self, array):
        a = np.ascontiguousarray(array)
        return self.createBuffer(max(a.nbytes, 16)).write(a)

    def wrap(self, ptr, size):
        """Borrow an existing device allocation (e.g. torch.Tensor.data_ptr())."""
        return Buffer(self, int(ptr), size, owned=False)

    def createCommandEncoder(self):
        return CommandEncoder()

    def sync(self):
        check(self.lib.splat_sync(self.ctx), self.ctx)

    def setTiming(self, enabled=True):
        check(self.lib.splat_set_timing(self.ctx, int(enabled)), self.ctx)

    def stageTimeMs(self, stage):
        ms = C.c_float()
        check(self.lib.splat_stage_time_ms(self.ctx, stage, C.byref(ms)), self.ctx)
        return float(ms.value)

    def rankStatus(self):
        """How this context ranks equal digits in its sort kernels (include/splat.h, NOTE on ranking): dict(policy =
        'checked' | 'atomic' | 'ballot', atomicsOrdered = the start-up probe's verdict, orderFaults = frames whose tile
        lists failed the per-tile sort's order check and were rendered again)."""
        pol, ordered, faults = C.c_int(), C.c_int(), C.c_uint32()
        check(self.lib.splat_rank_status(self.ctx, C.byref(pol), C.byref(ordered), C.byref(faults)), self.ctx)
        return {"policy": ("checked", "atomic", "ballot")[pol.value], "atomicsOrdered": bool(ordered.value == 1),
                "orderFaults": int(faults.value)}

    def forgetCompositeHistory(self):
        """splat_composite_forget_history: the lane-efficient composite's next launches run as a context's first do (row-major,
        no look-ahead bound from an earlier launch's per-tile costs)."""
        check(self.lib.splat_composite_forget_history(self.ctx), self.ctx)

    def compositeOptions(self, kernel=None, ahead=0, predict=None):
        """splat_composite_options: kernel None (process default) | 'quadrant' | 'pixel'; ahead 0 (default) | 1 | 2; predict None |
        False | True.  EVERY call sets all three (None / 0 = the process default, i.e. the environment's — not "as it was"); a choice
        made here takes precedence over the environment variable.  ahead / predict change the schedule only (same bytes); the two
        kernels agree within the composite's stated tolerance."""
        k = -1 if kernel is None else {"quadrant": 0, "pixel": 1}[kernel]
        check(self.lib.splat_composite_options(self.ctx, k, int(ahead), -1 if predict is None else int(bool(predict))), self.ctx)

    def injectOrderFault(self, tile, position=0):
        """TEST HOOK (splat_debug_inject_order_fault; the test build of the library only: SPLAT_LIB_PATH=libsplat_hip_hooks.so):
        the next per-tile sort leaves entries position, position + 1 of tile `tile`'s list swapped."""
        check(self.lib.splat_debug_inject_order_fault(self.ctx, int(tile), int(position)), self.ctx)

    def tileSortLaunches(self):
        """TEST HOOK (splat_debug_tile_sort_launches; the test build only): k_tile_sort launches of the last per-tile sort."""
        v = C.c_uint32(0)
        check(self.lib.splat_debug_tile_sort_launches(self.ctx, C.byref(v)), self.ctx)
        return int(v.value)

    def destroy(self):
        if self.ctx:
            self.lib.splat_ctx_destroy(self.ctx)
            self.ctx = None


def _uniform_floats(u):
    """Accepts a host float array or a small uniform Buffer written through queue.writeBuffer."""
    if isinstance(u, Buffer):
        if u._host_shadow is None:
            raise SplatError(-1, "uniform buffer was never written")
        u = u._host_shadow.view(np.float32)
    u = np.ascontiguousarray(u, dtype=np.float32)
    return u


# ------------------------------------------------------------------------------------------------
class PointManager:
    """src/PointManager.ts:10-252 — ping-pong position buffers.  `scene` is
      * an SDFScene (splat_renderer_amd.sdf): the reference's constructor — point count from the primitives' surface
        areas (:22-39), positions on the faces of the scene's global AABB (:96-189), fresh ones at every
        reinitialize() (:220-231) — from a SEEDED generator (`seed`, then seed+1, ...: the reference's Math.random
        clouds cannot be reproduced).  seeding="device" (default): the cloud is drawn by a kernel
        (splat_sdf_seed_positions: point i a pure function of (seed, i); nothing crosses PCIe — the reference's
        per-frame CPU draw + upload was 4.6 of the 4.7 ms its own demo frame took here); seeding="host": NumPy's
        PCG64 on the CPU and an upload (sdf.seed_positions);
      * an (n,4) f32 position array, which reinitialize() uploads again; or an int: the bench scene of that seed."""

    def __init__(self, device, scene, seed=0, seeding="device"):
        self.device = device
        self.scene, self._seed, self._positions = None, seed, None
        if seeding not in ("device", "host"):
            raise SplatError(-1, f"seeding must be 'device' or 'host', not {seeding!r}")
        self.seeding = seeding
        if hasattr(scene, "getPrimitives"):
            from . import sdf
            if not scene.getPrimitives():
                raise SplatError(-1, "Scene must have at least one primitive")  # :47-49
            self.scene = scene
            self.numPoints = sdf.point_count(scene)
        else:
            if isinstance(scene, (int, np.integer)):
                from .scene import make_scene
                props, _ = make_scene(int(scene))
                scene = np.concatenate([props[:, :3], np.ones((props.shape[0], 1), np.float32)], axis=1)
            self._positions = np.ascontiguousarray(scene, dtype=np.float32)
            self.numPoints = self._positions.shape[0]
        self._buffers = [device.createBuffer(self.numPoints * 16), device.createBuffer(self.numPoints * 16)]
        self._current = 0
        self.reinitialize()

    def reinitialize(self):  # :220-231
        if self.scene is not None:
            from . import sdf
            if self.seeding == "device":
                mn, mx = sdf.seeding_box(self.scene)
                fp = C.POINTER(C.c_float)
                check(self.device.lib.splat_sdf_seed_positions(self.device.ctx, mn.ctypes.data_as(fp), mx.ctypes.data_as(fp), self.numPoints,
                                                               int(self._seed) & 0xFFFFFFFFFFFFFFFF, self._buffers[self._current].ptr),
                      self.device.ctx)
                self._seed += 1
                return
            self._positions = sdf.seed_positions(self.scene, self.numPoints, self._seed)
            self._seed += 1
        self._buffers[self._current].write(self._positions)

    def swap(self):  # :240-242
        self._current = 1 - self._current

    def getCurrentPositionBuffer(self):  # :233-235
        return self._buffers[self._current]

    def getNextPositionBuffer(self):  # :236-238
        return self._buffers[1 - self._current]

    def getNumPoints(self):  # :244-246
        return self.numPoints

    def destroy(self):  # :248-252
        for b in self._buffers:
            b.destroy()


class PropertyPlanes:
    """Two device planes of vec4 per splat: (pos, radius) and (rgb, opacity).  prelit: the colour plane
    already carries the reference's shading (SplatPropertyManager.getLitPlanes)."""

    def __init__(self, posRadius, colorOpacity, prelit=False, ownsPosRadius=True):
        self.posRadius, self.colorOpacity, self.prelit, self._owns_pos = posRadius, colorOpacity, prelit, ownsPosRadius

    def destroy(self):
        if self._owns_pos:
            self.posRadius.destroy()
        self.colorOpacity.destroy()


def default_properties(numSplats):
    """SplatPropertyManager.initializeDefaults (src/SplatPropertyManager.ts:33-50): position 0, radius 0.04, white, opacity
    0.7 — (numSplats, 8) f32.  Held to an execution of the reference's own loop (tests/golden/ref_host.json)."""
    data = np.zeros((numSplats, 8), np.float32)
    data[:, 3] = 0.04
    data[:, 4:7] = 1.0
    data[:, 7] = 0.7
    return data


class SplatPropertyManager:
    """src/SplatPropertyManager.ts:13-181 — owns the 32 B/splat interleaved property buffer."""

    def __init__(self, device, numSplats):
        self.device, self.numSplats = device, numSplats
        self.propertyBuffer = device.createBuffer(numSplats * 32)
        self.propertyBuffer.write(default_properties(numSplats))  # initializeDefaults :33-50
        self._planes, self._planes_valid = None, False
        self._lit, self._lit_valid, self._lit_normals = None, False, None

    def updateFromCurvature(self, commandEncoder, positionBuffer, curvatureBuffer):  # :153-173
        d = self.device
        check(d.lib.splat_update_props(d.ctx, positionBuffer.ptr, curvatureBuffer.ptr, self.numSplats,
                                       self.propertyBuffer.ptr), d.ctx)
        self._planes_valid = False
        self._lit_valid = False

    def setFromArrays(self, props):
        """Synthetic scenes: upload (n,8) interleaved records directly."""
        self.propertyBuffer.write(np.ascontiguousarray(props, np.float32))
        self._planes_valid = False
        self._lit_valid = False

    def getPropertyBuffer(self):  # :175-177
        return self.propertyBuffer

    def getPropertyPlanes(self):
        """The MI355X-native layout of the same properties: (vec4(pos, radius) plane, vec4(rgb, opacity)
        plane), converted on the device when the interleaved buffer changed.
        Renderer.render accepts the pair in place of the interleaved buffer."""
        d = self.device
        if getattr(self, "_planes", None) is None:
            self._planes = PropertyPlanes(d.createBuffer(self.numSplats * 16), d.createBuffer(self.numSplats * 16))
            self._planes_valid = False
        if not self._planes_valid:
            check(d.lib.splat_props_to_planes(d.ctx, self.propertyBuffer.ptr, self.numSplats, self._planes.posRadius.ptr,
                                              self._planes.colorOpacity.ptr), d.ctx)
            self._planes_valid = True
        return self._planes

    def getLitPlanes(self, normalsBuffer):
        """The planes with the colour plane already lit by `normalsBuffer` — kd = 0.85 + 0.15 max(n.l, 0)
        (src/ComputeShaderRenderer.ts:143-145) applied once per property update instead of once per staged
        list entry: the composite then gathers two lines per entry (record, lit colour) instead of three.
        Same bits as shading in the composite.  Recomputed when the properties or the normals buffer change
        (call invalidateLighting() after rewriting the normals in place)."""
        d = self.device
        planes = self.getPropertyPlanes()
        if getattr(self, "_lit", None) is None:
            self._lit = PropertyPlanes(planes.posRadius, d.createBuffer(self.numSplats * 16), prelit=True, ownsPosRadius=False)
            self._lit_valid = False
        if not self._lit_valid or self._lit_normals != normalsBuffer.ptr:
            check(d.lib.splat_lit_colors(d.ctx, planes.colorOpacity.ptr, 1, normalsBuffer.ptr, 1, self.numSplats,
                                         self._lit.colorOpacity.ptr), d.ctx)
            self._lit_valid, self._lit_normals = True, normalsBuffer.ptr
        return self._lit

    def invalidateLighting(self):
        self._lit_valid = False

    def updatePlanesFromCurvature(self, commandEncoder, positionBuffer, curvatureBuffer):
        """updateFromCurvature writing the two planes directly (no interleaved copy is touched)."""
        d = self.device
        planes = self.getPropertyPlanes()
        check(d.lib.splat_update_props_planes(d.ctx, positionBuffer.ptr, curvatureBuffer.ptr, self.numSplats,
                                              planes.posRadius.ptr, planes.colorOpacity.ptr), d.ctx)
        self._lit_valid = False
        return planes

    def destroy(self):  # :179-181
        self.propertyBuffer.destroy()
        if getattr(self, "_lit", None) is not None:
            self._lit.destroy()
            self._lit = None
        if getattr(self, "_planes", None) is not None:
            self._planes.destroy()
            self._planes = None


def _footprint(value):
    if value in (None, "isotropic", _lib.FOOTPRINT_ISOTROPIC):
        return _lib.FOOTPRINT_ISOTROPIC
    if value in ("disc", _lib.FOOTPRINT_DISC):
        return _lib.FOOTPRINT_DISC
    raise SplatError(-1, f"footprint must be 'isotropic' or 'disc', not {value!r}")


class SplatProjector:
    """src/SplatProjector.ts:5-203.

    footprint="disc" (extension): project SequentialRenderer's oriented disc instead of the
    isotropic screen-space Gaussian — project() then needs the normals, the ProjectedSplat bounds are the disc's
    exact screen extent and getDiscBuffer() holds the 32-byte disc records the composite evaluates."""

    def __init__(self, device, numSplats, footprint="isotropic"):
        self.device, self.numSplats = device, numSplats
        self.footprint = _footprint(footprint)
        self.projectedBuffer = device.createBuffer(numSplats * 32)  # :19-23
        self.contents = "projected"  # what projectedBuffer holds: ProjectedSplat records, or a frame's lit composite records
        self.discBuffer = device.createBuffer(numSplats * 32) if self.footprint == _lib.FOOTPRINT_DISC else None

    def project(self, commandEncoder, uniformBuffer, splatPropertyBuffer, keysBuffer=None, payloadBuffer=None,
                paddedSize=0, normalsBuffer=None):  # :174-194
        """keysBuffer/payloadBuffer (extension): fuse DepthKeyExtractor.extract into the same kernel."""
        d = self.device
        u = _uniform_floats(uniformBuffer)
        if u.shape[0] < 22:
            raise SplatError(-1, "uniform block needs 22 floats (VP, eye, time, screenW, screenH)")
        uptr = u.ctypes.data_as(C.POINTER(C.c_float))
        keys, payload = keysBuffer.ptr if keysBuffer else None, payloadBuffer.ptr if payloadBuffer else None
        d.lastProjector = self
        self.contents = "projected"
        if self.footprint == _lib.FOOTPRINT_DISC:
            if normalsBuffer is None:
                raise SplatError(-1, "SplatProjector(footprint='disc').project needs normalsBuffer")
            check(d.lib.splat_project_disc(d.ctx, uptr, splatPropertyBuffer.ptr, 2, normalsBuffer.ptr, 1, self.numSplats,
                                           self.projectedBuffer.ptr, self.discBuffer.ptr, keys, payload, paddedSize), d.ctx)
            return
        check(d.lib.splat_project(d.ctx, uptr, splatPropertyBuffer.ptr, 2, self.numSplats, self.projectedBuffer.ptr, keys, payload,
                                  paddedSize), d.ctx)

    def getProjectedBuffer(self):  # :196-198
        """The reference's ProjectedSplat records {min.xy, max.xy | depth, radius, originalIndex, pad}
        (src/SplatProjector.ts:119-128).  Raises when the last frame left the 32-byte LIT composite records {centre.xy,
        radius, depth | lit rgb, opacity} here instead (Renderer(records='lit'), the default of the whole-frame facade):
        code written against the reference's layout must not read those by accident — ask for them with
        getRecordsBuffer(), or construct Renderer(records='projected')."""
        if self.contents == "lit":
            raise SplatError(-5, "the projector's buffer holds lit composite records (Renderer records='lit'), not ProjectedSplat "
                                 "records: use getRecordsBuffer() and Renderer.recordFormat, or Renderer(records='projected')")
        return self.projectedBuffer

    def getRecordsBuffer(self):
        """The record buffer whatever the last frame wrote into it (`contents`: 'projected' | 'lit')."""
        return self.projectedBuffer

    def getDiscBuffer(self):
        if self.discBuffer is None:
            raise SplatError(-5, "getDiscBuffer: this projector was not created with footprint='disc'")
        return self.discBuffer

    def destroy(self):  # :200-202
        if self.device.lastProjector is self:
            self.device.lastProjector = None
        self.projectedBuffer.destroy()
        if self.discBuffer is not None:
            self.discBuffer.destroy()


class DepthKeyExtractor:
    """src/DepthKeyExtractor.ts:5-115."""

    def __init__(self, device):
        self.device = device

    def extract(self, commandEncoder, projectedBuffer, keysBuffer, payloadBuffer, numSplats, paddedSize):  # :71-109
        d = self.device
        check(d.lib.splat_extract_keys(d.ctx, projectedBuffer.ptr, numSplats, paddedSize, keysBuffer.ptr, payloadBuffer.ptr),
              d.ctx)

    def cleanupTempBuffers(self):  # :111-114 (no per-call uniform buffers exist here)
        pass


class RadixSorter:
    """src/RadixSorter.ts:21-301 — stable ascending sort of (key, payload) u32 pairs."""

    def __init__(self, device, numSplats):
        self.device, self.numSplats = device, numSplats
        p = C.c_void_p()
        check(device.lib.splat_sort_create(device.ctx, numSplats, C.byref(p)), device.ctx)
        self._s = p
        self.paddedSize = device.lib.splat_sort_capacity(p)  # :46-52

    def sort(self, numKeys=None, bitBegin=0, bitEnd=32):  # :197-264
        d = self.device
        n = self.numSplats if numKeys is None else numKeys
        check(d.lib.splat_sort_run(self._s, n, bitBegin, bitEnd), d.ctx)

    def setMode(self, mode):
        """0 = rank equal digits as the context's policy says (default), 2 = always with ballots."""
        check(self.device.lib.splat_sort_set_mode(self._s, mode), self.device.ctx)

    def getSortedIndicesBuffer(self):  # :269-271
        return Buffer(self.device, self.device.lib.splat_sort_sorted_payload(self._s), self.paddedSize * 4, owned=False)

    def getSortedKeysBuffer(self):
        return Buffer(self.device, self.device.lib.splat_sort_sorted_keys(self._s), self.paddedSize * 4, owned=False)

    def getKeysBuffer(self):  # :273-275
        return Buffer(self.device, self.device.lib.splat_sort_keys(self._s), self.paddedSize * 4, owned=False)

    def getPayloadBuffer(self):  # :277-279
        return Buffer(self.device, self.device.lib.splat_sort_payload(self._s), self.paddedSize * 4, owned=False)

    def cleanupTempBuffers(self):  # :281-284
        pass

    def destroy(self):  # :286-300
        if self._s:
            self.device.lib.splat_sort_destroy(self._s)
            self._s = None


class PrefixSumScanner:
    """src/PrefixSumScanner.ts:8-168 — exclusive scan; never leaves the device here."""

    def __init__(self, device):
        self.device = device

    def scan(self, commandEncoder, inputBuffer, outputBuffer, numElements, totalBuffer=None):  # :74-87
        d = self.device
        check(d.lib.splat_scan_u32(d.ctx, inputBuffer.ptr, outputBuffer.ptr, numElements,
                                   totalBuffer.ptr if totalBuffer else None), d.ctx)

    def cleanupTempBuffers(self):  # :164-167
        pass


class GPUTileBinner:
    """src/GPUTileBinner.ts:11-378."""

    def __init__(self, device, tileSize):
        self.device, self.tileSize = device, tileSize
        p = C.c_void_p()
        check(device.lib.splat_bin_create(device.ctx, tileSize, C.byref(p)), device.ctx)
        self._b = p
        self.prefixSumScanner = PrefixSumScanner(device)  # :49
        self._tiles = 0

    def binSplats(self, commandEncoder, projectedBuffer, sortedIndicesBuffer, numSplats, screenWidth, screenHeight,
                  tileRow0=0, tileRow1=U32_MAX, numSorted=None):  # :190-338
        d = self.device
        n_sorted = numSplats if numSorted is None else numSorted
        check(d.lib.splat_bin_run(self._b, projectedBuffer.ptr, numSplats, sortedIndicesBuffer.ptr, n_sorted, screenWidth,
                                  screenHeight, tileRow0, tileRow1), d.ctx)
        self._tiles = -(-screenWidth // self.tileSize) * -(-screenHeight // self.tileSize)
        d.lastBinner = self

    def _get(self, fn, size):
        p = C.c_void_p()
        rc = fn(self._b, C.byref(p))
        if rc != 0:
            # the reference throws Error("Tile ... buffer not initialized") (:340-359)
            raise SplatError(rc, self.device.lib.splat_last_error(self.device.ctx).decode())
        return Buffer(self.device, p.value, size, owned=False)

    def getTileOffsetsBuffer(self):  # :340-345
        return self._get(self.device.lib.splat_bin_offsets, self._tiles * 4)

    def getTileIndicesBuffer(self):  # :347-352
        buf = self._get(self.device.lib.splat_bin_indices, 4)  # raises "not initialized" first
        buf.size = max(self.getTotalIndices(), 1) * 4  # "at least 4 bytes" (:288)
        return buf

    def getTileCountsBuffer(self):  # :354-359
        return self._get(self.device.lib.splat_bin_counts, self._tiles * 4)

    def getTotalIndices(self):
        t = C.c_uint64()
        check(self.device.lib.splat_bin_total(self._b, C.byref(t)), self.device.ctx)
        return int(t.value)

    def setFrameOrder(self, order):
        """Order of work of the whole-frame call: "sortFirst" (global depth sort, bin in sorted order),
        "tileFirst" (bin in index order, PerTileSorter-style depth sort of every tile's list) or
        "default".  Same lists either way."""
        code = {"default": -1, "sortFirst": 0, "tileFirst": 1}[order] if isinstance(order, str) else int(order)
        check(self.device.lib.splat_bin_set_frame_order(self._b, code), self.device.ctx)

    def getTileSize(self):  # :361-363
        return self.tileSize

    def cleanupTempBuffers(self):  # :365-369
        self.prefixSumScanner.cleanupTempBuffers()

    def destroy(self):  # :371-377
        if self.device.lastBinner is self:
            self.device.lastBinner = None
        if self._b:
            self.device.lib.splat_bin_destroy(self._b)
            self._b = None


class PerTileSorter:
    """src/PerTileSorter.ts:6-223.  The reference's per-tile LDS sort is racy and capped at 2048
    entries.  As a stage of the staged API it has nothing to reorder: binSplats bins an
    already sorted order, so every list leaves GPUTileBinner in (depth key, index) order; sort() keeps
    the reference's argument list and, with validate=True, runs the order CHECK on the device and
    returns the number of out-of-order neighbours (0).  The real per-tile depth sort (any list length,
    stable) lives inside the whole-frame call: Renderer.render bins in index order and sorts every
    tile's list in LDS (csrc/tile_first.hip, k_tile_sort)."""

    def __init__(self, device, validate=False):
        self.device, self.validate = device, validate
        self.violations = 0

    def sort(self, commandEncoder, projectedBuffer, tileListsBuffer, tileOffsetsBuffer, splatIndicesBuffer, numTiles,
             maxSplatsPerTile, totalPairs=None):  # :174-213
        if not self.validate:
            return None
        d = self.device
        if totalPairs is None:
            totalPairs = splatIndicesBuffer.size // 4
        v = C.c_uint64()
        check(d.lib.splat_validate_tile_order(d.ctx, projectedBuffer.ptr, tileOffsetsBuffer.ptr, numTiles,
                                              splatIndicesBuffer.ptr, totalPairs, C.byref(v)), d.ctx)
        self.violations = int(v.value)
        return self.violations

    def cleanupTempBuffers(self):
        pass

    def destroy(self):
        pass


class ComputeShaderRenderer:
    """src/ComputeShaderRenderer.ts:5-469 — the per-pixel composite.  The canvas blit (:268-338,
    :425-456) is out of scope (no canvas); the rgba8unorm output texture is exposed instead."""

    def __init__(self, device, context=None, presentationFormat="rgba8unorm", mode=_lib.MODE_FRONT_TO_BACK,
                 earlyOut=True, footprint="isotropic", recordFormat=_lib.RECORDS_PROJECTED):
        self.device = device
        self.mode, self.earlyOut = mode, earlyOut
        # recordFormat=RECORDS_LIT32: projectedBuffer in render() holds lit composite records (what a Renderer with
        # records="lit" leaves in its projector's buffer); colours and normals are then not read
        self.recordFormat = recordFormat
        # footprint="disc": projectedBuffer in render() is SplatProjector(footprint="disc").getDiscBuffer()
        self.footprint = _footprint(footprint)
        self.outputTexture = None
        self.outputFloat = None
        self._wh = (0, 0)
        self.tileRows = (0, U32_MAX)
        self.consumedBuffer = None

    def ensureOutputTexture(self, width, height, wantFloat=False):  # :340-360
        if self._wh != (width, height):
            if self.outputTexture:
                self.outputTexture.destroy()
            if self.outputFloat:
                self.outputFloat.destroy()
                self.outputFloat = None
            self.outputTexture = self.device.createBuffer(width * height * 4)
            self._wh = (width, height)
        if wantFloat and self.outputFloat is None:
            self.outputFloat = self.device.createBuffer(width * height * 16)

    def render(self, uniformData, splatPropertyBuffer, splatIndicesBuffer, curvatureBuffer, projectedBuffer,
               tileListsBuffer, tileOffsetsBuffer, tileSize, numTilesX, width, height, wantFloat=False):  # :362-462
        d = self.device
        if numTilesX != -(-width // tileSize):
            raise SplatError(-1, "numTilesX does not match ceil(width / tileSize)")
        self.ensureOutputTexture(width, height, wantFloat)
        cfg = CompositeCfg(self.mode, int(self.earlyOut), tileSize, self.tileRows[0], self.tileRows[1], self.recordFormat, 0,
                           self.footprint)
        check(d.lib.splat_composite(d.ctx, C.byref(cfg), splatPropertyBuffer.ptr + 16, 2, curvatureBuffer.ptr, 1,
                                    projectedBuffer.ptr, splatIndicesBuffer.ptr, tileListsBuffer.ptr, tileOffsetsBuffer.ptr,
                                    width, height, self.outputTexture.ptr,
                                    self.outputFloat.ptr if (wantFloat and self.outputFloat) else None,
                                    self.consumedBuffer.ptr if self.consumedBuffer else None), d.ctx)

    def readPixels(self):
        w, h = self._wh
        return self.outputTexture.read(np.uint8).reshape(h, w, 4)

    def readPixelsFloat(self):
        w, h = self._wh
        return self.outputFloat.read(np.float32).reshape(h, w, 4)

    def destroy(self):  # :464-468
        if self.outputTexture:
            self.outputTexture.destroy()
        if self.outputFloat:
            self.outputFloat.destroy()
        if self.consumedBuffer:
            self.consumedBuffer.destroy()
        self.outputTexture = self.outputFloat = self.consumedBuffer = None


class TileRenderer(ComputeShaderRenderer):
    """src/TileRenderer.ts:5-355.  The reference draws instanced oriented quads per tile in a CPU
    loop over a fixed-stride index layout (:291); north_star names this class for the per-pixel
    composite, so here it fronts the same HIP composite as ComputeShaderRenderer.  render() has the
    reference's eleven arguments (:234-246) and runs with nothing else: the projected records and the
    prefix-sum offsets the composite needs — which that signature does not name — are those of the
    SplatProjector and GPUTileBinner that last ran on the device (Device.lastProjector / lastBinner: their
    project() / binSplats(), or a Renderer's frame); bindTileData() overrides them.  tileCountsData is the
    reference's host array of counts per tile (Uint32Array: the reference's loop reads it on the CPU; here its
    length is checked and the device-resident counts are what the kernel reads) or, as an extension, the
    device buffer itself.  footprint="disc" gives the reference TileRenderer's own footprint (the oriented
    quad of its vertex shader, the same as SequentialRenderer's): the disc projector's getDiscBuffer() is
    then the record buffer."""

    def __init__(self, device, context=None, presentationFormat="rgba8unorm", **kw):
        self._format_given = "recordFormat" in kw
        super().__init__(device, context, presentationFormat, **kw)
        self._projected = self._offsets = self._counts = None

    def bindTileData(self, projectedBuffer, tileCountsBuffer, tileOffsetsBuffer):
        """Override: composite from these buffers (in this renderer's recordFormat) instead of the device's last projector / binner."""
        self._projected, self._counts, self._offsets = projectedBuffer, tileCountsBuffer, tileOffsetsBuffer

    def render(self, uniformData, splatPropertyBuffer, splatIndicesBuffer, curvatureBuffer, tileCountsData,
               numTilesX, numTilesY, tileSize, maxSplatsPerTile, width, height, wantFloat=False):  # :234-348
        if numTilesY != -(-height // tileSize):
            raise SplatError(-1, "numTilesY does not match ceil(height / tileSize)")
        if isinstance(tileCountsData, np.ndarray) and tileCountsData.size != numTilesX * numTilesY:
            raise SplatError(-1, "tileCountsData does not hold one count per tile (numTilesX * numTilesY)")
        if self._projected is not None:
            projected, counts, offsets = self._projected, self._counts, self._offsets
        else:
            p, b = self.device.lastProjector, self.device.lastBinner
            if p is None or b is None:
                raise SplatError(-5, "TileRenderer.render: no SplatProjector / GPUTileBinner has run on this device yet (and "
                                     "bindTileData was not called): there are no projected records and tile offsets to composite from")
            if self.footprint == _lib.FOOTPRINT_DISC:
                projected = p.getDiscBuffer()
            else:
                projected = p.getRecordsBuffer()
                if not self._format_given:  # what that projector left there: a frame's lit composite records, or ProjectedSplat records
                    self.recordFormat = _lib.RECORDS_LIT32 if p.contents == "lit" else _lib.RECORDS_PROJECTED
            counts = tileCountsData if isinstance(tileCountsData, Buffer) else b.getTileCountsBuffer()
            offsets = b.getTileOffsetsBuffer()
        return ComputeShaderRenderer.render(self, uniformData, splatPropertyBuffer, splatIndicesBuffer, curvatureBuffer,
                                            projected, counts, offsets, tileSize, numTilesX, width, height, wantFloat)


class SequentialRenderer:
    """src/SequentialRenderer.ts:5-321 — the ordering-exact path: one draw per splat in the order of a
    caller-supplied sorted index buffer (:268-307), each an oriented quad in the tangent plane of the
    splat's normal with a Gaussian cut at the unit disc (:91-142).  The reference does this with the
    hardware rasteriser, one draw call per splat; here the given order is binned and composited per
    pixel by the HIP kernel with the same footprint (footprint="disc", the default: the rasteriser's
    perspective-correct uv is the inverse plane-to-screen homography, evaluated per pixel), nearest
    first: pass near-to-far indices (RadixSorter's order) — the image is the one the reference's
    blend state (:189-200) gives for the reversed, back-to-front order.  Compared with the oracle's
    software rasteriser in tests/ (<= 1e-4 per channel, except pixels within 1e-3 of a disc's rim,
    where the discard is a step of 0.044).  footprint="isotropic" composites the same order with
    ComputeShaderRenderer's screen-space Gaussian instead."""

    def __init__(self, device, context=None, presentationFormat="rgba8unorm", numSplats=0, tileSize=16, footprint="disc",
                 earlyOut=True):
        self.device, self.numSplats, self.tileSize = device, numSplats, tileSize
        self.projector = SplatProjector(device, numSplats, footprint)
        self.binner = GPUTileBinner(device, tileSize)
        self.compositor = ComputeShaderRenderer(device, context, presentationFormat, earlyOut=earlyOut, footprint=footprint)

    def render(self, uniformData, splatPropertyBuffer, sortedIndexBuffer, curvatureBuffer, width, height,
               wantFloat=False):  # :233-314
        u = _uniform_floats(uniformData).copy()
        if u.shape[0] < 22:
            u = np.concatenate([u[:20], np.array([width, height], np.float32)])
        disc = self.projector.footprint == _lib.FOOTPRINT_DISC
        self.projector.project(None, u, splatPropertyBuffer, normalsBuffer=curvatureBuffer if disc else None)
        self.binner.binSplats(None, self.projector.getProjectedBuffer(), sortedIndexBuffer, self.numSplats, width, height)
        records = self.projector.getDiscBuffer() if disc else self.projector.getProjectedBuffer()
        self.compositor.render(u, splatPropertyBuffer, self.binner.getTileIndicesBuffer(), curvatureBuffer, records,
                               self.binner.getTileCountsBuffer(), self.binner.getTileOffsetsBuffer(), self.tileSize,
                               -(-width // self.tileSize), width, height, wantFloat)

    def readPixels(self):
        return self.compositor.readPixels()

    def readPixelsFloat(self):
        return self.compositor.readPixelsFloat()

    def destroy(self):  # :316-320
        self.projector.destroy()
        self.binner.destroy()
        self.compositor.destroy()


class Renderer:
    """src/Renderer.ts:13,250,311 keeps its name as the whole-frame facade: the app's render call
    (src/main.ts:183-190) becomes one call that runs project -> keys -> sort -> bin -> composite
    on the device.  (The reference's body — opaque depth-tested quads — is out of scope.)"""

    def __init__(self, device, context=None, presentationFormat="rgba8unorm", numPoints=0, tileSize=16,
                 mode=_lib.MODE_FRONT_TO_BACK, earlyOut=True, frameOrder=None, footprint="isotropic", writeProjected=True,
                 records="lit"):
        # footprint="disc": the frame is drawn with SequentialRenderer's oriented discs (normalsBuffer is then
        # required in render() even with pre-lit planes: the projector reads it).  writeProjected=False (disc
        # frames only): the ProjectedSplat records, which a disc frame's composite does not read, are not written.
        # records (isotropic frames): what the frame's projector leaves in projector.getProjectedBuffer() and the
        # composite gathers per staged list entry — "lit" (default): the 32-byte lit composite records (centre,
        # radius, depth | lit colour; SPLAT_RECORDS_LIT32), ONE line per entry; "projected": the reference's
        # ProjectedSplat records, with colour (and normal) gathered from the property buffers as the reference does.
        # Same image bit for bit.  Screens beyond 256 x 256 tiles always use "projected".
        self.footprint = _footprint(footprint)
        if not writeProjected and self.footprint != _lib.FOOTPRINT_DISC:
            raise SplatError(-1, "writeProjected=False: the isotropic composite reads the records the projector writes")
        if records not in ("lit", "projected"):
            raise SplatError(-1, f"records must be 'lit' or 'projected', not {records!r}")
        # (disc frames, "lit": the lit colour rides behind each disc record — 48-byte records inside the binner, one gathered
        # record per staged entry instead of disc record + colour + normal; the ProjectedSplat buffer is what it always was)
        self.records = records
        self.writeProjected = writeProjected
        self.device, self.numPoints, self.tileSize = device, numPoints, tileSize
        self.projector = SplatProjector(device, numPoints)
        self.sorter = RadixSorter(device, numPoints)
        self.binner = GPUTileBinner(device, tileSize)
        if frameOrder is not None:
            self.binner.setFrameOrder(frameOrder)
        self.mode, self.earlyOut = mode, earlyOut
        self.output = None
        self.outputFloat = None
        self._wh = (0, 0)
        self._last = None
        self.previousFrameOverflowed = False
        # frames whose tile lists failed the per-tile sort's order check (SPLAT_ERR_RETRY) and were rendered again with
        # ballots: NOT a capacity event — the context has changed its ranking for good (Device.rankStatus())
        self.framesMisranked = 0

    def _again(self, rc):
        """Books a SPLAT_ERR_CAPACITY / SPLAT_ERR_RETRY report about the previous sync-free frame."""
        if rc == _lib.ERR_RETRY:
            self.framesMisranked += 1
        else:
            self.previousFrameOverflowed = True

    def render(self, uniformData, propertyBuffer, normalsBuffer, scaleFactorsBuffer, width, height, tileRows=(0, U32_MAX),
               wantFloat=False):
        d = self.device
        u = _uniform_floats(uniformData).copy()
        if u.shape[0] < 22:
            u = np.concatenate([u[:20], np.array([width, height], np.float32)])
        if self._wh != (width, height):
            if self.output:
                self.output.destroy()
            if self.outputFloat:
                self.outputFloat.destroy()
                self.outputFloat = None
            self.output = d.createBuffer(width * height * 4)
            self._wh = (width, height)
        if wantFloat and self.outputFloat is None:
            self.outputFloat = d.createBuffer(width * height * 16)
        prelit = isinstance(propertyBuffer, PropertyPlanes) and propertyBuffer.prelit
        ts = self.tileSize
        lit = self.records == "lit" and -(-width // ts) <= 256 and -(-height // ts) <= 256
        # what the FRAME composites from (a disc frame with "lit": 48-byte lit disc records inside the binner) ...
        frame_format = self.frameRecordFormat = _lib.RECORDS_LIT32 if lit else _lib.RECORDS_PROJECTED
        # ... and what projector.getRecordsBuffer() holds after this frame — what a caller passes, with this format, to the staged
        # composite: lit composite records for an isotropic "lit" frame, ProjectedSplat records otherwise (a disc frame's too)
        iso_lit = lit and self.footprint == _lib.FOOTPRINT_ISOTROPIC
        self.recordFormat = _lib.RECORDS_LIT32 if iso_lit else _lib.RECORDS_PROJECTED
        self.projector.contents = "lit" if iso_lit else "projected"
        cfg = CompositeCfg(self.mode, int(self.earlyOut), self.tileSize, tileRows[0], tileRows[1], frame_format, int(prelit),
                           self.footprint)
        tail = (normalsBuffer.ptr if normalsBuffer is not None else None, self.numPoints, width, height,
                self.projector.projectedBuffer.ptr if self.writeProjected else None, self.output.ptr,
                self.outputFloat.ptr if wantFloat else None)
        head = (d.ctx, self.sorter._s, self.binner._b, C.byref(cfg), u.ctypes.data_as(C.POINTER(C.c_float)))
        if isinstance(propertyBuffer, PropertyPlanes):  # the native layout: SplatPropertyManager.getPropertyPlanes()
            fn, args = d.lib.splat_render_frame_planes, head + (propertyBuffer.posRadius.ptr, propertyBuffer.colorOpacity.ptr) + tail
        else:  # the reference's interleaved records
            fn, args = d.lib.splat_render_frame, head + (propertyBuffer.ptr,) + tail
        self._last = (fn, args, u, cfg)  # keeps u/cfg alive; finish() may have to render this frame again
        rc = fn(*args)
        if rc in _lib.RENDER_AGAIN:  # about the PREVIOUS (sync-free) frame: it outgrew its pair limit (room was made) or misranked
            self._again(rc)
            rc = fn(*args)
        check(rc, d.ctx)
        self.binner._tiles = -(-width // self.tileSize) * -(-height // self.tileSize)
        d.lastProjector, d.lastBinner = self.projector, self.binner
        return self.output

    def finish(self):
        """Settles a sync-free frame: waits for its pair total and, if the frame outgrew the limit
        sized from the frame before it, renders it again (now with room).  Called before results
        are read."""
        d = self.device
        t = C.c_uint64()
        rc = d.lib.splat_bin_total(self.binner._b, C.byref(t))
        if rc in _lib.RENDER_AGAIN and self._last is not None:
            self._again(rc)
            check(self._last[0](*self._last[1]), d.ctx)
            rc = d.lib.splat_bin_total(self.binner._b, C.byref(t))
        check(rc, d.ctx)
        return int(t.value)

    def readPixels(self):
        self.finish()
        w, h = self._wh
        return self.output.read(np.uint8).reshape(h, w, 4)

    def readPixelsFloat(self):
        self.finish()
        w, h = self._wh
        return self.outputFloat.read(np.float32).reshape(h, w, 4)

    def destroy(self):
        self.projector.destroy()
        self.sorter.destroy()
        self.binner.destroy()
        if self.output:
            self.output.destroy()
        if self.outputFloat:
            self.outputFloat.destroy()


class PipelinedRenderer:
    """`depth` frames in flight: frame k is enqueued on stream k % depth (one splat ctx, sorter, binner and record
    buffer per stream), so the device overlaps one frame's latency-bound kernels (per-tile sort, composite prologues)
    with the other's bandwidth- and ALU-bound ones.  Frames are unchanged — each is the same kernel sequence on its own
    stream — and so is a frame's latency; frames per second rise (C2: +14 %, C1: +25 % with two in flight on one
    MI355X).  Property and normal buffers are shared (any device pointer is valid on every stream of the device):
    the caller must not rewrite them while frames that read them are in flight (finish() waits for all)."""

    def __init__(self, ordinal=0, depth=2, numPoints=0, tileSize=16, **renderer_options):
        self.devices = [Device(ordinal) for _ in range(depth)]
        self.renderers = [Renderer(d, None, "rgba8unorm", numPoints, tileSize, **renderer_options) for d in self.devices]
        self.frame = 0
        self._last = None

    def render(self, uniformData, propertyBuffer, normalsBuffer, scaleFactorsBuffer, width, height, **kw):
        r = self.renderers[self.frame % len(self.renderers)]
        self.frame += 1
        self._last = r
        return r.render(uniformData, propertyBuffer, normalsBuffer, scaleFactorsBuffer, width, height, **kw)

    def finish(self):
        for d in self.devices:
            d.sync()
        return self._last.finish() if self._last is not None else 0

    def readPixels(self):
        """Pixels of the most recently submitted frame."""
        return self._last.readPixels()

    def destroy(self):
        for r in self.renderers:
            r.destroy()
        for d in self.devices:
            d.destroy()
