"""Headless frame loop: the reference's requestAnimationFrame loop
(/root/reference/src/main.ts:110-193) without a browser — camera controller, frame loop, PNG files.

  * OrbitCameraController mirrors /root/reference/src/OrbitCameraController.ts: the same speeds and the
    same mapping of mouse buttons and the wheel onto Camera.rotate / pan / zoom, fed synthetic events
    (there is no canvas: `canvas` is accepted and ignored).
  * FrameLoop renders a sequence of frames through Renderer.render with a camera that may move every
    frame.  Frames are enqueued back to back (sync-free after the first); a frame's pixels are read
    only when asked for.
  * write_png / read_png: 8-bit RGBA PNG, stored with zlib (no third-party imaging library).

Nothing here computes pixels on the CPU and nothing imports oracle/.
"""
import struct
import zlib

import numpy as np

from .camera import Camera
from .host import Renderer


class MouseEvent:
    """The fields of a DOM MouseEvent / WheelEvent the controller reads."""

    def __init__(self, clientX=0, clientY=0, button=0, deltaY=0.0):
        self.clientX, self.clientY, self.button, self.deltaY = clientX, clientY, button, deltaY

    def preventDefault(self):
        pass


class OrbitCameraController:
    """src/OrbitCameraController.ts:3-75."""

    def __init__(self, camera, canvas=None):
        self.camera, self.canvas = camera, canvas
        self.isDragging, self.dragButton = False, -1      # :7-8
        self.lastMouseX = self.lastMouseY = 0             # :9-10
        self.rotationSpeed, self.panSpeed, self.zoomSpeed = 0.005, 0.002, 0.001  # :12-14

    def onMouseDown(self, event):  # :35-40
        self.isDragging, self.dragButton = True, event.button
        self.lastMouseX, self.lastMouseY = event.clientX, event.clientY

    def onMouseMove(self, event):  # :42-58
        if not self.isDragging:
            return
        dx, dy = event.clientX - self.lastMouseX, event.clientY - self.lastMouseY
        if self.dragButton == 0:  # left button: rotate
            self.camera.rotate(dx * self.rotationSpeed, -dy * self.rotationSpeed)
        elif self.dragButton in (1, 2):  # middle or right button: pan
            self.camera.pan(-dx * self.panSpeed, dy * self.panSpeed)
        self.lastMouseX, self.lastMouseY = event.clientX, event.clientY

    def onMouseUp(self, _event=None):  # :60-63
        self.isDragging, self.dragButton = False, -1

    def onWheel(self, event):  # :65-70
        event.preventDefault()
        self.camera.zoom(event.deltaY * self.zoomSpeed)

    def destroy(self):  # :72-74
        pass


def write_png(path, rgba):
    """(H, W, 4) uint8 -> an 8-bit RGBA PNG (filter 0 on every scanline)."""
    rgba = np.ascontiguousarray(rgba, np.uint8)
    h, w, c = rgba.shape
    if c != 4:
        raise ValueError("write_png wants (H, W, 4) uint8")
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rgba.reshape(h, w * 4)], axis=1).tobytes()

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 3)) + chunk(b"IEND", b""))


def read_png(path):
    """An 8-bit RGBA, non-interlaced PNG -> (H, W, 4) uint8 (all five scanline filters; every chunk's CRC is checked)."""
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, w = 8, b"", None
    while pos < len(data):
        (length,), tag = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        (crc,) = struct.unpack(">I", data[pos + 8 + length:pos + 12 + length])
        if zlib.crc32(tag + body) & 0xFFFFFFFF != crc:
            raise ValueError(f"PNG chunk {tag!r}: CRC mismatch")
        if tag == b"IHDR":
            w, h, depth, colour, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if (depth, colour, interlace) != (8, 6, 0):
                raise ValueError("read_png reads 8-bit RGBA, non-interlaced files")
        elif tag == b"IDAT":
            idat += body
        elif tag == b"IEND":
            break
        pos += 12 + length
    if w is None:
        raise ValueError("PNG without IHDR")
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * 4)
    out = np.zeros((h, w * 4), np.uint8)
    for y in range(h):
        ft, line = int(rows[y, 0]), rows[y, 1:].astype(np.int32)
        up = out[y - 1].astype(np.int32) if y else np.zeros(w * 4, np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + up) & 255
        else:  # 1 (sub), 3 (average), 4 (Paeth) depend on the pixel to the left: byte by byte
            cur = np.zeros(w * 4, np.int32)
            for x in range(w * 4):
                a = int(cur[x - 4]) if x >= 4 else 0
                b = int(up[x])
                c = int(up[x - 4]) if x >= 4 else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                elif ft == 4:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                else:
                    raise ValueError(f"PNG filter type {ft}")
                cur[x] = (int(line[x]) + pred) & 255
        out[y] = cur.astype(np.uint8)
    return out.reshape(h, w, 4)


class SdfSplatSource:
    """The producer half of the reference's frame (src/main.ts:146-180): fresh points, five rounds of
    {evaluate gradients, step onto the surface, swap}, curvature scale factors — then, for the tile-raster path,
    SplatPropertyManager.updateFromCurvature (position + colour from the normal).  step() returns the property buffer
    and the vec4(normal, scale) buffer Renderer.render takes."""

    ITERATIONS = 5  # main.ts:149

    def __init__(self, device, scene, seed=0, seeding="device"):
        from .host import PointManager, SplatPropertyManager
        from .sdf import CurvatureSampler, GradientSampler, PositionUpdater
        self.device, self.scene = device, scene
        self.pointManager = PointManager(device, scene, seed, seeding)
        n = self.numPoints = self.pointManager.getNumPoints()
        self.gradientSampler = GradientSampler(device, scene, n)
        self.curvatureSampler = CurvatureSampler(device, scene, n)
        self.positionUpdater = PositionUpdater(device, None, n)
        self.properties = SplatPropertyManager(device, n)

    def step(self, reinitialize=True, fused=True):
        """One frame's splats.  fused=True (default): splat_sdf_generate, the whole producer in one launch; False: the
        reference's thirteen stage calls through the classes above.  Same bits either way, in the same buffers."""
        pm, gs, cs = self.pointManager, self.gradientSampler, self.curvatureSampler
        gs.updateSceneParameters()  # :119-120 (the caller may have animated the primitives)
        cs.updateSceneParameters()
        if fused:
            return self._step_fused(reinitialize)
        if reinitialize:
            pm.reinitialize()       # :147
        for _ in range(self.ITERATIONS):  # :149-172
            gs.evaluateGradients(None, None, pm.getCurrentPositionBuffer())
            self.positionUpdater.updatePositions(None, None, pm.getCurrentPositionBuffer(), gs.getGradientBuffer(), pm.getNextPositionBuffer())
            pm.swap()
        cs.computeScaleFactors(None, pm.getCurrentPositionBuffer())  # :175-180
        # (the gradients are those of the last evaluation, one step behind the positions, exactly as main.ts hands
        # gradientSampler.getGradientBuffer() to its renderer at :186)
        curvature = cs.getCurvatureBuffer(gs.getGradientBuffer())
        self.properties.updateFromCurvature(None, pm.getCurrentPositionBuffer(), curvature)
        return self.properties.getPropertyBuffer(), curvature

    def _step_fused(self, reinitialize):
        import ctypes as C
        from . import sdf
        from ._lib import check
        pm, gs, cs, d = self.pointManager, self.gradientSampler, self.curvatureSampler, self.device
        if cs.curvatureBuffer is None:
            cs.curvatureBuffer = d.createBuffer(self.numPoints * 16)
        fp = C.POINTER(C.c_float)
        mn = mx = None
        seed = 0
        if reinitialize and pm.scene is not None and pm.seeding == "device":  # the fresh cloud is drawn inside the launch
            lo, hi = sdf.seeding_box(pm.scene)
            mn, mx, seed = lo.ctypes.data_as(fp), hi.ctypes.data_as(fp), int(pm._seed) & 0xFFFFFFFFFFFFFFFF
            pm._seed += 1
        elif reinitialize:
            pm.reinitialize()
        check(d.lib.splat_sdf_generate(d.ctx, C.cast(gs._program, C.c_void_p), gs._count, mn, mx, seed, pm.getCurrentPositionBuffer().ptr,
                                       self.numPoints, self.ITERATIONS, pm.getNextPositionBuffer().ptr, gs.getGradientBuffer().ptr,
                                       cs.curvatureBuffer.ptr, self.properties.getPropertyBuffer().ptr), d.ctx)
        pm.swap()  # (five swaps in the staged form: the same buffer ends up current)
        return self.properties.getPropertyBuffer(), cs.curvatureBuffer

    def destroy(self):
        for o in (self.pointManager, self.gradientSampler, self.curvatureSampler, self.properties):
            o.destroy()


class FrameLoop:
    """The render loop of src/main.ts:110-193 for the tile-raster path: per frame the camera's uniform block
    (VP, eye, time, W, H — :126-144) and one Renderer.render call (:183-190)."""

    def __init__(self, device, numPoints, width, height, tileSize=16, camera=None, **renderer_options):
        self.device, self.width, self.height = device, width, height
        self.camera = camera if camera is not None else Camera()
        self.camera.setAspect(width / height)  # resizeCanvas, main.ts:97-101
        self.renderer = Renderer(device, None, "rgba8unorm", numPoints, tileSize, **renderer_options)
        self.frame = 0

    def render(self, propertyBuffer, normalsBuffer, time=None):
        """One frame with the camera as it stands; returns the output buffer (pixels stay on the device)."""
        t = self.frame / 60.0 if time is None else time
        out = self.renderer.render(self.camera.uniforms(self.width, self.height, time=t), propertyBuffer, normalsBuffer, None,
                                   self.width, self.height)
        self.frame += 1
        return out

    def readPixels(self):
        return self.renderer.readPixels()

    def turntable(self, propertyBuffer, normalsBuffer, frames, on_frame=None):
        """`frames` frames of a full orbit (Camera.rotate by 2 pi / frames after each); on_frame(k, rgba8) gets every
        frame's pixels (e.g. to write_png them)."""
        for k in range(frames):
            self.render(propertyBuffer, normalsBuffer)
            if on_frame is not None:
                on_frame(k, self.readPixels())
            self.camera.rotate(2.0 * np.pi / frames, 0.0)

    def destroy(self):
        self.renderer.destroy()
