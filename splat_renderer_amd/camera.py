"""Camera — host-side mirror of /root/reference/src/Camera.ts (same fields, same verbs).

gl-matrix 3.4.4 semantics are restated (the dependency is not vendored in the reference:
package-lock.json:866-871): vectors and matrices live in Float32Array (every store rounds to
f32), arithmetic is f64, matrices are column-major, mat4.perspective is the GL [-1,1]-z form.
"""
import math

import numpy as np

_EPS = 0.000001


def _vec3(x, y, z):
    return np.array([x, y, z], dtype=np.float32)  # vec3.fromValues -> Float32Array


def _look_at(eye, center, up):
    ex, ey, ez = (float(v) for v in eye)
    cx, cy, cz = (float(v) for v in center)
    ux, uy, uz = (float(v) for v in up)
    out = np.zeros(16, np.float32)
    if abs(ex - cx) < _EPS and abs(ey - cy) < _EPS and abs(ez - cz) < _EPS:
        out[0] = out[5] = out[10] = out[15] = 1.0
        return out
    z0, z1, z2 = ex - cx, ey - cy, ez - cz
    ln = 1.0 / math.sqrt(z0 * z0 + z1 * z1 + z2 * z2)
    z0, z1, z2 = z0 * ln, z1 * ln, z2 * ln
    x0, x1, x2 = uy * z2 - uz * z1, uz * z0 - ux * z2, ux * z1 - uy * z0
    ln = math.sqrt(x0 * x0 + x1 * x1 + x2 * x2)
    if not ln:
        x0 = x1 = x2 = 0.0
    else:
        ln = 1.0 / ln
        x0, x1, x2 = x0 * ln, x1 * ln, x2 * ln
    y0, y1, y2 = z1 * x2 - z2 * x1, z2 * x0 - z0 * x2, z0 * x1 - z1 * x0
    ln = math.sqrt(y0 * y0 + y1 * y1 + y2 * y2)
    if not ln:
        y0 = y1 = y2 = 0.0
    else:
        ln = 1.0 / ln
        y0, y1, y2 = y0 * ln, y1 * ln, y2 * ln
    out[:] = [x0, y0, z0, 0.0, x1, y1, z1, 0.0, x2, y2, z2, 0.0,
              -(x0 * ex + x1 * ey + x2 * ez), -(y0 * ex + y1 * ey + y2 * ez), -(z0 * ex + z1 * ey + z2 * ez), 1.0]
    return out


def _perspective(fovy, aspect, near, far):
    f = 1.0 / math.tan(fovy / 2.0)
    out = np.zeros(16, np.float32)
    out[0] = f / aspect
    out[5] = f
    out[11] = -1.0
    if far is not None and far != math.inf:
        nf = 1.0 / (near - far)
        out[10] = (far + near) * nf
        out[14] = 2.0 * far * near * nf
    else:
        out[10] = -1.0
        out[14] = -2.0 * near
    return out


def _multiply(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    out = np.zeros(16, np.float32)
    for c in range(4):
        b0, b1, b2, b3 = b[c * 4:c * 4 + 4]
        for k in range(4):
            out[c * 4 + k] = b0 * a[k] + b1 * a[4 + k] + b2 * a[8 + k] + b3 * a[12 + k]
    return out


class Camera:
    """Orbit camera (src/Camera.ts:3-139). Public fields and verbs keep the reference's names."""

    def __init__(self):
        self.target = _vec3(0, 0, 0)   # :24
        self.distance = 3.0
        self.azimuth = 0.5
        self.elevation = 0.5
        self.fov = 45
        self.aspect = 1.0
        self.near = 0.1
        self.far = 100.0
        self._view = np.zeros(16, np.float32)
        self._proj = np.zeros(16, np.float32)
        self._vp = np.zeros(16, np.float32)
        self._pos = _vec3(0, 0, 0)
        self._dirty = True

    def setAspect(self, aspect):  # :38-41
        self.aspect = aspect
        self._dirty = True

    def rotate(self, deltaAzimuth, deltaElevation):  # :43-53
        self.azimuth += deltaAzimuth
        self.elevation += deltaElevation
        max_el = math.pi / 2 - 0.01
        self.elevation = max(-max_el, min(max_el, self.elevation))
        self._dirty = True

    def zoom(self, deltaDistance):  # :55-59
        self.distance += deltaDistance
        self.distance = max(0.5, min(20.0, self.distance))
        self._dirty = True

    def pan(self, deltaX, deltaY):  # :61-83
        position = self._camera_position().astype(np.float64)
        tgt = self.target.astype(np.float64)

        def norm32(v):
            v = np.asarray(v, np.float32).astype(np.float64)  # results are stored in Float32Array
            ln = float(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
            if ln > 0:
                v = v * (1.0 / math.sqrt(ln))
            return v.astype(np.float32).astype(np.float64)

        def cross32(a, b):
            return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]],
                            np.float32).astype(np.float64)

        forward = norm32(tgt - position)
        right = norm32(cross32(forward, np.array([0.0, 1.0, 0.0])))
        up = norm32(cross32(right, forward))
        offset = np.zeros(3, np.float32)
        offset = (offset.astype(np.float64) + right * deltaX).astype(np.float32)
        offset = (offset.astype(np.float64) + up * deltaY).astype(np.float32)
        self.target = (self.target.astype(np.float64) + offset.astype(np.float64)).astype(np.float32)
        self._dirty = True

    def _camera_position(self):  # :85-95
        x = self.distance * math.cos(self.elevation) * math.sin(self.azimuth)
        y = self.distance * math.sin(self.elevation)
        z = self.distance * math.cos(self.elevation) * math.cos(self.azimuth)
        return _vec3(float(self.target[0]) + x, float(self.target[1]) + y, float(self.target[2]) + z)

    def _update(self):  # :97-128
        if not self._dirty:
            return
        self._pos = self._camera_position()
        self._view = _look_at(self._pos, self.target, _vec3(0, 1, 0))
        self._proj = _perspective((self.fov * math.pi) / 180, self.aspect, self.near, self.far)
        self._vp = _multiply(self._proj, self._view)
        self._dirty = False

    def getViewProjectionMatrix(self):  # :130-133
        self._update()
        return self._vp

    def getPosition(self):  # :135-138
        self._update()
        return self._pos

    def uniforms(self, width, height, time=0.0):
        """The 22-float frame block: VP | eye | time (src/main.ts:126-144) | screenW, screenH
        (src/SplatProjector.ts:35-41)."""
        u = np.zeros(22, np.float32)
        u[:16] = self.getViewProjectionMatrix()
        u[16:19] = self.getPosition()
        u[19] = time
        u[20] = width
        u[21] = height
        return u
