"""Host Camera (product code) against the oracle's restatement of src/Camera.ts + gl-matrix."""
import json
import math
import os

import numpy as np
import pytest

import splat_renderer_amd as sr
from oracle import np_oracle as NP
from oracle import oracle as O


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("aspect", [1.0, 16 / 9, 4 / 3, 0.5])
def test_default_camera_bit_exact(aspect):
    cam = sr.Camera()
    cam.setAspect(aspect)
    vp, eye = O.camera(aspect=aspect)
    assert np.array_equal(bits(cam.getViewProjectionMatrix()), bits(vp))
    assert np.array_equal(bits(cam.getPosition()), bits(eye))
    vp2, eye2 = NP.camera(aspect=aspect)
    assert np.array_equal(bits(vp), bits(vp2)) and np.array_equal(bits(eye), bits(eye2))


def test_known_answer_default_16_9():
    with open(os.path.join(os.path.dirname(__file__), "golden", "C0_digest.json")) as f:
        d = json.load(f)["camera_default_aspect_16_9"]
    cam = sr.Camera()
    cam.setAspect(16 / 9)
    assert np.allclose(cam.getViewProjectionMatrix(), d["vp"], rtol=0, atol=0)
    assert np.allclose(cam.getPosition(), d["eye"], rtol=0, atol=0)
    # structure of a GL perspective * lookAt: eye = 3*(cos.5 sin.5, sin.5, cos.5 cos.5), w row = -view z row
    assert np.allclose(cam.getPosition(), [3 * math.cos(.5) * math.sin(.5), 3 * math.sin(.5), 3 * math.cos(.5) ** 2], atol=1e-6)


def test_verbs_match_reference_clamps():
    cam = sr.Camera()
    cam.rotate(0.0, 10.0)  # src/Camera.ts:48-50
    assert cam.elevation == pytest.approx(math.pi / 2 - 0.01)
    cam.rotate(0.0, -20.0)
    assert cam.elevation == pytest.approx(-(math.pi / 2 - 0.01))
    cam.zoom(100)          # :57
    assert cam.distance == 20.0
    cam.zoom(-100)
    assert cam.distance == 0.5
    cam = sr.Camera()
    cam.rotate(0.3, -0.2)
    cam.zoom(1.0)
    vp, eye = O.camera(distance=4.0, azimuth=0.8, elevation=0.3)
    assert np.array_equal(bits(cam.getViewProjectionMatrix()), bits(vp))


def test_pan_moves_target_in_the_view_plane():
    cam = sr.Camera()
    before = cam.getPosition().copy()
    cam.pan(0.5, -0.25)
    assert np.linalg.norm(cam.target) == pytest.approx(math.hypot(0.5, 0.25), rel=1e-5)
    fwd = -before / np.linalg.norm(before)
    assert abs(float(np.dot(cam.target, fwd))) < 1e-6  # moved perpendicular to the view direction
    vp, eye = O.camera(target=cam.target)
    assert np.array_equal(bits(cam.getViewProjectionMatrix()), bits(vp))


def test_uniform_block_layout():
    cam = sr.Camera()
    cam.setAspect(2.0)
    u = cam.uniforms(640, 320, time=1.5)
    assert u.shape == (22,) and u[19] == 1.5 and u[20] == 640 and u[21] == 320  # main.ts:126-144, SplatProjector.ts:35-41
    assert np.array_equal(u[:16], cam.getViewProjectionMatrix()) and np.array_equal(u[16:19], cam.getPosition())
