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


def test_png_round_trip_and_foreign_filters(tmp_path):
    """write_png -> read_png gives the same bytes; read_png also undoes the scanline filters it never writes
    (sub / up / average / Paeth), checked on a file built here with every filter type, and rejects a corrupted file."""
    import struct
    import zlib
    from splat_renderer_amd import read_png, write_png
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    img[..., 3] = 255
    p = tmp_path / "a.png"
    write_png(p, img)
    assert np.array_equal(read_png(p), img)
    # the same image filtered by hand, one filter type per row in turn
    h, w, _ = img.shape
    flat = img.reshape(h, w * 4).astype(np.int32)
    raw = bytearray()
    for y in range(h):
        ft = y % 5
        up = flat[y - 1] if y else np.zeros(w * 4, np.int32)
        left = np.concatenate([np.zeros(4, np.int32), flat[y, :-4]])
        upleft = np.concatenate([np.zeros(4, np.int32), up[:-4]])
        if ft == 0:
            pred = np.zeros(w * 4, np.int32)
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = up
        elif ft == 3:
            pred = (left + up) >> 1
        else:
            pp = left + up - upleft
            pa, pb, pc = np.abs(pp - left), np.abs(pp - up), np.abs(pp - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
        raw += bytes([ft]) + ((flat[y] - pred) & 255).astype(np.uint8).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    q = tmp_path / "b.png"
    blob = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b"")
    q.write_bytes(blob)
    assert np.array_equal(read_png(q), img)
    bad = bytearray(blob)
    bad[60] ^= 0xFF
    q.write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        read_png(q)


def test_orbit_camera_controller_maps_events_like_the_reference():
    """src/OrbitCameraController.ts:35-70: left drag rotates by 0.005 rad per pixel (y inverted), middle / right drag
    pans by 0.002 per pixel (x inverted), the wheel zooms by 0.001 per unit; nothing moves without a button down."""
    from splat_renderer_amd import Camera, MouseEvent, OrbitCameraController
    cam, ref = Camera(), Camera()
    ctl = OrbitCameraController(cam, canvas=None)
    ctl.onMouseMove(MouseEvent(50, 50))  # not dragging
    assert (cam.azimuth, cam.elevation) == (ref.azimuth, ref.elevation)
    ctl.onMouseDown(MouseEvent(100, 100, button=0))
    ctl.onMouseMove(MouseEvent(140, 90))
    ref.rotate(40 * 0.005, 10 * 0.005)
    assert (cam.azimuth, cam.elevation) == (ref.azimuth, ref.elevation)
    ctl.onMouseUp()
    for button in (1, 2):
        ctl.onMouseDown(MouseEvent(10, 10, button=button))
        ctl.onMouseMove(MouseEvent(25, 4))
        ref.pan(-15 * 0.002, -6 * 0.002)
        ctl.onMouseUp()
    assert np.array_equal(cam.target, ref.target)
    ctl.onWheel(MouseEvent(deltaY=250.0))
    ref.zoom(0.25)
    assert cam.distance == ref.distance
    assert np.array_equal(cam.uniforms(64, 64).view(np.uint32), ref.uniforms(64, 64).view(np.uint32))
