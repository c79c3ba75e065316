#!/usr/bin/env python3
"""Headless turntable: orbits the reference's Camera around the synthetic scene and writes PNG (or
PPM) frames — the frame loop of /root/reference/src/main.ts:110-193 without a browser.

    python tools/turntable.py --config C1 --frames 8 --out gpurun_out/turntable
"""
import argparse
import os
import struct
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splat_renderer_amd as sr


def write_png(path, rgba):
    h, w, _ = rgba.shape
    raw = b"".join(b"\x00" + rgba[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 3)) + chunk(b"IEND", b""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C1", choices=sorted(sr.scene.CONFIGS))
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--out", default="gpurun_out/turntable")
    ap.add_argument("--footprint", default="isotropic", choices=["isotropic", "disc"],
                    help="isotropic = ComputeShaderRenderer's Gaussian, disc = SequentialRenderer's oriented disc")
    args = ap.parse_args()
    n, w, h = sr.scene.CONFIGS[args.config]
    props, normals = sr.scene.make_scene(n)
    dev = sr.Device(0)
    pbuf, nbuf = dev.createBufferFrom(props), dev.createBufferFrom(normals)
    r = sr.Renderer(dev, None, "rgba8unorm", n, footprint=args.footprint)
    cam = sr.Camera()
    cam.setAspect(w / h)
    os.makedirs(args.out, exist_ok=True)
    t0 = time.perf_counter()
    for f in range(args.frames):
        r.render(cam.uniforms(w, h, time=f / 60.0), pbuf, nbuf, None, w, h)
        img = r.readPixels()
        write_png(os.path.join(args.out, f"frame_{f:03d}.png"), img)
        cam.rotate(2 * np.pi / args.frames, 0.0)
    dt = time.perf_counter() - t0
    print(f"{args.frames} frames of {args.config} ({n} splats @{w}x{h}) rendered + read back + encoded in {dt:.2f} s -> {args.out}")


if __name__ == "__main__":
    main()
