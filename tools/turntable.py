#!/usr/bin/env python3
"""Headless turntable: orbits the reference's Camera around the synthetic scene and writes PNG (or
PPM) frames — the frame loop of /root/reference/src/main.ts:110-193 without a browser.

    python tools/turntable.py --config C1 --frames 8 --out gpurun_out/turntable
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splat_renderer_amd as sr


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
    loop = sr.FrameLoop(dev, n, w, h, footprint=args.footprint)
    os.makedirs(args.out, exist_ok=True)
    t0 = time.perf_counter()
    loop.turntable(pbuf, nbuf, args.frames, lambda k, img: sr.write_png(os.path.join(args.out, f"frame_{k:03d}.png"), img))
    dt = time.perf_counter() - t0
    print(f"{args.frames} frames of {args.config} ({n} splats @{w}x{h}) rendered + read back + encoded in {dt:.2f} s -> {args.out}")


if __name__ == "__main__":
    main()
