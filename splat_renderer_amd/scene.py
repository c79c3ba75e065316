"""Deterministic synthetic random-Gaussian scenes and the named configs.

The reference seeds its points with an unseeded Math.random on an SDF surface
(/root/reference/src/PointManager.ts:220-231) which cannot be reproduced; the bench/test scenes
are instead drawn from NumPy's PCG64 with a fixed seed, in 1M-splat chunks, in this order per
chunk: positions U[-1,1]^3, radius r0(N)*U[0.5,1.5], colour U[0,1]^3, normals = normalised
N(0,1)^3.  r0(N) = 0.04*sqrt(120000/N) keeps the total projected area at the reference's working
point (radius 0.04 at ~120k splats: src/SplatPropertyManager.ts:43).  Opacity is 1.0 (it is
ignored by the reference's footprints: src/ComputeShaderRenderer.ts:103-147).
"""
import math

import numpy as np

TILE = 16
CHUNK = 1_000_000

# name -> (N, W, H)   (BASELINE.json configs[0..3]; C4 = C3 over several GPUs)
CONFIGS = {
    "C0": (10_000, 256, 256),
    "C1": (1_000_000, 1920, 1080),
    "C2": (5_000_000, 1920, 1080),
    "C3": (10_000_000, 3840, 2160),
}


def base_radius(n):
    return 0.04 * math.sqrt(120000.0 / n)


def make_scene(n, seed=1234):
    """Returns (props (n,8) f32 interleaved [pos,radius | rgb,opacity], normals (n,4) f32)."""
    rng = np.random.default_rng(seed)
    props = np.empty((n, 8), np.float32)
    normals = np.empty((n, 4), np.float32)
    r0 = base_radius(n)
    for lo in range(0, n, CHUNK):
        m = min(CHUNK, n - lo)
        pos = rng.uniform(-1.0, 1.0, size=(m, 3))
        rad = r0 * rng.uniform(0.5, 1.5, size=m)
        col = rng.uniform(0.0, 1.0, size=(m, 3))
        nrm = rng.standard_normal(size=(m, 3))
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        props[lo:lo + m, 0:3] = pos
        props[lo:lo + m, 3] = rad
        props[lo:lo + m, 4:7] = col
        props[lo:lo + m, 7] = 1.0
        normals[lo:lo + m, 0:3] = nrm
        normals[lo:lo + m, 3] = 1.0
    return props, normals


def padded_size(n, block=3840):
    """RadixSorter pads the key buffer to whole 3840-key blocks
    (/root/reference/src/RadixSorter.ts:12-19,46-52)."""
    return ((n + block - 1) // block) * block
