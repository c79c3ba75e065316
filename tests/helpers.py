"""Shared by CPU and GPU tests: scene + oracle pipeline in one call."""
import numpy as np

from oracle import oracle as O
from splat_renderer_amd import scene


def make_case(n, w, h, seed=1234, radius_scale=1.0, camera=None):
    props, normals = scene.make_scene(n, seed=seed)
    props[:, 3] *= np.float32(radius_scale)
    cam = dict(aspect=w / h)
    if camera:
        cam.update(camera)
    vp, eye = O.camera(**cam)
    u = O.uniforms(vp, eye, w, h)
    return props, normals, u


def oracle_pipeline(props, normals, u, w, h, tile=16, n_padded=None):
    proj = O.project(u, props)
    keys, pay = O.extract_keys(proj, n_padded)
    skeys, order = O.sort_pairs(keys, pay)
    counts, offsets, idx = O.bin_sorted(proj, order, w, h, tile)
    return dict(proj=proj, keys=keys, payload=pay, sorted_keys=skeys, order=order, counts=counts, offsets=offsets,
                indices=idx)
