#!/usr/bin/env python3
"""Generates tests/golden/*.npz — input/output vectors of the hot path produced by the oracle
(oracle/oracle.c, cross-checked against oracle/np_oracle.py while generating).

The reference itself cannot produce vectors: it is TypeScript + WGSL that needs a WebGPU device and
holds no tests or fixtures, so these files pin the ORACLE, not the reference
("parity unpinned" — see oracle/oracle.h).  They exist so that a later change to the oracle, the
scene generator or the kernels that alters any result is caught without re-deriving anything.

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import np_oracle as NP  # noqa: E402
from oracle import oracle as O  # noqa: E402
from splat_renderer_amd import scene  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name: (n, width, height, seed, radius_scale)
CASES = {
    "tiny7": (7, 64, 48, 2, 1.0),
    "small300": (300, 80, 64, 5, 1.5),
    "ragged1000": (1000, 100, 70, 7, 2.0),
}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(n, w, h, seed, rs):
    props, normals = scene.make_scene(n, seed=seed)
    props[:, 3] *= np.float32(rs)
    vp, eye = O.camera(aspect=w / h)
    u = O.uniforms(vp, eye, w, h)
    proj = O.project(u, props)
    assert np.array_equal(proj.view(np.uint32), NP.project(u, props[:, :4]).view(np.uint32))
    n_pad = scene.padded_size(n)
    keys, pay = O.extract_keys(proj, n_pad)
    k2, p2 = NP.extract_keys(proj, n_pad)
    assert np.array_equal(keys, k2) and np.array_equal(pay, p2)
    skeys, order = O.sort_pairs(keys, pay)
    assert np.array_equal(order, NP.sort_pairs(keys, pay)[1])
    counts, offsets, idx = O.bin_sorted(proj, order, w, h)
    c2, o2, i2 = NP.bin_sorted(proj, order, w, h)
    assert np.array_equal(counts, c2) and np.array_equal(offsets, o2) and np.array_equal(idx, i2)
    img_ftb, img_ftb8, _ = O.composite(O.MODE_FRONT_TO_BACK, True, props[:, 4:], normals, proj, idx, counts, offsets, w, h)
    img_lit, img_lit8, _ = O.composite(O.MODE_REFERENCE_LITERAL, True, props[:, 4:], normals, proj, idx, counts, offsets, w, h)
    img_ftb_full, _, _ = O.composite(O.MODE_FRONT_TO_BACK, False, props[:, 4:], normals, proj, idx, counts, offsets, w, h)
    assert np.abs(img_ftb - NP.composite(0, True, props[:, 4:], normals, proj, idx, counts, offsets, w, h)).max() < 1e-6
    img_b, img_b8 = O.sequential(u, props, normals, order[:n][::-1].copy(), w, h)
    return dict(props=props, normals=normals, uniforms=u, projected=proj, keys=keys, payload=pay, sorted_keys=skeys,
                order=order, counts=counts, offsets=offsets, indices=idx, image_front_to_back=img_ftb,
                image_front_to_back_u8=img_ftb8, image_literal=img_lit, image_literal_u8=img_lit8,
                image_front_to_back_no_early_out=img_ftb_full, image_model_b=img_b, image_model_b_u8=img_b8,
                dims=np.array([n, w, h, seed], np.int64), radius_scale=np.float32(rs))


def main():
    for name, args in CASES.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **run_case(*args))
    # C0 (BASELINE configs[0]) is too large to commit as arrays: keep digests
    n, w, h = scene.CONFIGS["C0"]
    r = run_case(n, w, h, 1234, 1.0)
    summary = {"config": "C0", "n": n, "width": w, "height": h, "pairs": int(r["indices"].shape[0]),
               "sha256": {k: sha(r[k]) for k in ("projected", "keys", "order", "counts", "offsets", "indices",
                                                  "image_front_to_back_u8", "image_literal_u8", "image_model_b_u8")},
               "camera_default_aspect_16_9": {"vp": [float(x) for x in O.camera(aspect=16 / 9)[0]],
                                              "eye": [float(x) for x in O.camera(aspect=16 / 9)[1]]}}
    with open(os.path.join(HERE, "C0_digest.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
