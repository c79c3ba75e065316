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


# ---- diagnosable comparisons of index work --------------------------------------------------------------------------
# A bit-exactness failure must be explainable from ONE ordinary run: every list / offset / count
# comparison goes through assert_same, which on a mismatch names the path under test, the first differing position
# (and the tile it lies in when the tile offsets are given), shows both neighbourhoods, and saves both arrays.
def failure_dir():
    """Where mismatching arrays are kept: gpurun_out/test_failures/ (merged back from the GPU box) unless
    SPLAT_TEST_DUMP_DIR names another place."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.environ.get("SPLAT_TEST_DUMP_DIR") or os.path.join(root, "gpurun_out", "test_failures")
    os.makedirs(d, exist_ok=True)
    return d


def assert_same(got, want, what="", offsets=None, keys=None, extra=None):
    """Bit-exact comparison of two integer arrays.  what: the code path / case under test (any object, printed).
    offsets: the tile offsets the positions of an index list refer to (names the tile of the first difference).
    keys: per-splat depth keys (shown beside differing list entries: a pure tie-order swap implicates the ranking).
    extra: dict of small values worth having beside the dump (rank mode, block size ...)."""
    import os
    import re
    import time
    got, want = np.asarray(got), np.asarray(want)
    if got.shape == want.shape and np.array_equal(got, want):
        return
    tag = re.sub(r"[^A-Za-z0-9_.-]+", "_", str(what))[:80] or "unnamed"
    path = os.path.join(failure_dir(), f"{tag}_{int(time.time() * 1e3)}.npz")
    save = {"got": got, "want": want}
    if offsets is not None:
        save["offsets"] = np.asarray(offsets)
    if keys is not None:
        save["keys"] = np.asarray(keys)
    for k, v in (extra or {}).items():
        save["extra_" + k] = np.asarray(v)
    try:
        np.savez_compressed(path, **save)
    except Exception as e:  # (never let the dump hide the failure)
        path = f"<not saved: {e!r}>"
    msg = [f"arrays differ [{what}]: shapes got {got.shape} want {want.shape}; both saved in {path}"]
    if got.shape == want.shape and got.size:
        g, w = got.reshape(-1), want.reshape(-1)
        bad = np.nonzero(g != w)[0]
        p = int(bad[0])
        msg.append(f"{bad.size} of {g.size} elements differ, first at flat position {p}, last at {int(bad[-1])}")
        lo, hi = max(0, p - 4), min(g.size, p + 8)
        msg.append(f"got [{lo}:{hi}]  = {g[lo:hi].tolist()}")
        msg.append(f"want[{lo}:{hi}]  = {w[lo:hi].tolist()}")
        if offsets is not None:
            off = np.asarray(offsets).reshape(-1)
            t = int(np.searchsorted(off, p, side="right") - 1)
            end = int(off[t + 1]) if t + 1 < off.size else g.size
            msg.append(f"tile {t}: list positions [{int(off[t])}, {end}), first difference is entry {p - int(off[t])} of it")
            seg_g, seg_w = g[int(off[t]):end], w[int(off[t]):end]
            msg.append("same multiset inside the tile: %s" % bool(np.array_equal(np.sort(seg_g), np.sort(seg_w))))
        if keys is not None and np.issubdtype(g.dtype, np.integer):
            k = np.asarray(keys).reshape(-1)
            ok = (g[lo:hi] < k.size) & (w[lo:hi] < k.size)
            msg.append(f"depth keys got  = {[int(k[i]) if o else None for i, o in zip(g[lo:hi], ok)]}")
            msg.append(f"depth keys want = {[int(k[i]) if o else None for i, o in zip(w[lo:hi], ok)]}")
    for k, v in (extra or {}).items():
        msg.append(f"{k} = {v}")
    raise AssertionError("\n".join(msg))
