"""ctypes binding of oracle/_build/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (splat_renderer_amd) never does.  PARITY UNPINNED — see oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

MODE_FRONT_TO_BACK = 0
MODE_REFERENCE_LITERAL = 1


def build(force=False):
    if force or not os.path.exists(_SO) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO) for f in ("oracle.c", "oracle.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        fp, u32p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
        L.orc_camera.argtypes = [fp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_double, C.c_double, fp, fp]
        L.orc_project.argtypes = [fp, fp, C.c_size_t, C.c_uint32, fp]
        L.orc_project_compact.argtypes = [fp, fp, C.c_size_t, C.c_uint32, fp]
        L.orc_expand_compact.argtypes = [fp, C.c_uint32, C.c_uint32, fp]
        L.orc_extract_keys.argtypes = [fp, C.c_uint32, C.c_uint32, u32p, u32p]
        L.orc_sort_pairs.argtypes = [u32p, u32p, C.c_uint32]
        L.orc_scan_exclusive.argtypes = [u32p, u32p, C.c_uint32]
        L.orc_scan_exclusive.restype = C.c_uint64
        L.orc_bin_sorted.argtypes = [fp, C.c_uint32, u32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_uint32, u32p, u32p, u32p, C.c_uint64]
        L.orc_bin_sorted.restype = C.c_uint64
        L.orc_gpu_tile_range.argtypes = [fp, C.c_uint32, C.c_uint32, C.c_uint32, u32p]
        L.orc_composite.argtypes = [C.c_int, C.c_int, fp, C.c_size_t, fp, C.c_size_t, fp, u32p, u32p,
                                    u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_uint32, fp, u8p]
        L.orc_composite.restype = C.c_uint64
        L.orc_composite_ex.argtypes = L.orc_composite.argtypes + [u32p, u8p]
        L.orc_composite_ex.restype = C.c_uint64
        L.orc_sequential.argtypes = [fp, fp, C.c_size_t, fp, C.c_size_t, fp, C.c_size_t, u32p,
                                     C.c_uint32, C.c_uint32, C.c_uint32, fp, u8p]
        L.orc_update_props.argtypes = [fp, fp, C.c_uint32, fp]
        L.orc_disc_bounds.argtypes = [fp, fp]
        L.orc_disc_bounds.restype = C.c_int
        L.orc_project_disc.argtypes = [fp, fp, C.c_size_t, fp, C.c_size_t, C.c_uint32, fp, fp]
        L.orc_composite_disc.argtypes = [C.c_int, fp, C.c_size_t, fp, C.c_size_t, fp, u32p, u32p, u32p, C.c_uint32,
                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, fp, u8p, u8p]
        L.orc_composite_disc.restype = C.c_uint64
        L.orc_frame.argtypes = [C.c_int, C.c_int, fp, fp, fp, C.c_uint32, C.c_uint32, C.c_uint32,
                                C.c_uint32, C.c_int, fp, u8p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        L.orc_frame.restype = C.c_int
        L.orc_sdf_gradients.argtypes = [C.c_void_p, C.c_uint32, fp, C.c_uint32, fp]
        L.orc_sdf_update_positions.argtypes = [fp, fp, C.c_uint32, fp]
        L.orc_sdf_scale_factors.argtypes = [C.c_void_p, C.c_uint32, fp, C.c_uint32, fp]
        L.orc_sdf_curvature.argtypes = [fp, fp, C.c_uint32, fp]
        L.orc_sdf_seed_positions.argtypes = [fp, fp, C.c_uint32, C.c_uint64, fp]
        L.orc_unorm8.argtypes = [C.c_float]
        L.orc_unorm8.restype = C.c_uint8
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _b(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _c32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def camera(target=(0.0, 0.0, 0.0), distance=3.0, azimuth=0.5, elevation=0.5, fov=45.0, aspect=1.0,
           near=0.1, far=100.0):
    """Returns (vp[16] column-major f32, eye[3] f32) — src/Camera.ts:85-128."""
    t = np.asarray(target, dtype=np.float32)
    vp = np.zeros(16, np.float32)
    eye = np.zeros(3, np.float32)
    lib().orc_camera(_f(t), distance, azimuth, elevation, fov, aspect, near, far, _f(vp), _f(eye))
    return vp, eye


def uniforms(vp, eye, width, height, time=0.0):
    u = np.zeros(22, np.float32)
    u[:16] = vp
    u[16:19] = eye
    u[19] = time
    u[20] = width
    u[21] = height
    return u


def project(u, props):
    """props: (n,8) interleaved or (n,4) pos_radius plane. Returns (n,8) f32 records."""
    props = _c32(props)
    n, stride = props.shape
    out = np.zeros((n, 8), np.float32)
    lib().orc_project(_f(_c32(u)), _f(props), stride, n, _f(out))
    return out


def project_compact(u, props):
    """The multi-GPU exchange records: (n,4) f32 {screen centre x, y, screen radius, depth}."""
    props = _c32(props)
    n, stride = props.shape
    out = np.zeros((n, 4), np.float32)
    lib().orc_project_compact(_f(_c32(u)), _f(props), stride, n, _f(out))
    return out


def expand_compact(records16, index_base=0):
    """(n,4) exchange records -> (n,8) ProjectedSplat records with originalIndex = index_base + i."""
    rec = _c32(records16)
    out = np.zeros((rec.shape[0], 8), np.float32)
    lib().orc_expand_compact(_f(rec), rec.shape[0], index_base, _f(out))
    return out


def extract_keys(projected, n_padded=None):
    projected = _c32(projected)
    n = projected.shape[0]
    n_padded = n if n_padded is None else n_padded
    keys = np.zeros(n_padded, np.uint32)
    payload = np.zeros(n_padded, np.uint32)
    lib().orc_extract_keys(_f(projected), n, n_padded, _u(keys), _u(payload))
    return keys, payload


def sort_pairs(keys, payload):
    k = np.ascontiguousarray(keys, dtype=np.uint32).copy()
    p = np.ascontiguousarray(payload, dtype=np.uint32).copy()
    lib().orc_sort_pairs(_u(k), _u(p), k.shape[0])
    return k, p


def scan_exclusive(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    out = np.zeros_like(a)
    total = lib().orc_scan_exclusive(_u(a), _u(out), a.shape[0])
    return out, int(total)


def bin_sorted(projected, sorted_idx, width, height, tile=16):
    projected = _c32(projected)
    sorted_idx = np.ascontiguousarray(sorted_idx, dtype=np.uint32)
    ntx, nty = -(-width // tile), -(-height // tile)
    counts = np.zeros(ntx * nty, np.uint32)
    offsets = np.zeros(ntx * nty, np.uint32)
    total = lib().orc_bin_sorted(_f(projected), projected.shape[0], _u(sorted_idx), sorted_idx.shape[0],
                                 width, height, tile, _u(counts), _u(offsets), None, 0)
    indices = np.zeros(max(int(total), 1), np.uint32)
    lib().orc_bin_sorted(_f(projected), projected.shape[0], _u(sorted_idx), sorted_idx.shape[0],
                         width, height, tile, _u(counts), _u(offsets), _u(indices), int(total))
    return counts, offsets, indices[: int(total)]


def gpu_tile_range(rec, tile, ntx, nty):
    rec = _c32(rec)
    out = np.zeros(4, np.uint32)
    lib().orc_gpu_tile_range(_f(rec), tile, ntx, nty, _u(out))
    return out


def composite(mode, early_out, color_opacity, normals, projected, indices, counts, offsets, width,
              height, tile=16, rows=None, want_u8=True, want_stops=False):
    """color_opacity: (n,4) plane or a view into (n,8) props[:,4:]; normals: (n,4).
    want_stops: also return (stop, near) per pixel — entries visited, and whether the pixel's alpha came within
    2e-5 of the 0.99 threshold (see orc_composite_ex)."""
    color_opacity = np.asarray(color_opacity, dtype=np.float32)
    cs = color_opacity.strides[0] // 4
    normals = _c32(normals)
    projected = _c32(projected)
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    if indices.shape[0] == 0:
        indices = np.zeros(1, np.uint32)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    ntx = -(-width // tile)
    r0, r1 = (0, height) if rows is None else rows
    out = np.zeros((height, width, 4), np.float32)
    out8 = np.zeros((height, width, 4), np.uint8) if want_u8 else None
    cptr = C.cast(color_opacity.ctypes.data, C.POINTER(C.c_float))
    stop = np.zeros((height, width), np.uint32) if want_stops else None
    near = np.zeros((height, width), np.uint8) if want_stops else None
    consumed = lib().orc_composite_ex(mode, int(early_out), cptr, cs, _f(normals), normals.shape[1], _f(projected),
                                      _u(indices), _u(counts), _u(offsets), tile, ntx, width, height, r0, r1,
                                      _f(out), _b(out8) if want_u8 else None, _u(stop) if want_stops else None,
                                      _b(near) if want_stops else None)
    if want_stops:
        return out, out8, int(consumed), stop, near
    return out, out8, int(consumed)


def sequential(u, props, normals, order, width, height):
    props = _c32(props)
    assert props.shape[1] == 8
    normals = _c32(normals)
    order = np.ascontiguousarray(order, dtype=np.uint32)
    out = np.zeros((height, width, 4), np.float32)
    out8 = np.zeros((height, width, 4), np.uint8)
    base = props.ctypes.data
    lib().orc_sequential(_f(_c32(u)), C.cast(base, C.POINTER(C.c_float)), 8,
                         C.cast(base + 16, C.POINTER(C.c_float)), 8, _f(normals), normals.shape[1],
                         _u(order), order.shape[0], width, height, _f(out), _b(out8))
    return out, out8


def project_disc(u, props, normals):
    """Oriented-disc footprint: returns ((n,8) ProjectedSplat records with the disc's exact bounds, (n,8) disc records)."""
    props = _c32(props)
    normals = _c32(normals)
    n, stride = props.shape
    proj = np.zeros((n, 8), np.float32)
    discs = np.zeros((n, 8), np.float32)
    lib().orc_project_disc(_f(_c32(u)), _f(props), stride, _f(normals), normals.shape[1], n, _f(proj), _f(discs))
    return proj, discs


def disc_bounds(rec):
    rec = _c32(rec)
    out = np.zeros(4, np.float32)
    ok = lib().orc_disc_bounds(_f(rec), _f(out))
    return bool(ok), out


def composite_disc(early_out, color_opacity, normals, discs, indices, counts, offsets, width, height, tile=16,
                   rows=None):
    """normals=None: color_opacity holds lit colours.  Returns (f32 image, u8 image, consumed, rim mask)."""
    color_opacity = np.asarray(color_opacity, dtype=np.float32)
    cs = color_opacity.strides[0] // 4
    discs = _c32(discs)
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    if indices.shape[0] == 0:
        indices = np.zeros(1, np.uint32)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    ntx = -(-width // tile)
    r0, r1 = (0, height) if rows is None else rows
    out = np.zeros((height, width, 4), np.float32)
    out8 = np.zeros((height, width, 4), np.uint8)
    rim = np.zeros((height, width), np.uint8)
    cptr = C.cast(color_opacity.ctypes.data, C.POINTER(C.c_float))
    if normals is not None:
        normals = _c32(normals)
    consumed = lib().orc_composite_disc(int(early_out), cptr, cs, _f(normals) if normals is not None else None,
                                        normals.shape[1] if normals is not None else 0, _f(discs), _u(indices),
                                        _u(counts), _u(offsets), tile, ntx, width, height, r0, r1, _f(out), _b(out8),
                                        _b(rim))
    return out, out8, int(consumed), rim


def update_props(positions, curvature):
    positions = _c32(positions)
    curvature = _c32(curvature)
    n = positions.shape[0]
    props = np.zeros((n, 8), np.float32)
    lib().orc_update_props(_f(positions), _f(curvature), n, _f(props))
    return props


def frame(u, props, normals, width, height, tile=16, mode=MODE_FRONT_TO_BACK, early_out=True, threads=1,
          want_f32=True):
    """Whole CPU frame; returns dict(out_f32, out_u8, total_pairs, stage_ms[project,keys,sort,bin,composite])."""
    props = _c32(props)
    normals = _c32(normals)
    n = props.shape[0]
    out = np.zeros((height, width, 4), np.float32) if want_f32 else None
    out8 = np.zeros((height, width, 4), np.uint8)
    total = C.c_uint64(0)
    ms = (C.c_double * 5)()
    rc = lib().orc_frame(mode, int(early_out), _f(_c32(u)), _f(props), _f(normals), n, width, height, tile,
                         threads, _f(out) if want_f32 else None, _b(out8), C.byref(total), ms)
    if rc != 0:
        raise MemoryError("orc_frame allocation failed")
    return dict(out_f32=out, out_u8=out8, total_pairs=int(total.value), stage_ms=list(ms))


# ---- SDF splat generation -------------------------------------------------------------------------
def _sdf_program(program):
    """program: sequence of (op, [params]) in postfix order -> packed {u32 op, f32 a[7]} records."""
    rec = np.zeros(len(program), dtype=[("op", np.uint32), ("a", np.float32, 7)])
    for k, (op, a) in enumerate(program):
        rec[k]["op"] = op
        rec[k]["a"][:len(a)] = np.asarray(a, np.float32)
    return rec


def sdf_gradients(program, positions):
    rec = _sdf_program(program)
    positions = _c32(positions)
    out = np.zeros_like(positions)
    lib().orc_sdf_gradients(rec.ctypes.data, len(rec), _f(positions), positions.shape[0], _f(out))
    return out


def sdf_update_positions(positions, gradients):
    positions, gradients = _c32(positions), _c32(gradients)
    out = np.zeros_like(positions)
    lib().orc_sdf_update_positions(_f(positions), _f(gradients), positions.shape[0], _f(out))
    return out


def sdf_scale_factors(program, positions):
    rec = _sdf_program(program)
    positions = _c32(positions)
    out = np.zeros(positions.shape[0], np.float32)
    lib().orc_sdf_scale_factors(rec.ctypes.data, len(rec), _f(positions), positions.shape[0], _f(out))
    return out


def sdf_seed_positions(aabb_min, aabb_max, n, seed):
    """(n, 4) f32 points on the faces of the box, the product's seeded generator (splat_sdf_seed_positions)."""
    mn, mx = np.ascontiguousarray(aabb_min, np.float32), np.ascontiguousarray(aabb_max, np.float32)
    out = np.zeros((n, 4), np.float32)
    lib().orc_sdf_seed_positions(_f(mn), _f(mx), n, int(seed) & 0xFFFFFFFFFFFFFFFF, _f(out))
    return out


def sdf_curvature(gradients, scale_factors):
    gradients = _c32(gradients)
    scale_factors = np.ascontiguousarray(scale_factors, np.float32)
    out = np.zeros_like(gradients)
    lib().orc_sdf_curvature(_f(gradients), _f(scale_factors), gradients.shape[0], _f(out))
    return out
