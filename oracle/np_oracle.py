"""NumPy twin of oracle.c — an INDEPENDENT second restatement of the same reference text.

TEST INFRASTRUCTURE ONLY (imported by tests/ and by tests/golden/make_golden.py).
PARITY UNPINNED: see oracle.h.  Its job is to catch transcription mistakes in oracle.c: the two
must agree bit-for-bit on keys / sort order / tile lists / ProjectedSplat floats and to float
round-off on composited pixels.

Written from the reference text directly (citations = file:line under /root/reference), not from
oracle.c: vectorised float32 NumPy for the per-splat stages, plain Python loops for binning.
"""
import math

import numpy as np

F = np.float32


# ---- Camera (src/Camera.ts:85-128; gl-matrix 3.4.4: Float32Array storage, f64 arithmetic) ------
def camera(target=(0.0, 0.0, 0.0), distance=3.0, azimuth=0.5, elevation=0.5, fov=45.0, aspect=1.0,
           near=0.1, far=100.0):
    tgt = np.asarray(target, np.float32).astype(np.float64)
    eye = np.array([tgt[0] + distance * math.cos(elevation) * math.sin(azimuth),
                    tgt[1] + distance * math.sin(elevation),
                    tgt[2] + distance * math.cos(elevation) * math.cos(azimuth)]).astype(np.float32)
    e = eye.astype(np.float64)
    up = np.array([0.0, 1.0, 0.0])
    z = e - tgt
    z = z * (1.0 / math.sqrt(float(z[0] * z[0] + z[1] * z[1] + z[2] * z[2])))
    x = np.array([up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0]])
    ln = math.sqrt(float(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]))
    x = x * (1.0 / ln) if ln else x * 0
    y = np.array([z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]])
    ln = math.sqrt(float(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]))
    y = y * (1.0 / ln) if ln else y * 0
    view = np.array([x[0], y[0], z[0], 0, x[1], y[1], z[1], 0, x[2], y[2], z[2], 0,
                     -(x[0] * e[0] + x[1] * e[1] + x[2] * e[2]),
                     -(y[0] * e[0] + y[1] * e[1] + y[2] * e[2]),
                     -(z[0] * e[0] + z[1] * e[1] + z[2] * e[2]), 1]).astype(np.float32)
    f = 1.0 / math.tan(((fov * math.pi) / 180.0) / 2.0)
    nf = 1.0 / (near - far)
    proj = np.zeros(16)
    proj[0] = f / aspect
    proj[5] = f
    proj[10] = (far + near) * nf
    proj[11] = -1.0
    proj[14] = 2.0 * far * near * nf
    proj = proj.astype(np.float32)
    a = proj.astype(np.float64)
    b = view.astype(np.float64)
    vp = np.zeros(16)
    for c in range(4):
        for k in range(4):
            vp[c * 4 + k] = (b[c * 4] * a[k] + b[c * 4 + 1] * a[4 + k] + b[c * 4 + 2] * a[8 + k]
                             + b[c * 4 + 3] * a[12 + k])
    return vp.astype(np.float32), eye


# ---- SplatProjector (src/SplatProjector.ts:64-132) ---------------------------------------------
def _screen(u, x, y, z):
    m = u
    cx = ((m[0] * x + m[4] * y) + m[8] * z) + m[12]
    cy = ((m[1] * x + m[5] * y) + m[9] * z) + m[13]
    cw = ((m[3] * x + m[7] * y) + m[11] * z) + m[15]
    nx = cx / cw
    ny = cy / cw
    return ((nx + F(1.0)) * F(0.5)) * u[20], ((F(1.0) - ny) * F(0.5)) * u[21]


def project(u, pos_radius):
    u = np.asarray(u, np.float32)
    x, y, z, r = (np.ascontiguousarray(pos_radius[:, i], np.float32) for i in range(4))
    dx, dy, dz = x - u[16], y - u[17], z - u[18]
    depth = np.sqrt((dx * dx + dy * dy) + dz * dz)
    scx, scy = _screen(u, x, y, z)
    zero = np.zeros_like(r)
    maxr = np.zeros_like(r)
    for ox, oy, oz in ((r, zero, zero), (-r, zero, zero), (zero, r, zero), (zero, -r, zero),
                       (zero, zero, r), (zero, zero, -r)):
        sx, sy = _screen(u, x + ox, y + oy, z + oz)
        ex, ey = scx - sx, scy - sy
        maxr = np.fmax(maxr, np.sqrt(ex * ex + ey * ey))
    pad = maxr * F(1.5)
    out = np.zeros((x.shape[0], 8), np.float32)
    out[:, 0], out[:, 1], out[:, 2], out[:, 3] = scx - pad, scy - pad, scx + pad, scy + pad
    out[:, 4], out[:, 5] = depth, maxr
    out[:, 6] = np.arange(x.shape[0], dtype=np.uint32).view(np.float32)
    return out


# ---- DepthKeyExtractor (src/shaders/extract-depth-keys.wgsl:37-63) -----------------------------
def extract_keys(projected, n_padded=None):
    n = projected.shape[0]
    n_padded = n if n_padded is None else n_padded
    bits = np.ascontiguousarray(projected[:, 4]).view(np.uint32)
    mask = np.where((bits >> np.uint32(31)) == 1, np.uint32(0xFFFFFFFF), np.uint32(0x80000000))
    keys = np.full(n_padded, 0xFFFFFFFF, np.uint32)
    payload = np.full(n_padded, 0xFFFFFFFF, np.uint32)
    keys[:n] = bits ^ mask
    payload[:n] = np.arange(n, dtype=np.uint32)
    return keys, payload


# ---- RadixSorter contract (src/RadixSorter.ts:263-271) ----------------------------------------
def sort_pairs(keys, payload):
    order = np.argsort(keys, kind="stable")
    return keys[order], payload[order]


# ---- PrefixSumScanner (src/PrefixSumScanner.ts:150-155) -----------------------------------------
def scan_exclusive(a):
    a = np.asarray(a, np.uint32)
    c = np.cumsum(a, dtype=np.uint64)
    out = np.zeros_like(a)
    out[1:] = c[:-1].astype(np.uint32)
    return out, int(c[-1]) if a.size else 0


# ---- TileBinner.binSorted (src/TileBinner.ts:426-495) — Python floats are JS doubles -----------
def bin_sorted(projected, sorted_idx, width, height, tile=16):
    ntx, nty = math.ceil(width / tile), math.ceil(height / tile)
    lists = [[] for _ in range(ntx * nty)]
    n = projected.shape[0]
    for s in sorted_idx.tolist():
        if s >= n:
            continue
        r = projected[s]
        mnx, mny = max(float(r[0]), 0.0), max(float(r[1]), 0.0)
        mxx, mxy = min(float(r[2]), float(width)), min(float(r[3]), float(height))
        if math.isnan(mnx + mny + mxx + mxy) or mnx >= mxx or mny >= mxy:
            continue
        tx0, tx1 = math.floor(mnx / tile), min(math.floor(mxx / tile), ntx - 1)
        ty0, ty1 = math.floor(mny / tile), min(math.floor(mxy / tile), nty - 1)
        for ty in range(ty0, ty1 + 1):
            for tx in range(tx0, tx1 + 1):
                lists[ty * ntx + tx].append(s)
    counts = np.array([len(l) for l in lists], np.uint32)
    offsets, _ = scan_exclusive(counts)
    flat = [s for l in lists for s in l]
    return counts, offsets, np.array(flat, np.uint32)


# ---- ComputeShaderRenderer (src/ComputeShaderRenderer.ts:97-198), one tile of pixels at a time --
def composite(mode, early_out, color_opacity, normals, projected, indices, counts, offsets, width,
              height, tile=16):
    ntx = math.ceil(width / tile)
    out = np.zeros((height, width, 4), np.float32)
    inv3 = F(1.0) / np.sqrt(F(3.0))
    for t in range(counts.shape[0]):
        tx, ty = t % ntx, t // ntx
        x0, y0 = tx * tile, ty * tile
        x1, y1 = min(x0 + tile, width), min(y0 + tile, height)
        if x1 <= x0 or y1 <= y0:
            continue
        py, px = np.meshgrid(np.arange(y0, y1, dtype=np.float32) + F(0.5),
                             np.arange(x0, x1, dtype=np.float32) + F(0.5), indexing="ij")
        col = np.zeros(px.shape + (3,), np.float32)
        alpha = np.zeros(px.shape, np.float32)
        trans = np.ones(px.shape, np.float32)
        live = np.ones(px.shape, bool)
        for s in indices[offsets[t]:offsets[t] + counts[t]].tolist():
            if early_out and not live.any():
                break
            r = projected[s]
            inside = ~((px < r[0]) | (px > r[2]) | (py < r[1]) | (py > r[3]))
            g = np.zeros(px.shape, np.float32)
            lit = np.zeros(3, np.float32)
            if not (r[5] < F(0.5)):
                cx, cy = (r[0] + r[2]) * F(0.5), (r[1] + r[3]) * F(0.5)
                ox, oy = px - cx, py - cy
                nd = np.sqrt(ox * ox + oy * oy) / r[5]
                g = np.where(inside, np.exp(((F(-0.5) * nd) * nd) / (F(0.5) * F(0.5))), F(0.0)).astype(np.float32)
                nrm = normals[s]
                ndl = (nrm[0] * inv3 + nrm[1] * inv3) + nrm[2] * inv3
                k = F(0.85) + F(0.15) * max(ndl, F(0.0))
                lit = (color_opacity[s, :3] * k).astype(np.float32)
            g = np.where(live, g, F(0.0)).astype(np.float32)
            if mode == 1:
                col = np.where(live[..., None], col * (F(1.0) - g)[..., None] + lit * g[..., None], col)
                alpha = np.where(live, alpha * (F(1.0) - g) + g, alpha)
                if early_out:
                    live &= ~(alpha >= F(0.99))
            else:
                w = trans * g
                col = col + lit * w[..., None]
                trans = trans * (F(1.0) - g)
                if early_out:
                    live &= ~((F(1.0) - trans) >= F(0.99))
        rem = (F(1.0) - alpha) if mode == 1 else trans
        bg = np.array([0.05, 0.05, 0.1], np.float32)
        out[y0:y1, x0:x1, :3] = col + bg * rem[..., None]
        out[y0:y1, x0:x1, 3] = 1.0
    return out


# ---- oriented-disc footprint (src/SequentialRenderer.ts:68-71, 91-112, 125-141) ------------------
# The rasteriser's perspective-correct uv over the planar quad p + r*(t*u + b*v) is the inverse of the
# plane-to-screen homography; about the screen centre c it is (u,v) = B*d / (1 - q.d).
def project_disc(u, pos_radius, normals):
    """Vectorised float32 twin of oracle.c's disc_record / orc_disc_bounds / orc_project_disc."""
    with np.errstate(all="ignore"):
        m = np.asarray(u, np.float32)
        p = np.asarray(pos_radius, np.float32)
        nr = np.asarray(normals, np.float32)
        n = p.shape[0]
        x, y, z, r = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
        n0, n1, n2 = nr[:, 0], nr[:, 1], nr[:, 2]
        steep = np.abs(n1) > F(0.9)  # :69
        u0 = np.where(steep, F(1), F(0)).astype(np.float32)
        u1 = np.where(steep, F(0), F(1)).astype(np.float32)
        u2 = np.zeros(n, np.float32)
        t0, t1, t2 = u1 * n2 - u2 * n1, u2 * n0 - u0 * n2, u0 * n1 - u1 * n0  # cross(up, normal)
        tl = np.sqrt((t0 * t0 + t1 * t1) + t2 * t2)
        itl = F(1) / tl
        t0, t1, t2 = t0 * itl, t1 * itl, t2 * itl  # :70
        b0, b1, b2 = n1 * t2 - n2 * t1, n2 * t0 - n0 * t2, n0 * t1 - n1 * t0  # :96 cross(normal, tangent)
        e0 = (t0 * r, t1 * r, t2 * r)
        e1 = (b0 * r, b1 * r, b2 * r)

        def lin(row, e):
            return (m[row] * e[0] + m[4 + row] * e[1]) + m[8 + row] * e[2]

        ctx, cty, ctw = lin(0, e0), lin(1, e0), lin(3, e0)
        cbx, cby, cbw = lin(0, e1), lin(1, e1), lin(3, e1)
        cpx = ((m[0] * x + m[4] * y) + m[8] * z) + m[12]
        cpy = ((m[1] * x + m[5] * y) + m[9] * z) + m[13]
        cpw = ((m[3] * x + m[7] * y) + m[11] * z) + m[15]
        front = (cpw - (np.abs(ctw) + np.abs(cbw))) > F(0)
        hw, hh = F(0.5) * m[20], F(0.5) * m[21]
        m00, m01, m02 = hw * (ctx + ctw), hw * (cbx + cbw), hw * (cpx + cpw)
        m10, m11, m12 = hh * (ctw - cty), hh * (cbw - cby), hh * (cpw - cpy)
        icw = F(1) / cpw
        scx, scy = m02 * icw, m12 * icw
        a00, a01 = m00 - scx * ctw, m01 - scx * cbw
        a10, a11 = m10 - scy * ctw, m11 - scy * cbw
        det = a00 * a11 - a01 * a10
        ok = front & (np.abs(det) > F(0))
        idet = F(1) / det
        k = cpw * idet
        rec = np.stack([scx, scy, a11 * k, (-a01) * k, (-a10) * k, a00 * k,
                        (a11 * ctw - a10 * cbw) * idet, (a00 * cbw - a01 * ctw) * idet], axis=1).astype(np.float32)
        ok &= np.isfinite(rec).all(axis=1)
        rec[~ok] = 0
        bounds, _ = disc_bounds(rec)
        proj = np.zeros((n, 8), np.float32)
        proj[:, :4] = bounds
        dx, dy, dz = x - m[16], y - m[17], z - m[18]
        proj[:, 4] = np.sqrt((dx * dx + dy * dy) + dz * dz)
        proj[:, 5] = F(0.5) * np.maximum(bounds[:, 2] - bounds[:, 0], bounds[:, 3] - bounds[:, 1])
        proj[:, 6] = np.arange(n, dtype=np.uint32).view(np.float32)
        return proj, rec


def disc_bounds(rec):
    """Exact screen extent of the projected unit circle from the record (dual conic M diag(1,1,-1) M^T)."""
    with np.errstate(all="ignore"):
        rec = np.asarray(rec, np.float32).reshape(-1, 8)
        cx, cy, b00, b01, b10, b11, q0, q1 = (rec[:, i] for i in range(8))
        detb = b00 * b11 - b01 * b10
        inv = F(1) / detb
        a00, a01, a10, a11 = b11 * inv, (-b01) * inv, (-b10) * inv, b00 * inv
        g0, g1 = a00 * q0 + a10 * q1, a01 * q0 + a11 * q1
        q00, q11 = a00 * a00 + a01 * a01, a10 * a10 + a11 * a11
        q22 = (g0 * g0 + g1 * g1) - F(1)
        q02, q12 = a00 * g0 + a01 * g1, a10 * g0 + a11 * g1
        sx, sy = np.sqrt(q02 * q02 - q00 * q22), np.sqrt(q12 * q12 - q11 * q22)
        iq = F(1) / q22
        out = np.stack([cx + (q02 + sx) * iq, cy + (q12 + sy) * iq, cx + (q02 - sx) * iq, cy + (q12 - sy) * iq],
                       axis=1).astype(np.float32)
        ok = (q22 < F(0)) & np.isfinite(out).all(axis=1)
        out[~ok] = 0
        return out, ok


def disc_uv(rec, px, py):
    """(u, v) of pixel centre (px, py) on the disc of one record — float64, for checking the float32 paths."""
    r = np.asarray(rec, np.float64)
    dx, dy = px - r[0], py - r[1]
    den = 1.0 - (r[6] * dx + r[7] * dy)
    return (r[2] * dx + r[3] * dy) / den, (r[4] * dx + r[5] * dy) / den


def unorm8(img):
    v = np.clip(np.nan_to_num(img, nan=0.0), 0.0, 1.0).astype(np.float32)
    return (v * F(255.0) + F(0.5)).astype(np.uint8)


# ---- SplatPropertyManager update kernel (src/SplatPropertyManager.ts:82-107) -------------------
def update_props(positions, curvature):
    n = positions.shape[0]
    props = np.zeros((n, 8), np.float32)
    props[:, :3] = positions[:, :3]
    props[:, 3] = F(0.04)
    props[:, 4:7] = np.abs(curvature[:, :3]) * F(0.8) + F(0.2)
    props[:, 7] = F(1.0)
    return props
