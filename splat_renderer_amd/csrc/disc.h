// disc.h — the oriented-disc footprint (SequentialRenderer / TileRenderer semantics), shared by the
// projector (project.hip) and the composite (composite.hip).
//
// Reference: /root/reference/src/SequentialRenderer.ts:68-71 (computeTangent), :91-112 (quad in the tangent
// plane of the normal), :125-141 (fragment: discard at u^2+v^2 > 1, sigma 0.4).  The rasteriser's
// perspective-correct (u,v) over that planar quad is the inverse of the quad's plane-to-screen homography,
// so it can be evaluated per pixel from tile lists:
//      (u, v) = B*d / (1 - q.d),   d = pixel centre - c
// with the 8-float record {c.x, c.y, B00, B01, B10, B11, q0, q1} (all zeros = culled).  B is the inverse of
// the 2x2 screen Jacobian of the quad's axes (the "3D -> 2D covariance via view/proj Jacobian" of the splat,
// kept as its square-root factor), q the perspective term (q = 0 is the affine EWA footprint).
//
// Every function here is one IEEE binary32 operation per operator in the order written (`contract(off)`
// is set inside each, because composite.hip is compiled with contraction on): the bounds are compared bit
// for bit with oracle/oracle.c (orc_disc_bounds, disc_record).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct DiscRecord {
    float4 a; // c.x, c.y, B00, B01
    float4 b; // B10, B11, q0, q1
};

// 0 iff all four are finite (NaN otherwise)
__device__ __forceinline__ bool disc_finite4(float a, float b, float c, float d) {
#pragma clang fp contract(off)
    const float s = ((a - a) + (b - b)) + ((c - c) + (d - d));
    return s == 0.0f;
}

// Exact screen extent of the disc a record describes: the tangent lines of the dual conic
// M * diag(1,1,-1) * M^T of the projected unit circle, M rebuilt (up to scale) from the record.  All
// zero — which bins nowhere (TileBinner.ts:437 skips min >= max) — when the record is degenerate.
__device__ __forceinline__ bool disc_bounds(const DiscRecord &r, float4 &bounds) {
#pragma clang fp contract(off)
    const float b00 = r.a.z, b01 = r.a.w, b10 = r.b.x, b11 = r.b.y, q0 = r.b.z, q1 = r.b.w;
    const float detb = b00 * b11 - b01 * b10;
    const float inv = 1.0f / detb;
    const float a00 = b11 * inv, a01 = (-b01) * inv, a10 = (-b10) * inv, a11 = b00 * inv; // A / w_c
    const float g0 = a00 * q0 + a10 * q1, g1 = a01 * q0 + a11 * q1;                       // g / w_c
    const float q00 = a00 * a00 + a01 * a01, q11 = a10 * a10 + a11 * a11;
    const float q22 = (g0 * g0 + g1 * g1) - 1.0f;
    const float q02 = a00 * g0 + a01 * g1, q12 = a10 * g0 + a11 * g1;
    const float sx = sqrtf(q02 * q02 - q00 * q22), sy = sqrtf(q12 * q12 - q11 * q22);
    const float iq = 1.0f / q22;
    const float x0 = r.a.x + (q02 + sx) * iq, x1 = r.a.x + (q02 - sx) * iq;
    const float y0 = r.a.y + (q12 + sy) * iq, y1 = r.a.y + (q12 - sy) * iq;
    const bool ok = q22 < 0.0f && disc_finite4(x0, y0, x1, y1);
    bounds = ok ? make_float4(x0, y0, x1, y1) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    return ok;
}

// The record of one splat.  m = view-projection (column-major), w/h = screen size, pr = (pos, radius),
// n = normal.  Same operation order as oracle.c's disc_record.
__device__ __forceinline__ DiscRecord disc_record(const float *m, float w, float h, float4 pr, float4 n) {
#pragma clang fp contract(off)
    DiscRecord zero = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    // computeTangent :68-71, bitangent :96
    const bool steep = fabsf(n.y) > 0.9f;
    const float ux = steep ? 1.0f : 0.0f, uy = steep ? 0.0f : 1.0f, uz = 0.0f;
    float tx = uy * n.z - uz * n.y, ty = uz * n.x - ux * n.z, tz = ux * n.y - uy * n.x; // cross(up, n)
    const float tl = sqrtf((tx * tx + ty * ty) + tz * tz);
    const float itl = 1.0f / tl; // normalize(): one reciprocal, three products
    tx *= itl; ty *= itl; tz *= itl;
    const float bx = n.y * tz - n.z * ty, by = n.z * tx - n.x * tz, bz = n.x * ty - n.y * tx; // cross(n, t)
    // half-axes of the quad in world space (:107-109) and their clip-space images (x, y, w rows)
    const float r = pr.w;
    const float e0x = tx * r, e0y = ty * r, e0z = tz * r, e1x = bx * r, e1y = by * r, e1z = bz * r;
    const float ctx = (m[0] * e0x + m[4] * e0y) + m[8] * e0z;
    const float cty = (m[1] * e0x + m[5] * e0y) + m[9] * e0z;
    const float ctw = (m[3] * e0x + m[7] * e0y) + m[11] * e0z;
    const float cbx = (m[0] * e1x + m[4] * e1y) + m[8] * e1z;
    const float cby = (m[1] * e1x + m[5] * e1y) + m[9] * e1z;
    const float cbw = (m[3] * e1x + m[7] * e1y) + m[11] * e1z;
    const float cpx = ((m[0] * pr.x + m[4] * pr.y) + m[8] * pr.z) + m[12];
    const float cpy = ((m[1] * pr.x + m[5] * pr.y) + m[9] * pr.z) + m[13];
    const float cpw = ((m[3] * pr.x + m[7] * pr.y) + m[11] * pr.z) + m[15];
    // a corner at w <= 0: the quad is skipped (no clipper, as the oracle's rasteriser)
    if (!(cpw - (fabsf(ctw) + fabsf(cbw)) > 0.0f)) return zero;
    const float hw = 0.5f * w, hh = 0.5f * h;
    const float m00 = hw * (ctx + ctw), m01 = hw * (cbx + cbw), m02 = hw * (cpx + cpw);
    const float m10 = hh * (ctw - cty), m11 = hh * (cbw - cby), m12 = hh * (cpw - cpy);
    const float icw = 1.0f / cpw;
    const float scx = m02 * icw, scy = m12 * icw;
    const float a00 = m00 - scx * ctw, a01 = m01 - scx * cbw;
    const float a10 = m10 - scy * ctw, a11 = m11 - scy * cbw;
    const float det = a00 * a11 - a01 * a10;
    if (!(fabsf(det) > 0.0f)) return zero; // edge-on (or NaN): covers no pixel
    const float idet = 1.0f / det, k = cpw * idet;
    DiscRecord o;
    o.a = make_float4(scx, scy, a11 * k, (-a01) * k);
    o.b = make_float4((-a10) * k, a00 * k, (a11 * ctw - a10 * cbw) * idet, (a00 * cbw - a01 * ctw) * idet);
    if (!disc_finite4(o.a.z, o.a.w, o.b.x, o.b.y) || !disc_finite4(o.a.x, o.a.y, o.b.z, o.b.w)) return zero;
    return o;
}
