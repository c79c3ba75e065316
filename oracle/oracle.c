/*
 * oracle.c — CPU restatement of the ath92/splat-renderer tile-raster hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED (see oracle.h): the reference has
 * no golden vectors for this path; this file follows the reference's shader/host text line by
 * line and is cross-checked by oracle/np_oracle.py.
 *
 * Arithmetic policy: every float operation below is a single IEEE-754 binary32 operation in
 * the order written (compile with -ffp-contract=off, no -ffast-math).  The HIP projector is
 * compiled the same way, which is what makes ProjectedSplat records / keys / sort order / tile
 * lists bit-exact between this file and the GPU.  The composite uses libm expf/sqrtf; the GPU
 * composite is compared to it within a stated tolerance (tests/test_gpu_stages.py).
 *
 * Citations are file:line under /root/reference.
 */
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------ */
/* Camera (src/Camera.ts:85-128) with gl-matrix 3.4.4 semantics (package-lock.json:866-871).   */
/* gl-matrix stores into Float32Array (each store rounds to f32) and computes in f64.          */
/* ------------------------------------------------------------------------------------------ */

static void mat4_look_at(float out[16], const float eye[3], const float center[3], const float up[3]) {
    /* gl-matrix mat4.lookAt: z = normalize(eye-center), x = normalize(up x z), y = normalize(z x x) */
    double eyex = eye[0], eyey = eye[1], eyez = eye[2];
    double upx = up[0], upy = up[1], upz = up[2];
    double cx = center[0], cy = center[1], cz = center[2];
    const double EPS = 0.000001;
    if (fabs(eyex - cx) < EPS && fabs(eyey - cy) < EPS && fabs(eyez - cz) < EPS) {
        memset(out, 0, 16 * sizeof(float));
        out[0] = out[5] = out[10] = out[15] = 1.0f;
        return;
    }
    double z0 = eyex - cx, z1 = eyey - cy, z2 = eyez - cz;
    double len = 1.0 / sqrt(z0 * z0 + z1 * z1 + z2 * z2);
    z0 *= len; z1 *= len; z2 *= len;
    double x0 = upy * z2 - upz * z1, x1 = upz * z0 - upx * z2, x2 = upx * z1 - upy * z0;
    len = sqrt(x0 * x0 + x1 * x1 + x2 * x2);
    if (!len) { x0 = x1 = x2 = 0; } else { len = 1.0 / len; x0 *= len; x1 *= len; x2 *= len; }
    double y0 = z1 * x2 - z2 * x1, y1 = z2 * x0 - z0 * x2, y2 = z0 * x1 - z1 * x0;
    len = sqrt(y0 * y0 + y1 * y1 + y2 * y2);
    if (!len) { y0 = y1 = y2 = 0; } else { len = 1.0 / len; y0 *= len; y1 *= len; y2 *= len; }
    out[0] = (float)x0; out[1] = (float)y0; out[2] = (float)z0; out[3] = 0.0f;
    out[4] = (float)x1; out[5] = (float)y1; out[6] = (float)z1; out[7] = 0.0f;
    out[8] = (float)x2; out[9] = (float)y2; out[10] = (float)z2; out[11] = 0.0f;
    out[12] = (float)(-(x0 * eyex + x1 * eyey + x2 * eyez));
    out[13] = (float)(-(y0 * eyex + y1 * eyey + y2 * eyez));
    out[14] = (float)(-(z0 * eyex + z1 * eyey + z2 * eyez));
    out[15] = 1.0f;
}

static void mat4_perspective_no(float out[16], double fovy, double aspect, double near_, double far_) {
    /* gl-matrix mat4.perspective == perspectiveNO: GL clip z in [-1,1] */
    double f = 1.0 / tan(fovy / 2.0);
    memset(out, 0, 16 * sizeof(float));
    out[0] = (float)(f / aspect);
    out[5] = (float)f;
    out[11] = -1.0f;
    if (isfinite(far_)) {
        double nf = 1.0 / (near_ - far_);
        out[10] = (float)((far_ + near_) * nf);
        out[14] = (float)(2.0 * far_ * near_ * nf);
    } else {
        out[10] = -1.0f;
        out[14] = (float)(-2.0 * near_);
    }
}

static void mat4_multiply(float out[16], const float a[16], const float b[16]) {
    /* gl-matrix mat4.multiply(out,a,b) = a*b, column-major, f64 accumulate left to right */
    float r[16];
    for (int c = 0; c < 4; ++c) {
        double b0 = b[c * 4 + 0], b1 = b[c * 4 + 1], b2 = b[c * 4 + 2], b3 = b[c * 4 + 3];
        for (int k = 0; k < 4; ++k)
            r[c * 4 + k] = (float)(b0 * (double)a[k] + b1 * (double)a[4 + k] + b2 * (double)a[8 + k] +
                                   b3 * (double)a[12 + k]);
    }
    memcpy(out, r, sizeof r);
}

void orc_camera(const float target[3], double distance, double azimuth, double elevation,
                double fov_deg, double aspect, double near_, double far_, float vp_out[16],
                float eye_out[3]) {
    /* src/Camera.ts:85-95 getCameraPosition; vec3.fromValues rounds to f32 */
    double x = distance * cos(elevation) * sin(azimuth);
    double y = distance * sin(elevation);
    double z = distance * cos(elevation) * cos(azimuth);
    float eye[3] = {(float)((double)target[0] + x), (float)((double)target[1] + y),
                    (float)((double)target[2] + z)};
    const float up[3] = {0.0f, 1.0f, 0.0f};
    float view[16], proj[16];
    mat4_look_at(view, eye, target, up);                                           /* :104-109 */
    mat4_perspective_no(proj, (fov_deg * 3.141592653589793) / 180.0, aspect, near_, far_); /* :112-118 */
    mat4_multiply(vp_out, proj, view);                                             /* :121-125 */
    eye_out[0] = eye[0]; eye_out[1] = eye[1]; eye_out[2] = eye[2];
}

/* ------------------------------------------------------------------------------------------ */
/* SplatProjector (src/SplatProjector.ts:64-132)                                               */
/* ------------------------------------------------------------------------------------------ */

/* clip = VP * vec4(p,1): column-major, each component summed left to right, no FMA */
static inline void vp_mul(const float *m, float x, float y, float z, float *cx, float *cy,
                          float *cz, float *cw) {
    *cx = ((m[0] * x + m[4] * y) + m[8] * z) + m[12];
    *cy = ((m[1] * x + m[5] * y) + m[9] * z) + m[13];
    *cz = ((m[2] * x + m[6] * y) + m[10] * z) + m[14];
    *cw = ((m[3] * x + m[7] * y) + m[11] * z) + m[15];
}

static inline void to_screen(const float *u, float x, float y, float z, float *sx, float *sy) {
    float cx, cy, cz, cw;
    vp_mul(u, x, y, z, &cx, &cy, &cz, &cw);
    float nx = cx / cw, ny = cy / cw; /* :83 ndc = clip.xyz / clip.w */
    (void)cz;
    *sx = ((nx + 1.0f) * 0.5f) * u[20]; /* :86-89 */
    *sy = ((1.0f - ny) * 0.5f) * u[21];
}

/* the projector up to the point where the record is formed: screen centre, screen radius, depth */
static void project_centre(const float *u, const float *p, float *scx_out, float *scy_out, float *max_r_out,
                           float *depth_out) {
    float x = p[0], y = p[1], z = p[2], radius = p[3];
    /* :77 depth = distance(worldPos, cameraPosition) */
    float dx = x - u[16], dy = y - u[17], dz = z - u[18];
    float depth = sqrtf((dx * dx + dy * dy) + dz * dz);
    float scx, scy;
    to_screen(u, x, y, z, &scx, &scy);
    /* :93-113 six axis-aligned offsets, in the shader's order */
    const float off[6][3] = {{radius, 0, 0}, {-radius, 0, 0}, {0, radius, 0},
                             {0, -radius, 0}, {0, 0, radius}, {0, 0, -radius}};
    float max_r = 0.0f;
    for (int k = 0; k < 6; ++k) {
        float ox, oy;
        to_screen(u, x + off[k][0], y + off[k][1], z + off[k][2], &ox, &oy);
        float ex = scx - ox, ey = scy - oy;
        float dist = sqrtf(ex * ex + ey * ey);
        max_r = fmaxf(max_r, dist); /* WGSL max(): NaN handling is implementation-defined;
                                       fmaxf drops a NaN operand, the HIP kernel does the same */
    }
    *scx_out = scx; *scy_out = scy; *max_r_out = max_r; *depth_out = depth;
}

/* record = f(centre, radius, depth, index): :119-128 */
static void form_record(float scx, float scy, float max_r, float depth, uint32_t idx, float *o) {
    float padded = max_r * 1.5f; /* :119 */
    o[0] = scx - padded; o[1] = scy - padded; /* :120 */
    o[2] = scx + padded; o[3] = scy + padded; /* :121 */
    o[4] = depth;
    o[5] = max_r;
    memcpy(&o[6], &idx, 4); /* :128 originalIndex */
    o[7] = 0.0f;
}

void orc_project(const float uniforms[22], const float *pos_radius, size_t stride, uint32_t n,
                 float *projected) {
    for (uint32_t i = 0; i < n; ++i) {
        float scx, scy, max_r, depth;
        project_centre(uniforms, pos_radius + (size_t)i * stride, &scx, &scy, &max_r, &depth);
        form_record(scx, scy, max_r, depth, i, projected + (size_t)i * ORC_PROJ_FLOATS);
    }
}

/* The multi-GPU exchange format (no reference counterpart: the reference is single-device): 16 bytes
 * per splat {screen centre x, y, screen radius, depth}; the 32-byte record is a pure function of it. */
void orc_project_compact(const float uniforms[22], const float *pos_radius, size_t stride, uint32_t n,
                         float *records16) {
    for (uint32_t i = 0; i < n; ++i) {
        float *o = records16 + (size_t)i * 4;
        project_centre(uniforms, pos_radius + (size_t)i * stride, &o[0], &o[1], &o[2], &o[3]);
    }
}

void orc_expand_compact(const float *records16, uint32_t n, uint32_t index_base, float *projected) {
    for (uint32_t i = 0; i < n; ++i) {
        const float *c = records16 + (size_t)i * 4;
        form_record(c[0], c[1], c[2], c[3], index_base + i, projected + (size_t)i * ORC_PROJ_FLOATS);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* DepthKeyExtractor (src/shaders/extract-depth-keys.wgsl:37-63)                               */
/* ------------------------------------------------------------------------------------------ */

void orc_extract_keys(const float *projected, uint32_t n, uint32_t n_padded, uint32_t *keys,
                      uint32_t *payload) {
    for (uint32_t i = 0; i < n_padded; ++i) {
        if (i >= n) { /* :46-50 */
            keys[i] = 0xffffffffu;
            payload[i] = 0xffffffffu;
            continue;
        }
        uint32_t bits;
        memcpy(&bits, &projected[(size_t)i * ORC_PROJ_FLOATS + 4], 4);
        uint32_t mask = ((bits >> 31) == 1u) ? 0xffffffffu : 0x80000000u; /* :57-58 */
        keys[i] = bits ^ mask;
        payload[i] = i; /* :62 */
    }
}

/* ------------------------------------------------------------------------------------------ */
/* RadixSorter contract: stable ascending (src/RadixSorter.ts:263-271)                         */
/* A plain 4x8-bit LSD counting sort — stable by construction. The reference's WGSL internals   */
/* (onesweep look-back) are not mirrored; only the result contract is.                         */
/* ------------------------------------------------------------------------------------------ */

void orc_sort_pairs(uint32_t *keys, uint32_t *payload, uint32_t n) {
    if (n < 2) return;
    uint32_t *k2 = (uint32_t *)malloc((size_t)n * 4), *p2 = (uint32_t *)malloc((size_t)n * 4);
    uint32_t *ka = keys, *pa = payload, *kb = k2, *pb = p2;
    for (int pass = 0; pass < 4; ++pass) {
        size_t hist[256] = {0};
        int sh = pass * 8;
        for (uint32_t i = 0; i < n; ++i) hist[(ka[i] >> sh) & 255]++;
        size_t run = 0;
        for (int d = 0; d < 256; ++d) { size_t c = hist[d]; hist[d] = run; run += c; }
        for (uint32_t i = 0; i < n; ++i) {
            size_t dst = hist[(ka[i] >> sh) & 255]++;
            kb[dst] = ka[i];
            pb[dst] = pa[i];
        }
        uint32_t *t = ka; ka = kb; kb = t;
        t = pa; pa = pb; pb = t;
    }
    /* 4 passes: result is back in the caller's arrays */
    free(k2);
    free(p2);
}

/* ------------------------------------------------------------------------------------------ */
/* PrefixSumScanner (src/PrefixSumScanner.ts:150-155)                                          */
/* ------------------------------------------------------------------------------------------ */

uint64_t orc_scan_exclusive(const uint32_t *in, uint32_t *out, uint32_t n) {
    uint64_t run = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t v = in[i]; /* in may alias out */
        out[i] = (uint32_t)run;
        run += v;
    }
    return run;
}

/* ------------------------------------------------------------------------------------------ */
/* TileBinner.binSorted (src/TileBinner.ts:426-495)                                            */
/* JS does the arithmetic in f64 on f32 inputs; max/min/floor and the division by the tile      */
/* size are done in double here too so the result is the JS result for any tile size.          */
/* ------------------------------------------------------------------------------------------ */

static inline int bin_range(const float *rec, uint32_t width, uint32_t height, uint32_t tile,
                            uint32_t ntx, uint32_t nty, uint32_t *tx0, uint32_t *tx1,
                            uint32_t *ty0, uint32_t *ty1) {
    double min_x = fmax((double)rec[0], 0.0), min_y = fmax((double)rec[1], 0.0);
    double max_x = fmin((double)rec[2], (double)width), max_y = fmin((double)rec[3], (double)height);
    /* Math.max/min propagate NaN; `NaN >= x` is false and the tile loops then run zero times:
       a NaN bound bins nowhere. */
    if (isnan(rec[0]) || isnan(rec[1]) || isnan(rec[2]) || isnan(rec[3])) return 0;
    if (min_x >= max_x || min_y >= max_y) return 0; /* :437 */
    double a = floor(min_x / tile), b = fmin(floor(max_x / tile), (double)ntx - 1.0);
    double c = floor(min_y / tile), d = fmin(floor(max_y / tile), (double)nty - 1.0);
    if (a > b || c > d) return 0; /* loops run zero times */
    *tx0 = (uint32_t)a; *tx1 = (uint32_t)b; *ty0 = (uint32_t)c; *ty1 = (uint32_t)d;
    return 1;
}

uint64_t orc_bin_sorted(const float *projected, uint32_t n_splats, const uint32_t *sorted,
                        uint32_t n_sorted, uint32_t width, uint32_t height, uint32_t tile,
                        uint32_t *counts, uint32_t *offsets, uint32_t *indices, uint64_t cap) {
    uint32_t ntx = (width + tile - 1) / tile, nty = (height + tile - 1) / tile; /* ensureBuffers */
    uint32_t nt = ntx * nty;
    memset(counts, 0, (size_t)nt * 4);
    for (uint32_t i = 0; i < n_sorted; ++i) { /* :426-450 first pass: count */
        uint32_t s = sorted[i];
        if (s >= n_splats) continue; /* padding entries index past the readback -> NaN -> no tiles */
        uint32_t a, b, c, d;
        if (!bin_range(projected + (size_t)s * ORC_PROJ_FLOATS, width, height, tile, ntx, nty, &a, &b, &c, &d))
            continue;
        for (uint32_t ty = c; ty <= d; ++ty)
            for (uint32_t tx = a; tx <= b; ++tx) counts[ty * ntx + tx]++;
    }
    uint64_t total = orc_scan_exclusive(counts, offsets, nt); /* :452-459 */
    if (!indices || total > cap) return total;
    uint32_t *cur = (uint32_t *)malloc((size_t)nt * 4);
    memcpy(cur, offsets, (size_t)nt * 4);
    for (uint32_t i = 0; i < n_sorted; ++i) { /* :470-495 second pass: fill in sorted order */
        uint32_t s = sorted[i];
        if (s >= n_splats) continue;
        uint32_t a, b, c, d;
        if (!bin_range(projected + (size_t)s * ORC_PROJ_FLOATS, width, height, tile, ntx, nty, &a, &b, &c, &d))
            continue;
        for (uint32_t ty = c; ty <= d; ++ty)
            for (uint32_t tx = a; tx <= b; ++tx) indices[cur[ty * ntx + tx]++] = s;
    }
    free(cur);
    return total;
}

void orc_gpu_tile_range(const float *rec, uint32_t tile, uint32_t ntx, uint32_t nty, uint32_t out[4]) {
    /* src/shaders/count-tile-hits.wgsl:53-56, f32 arithmetic */
    float ts = (float)tile;
    out[0] = (uint32_t)fmaxf(0.0f, floorf(rec[0] / ts));
    out[1] = (uint32_t)fmaxf(0.0f, floorf(rec[1] / ts));
    out[2] = (uint32_t)fminf((float)(ntx - 1u), floorf(rec[2] / ts));
    out[3] = (uint32_t)fminf((float)(nty - 1u), floorf(rec[3] / ts));
}

/* ------------------------------------------------------------------------------------------ */
/* ComputeShaderRenderer (src/ComputeShaderRenderer.ts:97-198) — "model A"                     */
/* ------------------------------------------------------------------------------------------ */

uint8_t orc_unorm8(float v) {
    if (!(v > 0.0f)) v = 0.0f; /* also maps NaN to 0 */
    if (v > 1.0f) v = 1.0f;
    return (uint8_t)(v * 255.0f + 0.5f);
}

typedef struct {
    int mode, early_out;
    const float *color; size_t cs;
    const float *normals; size_t ns;
    const float *proj;
    const uint32_t *indices, *counts, *offsets;
    uint32_t tile, ntx, width, height, row0, row1;
    float *out_f32; uint8_t *out_u8;
    uint64_t consumed;
    uint32_t *stop;  /* optional, per pixel: list entries visited (index of the break + 1, or the whole list) */
    uint8_t *near;   /* optional, per pixel: 1 if its alpha came within ORC_STOP_MARGIN of the 0.99 threshold at some entry */
} comp_job;

/* a pixel whose alpha lands this close to the threshold may legitimately stop one entry earlier or later in
 * another correct float evaluation (different exp, fused multiply-adds): tests hold such pixels to the loose
 * bound (1 - 0.99) * max colour and every other pixel to the tight one */
#define ORC_STOP_MARGIN 2e-5f

static void composite_rows(comp_job *j) {
    const float inv_sqrt3 = 1.0f / sqrtf(3.0f); /* normalize(vec3(1,1,1)) :143 */
    uint64_t consumed = 0;
    for (uint32_t py = j->row0; py < j->row1; ++py) {
        for (uint32_t px = 0; px < j->width; ++px) {
            uint32_t tile_idx = (py / j->tile) * j->ntx + (px / j->tile); /* :161-163 */
            float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;       /* :169 */
            uint32_t off = j->offsets[tile_idx], cnt = j->counts[tile_idx];
            float cr = 0.0f, cg = 0.0f, cb = 0.0f;
            float alpha = 0.0f; /* literal mode */
            float trans = 1.0f; /* front-to-back mode: T = prod(1-g) */
            uint32_t visited = 0;
            uint8_t near = 0;
            for (uint32_t i = 0; i < cnt; ++i) {
                uint32_t s = j->indices[off + i];
                ++consumed;
                ++visited;
                const float *rec = j->proj + (size_t)s * ORC_PROJ_FLOATS;
                float g = 0.0f, lr = 0.0f, lg = 0.0f, lb = 0.0f;
                /* evaluateSplat :98-148 */
                if (!(pxf < rec[0] || pxf > rec[2] || pyf < rec[1] || pyf > rec[3])) {
                    float scx = (rec[0] + rec[2]) * 0.5f, scy = (rec[1] + rec[3]) * 0.5f; /* :124 */
                    float r = rec[5];
                    if (!(r < 0.5f)) { /* :127-129 */
                        float ox = pxf - scx, oy = pyf - scy;
                        float dist = sqrtf(ox * ox + oy * oy);
                        float nd = dist / r;
                        g = expf(((-0.5f * nd) * nd) / (0.5f * 0.5f)); /* :139-140 */
                        const float *c = j->color + (size_t)s * j->cs;
                        const float *nrm = j->normals + (size_t)s * j->ns;
                        float ndl = (nrm[0] * inv_sqrt3 + nrm[1] * inv_sqrt3) + nrm[2] * inv_sqrt3;
                        float diffuse = fmaxf(ndl, 0.0f);
                        float k = 0.85f + 0.15f * diffuse; /* :145 */
                        lr = c[0] * k; lg = c[1] * k; lb = c[2] * k;
                    }
                }
                if (j->mode == ORC_MODE_REFERENCE_LITERAL) {
                    /* :183-185 exactly as written */
                    cr = cr * (1.0f - g) + lr * g;
                    cg = cg * (1.0f - g) + lg * g;
                    cb = cb * (1.0f - g) + lb * g;
                    alpha = alpha * (1.0f - g) + g;
                    if (fabsf(alpha - 0.99f) < ORC_STOP_MARGIN) near = 1;
                    if (j->early_out && alpha >= 0.99f) break; /* :187-190 */
                } else {
                    /* SURVEY §8a contract 3: nearest on top, C += c*g*T, T *= (1-g) */
                    float w = trans * g;
                    cr = cr + lr * w; cg = cg + lg * w; cb = cb + lb * w;
                    trans = trans * (1.0f - g);
                    if (fabsf((1.0f - trans) - 0.99f) < ORC_STOP_MARGIN) near = 1;
                    if (j->early_out && (1.0f - trans) >= 0.99f) break;
                }
            }
            if (j->stop) j->stop[(size_t)py * j->width + px] = visited;
            if (j->near) j->near[(size_t)py * j->width + px] = near;
            float rem = (j->mode == ORC_MODE_REFERENCE_LITERAL) ? (1.0f - alpha) : trans;
            float fr = cr + 0.05f * rem, fg = cg + 0.05f * rem, fb = cb + 0.1f * rem; /* :193-195 */
            size_t o = ((size_t)py * j->width + px) * 4;
            if (j->out_f32) { j->out_f32[o] = fr; j->out_f32[o + 1] = fg; j->out_f32[o + 2] = fb; j->out_f32[o + 3] = 1.0f; }
            if (j->out_u8) { j->out_u8[o] = orc_unorm8(fr); j->out_u8[o + 1] = orc_unorm8(fg); j->out_u8[o + 2] = orc_unorm8(fb); j->out_u8[o + 3] = 255; }
        }
    }
    j->consumed = consumed;
}

uint64_t orc_composite(int mode, int early_out, const float *color_opacity, size_t color_stride,
                       const float *normals, size_t normal_stride, const float *projected,
                       const uint32_t *indices, const uint32_t *counts, const uint32_t *offsets,
                       uint32_t tile, uint32_t ntx, uint32_t width, uint32_t height,
                       uint32_t row0, uint32_t row1, float *out_f32, uint8_t *out_u8) {
    return orc_composite_ex(mode, early_out, color_opacity, color_stride, normals, normal_stride, projected, indices, counts,
                            offsets, tile, ntx, width, height, row0, row1, out_f32, out_u8, NULL, NULL);
}

uint64_t orc_composite_ex(int mode, int early_out, const float *color_opacity, size_t color_stride,
                          const float *normals, size_t normal_stride, const float *projected,
                          const uint32_t *indices, const uint32_t *counts, const uint32_t *offsets,
                          uint32_t tile, uint32_t ntx, uint32_t width, uint32_t height,
                          uint32_t row0, uint32_t row1, float *out_f32, uint8_t *out_u8, uint32_t *stop, uint8_t *near) {
    comp_job j = {mode, early_out, color_opacity, color_stride, normals, normal_stride, projected,
                  indices, counts, offsets, tile, ntx, width, height, row0, row1, out_f32, out_u8, 0, stop, near};
    if (row1 > height) j.row1 = height;
    composite_rows(&j);
    return j.consumed;
}

/* ------------------------------------------------------------------------------------------ */
/* SequentialRenderer (src/SequentialRenderer.ts:68-142,186-209,246-307) — "model B"          */
/* Software restatement of what the fixed-function pipeline does with that shader pair:        */
/* two triangles per splat, pixel centres at +0.5, top-left fill rule, perspective-correct uv, */
/* blend src-alpha/one-minus-src-alpha in float (no 8-bit intermediate quantisation).          */
/* ------------------------------------------------------------------------------------------ */

static inline void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

static inline int edge_top_left(double ax, double ay, double bx, double by) {
    /* Framebuffer y grows downward; the triangle is oriented so that every edge function
       w(p) = dx*(py-ay) - dy*(px-ax) is >= 0 inside.  dw/dpx = -dy, so the interior lies to the
       right of an edge with dy < 0 (a left edge); for dy == 0, w = dx*(py-ay) and the interior
       lies below when dx > 0 (a top edge). */
    double dx = bx - ax, dy = by - ay;
    return (dy < 0) || (dy == 0 && dx > 0);
}

typedef struct { double x, y, invw; float u, v; } rvert;

static void raster_tri(const rvert *a, const rvert *b, const rvert *c, const float lit[3],
                       uint32_t width, uint32_t height, float *fb) {
    double area = (b->x - a->x) * (c->y - a->y) - (b->y - a->y) * (c->x - a->x);
    if (area == 0) return;
    const rvert *v0 = a, *v1 = b, *v2 = c;
    if (area < 0) { v1 = c; v2 = b; area = -area; }
    double minx = fmin(v0->x, fmin(v1->x, v2->x)), maxx = fmax(v0->x, fmax(v1->x, v2->x));
    double miny = fmin(v0->y, fmin(v1->y, v2->y)), maxy = fmax(v0->y, fmax(v1->y, v2->y));
    if (maxx < 0 || maxy < 0 || minx > width || miny > height) return;
    long x0 = (long)floor(fmax(minx, 0.0)), x1 = (long)ceil(fmin(maxx, (double)width));
    long y0 = (long)floor(fmax(miny, 0.0)), y1 = (long)ceil(fmin(maxy, (double)height));
    int tl0 = edge_top_left(v1->x, v1->y, v2->x, v2->y);
    int tl1 = edge_top_left(v2->x, v2->y, v0->x, v0->y);
    int tl2 = edge_top_left(v0->x, v0->y, v1->x, v1->y);
    for (long y = y0; y < y1 && y < (long)height; ++y) {
        for (long x = x0; x < x1 && x < (long)width; ++x) {
            double px = x + 0.5, py = y + 0.5;
            double w0 = (v2->x - v1->x) * (py - v1->y) - (v2->y - v1->y) * (px - v1->x);
            double w1 = (v0->x - v2->x) * (py - v2->y) - (v0->y - v2->y) * (px - v2->x);
            double w2 = (v1->x - v0->x) * (py - v0->y) - (v1->y - v0->y) * (px - v0->x);
            if (w0 < 0 || w1 < 0 || w2 < 0) continue;
            if ((w0 == 0 && !tl0) || (w1 == 0 && !tl1) || (w2 == 0 && !tl2)) continue;
            /* perspective-correct interpolation */
            double b0 = w0 * v0->invw, b1 = w1 * v1->invw, b2 = w2 * v2->invw;
            double s = b0 + b1 + b2;
            float u = (float)((b0 * v0->u + b1 * v1->u + b2 * v2->u) / s);
            float v = (float)((b0 * v0->v + b1 * v1->v + b2 * v2->v) / s);
            float dist2 = u * u + v * v;                 /* :126 */
            if (dist2 > 1.0f) continue;                  /* :128-130 discard */
            float g = expf((-0.5f * dist2) / (0.4f * 0.4f)); /* :132-133 */
            float *p = fb + ((size_t)y * width + (size_t)x) * 4;
            /* :189-200 color: src*srcA + dst*(1-srcA); alpha: src*1 + dst*(1-srcA) */
            p[0] = lit[0] * g + p[0] * (1.0f - g);
            p[1] = lit[1] * g + p[1] * (1.0f - g);
            p[2] = lit[2] * g + p[2] * (1.0f - g);
            p[3] = g + p[3] * (1.0f - g);
        }
    }
}

void orc_sequential(const float uniforms[22], const float *pos_radius, size_t pr_stride,
                    const float *color_opacity, size_t color_stride, const float *normals,
                    size_t normal_stride, const uint32_t *order, uint32_t n_order, uint32_t width,
                    uint32_t height, float *out_f32, uint8_t *out_u8) {
    size_t npx = (size_t)width * height;
    float *fb = out_f32 ? out_f32 : (float *)malloc(npx * 16);
    for (size_t i = 0; i < npx; ++i) { /* :251 clear colour */
        fb[i * 4] = 0.05f; fb[i * 4 + 1] = 0.05f; fb[i * 4 + 2] = 0.1f; fb[i * 4 + 3] = 1.0f;
    }
    const float inv_sqrt3 = 1.0f / sqrtf(3.0f);
    static const float quad[6][2] = {{-1, -1}, {1, -1}, {-1, 1}, {-1, 1}, {1, -1}, {1, 1}}; /* :99-102 */
    for (uint32_t i = 0; i < n_order; ++i) {
        uint32_t s = order[i];
        const float *p = pos_radius + (size_t)s * pr_stride;
        const float *c = color_opacity + (size_t)s * color_stride;
        const float *n = normals + (size_t)s * normal_stride;
        float radius = p[3];
        /* computeTangent :68-71 */
        float up[3] = {0.0f, 1.0f, 0.0f};
        if (fabsf(n[1]) > 0.9f) { up[0] = 1.0f; up[1] = 0.0f; }
        float t[3], bt[3];
        cross3(up, n, t);
        float tl = sqrtf((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
        t[0] /= tl; t[1] /= tl; t[2] /= tl;
        cross3(n, t, bt); /* :96 */
        rvert rv[6];
        int ok = 1;
        for (int k = 0; k < 6; ++k) {
            float ox = quad[k][0], oy = quad[k][1];
            /* :107-111 */
            float wx = p[0] + ((t[0] * ox) * radius + (bt[0] * oy) * radius);
            float wy = p[1] + ((t[1] * ox) * radius + (bt[1] * oy) * radius);
            float wz = p[2] + ((t[2] * ox) * radius + (bt[2] * oy) * radius);
            float cx, cy, cz, cw;
            vp_mul(uniforms, wx, wy, wz, &cx, &cy, &cz, &cw);
            if (!(cw > 0.0f)) { ok = 0; break; }
            rv[k].invw = 1.0 / (double)cw;
            rv[k].x = ((double)cx * rv[k].invw + 1.0) * 0.5 * width;
            rv[k].y = (1.0 - (double)cy * rv[k].invw) * 0.5 * height;
            rv[k].u = ox; rv[k].v = oy;
        }
        if (!ok) continue;
        float ndl = (n[0] * inv_sqrt3 + n[1] * inv_sqrt3) + n[2] * inv_sqrt3;
        float kdiff = 0.85f + 0.15f * fmaxf(ndl, 0.0f); /* :135-137 */
        float lit[3] = {c[0] * kdiff, c[1] * kdiff, c[2] * kdiff};
        raster_tri(&rv[0], &rv[1], &rv[2], lit, width, height, fb);
        raster_tri(&rv[3], &rv[4], &rv[5], lit, width, height, fb);
    }
    if (out_u8)
        for (size_t i = 0; i < npx * 4; ++i) out_u8[i] = orc_unorm8(fb[i]);
    if (!out_f32) free(fb);
}

/* ------------------------------------------------------------------------------------------ */
/* Oriented-disc footprint — SequentialRenderer's splat, evaluated per pixel from tile lists    */
/* (closes I6).                                                               */
/*                                                                                              */
/* The vertex shader places a quad p + r*(t*u + b*v), (u,v) in [-1,1]^2, in the tangent plane of */
/* the normal (:91-112); the rasteriser interpolates (u,v) perspective-correctly; the fragment  */
/* shader keeps u^2+v^2 <= 1 (:126-130).  Perspective-correct interpolation over a planar quad */
/* IS the plane-to-screen homography inverted, so the same footprint can be evaluated at any    */
/* pixel without a rasteriser:  [X, Y, w] = M * [u, v, 1] with X = W/2*(cx+cw), Y = H/2*(cw-cy), */
/* and relative to the screen centre c = (X/w, Y/w) at (0,0):                                   */
/*      (u, v) = B*d / (1 - q.d),   d = pixel - c,                                              */
/* B = w_c * A^-1 with A the 2x2 Jacobian numerators, q = A^-T g with g the w-row of M (the      */
/* perspective term; q = 0 gives the affine "EWA" footprint).  The 8-float record is             */
/* {c.x, c.y, B00, B01, B10, B11, q0, q1}.  Its screen bounding box — what the binner bins by — */
/* is the exact extent of the projected unit circle (tangent lines of the dual conic             */
/* M*diag(1,1,-1)*M^T), computed FROM THE RECORD so that it is a pure function of it.            */
/* All of it in binary32, one rounding per operator, in the order written (the HIP projector     */
/* follows the same order with contraction off: records and bounds are compared bit for bit).    */
/* ------------------------------------------------------------------------------------------ */

static inline int finite4(float a, float b, float c, float d) {
    float s = ((a - a) + (b - b)) + ((c - c) + (d - d)); /* 0 iff all finite, NaN otherwise */
    return s == 0.0f;
}

/* bounds = {min.x, min.y, max.x, max.y} of the disc a record describes; all zero (bins nowhere:
 * TileBinner.ts:437 skips min >= max) when the record is degenerate.  Returns 1 when valid. */
int orc_disc_bounds(const float *rec, float out[4]) {
    float detb = rec[2] * rec[5] - rec[3] * rec[4];
    float inv = 1.0f / detb;
    float a00 = rec[5] * inv, a01 = (-rec[3]) * inv, a10 = (-rec[4]) * inv, a11 = rec[2] * inv; /* A / w_c */
    float g0 = a00 * rec[6] + a10 * rec[7], g1 = a01 * rec[6] + a11 * rec[7];                   /* g / w_c */
    float q00 = a00 * a00 + a01 * a01, q11 = a10 * a10 + a11 * a11;
    float q22 = (g0 * g0 + g1 * g1) - 1.0f;
    float q02 = a00 * g0 + a01 * g1, q12 = a10 * g0 + a11 * g1;
    float sx = sqrtf(q02 * q02 - q00 * q22), sy = sqrtf(q12 * q12 - q11 * q22);
    float iq = 1.0f / q22;
    float x0 = rec[0] + (q02 + sx) * iq, x1 = rec[0] + (q02 - sx) * iq;
    float y0 = rec[1] + (q12 + sy) * iq, y1 = rec[1] + (q12 - sy) * iq;
    if (q22 < 0.0f && finite4(x0, y0, x1, y1)) {
        out[0] = x0; out[1] = y0; out[2] = x1; out[3] = y1;
        return 1;
    }
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    return 0;
}

static void disc_record(const float *u, const float *p, const float *n, float *rec) {
    const float *m = u;
    for (int k = 0; k < ORC_DISC_FLOATS; ++k) rec[k] = 0.0f;
    /* computeTangent :68-71, bitangent :96 */
    float up[3] = {0.0f, 1.0f, 0.0f};
    if (fabsf(n[1]) > 0.9f) { up[0] = 1.0f; up[1] = 0.0f; }
    float t[3], b[3];
    cross3(up, n, t);
    float tl = sqrtf((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
    float itl = 1.0f / tl; /* normalize(): one reciprocal, three products */
    t[0] *= itl; t[1] *= itl; t[2] *= itl;
    cross3(n, t, b);
    /* half-axes of the quad in world space (:107-109), then their clip-space images (x, y, w rows) */
    float r = p[3];
    float e0[3] = {t[0] * r, t[1] * r, t[2] * r}, e1[3] = {b[0] * r, b[1] * r, b[2] * r};
    float ctx = (m[0] * e0[0] + m[4] * e0[1]) + m[8] * e0[2];
    float cty = (m[1] * e0[0] + m[5] * e0[1]) + m[9] * e0[2];
    float ctw = (m[3] * e0[0] + m[7] * e0[1]) + m[11] * e0[2];
    float cbx = (m[0] * e1[0] + m[4] * e1[1]) + m[8] * e1[2];
    float cby = (m[1] * e1[0] + m[5] * e1[1]) + m[9] * e1[2];
    float cbw = (m[3] * e1[0] + m[7] * e1[1]) + m[11] * e1[2];
    float cpx, cpy, cpz, cpw;
    vp_mul(m, p[0], p[1], p[2], &cpx, &cpy, &cpz, &cpw);
    (void)cpz;
    /* a corner at w <= 0: the quad is skipped (orc_sequential does the same; no clipper) */
    if (!(cpw - (fabsf(ctw) + fabsf(cbw)) > 0.0f)) return;
    float hw = 0.5f * u[20], hh = 0.5f * u[21];
    float m00 = hw * (ctx + ctw), m01 = hw * (cbx + cbw), m02 = hw * (cpx + cpw);
    float m10 = hh * (ctw - cty), m11 = hh * (cbw - cby), m12 = hh * (cpw - cpy);
    float icw = 1.0f / cpw;
    float scx = m02 * icw, scy = m12 * icw;
    float a00 = m00 - scx * ctw, a01 = m01 - scx * cbw;
    float a10 = m10 - scy * ctw, a11 = m11 - scy * cbw;
    float det = a00 * a11 - a01 * a10;
    if (!(fabsf(det) > 0.0f)) return; /* edge-on (or NaN): covers no pixel */
    float idet = 1.0f / det, k = cpw * idet;
    rec[0] = scx; rec[1] = scy;
    rec[2] = a11 * k; rec[3] = (-a01) * k; rec[4] = (-a10) * k; rec[5] = a00 * k;
    rec[6] = (a11 * ctw - a10 * cbw) * idet;
    rec[7] = (a00 * cbw - a01 * ctw) * idet;
    if (!finite4(rec[2], rec[3], rec[4], rec[5]) || !finite4(rec[0], rec[1], rec[6], rec[7]))
        for (int j = 0; j < ORC_DISC_FLOATS; ++j) rec[j] = 0.0f;
}

void orc_project_disc(const float uniforms[22], const float *pos_radius, size_t pr_stride, const float *normals,
                      size_t normal_stride, uint32_t n, float *projected, float *discs) {
    for (uint32_t i = 0; i < n; ++i) {
        const float *p = pos_radius + (size_t)i * pr_stride;
        float *rec = discs + (size_t)i * ORC_DISC_FLOATS, *o = projected + (size_t)i * ORC_PROJ_FLOATS;
        disc_record(uniforms, p, normals + (size_t)i * normal_stride, rec);
        orc_disc_bounds(rec, o);
        float dx = p[0] - uniforms[16], dy = p[1] - uniforms[17], dz = p[2] - uniforms[18];
        o[4] = sqrtf((dx * dx + dy * dy) + dz * dz); /* same depth, same order as SplatProjector.ts:77 */
        o[5] = 0.5f * fmaxf(o[2] - o[0], o[3] - o[1]);
        memcpy(&o[6], &i, 4);
        o[7] = 0.0f;
    }
}

/* The per-pixel loop over tile lists (nearest first, C += c*g*T, T *= 1-g: the "over" operator of the
 * blend state :189-200 applied back to front, re-associated) with SequentialRenderer's fragment
 * (:125-141).  rim (may be NULL): set to 1 for pixels where some evaluated entry has |d2 - 1| <= 1e-3 —
 * the discard at d2 > 1 is a step of exp(-3.125) = 0.044 in alpha, so on those pixels (and only
 * those) two correct evaluations that round differently may differ by up to 0.044. */
uint64_t orc_composite_disc(int early_out, const float *lit_or_color, size_t color_stride, const float *normals,
                            size_t normal_stride, const float *discs, const uint32_t *indices, const uint32_t *counts,
                            const uint32_t *offsets, uint32_t tile, uint32_t ntx, uint32_t width, uint32_t height,
                            uint32_t row0, uint32_t row1, float *out_f32, uint8_t *out_u8, uint8_t *rim) {
    const float inv_sqrt3 = 1.0f / sqrtf(3.0f);
    uint64_t consumed = 0;
    if (row1 > height) row1 = height;
    for (uint32_t py = row0; py < row1; ++py) {
        for (uint32_t px = 0; px < width; ++px) {
            uint32_t tile_idx = (py / tile) * ntx + (px / tile);
            float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
            uint32_t off = offsets[tile_idx], cnt = counts[tile_idx];
            float cr = 0.0f, cg = 0.0f, cb = 0.0f, trans = 1.0f;
            int on_rim = 0;
            for (uint32_t i = 0; i < cnt; ++i) {
                uint32_t s = indices[off + i];
                ++consumed;
                const float *rec = discs + (size_t)s * ORC_DISC_FLOATS;
                float bnd[4];
                if (!orc_disc_bounds(rec, bnd)) continue;
                float dx = pxf - rec[0], dy = pyf - rec[1];
                float den = 1.0f - (rec[6] * dx + rec[7] * dy);
                float nu = rec[2] * dx + rec[3] * dy, nv = rec[4] * dx + rec[5] * dy;
                float uu = nu / den, vv = nv / den;
                float d2 = uu * uu + vv * vv; /* :126 */
                if (fabsf(d2 - 1.0f) <= 1e-3f) on_rim = 1;
                if (pxf < bnd[0] || pxf > bnd[2] || pyf < bnd[1] || pyf > bnd[3]) continue;
                if (!(d2 <= 1.0f)) continue; /* :128-130 discard */
                float g = expf((-0.5f * d2) / (0.4f * 0.4f)); /* :132-133 */
                const float *c = lit_or_color + (size_t)s * color_stride;
                float k = 1.0f;
                if (normals) {
                    const float *nrm = normals + (size_t)s * normal_stride;
                    float ndl = (nrm[0] * inv_sqrt3 + nrm[1] * inv_sqrt3) + nrm[2] * inv_sqrt3;
                    k = 0.85f + 0.15f * fmaxf(ndl, 0.0f); /* :135-137 */
                }
                float w = trans * g;
                cr = cr + (c[0] * k) * w; cg = cg + (c[1] * k) * w; cb = cb + (c[2] * k) * w;
                trans = trans * (1.0f - g);
                if (early_out && (1.0f - trans) >= 0.99f) break;
            }
            float fr = cr + 0.05f * trans, fg = cg + 0.05f * trans, fb = cb + 0.1f * trans; /* clear colour :251 */
            size_t o = ((size_t)py * width + px) * 4;
            if (out_f32) { out_f32[o] = fr; out_f32[o + 1] = fg; out_f32[o + 2] = fb; out_f32[o + 3] = 1.0f; }
            if (out_u8) { out_u8[o] = orc_unorm8(fr); out_u8[o + 1] = orc_unorm8(fg); out_u8[o + 2] = orc_unorm8(fb); out_u8[o + 3] = 255; }
            if (rim) rim[(size_t)py * width + px] = (uint8_t)on_rim;
        }
    }
    return consumed;
}

/* ------------------------------------------------------------------------------------------ */
/* SplatPropertyManager update kernel (src/SplatPropertyManager.ts:82-107)                     */
/* ------------------------------------------------------------------------------------------ */

void orc_update_props(const float *positions, const float *curvature, uint32_t n, float *props) {
    for (uint32_t i = 0; i < n; ++i) {
        const float *p = positions + (size_t)i * 4, *cv = curvature + (size_t)i * 4;
        float *o = props + (size_t)i * 8;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
        o[3] = 0.04f;                           /* :94 */
        o[4] = fabsf(cv[0]) * 0.8f + 0.2f;      /* :97 */
        o[5] = fabsf(cv[1]) * 0.8f + 0.2f;
        o[6] = fabsf(cv[2]) * 0.8f + 0.2f;
        o[7] = 1.0f;                            /* :101 */
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Whole frame, timed per stage (bench.py cpu_baseline leg).  `threads` bands the composite's   */
/* rows and the projector's splat range over pthreads; sort and binning stay single-threaded    */
/* like the reference's JS loops (src/TileBinner.ts:426-495).                                   */
/* ------------------------------------------------------------------------------------------ */

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

typedef struct { const float *u; const float *props; uint32_t i0, i1; float *proj; } proj_job;
static void *proj_thread(void *a) {
    proj_job *j = (proj_job *)a;
    /* orc_project writes originalIndex = local i; redo with global index afterwards */
    orc_project(j->u, j->props + (size_t)j->i0 * 8, 8, j->i1 - j->i0, j->proj + (size_t)j->i0 * ORC_PROJ_FLOATS);
    for (uint32_t i = j->i0; i < j->i1; ++i) memcpy(&j->proj[(size_t)i * ORC_PROJ_FLOATS + 6], &i, 4);
    return NULL;
}
static void *comp_thread(void *a) { composite_rows((comp_job *)a); return NULL; }

int orc_frame(int mode, int early_out, const float uniforms[22], const float *props,
              const float *normals, uint32_t n, uint32_t width, uint32_t height, uint32_t tile,
              int threads, float *out_f32, uint8_t *out_u8, uint64_t *total_pairs, double stage_ms[5]) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    uint32_t ntx = (width + tile - 1) / tile, nty = (height + tile - 1) / tile, nt = ntx * nty;
    float *proj = (float *)malloc((size_t)n * 32 + 32);
    uint32_t *keys = (uint32_t *)malloc((size_t)n * 4 + 4), *pay = (uint32_t *)malloc((size_t)n * 4 + 4);
    uint32_t *counts = (uint32_t *)malloc((size_t)nt * 4), *offsets = (uint32_t *)malloc((size_t)nt * 4);
    if (!proj || !keys || !pay || !counts || !offsets) return -1;
    pthread_t th[256];
    double t0 = now_ms();
    {
        proj_job pj[256];
        for (int t = 0; t < threads; ++t) {
            pj[t] = (proj_job){uniforms, props, (uint32_t)((uint64_t)n * t / threads),
                               (uint32_t)((uint64_t)n * (t + 1) / threads), proj};
            pthread_create(&th[t], NULL, proj_thread, &pj[t]);
        }
        for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    }
    double t1 = now_ms();
    orc_extract_keys(proj, n, n, keys, pay);
    double t2 = now_ms();
    orc_sort_pairs(keys, pay, n);
    double t3 = now_ms();
    uint64_t total = orc_bin_sorted(proj, n, pay, n, width, height, tile, counts, offsets, NULL, 0);
    uint32_t *indices = (uint32_t *)malloc((size_t)(total ? total : 1) * 4);
    if (!indices) return -1;
    orc_bin_sorted(proj, n, pay, n, width, height, tile, counts, offsets, indices, total);
    double t4 = now_ms();
    {
        comp_job cj[256];
        for (int t = 0; t < threads; ++t) {
            /* band whole tile rows so thread boundaries fall on tile boundaries */
            uint32_t r0 = (uint32_t)((uint64_t)nty * t / threads) * tile;
            uint32_t r1 = (uint32_t)((uint64_t)nty * (t + 1) / threads) * tile;
            if (r1 > height) r1 = height;
            if (r0 > height) r0 = height;
            cj[t] = (comp_job){mode, early_out, props + 4, 8, normals, 4, proj, indices, counts, offsets,
                               tile, ntx, width, height, r0, r1, out_f32, out_u8, 0, NULL, NULL};
            pthread_create(&th[t], NULL, comp_thread, &cj[t]);
        }
        for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    }
    double t5 = now_ms();
    if (total_pairs) *total_pairs = total;
    if (stage_ms) {
        stage_ms[0] = t1 - t0; stage_ms[1] = t2 - t1; stage_ms[2] = t3 - t2;
        stage_ms[3] = t4 - t3; stage_ms[4] = t5 - t4;
    }
    free(proj); free(keys); free(pay); free(counts); free(offsets); free(indices);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* SDF splat generation (src/sdf/CodeGenerator.ts, GradientSampler.ts, update-positions.wgsl,   */
/* CurvatureSampler.ts) — SURVEY §8f row 4.  WGSL built-ins spelled out: length(v) =            */
/* sqrt((x*x + y*y) + z*z), normalize(v) = v / length(v), mix(a,b,t) = a*(1-t) + b*t,           */
/* smoothstep(lo,hi,x) = t*t*(3 - 2t), t = clamp((x-lo)/(hi-lo), 0, 1), sign(0) = 0.            */
/* ------------------------------------------------------------------------------------------ */
static float sdf_len3(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }
static float sdf_len2(float x, float y) { return sqrtf(x * x + y * y); }
static float sdf_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

static void sdg_sphere(const float p[3], float r, float o[4]) { /* CodeGenerator.ts:100-106 */
    float d = sdf_len3(p[0], p[1], p[2]);
    float m = fmaxf(d, 0.0001f);
    o[0] = d - r; o[1] = p[0] / m; o[2] = p[1] / m; o[3] = p[2] / m;
}

static void sdg_box(const float p[3], const float b[3], float o[4]) { /* :109-133 */
    float qx = fabsf(p[0]) - b[0], qy = fabsf(p[1]) - b[1], qz = fabsf(p[2]) - b[2];
    float wx = fmaxf(qx, 0.0f), wy = fmaxf(qy, 0.0f), wz = fmaxf(qz, 0.0f);
    float g = fmaxf(qx, fmaxf(qy, qz));
    float sx = sdf_sign(p[0]), sy = sdf_sign(p[1]), sz = sdf_sign(p[2]);
    o[0] = sdf_len3(wx, wy, wz) + fminf(g, 0.0f);
    if (g > 0.0f) {
        float l = sdf_len3(wx, wy, wz);
        o[1] = sx * (wx / l); o[2] = sy * (wy / l); o[3] = sz * (wz / l);
    } else if (qx > qy && qx > qz) {
        o[1] = sx; o[2] = 0.0f; o[3] = 0.0f;
    } else if (qy > qz) {
        o[1] = 0.0f; o[2] = sy; o[3] = 0.0f;
    } else {
        o[1] = 0.0f; o[2] = 0.0f; o[3] = sz;
    }
}

static void sdg_torus(const float p[3], float major, float minor, float o[4]) { /* :136-157 */
    float lxz = sdf_len2(p[0], p[2]);
    float dx = lxz - major, dy = p[1];
    float ldir = sdf_len2(dx, dy);
    o[0] = ldir - minor; o[1] = 0.0f; o[2] = 1.0f; o[3] = 0.0f;
    if (lxz > 0.0001f && ldir > 0.0001f) {
        float ux = p[0] / lxz, uz = p[2] / lxz, ddx = dx / ldir, ddy = dy / ldir;
        o[1] = ux * ddx; o[2] = ddy; o[3] = uz * ddx;
    }
}

static void sdg_capsule(const float p[3], float h, float r, float o[4]) { /* :160-176 */
    float half = h * 0.5f;
    float cy = fminf(fmaxf(p[1], -half), half);
    float qx = p[0], qy = p[1] - cy, qz = p[2];
    float d = sdf_len3(qx, qy, qz);
    o[0] = d - r; o[1] = 0.0f; o[2] = sdf_sign(p[1]); o[3] = 0.0f;
    if (d > 0.0001f) { o[1] = qx / d; o[2] = qy / d; o[3] = qz / d; }
}

static void op_smooth_union(const float a[4], const float b[4], float k, float o[4]) { /* :206-224 */
    float k4 = k * 4.0f;
    float diff = fabsf(a[0] - b[0]);
    float h = fmaxf(k4 - diff, 0.0f) / k4;
    float hg = fmaxf(k4 - diff, 0.0f) / (2.0f * k4);
    float t = (a[0] < b[0]) ? hg : (1.0f - hg);
    float u = 1.0f - t;
    o[0] = fminf(a[0], b[0]) - ((h * h) * k4) * 0.25f;
    for (int c = 1; c < 4; ++c) o[c] = a[c] * u + b[c] * t;
}

void orc_sdf_scene(const orc_sdf_instr *prog, uint32_t n_instr, const float p[3], float out[4]) { /* :276-353 */
    float stack[16][4];
    int sp = 0;
    for (uint32_t k = 0; k < n_instr; ++k) {
        const orc_sdf_instr *in = &prog[k];
        float v[4];
        if (in->op < 16u) {
            float q[3] = {p[0] - in->a[0], p[1] - in->a[1], p[2] - in->a[2]};
            if (in->op == 0u) sdg_sphere(q, in->a[3], v);
            else if (in->op == 1u) sdg_box(q, &in->a[3], v);
            else if (in->op == 2u) sdg_torus(q, in->a[3], in->a[4], v);
            else sdg_capsule(q, in->a[3], in->a[4], v);
        } else {
            const float *b = stack[--sp];
            const float *a = stack[--sp];
            if (in->op == 16u) memcpy(v, (a[0] < b[0]) ? a : b, sizeof v);      /* opUnion :181-187 */
            else if (in->op == 17u) memcpy(v, (a[0] > b[0]) ? a : b, sizeof v); /* opIntersection :190-196 */
            else if (in->op == 18u) {                                           /* opSubtraction :199-202 */
                float nb[4] = {-b[0], -b[1], -b[2], -b[3]};
                memcpy(v, (a[0] > nb[0]) ? a : nb, sizeof v);
            } else op_smooth_union(a, b, in->a[0], v);
        }
        memcpy(stack[sp++], v, sizeof v);
    }
    if (sp == 0) { out[0] = 1000.0f; out[1] = 0.0f; out[2] = 1.0f; out[3] = 0.0f; return; } /* empty scene :282-286 */
    memcpy(out, stack[sp - 1], 4 * sizeof(float));
}

void orc_sdf_gradients(const orc_sdf_instr *prog, uint32_t n_instr, const float *positions, uint32_t n, float *gradients) {
    for (uint32_t i = 0; i < n; ++i) orc_sdf_scene(prog, n_instr, positions + (size_t)i * 4, gradients + (size_t)i * 4);
}

void orc_sdf_update_positions(const float *positions, const float *gradients, uint32_t n, float *next_positions) {
    for (uint32_t i = 0; i < n; ++i) { /* update-positions.wgsl:22-50 */
        const float *p = positions + (size_t)i * 4, *g = gradients + (size_t)i * 4;
        float *o = next_positions + (size_t)i * 4;
        float len = sdf_len3(g[1], g[2], g[3]);
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = 0.0f;
        if (len > 0.0001f) {
            o[0] = p[0] - (g[1] / len) * g[0];
            o[1] = p[1] - (g[2] / len) * g[0];
            o[2] = p[2] - (g[3] / len) * g[0];
        }
    }
}

void orc_sdf_scale_factors(const orc_sdf_instr *prog, uint32_t n_instr, const float *positions, uint32_t n, float *scale_factors) {
    const float r = 0.02f; /* sampleRadius, CurvatureSampler.ts:98 */
    for (uint32_t i = 0; i < n; ++i) {
        const float *c = positions + (size_t)i * 4;
        float cr[4];
        orc_sdf_scene(prog, n_instr, c, cr);
        float cl = sdf_len3(cr[1], cr[2], cr[3]);
        float nx = cr[1] / cl, ny = cr[2] / cl, nz = cr[3] / cl;
        float total = 0.0f;
        for (int k = 0; k < 6; ++k) { /* offsets in the reference's order :100-107 */
            float q[3] = {c[0] + ((k == 0) ? r : (k == 1) ? -r : 0.0f), c[1] + ((k == 2) ? r : (k == 3) ? -r : 0.0f),
                          c[2] + ((k == 4) ? r : (k == 5) ? -r : 0.0f)};
            float s[4];
            orc_sdf_scene(prog, n_instr, q, s);
            float sl = sdf_len3(s[1], s[2], s[3]);
            float d = (nx * (s[1] / sl) + ny * (s[2] / sl)) + nz * (s[3] / sl);
            total = total + (1.0f - d);
        }
        float avg = total / 6.0f;
        float t = fminf(fmaxf((avg - 0.0f) / (0.5f - 0.0f), 0.0f), 1.0f);
        float sm = (t * t) * (3.0f - 2.0f * t);
        float sf = 1.0f - sm;
        scale_factors[i] = 0.01f * (1.0f - sf) + 1.0f * sf;
    }
}

void orc_sdf_curvature(const float *gradients, const float *scale_factors, uint32_t n, float *curvature) {
    for (uint32_t i = 0; i < n; ++i) {
        const float *g = gradients + (size_t)i * 4;
        float *o = curvature + (size_t)i * 4;
        float len = sdf_len3(g[1], g[2], g[3]);
        o[0] = 0.0f; o[1] = 1.0f; o[2] = 0.0f; o[3] = scale_factors[i];
        if (len > 0.0001f) { o[0] = g[1] / len; o[1] = g[2] / len; o[2] = g[3] / len; }
    }
}

/* PointManager.generateRandomPositions (/root/reference/src/PointManager.ts:96-189) with the product's seeded generator
 * (include/splat.h: splat_sdf_seed_positions): the reference draws from an unseeded Math.random, so only the
 * distribution can follow it — a face of the box by area (:108-131), uniform on it (:133-186), w = 0 — and the draw itself
 * is defined here: point i of cloud `seed` uses two splitmix64 outputs of the counter seed * 0x9E3779B97F4A7C15 + 2 i (+ 1),
 * 24 bits per uniform.  One rounding per operation (built with -ffp-contract=off). */
static uint64_t orc_splitmix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_sdf_seed_positions(const float *mn, const float *mx, uint32_t n, uint64_t seed, float *positions) {
    const float k24 = 1.0f / 16777216.0f;
    const float d[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    const float ax = d[1] * d[2], ay = d[0] * d[2], az = d[0] * d[1];
    const float c0 = ax, c1 = c0 + ax, c2 = c1 + ay, c3 = c2 + ay, c4 = c3 + az, c5 = c4 + az;
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t base = seed * 0x9E3779B97F4A7C15ull + 2ull * i;
        const uint64_t a = orc_splitmix64(base), b = orc_splitmix64(base + 1ull);
        const float u0 = (float)(uint32_t)(a >> 40) * k24, u1 = (float)(uint32_t)((a >> 16) & 0xffffffu) * k24;
        const float u2 = (float)(uint32_t)(b >> 40) * k24, u3 = (float)(uint32_t)((b >> 16) & 0xffffffu) * k24;
        const float t = u0 * c5;
        const uint32_t face = t < c0 ? 0u : t < c1 ? 1u : t < c2 ? 2u : t < c3 ? 3u : t < c4 ? 4u : 5u;
        float p[3] = {mn[0] + u1 * d[0], mn[1] + u2 * d[1], mn[2] + u3 * d[2]};
        p[face >> 1] = (face & 1u) ? mx[face >> 1] : mn[face >> 1];
        positions[4 * i] = p[0];
        positions[4 * i + 1] = p[1];
        positions[4 * i + 2] = p[2];
        positions[4 * i + 3] = 0.0f;
    }
}
