/*
 * splat_napi.c — thin N-API addon over the C ABI of libsplat_hip.so (include/splat.h).
 *
 * One JS function per ABI entry point, same name without the "splat_" prefix.  Handles (ctx,
 * sorter, binner) are napi_externals; device pointers cross as JS numbers (GPU virtual addresses
 * are < 2^53); host data crosses as TypedArrays/ArrayBuffers.  A negative status becomes a thrown
 * JS Error carrying splat_last_error(), which is the reference's error behaviour (its getters
 * `throw new Error(...)`: /root/reference/src/GPUTileBinner.ts:340-359).
 *
 * Plain C, no node-addon-api, no node-gyp: built by the Makefile next to this file with
 *   gcc -shared -fPIC -I/usr/include/node splat_napi.c -o splat_napi.node -L.. -lsplat_hip
 */
#define NAPI_VERSION 6
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/splat.h"

#define MAX_ARGS 16

typedef struct {
    napi_env env;
    size_t argc;
    napi_value argv[MAX_ARGS];
    int failed;
} call_t;

static int get_args(napi_env env, napi_callback_info info, call_t *c, size_t want) {
    c->env = env;
    c->argc = MAX_ARGS;
    c->failed = 0;
    if (napi_get_cb_info(env, info, &c->argc, c->argv, NULL, NULL) != napi_ok || c->argc < want) {
        napi_throw_type_error(env, NULL, "wrong number of arguments");
        return 0;
    }
    return 1;
}

static void *arg_external(call_t *c, size_t i) {
    void *p = NULL;
    napi_valuetype t;
    napi_typeof(c->env, c->argv[i], &t);
    if (t == napi_null || t == napi_undefined) return NULL;
    if (napi_get_value_external(c->env, c->argv[i], &p) != napi_ok) {
        c->failed = 1;
        napi_throw_type_error(c->env, NULL, "expected a handle");
    }
    return p;
}

static double arg_number(call_t *c, size_t i) {
    double d = 0;
    if (napi_get_value_double(c->env, c->argv[i], &d) != napi_ok) {
        c->failed = 1;
        napi_throw_type_error(c->env, NULL, "expected a number");
    }
    return d;
}

static void *arg_dptr(call_t *c, size_t i) { /* device pointer as a number; null/undefined -> NULL */
    napi_valuetype t;
    napi_typeof(c->env, c->argv[i], &t);
    if (t == napi_null || t == napi_undefined) return NULL;
    return (void *)(uintptr_t)arg_number(c, i);
}

static void *arg_hostbuf(call_t *c, size_t i, size_t *bytes) { /* TypedArray, DataView or ArrayBuffer */
    bool is = false;
    void *data = NULL;
    size_t len = 0;
    napi_is_typedarray(c->env, c->argv[i], &is);
    if (is) {
        napi_typedarray_type tt;
        napi_value ab;
        size_t off;
        napi_get_typedarray_info(c->env, c->argv[i], &tt, &len, &data, &ab, &off);
        static const size_t esz[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
        len *= esz[tt];
    } else {
        napi_is_arraybuffer(c->env, c->argv[i], &is);
        if (is) {
            napi_get_arraybuffer_info(c->env, c->argv[i], &data, &len);
        } else {
            c->failed = 1;
            napi_throw_type_error(c->env, NULL, "expected a TypedArray or ArrayBuffer");
        }
    }
    if (bytes) *bytes = len;
    return data;
}

static napi_value mk_number(napi_env env, double v) {
    napi_value r;
    napi_create_double(env, v, &r);
    return r;
}

static napi_value mk_undefined(napi_env env) {
    napi_value r;
    napi_get_undefined(env, &r);
    return r;
}

static napi_value mk_external(napi_env env, void *p) {
    napi_value r;
    napi_create_external(env, p, NULL, NULL, &r);
    return r;
}

/* status -> thrown Error("libsplat_hip <code>: <message>") */
static napi_value check(napi_env env, splat_ctx *ctx, int rc, napi_value ok) {
    if (rc == SPLAT_OK) return ok;
    char msg[600];
    const char *m = splat_last_error(ctx);
    strcpy(msg, "libsplat_hip ");
    char num[16];
    int n = rc, k = 0;
    if (n < 0) { msg[strlen(msg) + 1] = 0; msg[strlen(msg)] = '-'; n = -n; }
    do { num[k++] = (char)('0' + n % 10); n /= 10; } while (n);
    while (k) { size_t l = strlen(msg); msg[l] = num[--k]; msg[l + 1] = 0; }
    strcat(msg, ": ");
    strncat(msg, m ? m : "", sizeof msg - strlen(msg) - 1);
    napi_throw_error(env, NULL, msg);
    return NULL;
}

/* A frame function that returns SPLAT_ERR_CAPACITY / SPLAT_ERR_RETRY is reporting the PREVIOUS (sync-free) frame — it
 * outgrew its pair limit, or its lists failed the order check; room has been made / the ranking switched — and has not
 * rendered this one: call it again (include/splat.h).  The Python facades do the same (host.py Renderer.render). */
#define AGAIN(rc) ((rc) == SPLAT_ERR_CAPACITY || (rc) == SPLAT_ERR_RETRY)

#define FN(name) static napi_value name(napi_env env, napi_callback_info info)
#define ARGS(n) call_t c; if (!get_args(env, info, &c, n)) return NULL
#define BAIL if (c.failed) return NULL

FN(abi_version) { (void)info; return mk_number(env, splat_abi_version()); }

FN(ctx_create) {
    ARGS(1);
    int dev = (int)arg_number(&c, 0); BAIL;
    splat_ctx *ctx = NULL;
    int rc = splat_ctx_create(dev, &ctx);
    return check(env, NULL, rc, rc == SPLAT_OK ? mk_external(env, ctx) : NULL);
}
FN(ctx_destroy) { ARGS(1); splat_ctx_destroy((splat_ctx *)arg_external(&c, 0)); return mk_undefined(env); }
FN(sync) { ARGS(1); splat_ctx *x = arg_external(&c, 0); BAIL; return check(env, x, splat_sync(x), mk_undefined(env)); }
FN(rank_status) { /* (ctx) -> [policy (0 checked | 1 atomic | 2 ballot), atomicsOrdered (1 | 0 | -1), orderFaults]: splat_rank_status */
    ARGS(1); splat_ctx *x = arg_external(&c, 0); BAIL;
    int pol = 0, ordered = 0; uint32_t faults = 0;
    int rc = splat_rank_status(x, &pol, &ordered, &faults);
    if (rc != SPLAT_OK) return check(env, x, rc, NULL);
    napi_value arr;
    if (napi_create_array_with_length(env, 3, &arr) != napi_ok) return NULL;
    napi_set_element(env, arr, 0, mk_number(env, pol));
    napi_set_element(env, arr, 1, mk_number(env, ordered));
    napi_set_element(env, arr, 2, mk_number(env, (double)faults));
    return arr;
}
FN(set_timing) { ARGS(2); splat_ctx *x = arg_external(&c, 0); int e = (int)arg_number(&c, 1); BAIL; return check(env, x, splat_set_timing(x, e), mk_undefined(env)); }
FN(stage_time_ms) {
    ARGS(2); splat_ctx *x = arg_external(&c, 0); int st = (int)arg_number(&c, 1); BAIL;
    float ms = 0; int rc = splat_stage_time_ms(x, st, &ms);
    return check(env, x, rc, mk_number(env, ms));
}
FN(buf_alloc) {
    ARGS(2); splat_ctx *x = arg_external(&c, 0); size_t b = (size_t)arg_number(&c, 1); BAIL;
    void *p = NULL; int rc = splat_buf_alloc(x, b, &p);
    return check(env, x, rc, mk_number(env, (double)(uintptr_t)p));
}
FN(buf_free) { ARGS(2); splat_ctx *x = arg_external(&c, 0); void *p = arg_dptr(&c, 1); BAIL; return check(env, x, splat_buf_free(x, p), mk_undefined(env)); }
FN(buf_zero) { ARGS(3); splat_ctx *x = arg_external(&c, 0); void *p = arg_dptr(&c, 1); size_t b = (size_t)arg_number(&c, 2); BAIL; return check(env, x, splat_buf_zero(x, p, b), mk_undefined(env)); }
FN(buf_upload) { /* (ctx, dptr, hostTypedArray) */
    ARGS(3); splat_ctx *x = arg_external(&c, 0); void *p = arg_dptr(&c, 1); size_t n = 0; void *h = arg_hostbuf(&c, 2, &n); BAIL;
    return check(env, x, splat_buf_upload(x, p, h, n), mk_undefined(env));
}
FN(buf_download) { /* (ctx, hostTypedArray, dptr) fills the whole array */
    ARGS(3); splat_ctx *x = arg_external(&c, 0); size_t n = 0; void *h = arg_hostbuf(&c, 1, &n); void *p = arg_dptr(&c, 2); BAIL;
    return check(env, x, splat_buf_download(x, h, p, n), mk_undefined(env));
}
FN(update_props) {
    ARGS(5); splat_ctx *x = arg_external(&c, 0); void *pos = arg_dptr(&c, 1), *cur = arg_dptr(&c, 2);
    uint32_t n = (uint32_t)arg_number(&c, 3); void *props = arg_dptr(&c, 4); BAIL;
    return check(env, x, splat_update_props(x, pos, cur, n, props), mk_undefined(env));
}
FN(update_props_planes) { /* (ctx, positions, curvature, n, posRadius, colorOpacity) */
    ARGS(6); splat_ctx *x = arg_external(&c, 0); void *pos = arg_dptr(&c, 1), *cur = arg_dptr(&c, 2);
    uint32_t n = (uint32_t)arg_number(&c, 3); void *pr = arg_dptr(&c, 4), *co = arg_dptr(&c, 5); BAIL;
    return check(env, x, splat_update_props_planes(x, pos, cur, n, pr, co), mk_undefined(env));
}
FN(props_to_planes) { /* (ctx, props, n, posRadius, colorOpacity) */
    ARGS(5); splat_ctx *x = arg_external(&c, 0); void *props = arg_dptr(&c, 1); uint32_t n = (uint32_t)arg_number(&c, 2);
    void *pr = arg_dptr(&c, 3), *co = arg_dptr(&c, 4); BAIL;
    return check(env, x, splat_props_to_planes(x, props, n, pr, co), mk_undefined(env));
}
FN(lit_colors) { /* (ctx, colorOpacity, cStride, normals, nStride, n, lit) */
    ARGS(7); splat_ctx *x = arg_external(&c, 0); void *col = arg_dptr(&c, 1); uint32_t cs = (uint32_t)arg_number(&c, 2);
    void *nrm = arg_dptr(&c, 3); uint32_t ns = (uint32_t)arg_number(&c, 4), n = (uint32_t)arg_number(&c, 5); void *lit = arg_dptr(&c, 6); BAIL;
    return check(env, x, splat_lit_colors(x, col, cs, nrm, ns, n, lit), mk_undefined(env));
}
FN(project) { /* (ctx, Float32Array(22), posRadius, strideVec4, n, projected, keys|null, payload|null, nPadded) */
    ARGS(9); splat_ctx *x = arg_external(&c, 0); size_t ub = 0; float *u = arg_hostbuf(&c, 1, &ub);
    void *pr = arg_dptr(&c, 2); uint32_t st = (uint32_t)arg_number(&c, 3), n = (uint32_t)arg_number(&c, 4);
    void *proj = arg_dptr(&c, 5), *keys = arg_dptr(&c, 6), *pay = arg_dptr(&c, 7); uint32_t np = (uint32_t)arg_number(&c, 8); BAIL;
    if (ub < 22 * sizeof(float)) { napi_throw_range_error(env, NULL, "uniform block needs 22 floats"); return NULL; }
    return check(env, x, splat_project(x, u, pr, st, n, proj, keys, pay, np), mk_undefined(env));
}
FN(project_disc) { /* (ctx, Float32Array(22), posRadius, strideVec4, normals, normalStrideVec4, n, projected, discs, keys|null, payload|null, nPadded) */
    ARGS(12); splat_ctx *x = arg_external(&c, 0); size_t ub = 0; float *u = arg_hostbuf(&c, 1, &ub);
    void *pr = arg_dptr(&c, 2); uint32_t st = (uint32_t)arg_number(&c, 3); void *nrm = arg_dptr(&c, 4);
    uint32_t ns = (uint32_t)arg_number(&c, 5), n = (uint32_t)arg_number(&c, 6);
    void *proj = arg_dptr(&c, 7), *discs = arg_dptr(&c, 8), *keys = arg_dptr(&c, 9), *pay = arg_dptr(&c, 10); uint32_t np = (uint32_t)arg_number(&c, 11); BAIL;
    if (ub < 22 * sizeof(float)) { napi_throw_range_error(env, NULL, "uniform block needs 22 floats"); return NULL; }
    return check(env, x, splat_project_disc(x, u, pr, st, nrm, ns, n, proj, discs, keys, pay, np), mk_undefined(env));
}
FN(extract_keys) {
    ARGS(6); splat_ctx *x = arg_external(&c, 0); void *proj = arg_dptr(&c, 1); uint32_t n = (uint32_t)arg_number(&c, 2), np = (uint32_t)arg_number(&c, 3);
    void *k = arg_dptr(&c, 4), *p = arg_dptr(&c, 5); BAIL;
    return check(env, x, splat_extract_keys(x, proj, n, np, k, p), mk_undefined(env));
}
FN(sort_create) {
    ARGS(2); splat_ctx *x = arg_external(&c, 0); uint32_t cap = (uint32_t)arg_number(&c, 1); BAIL;
    splat_sorter *s = NULL; int rc = splat_sort_create(x, cap, &s);
    return check(env, x, rc, rc == SPLAT_OK ? mk_external(env, s) : NULL);
}
FN(sort_destroy) { ARGS(1); splat_sort_destroy(arg_external(&c, 0)); return mk_undefined(env); }
FN(sort_capacity) { ARGS(1); return mk_number(env, splat_sort_capacity(arg_external(&c, 0))); }
FN(sort_keys) { ARGS(1); return mk_number(env, (double)(uintptr_t)splat_sort_keys(arg_external(&c, 0))); }
FN(sort_payload) { ARGS(1); return mk_number(env, (double)(uintptr_t)splat_sort_payload(arg_external(&c, 0))); }
FN(sort_sorted_payload) { ARGS(1); return mk_number(env, (double)(uintptr_t)splat_sort_sorted_payload(arg_external(&c, 0))); }
FN(sort_sorted_keys) { ARGS(1); return mk_number(env, (double)(uintptr_t)splat_sort_sorted_keys(arg_external(&c, 0))); }
FN(sort_run) { /* (ctx, sorter, n, bitBegin, bitEnd) */
    ARGS(5); splat_ctx *x = arg_external(&c, 0); splat_sorter *s = arg_external(&c, 1);
    uint32_t n = (uint32_t)arg_number(&c, 2), b0 = (uint32_t)arg_number(&c, 3), b1 = (uint32_t)arg_number(&c, 4); BAIL;
    return check(env, x, splat_sort_run(s, n, b0, b1), mk_undefined(env));
}
FN(sort_set_mode) { ARGS(3); splat_ctx *x = arg_external(&c, 0); splat_sorter *s = arg_external(&c, 1); int m = (int)arg_number(&c, 2); BAIL; return check(env, x, splat_sort_set_mode(s, m), mk_undefined(env)); }
FN(bin_set_frame_order) { ARGS(3); splat_ctx *x = arg_external(&c, 0); splat_binner *b = arg_external(&c, 1); int o = (int)arg_number(&c, 2); BAIL; return check(env, x, splat_bin_set_frame_order(b, o), mk_undefined(env)); }
FN(scan_u32) {
    ARGS(5); splat_ctx *x = arg_external(&c, 0); void *in = arg_dptr(&c, 1), *out = arg_dptr(&c, 2); uint32_t n = (uint32_t)arg_number(&c, 3);
    void *tot = arg_dptr(&c, 4); BAIL;
    return check(env, x, splat_scan_u32(x, in, out, n, tot), mk_undefined(env));
}
FN(bin_create) {
    ARGS(2); splat_ctx *x = arg_external(&c, 0); uint32_t t = (uint32_t)arg_number(&c, 1); BAIL;
    splat_binner *b = NULL; int rc = splat_bin_create(x, t, &b);
    return check(env, x, rc, rc == SPLAT_OK ? mk_external(env, b) : NULL);
}
FN(bin_destroy) { ARGS(1); splat_bin_destroy(arg_external(&c, 0)); return mk_undefined(env); }
FN(bin_run) { /* (ctx, binner, projected, nSplats, sorted, nSorted, W, H, row0, row1) */
    ARGS(10); splat_ctx *x = arg_external(&c, 0); splat_binner *b = arg_external(&c, 1); void *proj = arg_dptr(&c, 2);
    uint32_t ns = (uint32_t)arg_number(&c, 3); void *sorted = arg_dptr(&c, 4); uint32_t nso = (uint32_t)arg_number(&c, 5);
    uint32_t w = (uint32_t)arg_number(&c, 6), h = (uint32_t)arg_number(&c, 7), r0 = (uint32_t)arg_number(&c, 8), r1 = (uint32_t)arg_number(&c, 9); BAIL;
    return check(env, x, splat_bin_run(b, proj, ns, sorted, nso, w, h, r0, r1), mk_undefined(env));
}
#define BIN_GETTER(jsname, cfn)                                                                        \
    FN(jsname) {                                                                                       \
        ARGS(2); splat_ctx *x = arg_external(&c, 0); splat_binner *b = arg_external(&c, 1); BAIL;      \
        void *p = NULL; int rc = cfn(b, &p);                                                           \
        return check(env, x, rc, mk_number(env, (double)(uintptr_t)p));                                \
    }
BIN_GETTER(bin_counts, splat_bin_counts)
BIN_GETTER(bin_offsets, splat_bin_offsets)
BIN_GETTER(bin_indices, splat_bin_indices)
FN(bin_total) {
    ARGS(2); splat_ctx *x = arg_external(&c, 0); splat_binner *b = arg_external(&c, 1); BAIL;
    uint64_t t = 0; int rc = splat_bin_total(b, &t);
    return check(env, x, rc, mk_number(env, (double)t));
}
FN(validate_tile_order) { /* (ctx, projected, offsets, numTiles, indices, totalPairs) -> violations */
    ARGS(6); splat_ctx *x = arg_external(&c, 0); void *proj = arg_dptr(&c, 1), *off = arg_dptr(&c, 2); uint32_t nt = (uint32_t)arg_number(&c, 3);
    void *idx = arg_dptr(&c, 4); uint64_t total = (uint64_t)arg_number(&c, 5); BAIL;
    uint64_t v = 0; int rc = splat_validate_tile_order(x, proj, off, nt, idx, total, &v);
    return check(env, x, rc, mk_number(env, (double)v));
}
static void fill_cfg(call_t *c, size_t i, splat_composite_cfg *cfg) { /* [mode, earlyOut, tile, row0, row1, recordFormat?, prelit?, footprint?] */
    memset(cfg, 0, sizeof *cfg);
    uint32_t v[8] = {0, 1, 16, 0, 0xffffffffu, 0, 0, 0};
    bool is = false;
    napi_is_array(c->env, c->argv[i], &is);
    if (is)
        for (uint32_t k = 0; k < 8; ++k) {
            napi_value e;
            double d;
            if (napi_get_element(c->env, c->argv[i], k, &e) == napi_ok && napi_get_value_double(c->env, e, &d) == napi_ok) v[k] = (uint32_t)d;
        }
    cfg->mode = v[0]; cfg->early_out = v[1]; cfg->tile_size = v[2]; cfg->tile_row0 = v[3]; cfg->tile_row1 = v[4];
    cfg->record_format = v[5]; cfg->prelit = v[6]; cfg->footprint = v[7];
}
FN(composite) { /* (ctx, cfg[5], color, cStride, normals, nStride, projected, indices, counts, offsets, W, H, out8|null, outF|null) */
    ARGS(14); splat_ctx *x = arg_external(&c, 0); splat_composite_cfg cfg; fill_cfg(&c, 1, &cfg);
    void *col = arg_dptr(&c, 2); uint32_t cs = (uint32_t)arg_number(&c, 3); void *nrm = arg_dptr(&c, 4); uint32_t ns = (uint32_t)arg_number(&c, 5);
    void *proj = arg_dptr(&c, 6), *idx = arg_dptr(&c, 7), *cnt = arg_dptr(&c, 8), *off = arg_dptr(&c, 9);
    uint32_t w = (uint32_t)arg_number(&c, 10), h = (uint32_t)arg_number(&c, 11); void *o8 = arg_dptr(&c, 12), *of = arg_dptr(&c, 13); BAIL;
    return check(env, x, splat_composite(x, &cfg, col, cs, nrm, ns, proj, idx, cnt, off, w, h, o8, of, NULL), mk_undefined(env));
}
FN(render_frame) { /* (ctx, sorter, binner, cfg[5], Float32Array(22), props, normals, n, W, H, projected, out8|null, outF|null) */
    ARGS(13); splat_ctx *x = arg_external(&c, 0); splat_sorter *s = arg_external(&c, 1); splat_binner *b = arg_external(&c, 2);
    splat_composite_cfg cfg; fill_cfg(&c, 3, &cfg); size_t ub = 0; float *u = arg_hostbuf(&c, 4, &ub);
    void *props = arg_dptr(&c, 5), *nrm = arg_dptr(&c, 6); uint32_t n = (uint32_t)arg_number(&c, 7), w = (uint32_t)arg_number(&c, 8), h = (uint32_t)arg_number(&c, 9);
    void *proj = arg_dptr(&c, 10), *o8 = arg_dptr(&c, 11), *of = arg_dptr(&c, 12); BAIL;
    if (ub < 22 * sizeof(float)) { napi_throw_range_error(env, NULL, "uniform block needs 22 floats"); return NULL; }
    int rc = splat_render_frame(x, s, b, &cfg, u, props, nrm, n, w, h, proj, o8, of);
    if (AGAIN(rc)) rc = splat_render_frame(x, s, b, &cfg, u, props, nrm, n, w, h, proj, o8, of);
    return check(env, x, rc, mk_undefined(env));
}

FN(render_frame_planes) { /* (ctx, sorter, binner, cfg[5], Float32Array(22), posRadius, colorOpacity, normals, n, W, H, projected, out8|null, outF|null) */
    ARGS(14); splat_ctx *x = arg_external(&c, 0); splat_sorter *s = arg_external(&c, 1); splat_binner *b = arg_external(&c, 2);
    splat_composite_cfg cfg; fill_cfg(&c, 3, &cfg); size_t ub = 0; float *u = arg_hostbuf(&c, 4, &ub);
    void *pr = arg_dptr(&c, 5), *co = arg_dptr(&c, 6), *nrm = arg_dptr(&c, 7);
    uint32_t n = (uint32_t)arg_number(&c, 8), w = (uint32_t)arg_number(&c, 9), h = (uint32_t)arg_number(&c, 10);
    void *proj = arg_dptr(&c, 11), *o8 = arg_dptr(&c, 12), *of = arg_dptr(&c, 13); BAIL;
    if (ub < 22 * sizeof(float)) { napi_throw_range_error(env, NULL, "uniform block needs 22 floats"); return NULL; }
    int rc = splat_render_frame_planes(x, s, b, &cfg, u, pr, co, nrm, n, w, h, proj, o8, of);
    if (AGAIN(rc)) rc = splat_render_frame_planes(x, s, b, &cfg, u, pr, co, nrm, n, w, h, proj, o8, of);
    return check(env, x, rc, mk_undefined(env));
}

/* ---- SDF splat generation (include/splat.h: "SDF splat generation") ---- */
static int sdf_program(call_t *c, size_t i, splat_sdf_instr *prog, uint32_t *count) { /* Float32Array of 8 floats per instruction: [op, a0..a6] */
    size_t nb = 0; float *f = arg_hostbuf(c, i, &nb);
    if (c->failed) return 0;
    size_t n = nb / (8 * sizeof(float));
    if (n > SPLAT_SDF_MAX_INSTR) { c->failed = 1; napi_throw_range_error(c->env, NULL, "SDF program: too many instructions"); return 0; }
    for (size_t k = 0; k < n; k++) { prog[k].op = (uint32_t)f[k * 8]; for (int j = 0; j < 7; j++) prog[k].a[j] = f[k * 8 + 1 + j]; }
    *count = (uint32_t)n;
    return 1;
}
FN(sdf_gradients) { /* (ctx, program, positions, n, gradients) */
    ARGS(5); splat_ctx *x = arg_external(&c, 0); splat_sdf_instr prog[SPLAT_SDF_MAX_INSTR]; uint32_t cnt = 0;
    if (!sdf_program(&c, 1, prog, &cnt)) return NULL;
    void *pos = arg_dptr(&c, 2); uint32_t n = (uint32_t)arg_number(&c, 3); void *g = arg_dptr(&c, 4); BAIL;
    return check(env, x, splat_sdf_gradients(x, prog, cnt, pos, n, g), mk_undefined(env));
}
FN(sdf_update_positions) { /* (ctx, positions, gradients, n, nextPositions) */
    ARGS(5); splat_ctx *x = arg_external(&c, 0); void *pos = arg_dptr(&c, 1), *g = arg_dptr(&c, 2); uint32_t n = (uint32_t)arg_number(&c, 3);
    void *nx = arg_dptr(&c, 4); BAIL;
    return check(env, x, splat_sdf_update_positions(x, pos, g, n, nx), mk_undefined(env));
}
FN(sdf_scale_factors) { /* (ctx, program, positions, n, scaleFactors) */
    ARGS(5); splat_ctx *x = arg_external(&c, 0); splat_sdf_instr prog[SPLAT_SDF_MAX_INSTR]; uint32_t cnt = 0;
    if (!sdf_program(&c, 1, prog, &cnt)) return NULL;
    void *pos = arg_dptr(&c, 2); uint32_t n = (uint32_t)arg_number(&c, 3); void *sf = arg_dptr(&c, 4); BAIL;
    return check(env, x, splat_sdf_scale_factors(x, prog, cnt, pos, n, sf), mk_undefined(env));
}
FN(sdf_curvature) { /* (ctx, gradients, scaleFactors, n, curvature) */
    ARGS(5); splat_ctx *x = arg_external(&c, 0); void *g = arg_dptr(&c, 1), *sf = arg_dptr(&c, 2); uint32_t n = (uint32_t)arg_number(&c, 3);
    void *cur = arg_dptr(&c, 4); BAIL;
    return check(env, x, splat_sdf_curvature(x, g, sf, n, cur), mk_undefined(env));
}
FN(sdf_seed_positions) { /* (ctx, Float32Array(3) boxMin, Float32Array(3) boxMax, n, seed (integer < 2^53), positions) */
    ARGS(6); splat_ctx *x = arg_external(&c, 0); size_t b0 = 0, b1 = 0; float *mn = arg_hostbuf(&c, 1, &b0), *mx = arg_hostbuf(&c, 2, &b1);
    uint32_t n = (uint32_t)arg_number(&c, 3); double seed = arg_number(&c, 4); void *pos = arg_dptr(&c, 5); BAIL;
    if (b0 < 12 || b1 < 12) { napi_throw_range_error(env, NULL, "sdf_seed_positions: the box corners are three floats each"); return NULL; }
    return check(env, x, splat_sdf_seed_positions(x, mn, mx, n, (uint64_t)seed, pos), mk_undefined(env));
}
FN(sdf_generate) { /* (ctx, program, boxMin | null, boxMax | null, seed, positionsIn | null, n, steps, positionsOut, gradientsOut | null, curvatureOut, propsOut | null) */
    ARGS(12); splat_ctx *x = arg_external(&c, 0); splat_sdf_instr prog[SPLAT_SDF_MAX_INSTR]; uint32_t cnt = 0;
    if (!sdf_program(&c, 1, prog, &cnt)) return NULL;
    napi_valuetype t; napi_typeof(env, c.argv[2], &t);
    float *mn = NULL, *mx = NULL; size_t b0 = 12, b1 = 12;
    if (t != napi_null && t != napi_undefined) { mn = arg_hostbuf(&c, 2, &b0); mx = arg_hostbuf(&c, 3, &b1); }
    double seed = arg_number(&c, 4); void *pin = arg_dptr(&c, 5); uint32_t n = (uint32_t)arg_number(&c, 6), steps = (uint32_t)arg_number(&c, 7);
    void *pout = arg_dptr(&c, 8), *gout = arg_dptr(&c, 9), *cout = arg_dptr(&c, 10), *props = arg_dptr(&c, 11); BAIL;
    if (b0 < 12 || b1 < 12) { napi_throw_range_error(env, NULL, "sdf_generate: the box corners are three floats each"); return NULL; }
    return check(env, x, splat_sdf_generate(x, prog, cnt, mn, mx, (uint64_t)seed, pin, n, steps, pout, gout, cout, props), mk_undefined(env));
}

/* ---- multi-GPU band path (include/splat.h: "multi-GPU band path", "the multi-GPU frame's one exchange") ---- */
FN(project_slice_compact) { /* (ctx, Float32Array(22), posRadius, strideVec4, first, count, records16) */
    ARGS(7); splat_ctx *x = arg_external(&c, 0); size_t ub = 0; float *u = arg_hostbuf(&c, 1, &ub);
    void *pr = arg_dptr(&c, 2); uint32_t st = (uint32_t)arg_number(&c, 3), first = (uint32_t)arg_number(&c, 4), count = (uint32_t)arg_number(&c, 5);
    void *rec = arg_dptr(&c, 6); BAIL;
    if (ub < 22 * sizeof(float)) { napi_throw_range_error(env, NULL, "uniform block needs 22 floats"); return NULL; }
    return check(env, x, splat_project_slice_compact(x, u, pr, st, first, count, rec), mk_undefined(env));
}
FN(band_frame) { /* (ctx, sorter, binner, cfg[8], props, normals|null, records, nRecords, W, H, out8|null, outF|null) */
    ARGS(12); splat_ctx *x = arg_external(&c, 0); splat_sorter *s = arg_external(&c, 1); splat_binner *b = arg_external(&c, 2);
    splat_composite_cfg cfg; fill_cfg(&c, 3, &cfg);
    void *props = arg_dptr(&c, 4), *nrm = arg_dptr(&c, 5), *rec = arg_dptr(&c, 6);
    uint32_t n = (uint32_t)arg_number(&c, 7), w = (uint32_t)arg_number(&c, 8), h = (uint32_t)arg_number(&c, 9);
    void *o8 = arg_dptr(&c, 10), *of = arg_dptr(&c, 11); BAIL;
    int rc = splat_band_frame(x, s, b, &cfg, props, nrm, rec, n, w, h, o8, of, NULL);
    if (AGAIN(rc)) rc = splat_band_frame(x, s, b, &cfg, props, nrm, rec, n, w, h, o8, of, NULL);
    return check(env, x, rc, mk_undefined(env));
}
FN(band_settle) { /* (ctx, sorter, binner) -> pairs of the last band frame (waits for it; throws if it overflowed: render it again) */
    ARGS(3); splat_ctx *x = arg_external(&c, 0); splat_sorter *s = arg_external(&c, 1); splat_binner *b = arg_external(&c, 2); BAIL;
    uint32_t kept = 0; uint64_t pairs = 0;
    int rc = splat_band_settle(x, s, b, &kept, &pairs);
    return check(env, x, rc, rc == SPLAT_OK ? mk_number(env, (double)pairs) : NULL);
}
FN(comm_unique_id) { /* () -> ArrayBuffer(128): rank 0 makes it, every rank passes the same bytes to comm_init */
    (void)info;
    void *data = NULL; napi_value ab;
    if (napi_create_arraybuffer(env, SPLAT_COMM_ID_BYTES, &data, &ab) != napi_ok) return NULL;
    return check(env, NULL, splat_comm_unique_id(data), ab);
}
FN(comm_init) { /* (ctx, rank, world, idBytes) -> comm */
    ARGS(4); splat_ctx *x = arg_external(&c, 0); int rank = (int)arg_number(&c, 1), world = (int)arg_number(&c, 2);
    size_t nb = 0; void *id = arg_hostbuf(&c, 3, &nb); BAIL;
    if (nb < SPLAT_COMM_ID_BYTES) { napi_throw_range_error(env, NULL, "the communicator id is 128 bytes"); return NULL; }
    splat_comm *comm = NULL;
    int rc = splat_comm_init(x, rank, world, id, &comm);
    return check(env, x, rc, rc == SPLAT_OK ? mk_external(env, comm) : NULL);
}
FN(comm_destroy) { ARGS(1); splat_comm_destroy((splat_comm *)arg_external(&c, 0)); return mk_undefined(env); }
FN(allgather_records) { /* (ctx, comm, shard, gathered, bytesPerRank) */
    ARGS(5); splat_ctx *x = arg_external(&c, 0); splat_comm *comm = arg_external(&c, 1);
    void *shard = arg_dptr(&c, 2), *all = arg_dptr(&c, 3); size_t nb = (size_t)arg_number(&c, 4); BAIL;
    return check(env, x, splat_allgather_records(x, comm, shard, all, nb), mk_undefined(env));
}

/* ---- the rest of the ABI: timing detail, composite options, multi-GPU helpers, diagnostics ---- */
static napi_value mk_pair(napi_env env, double a, double b) {
    napi_value arr;
    if (napi_create_array_with_length(env, 2, &arr) != napi_ok) return NULL;
    napi_set_element(env, arr, 0, mk_number(env, a));
    napi_set_element(env, arr, 1, mk_number(env, b));
    return arr;
}
FN(ctx_create_on_stream) { /* (device, hipStream as a number) -> ctx */
    ARGS(2); int dev = (int)arg_number(&c, 0); void *stream = arg_dptr(&c, 1); BAIL;
    splat_ctx *ctx = NULL;
    int rc = splat_ctx_create_on_stream(dev, stream, &ctx);
    return check(env, NULL, rc, rc == SPLAT_OK ? mk_external(env, ctx) : NULL);
}
FN(last_error) { /* (ctx|null) -> string */
    ARGS(1); splat_ctx *x = arg_external(&c, 0); BAIL;
    const char *m = splat_last_error(x);
    napi_value r;
    napi_create_string_utf8(env, m ? m : "", NAPI_AUTO_LENGTH, &r);
    return r;
}
FN(set_timing_stages) { ARGS(2); splat_ctx *x = arg_external(&c, 0); uint32_t m = (uint32_t)arg_number(&c, 1); BAIL; return check(env, x, splat_set_timing_stages(x, m), mk_undefined(env)); }
FN(set_timing_sampling) { ARGS(2); splat_ctx *x = arg_external(&c, 0); uint32_t e = (uint32_t)arg_number(&c, 1); BAIL; return check(env, x, splat_set_timing_sampling(x, e), mk_undefined(env)); }
FN(stage_time_stats) { /* (ctx, stage) -> [samples, totalMs] */
    ARGS(2); splat_ctx *x = arg_external(&c, 0); int st = (int)arg_number(&c, 1); BAIL;
    uint32_t n = 0; double ms = 0;
    int rc = splat_stage_time_stats(x, st, &n, &ms);
    return rc == SPLAT_OK ? mk_pair(env, n, ms) : check(env, x, rc, NULL);
}
FN(timing_consumed) { /* (ctx) -> [staged, consumed] */
    ARGS(1); splat_ctx *x = arg_external(&c, 0); BAIL;
    uint64_t a = 0, b = 0;
    int rc = splat_timing_consumed(x, &a, &b);
    return rc == SPLAT_OK ? mk_pair(env, (double)a, (double)b) : check(env, x, rc, NULL);
}
FN(buf_copy) { /* (ctx, dst, src, bytes) */
    ARGS(4); splat_ctx *x = arg_external(&c, 0); void *d = arg_dptr(&c, 1), *s = arg_dptr(&c, 2); size_t nb = (size_t)arg_number(&c, 3); BAIL;
    return check(env, x, splat_buf_copy(x, d, s, nb), mk_undefined(env));
}
FN(probe_lds_atomic_order) { /* (ctx) -> mismatches */
    ARGS(1); splat_ctx *x = arg_external(&c, 0); BAIL;
    uint64_t m = 0;
    int rc = splat_probe_lds_atomic_order(x, &m);
    return check(env, x, rc, rc == SPLAT_OK ? mk_number(env, (double)m) : NULL);
}
FN(composite_forget_history) { ARGS(1); splat_ctx *x = arg_external(&c, 0); BAIL; return check(env, x, splat_composite_forget_history(x), mk_undefined(env)); }
FN(composite_options) { /* (ctx, kernel, ahead, predict): every call sets all three; -1 (kernel, predict) / 0 (ahead) = the process default (the environment's), as splat.h says */
    ARGS(4); splat_ctx *x = arg_external(&c, 0);
    int k = (int)arg_number(&c, 1), a = (int)arg_number(&c, 2), p = (int)arg_number(&c, 3); BAIL;
    return check(env, x, splat_composite_options(x, k, a, p), mk_undefined(env));
}
FN(bin_dims) { /* (binner) -> [tilesX, tilesY] */
    ARGS(1); splat_binner *b = arg_external(&c, 0); BAIL;
    uint32_t ntx = 0, nty = 0;
    int rc = splat_bin_dims(b, &ntx, &nty);
    return rc == SPLAT_OK ? mk_pair(env, ntx, nty) : check(env, NULL, rc, NULL);
}
FN(bin_tile_size) { ARGS(1); splat_binner *b = arg_external(&c, 0); BAIL; return mk_number(env, splat_bin_tile_size(b)); }
FN(project_slice) { /* (ctx, Float32Array(22), posRadius, strideVec4, first, count, projectedSlice) */
    ARGS(7); splat_ctx *x = arg_external(&c, 0); size_t ub = 0; float *u = arg_hostbuf(&c, 1, &ub);
    void *pr = arg_dptr(&c, 2); uint32_t st = (uint32_t)arg_number(&c, 3), first = (uint32_t)arg_number(&c, 4), count = (uint32_t)arg_number(&c, 5);
    void *out = arg_dptr(&c, 6); BAIL;
    if (ub < 22 * sizeof(float)) { napi_throw_range_error(env, NULL, "uniform block needs 22 floats"); return NULL; }
    return check(env, x, splat_project_slice(x, u, pr, st, first, count, out), mk_undefined(env));
}
FN(project_slice_disc) { /* (ctx, Float32Array(22), posRadius, strideVec4, normals, normalStrideVec4, first, count, records48Slice) */
    ARGS(9); splat_ctx *x = arg_external(&c, 0); size_t ub = 0; float *u = arg_hostbuf(&c, 1, &ub);
    void *pr = arg_dptr(&c, 2); uint32_t st = (uint32_t)arg_number(&c, 3); void *nrm = arg_dptr(&c, 4);
    uint32_t ns = (uint32_t)arg_number(&c, 5), first = (uint32_t)arg_number(&c, 6), count = (uint32_t)arg_number(&c, 7);
    void *out = arg_dptr(&c, 8); BAIL;
    if (ub < 22 * sizeof(float)) { napi_throw_range_error(env, NULL, "uniform block needs 22 floats"); return NULL; }
    return check(env, x, splat_project_slice_disc(x, u, pr, st, nrm, ns, first, count, out), mk_undefined(env));
}
FN(expand_compact) { /* (ctx, records16, n, indexBase, projected) */
    ARGS(5); splat_ctx *x = arg_external(&c, 0); void *rec = arg_dptr(&c, 1);
    uint32_t n = (uint32_t)arg_number(&c, 2), base = (uint32_t)arg_number(&c, 3); void *out = arg_dptr(&c, 4); BAIL;
    return check(env, x, splat_expand_compact(x, rec, n, base, out), mk_undefined(env));
}
FN(band_keys) { /* (ctx, sorter, projected, n, W, H, tile, row0, row1) -> kept */
    ARGS(9); splat_ctx *x = arg_external(&c, 0); splat_sorter *s = arg_external(&c, 1); void *proj = arg_dptr(&c, 2);
    uint32_t n = (uint32_t)arg_number(&c, 3), w = (uint32_t)arg_number(&c, 4), h = (uint32_t)arg_number(&c, 5), t = (uint32_t)arg_number(&c, 6);
    uint32_t r0 = (uint32_t)arg_number(&c, 7), r1 = (uint32_t)arg_number(&c, 8); BAIL;
    uint32_t kept = 0;
    int rc = splat_band_keys(x, s, proj, n, w, h, t, r0, r1, &kept);
    return check(env, x, rc, rc == SPLAT_OK ? mk_number(env, kept) : NULL);
}
FN(band_kept) { /* (ctx, sorter) -> kept */
    ARGS(2); splat_ctx *x = arg_external(&c, 0); splat_sorter *s = arg_external(&c, 1); BAIL;
    uint32_t kept = 0;
    int rc = splat_band_kept(x, s, &kept);
    return check(env, x, rc, rc == SPLAT_OK ? mk_number(env, kept) : NULL);
}
FN(comm_rank) { /* (comm) -> [rank, world] */
    ARGS(1); splat_comm *cm = arg_external(&c, 0); BAIL;
    int r = 0, w = 0;
    int rc = splat_comm_rank(cm, &r, &w);
    return rc == SPLAT_OK ? mk_pair(env, r, w) : check(env, NULL, rc, NULL);
}
FN(comm_count) { /* (comm) -> [ranks RCCL reports, this rank as RCCL reports it] */
    ARGS(1); splat_comm *cm = arg_external(&c, 0); BAIL;
    int n = 0, r = 0;
    int rc = splat_comm_count(cm, &n, &r);
    return rc == SPLAT_OK ? mk_pair(env, n, r) : check(env, NULL, rc, NULL);
}

static napi_value init(napi_env env, napi_value exports) {
#define EXPORT(name) { #name, NULL, name, NULL, NULL, NULL, napi_enumerable, NULL }
    napi_property_descriptor d[] = {
        EXPORT(abi_version), EXPORT(ctx_create), EXPORT(ctx_destroy), EXPORT(sync), EXPORT(rank_status), EXPORT(set_timing), EXPORT(stage_time_ms),
        EXPORT(buf_alloc), EXPORT(buf_free), EXPORT(buf_zero), EXPORT(buf_upload), EXPORT(buf_download), EXPORT(update_props), EXPORT(update_props_planes), EXPORT(props_to_planes), EXPORT(lit_colors),
        EXPORT(project), EXPORT(project_disc), EXPORT(extract_keys), EXPORT(sort_create), EXPORT(sort_destroy), EXPORT(sort_capacity), EXPORT(sort_keys),
        EXPORT(sort_payload), EXPORT(sort_sorted_payload), EXPORT(sort_sorted_keys), EXPORT(sort_run), EXPORT(sort_set_mode),
        EXPORT(scan_u32), EXPORT(bin_create), EXPORT(bin_destroy), EXPORT(bin_run), EXPORT(bin_counts), EXPORT(bin_offsets),
        EXPORT(bin_indices), EXPORT(bin_total), EXPORT(bin_set_frame_order), EXPORT(validate_tile_order), EXPORT(composite), EXPORT(render_frame), EXPORT(render_frame_planes),
        EXPORT(project_slice_compact), EXPORT(band_frame), EXPORT(band_settle), EXPORT(comm_unique_id), EXPORT(comm_init), EXPORT(comm_destroy),
        EXPORT(ctx_create_on_stream), EXPORT(last_error), EXPORT(set_timing_stages), EXPORT(set_timing_sampling), EXPORT(stage_time_stats), EXPORT(timing_consumed),
        EXPORT(buf_copy), EXPORT(probe_lds_atomic_order),
        EXPORT(composite_forget_history), EXPORT(composite_options), EXPORT(bin_dims), EXPORT(bin_tile_size), EXPORT(project_slice),
        EXPORT(project_slice_disc), EXPORT(expand_compact), EXPORT(band_keys), EXPORT(band_kept), EXPORT(comm_rank), EXPORT(comm_count),
        EXPORT(allgather_records), EXPORT(sdf_gradients), EXPORT(sdf_update_positions), EXPORT(sdf_scale_factors), EXPORT(sdf_curvature), EXPORT(sdf_seed_positions), EXPORT(sdf_generate),
    };
    napi_define_properties(env, exports, sizeof d / sizeof d[0], d);
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
