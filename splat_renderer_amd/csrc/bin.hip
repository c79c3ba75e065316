// bin.hip — GPUTileBinner: count -> scan -> order-preserving fill, all on the device.
//
// Reference: /root/reference/src/GPUTileBinner.ts:190-338 (binSplats), src/shaders/count-tile-hits.wgsl:41-65
// (K7), src/shaders/fill-tile-lists.wgsl:51-81 (K9).  The reference's fill appends with atomicAdd, so
// its per-tile order is racy; the only deterministic spec is the CPU loop
// TileBinner.binSorted (src/TileBinner.ts:426-495), which this file reproduces bit-exactly:
//   - bounds are clamped to the screen and a splat with an empty clamped box is culled (:432-437),
//   - each tile's list holds its splats in `sorted` order (:470-495).
//
// Order-preserving fill without a serial loop: pairs (tileId, splatIdx) are EXPANDED in sorted
// order (an exclusive scan of the per-splat hit counts gives every splat its output slot), then
// a STABLE radix sort on the tile-id bits (2 x 8-bit passes for up to 65 536 tiles) groups them
// by tile while keeping the depth order inside each tile.
//
// Roofline: HBM.  Algorithmic bytes: count N*(4+16), fill N*(4+16)+4P; the
// implementation's own extra traffic (hit counts, packed ranges, the pair sort's ping-pong) is
// accounted in DESIGN.md.
#include "common.h"
#include "tile_range.h"

struct splat_binner {
    splat_ctx *ctx = nullptr;
    uint32_t tile = 16;
    uint32_t ntx = 0, nty = 0;
    uint32_t tiles_cap = 0, splats_cap = 0;
    uint32_t *counts = nullptr, *offsets = nullptr; // per tile
    uint32_t *hits = nullptr;                       // per sorted position: #tiles, then exclusive pair offset
    uint2 *ranges = nullptr;                        // per sorted position: packed clamped tile range
    uint32_t *d_total = nullptr;
    splat_sorter pairs;                             // (tileId, splatIdx) ping-pong buffers
    uint64_t total = 0;
    bool ran = false;
};

__global__ __launch_bounds__(256) void k_bin_count(const float4 *__restrict__ projected, uint32_t n_splats,
                                                   const uint32_t *__restrict__ sorted, uint32_t n_sorted, uint32_t width,
                                                   uint32_t height, uint32_t tile, uint32_t ntx, uint32_t nty, uint32_t row0,
                                                   uint32_t row1, uint32_t *__restrict__ hits, uint2 *__restrict__ ranges) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_sorted) return;
    uint32_t s = sorted[i];
    uint32_t tx0 = 1, tx1 = 0, ty0 = 1, ty1 = 0, h = 0;
    if (s < n_splats) { // 0xFFFFFFFF padding bins nowhere
        float4 b = projected[(size_t)s * 2];
        if (tile_range(b, width, height, tile, ntx, nty, row0, row1, tx0, tx1, ty0, ty1)) {
            h = (tx1 - tx0 + 1) * (ty1 - ty0 + 1);
        }
    }
    hits[i] = h;
    ranges[i] = make_uint2(tx0 | (tx1 << 16), ty0 | (ty1 << 16));
}

__global__ __launch_bounds__(256) void k_bin_expand(const uint32_t *__restrict__ sorted, uint32_t n_sorted,
                                                    const uint32_t *__restrict__ pair_off, const uint2 *__restrict__ ranges,
                                                    uint32_t ntx, uint32_t *__restrict__ pair_tile,
                                                    uint32_t *__restrict__ pair_splat) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_sorted) return;
    uint2 r = ranges[i];
    uint32_t tx0 = r.x & 0xffffu, tx1 = r.x >> 16, ty0 = r.y & 0xffffu, ty1 = r.y >> 16;
    if (tx0 > tx1 || ty0 > ty1) return;
    uint32_t s = sorted[i];
    uint32_t o = pair_off[i];
    for (uint32_t ty = ty0; ty <= ty1; ++ty)
        for (uint32_t tx = tx0; tx <= tx1; ++tx) {
            pair_tile[o] = ty * ntx + tx;
            pair_splat[o] = s;
            ++o;
        }
}

// Tile offsets from the tile-sorted pair keys: the thread at the first pair of tile t writes
// offsets[u] = i for every u in (previous tile, t] (empty tiles in between start where t starts),
// the thread at the last pair closes the tail with offsets[u] = P.  offsets has T+1 entries.
// Same values as the exclusive scan of the counts (TileBinner.ts:452-459), with no atomics:
// 11M global atomic increments cost 1.03 ms at C2, this costs a few microseconds.
__global__ __launch_bounds__(256) void k_tile_offsets(const uint32_t *__restrict__ sorted_tiles, uint32_t pairs,
                                                      uint32_t tiles, uint32_t *__restrict__ offsets) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= pairs) return;
    const uint32_t t = sorted_tiles[i];
    const int64_t prev = (i == 0) ? -1 : (int64_t)sorted_tiles[i - 1];
    if ((int64_t)t != prev)
        for (int64_t u = prev + 1; u <= (int64_t)t; ++u) offsets[u] = i;
    if (i == pairs - 1)
        for (uint32_t u = t + 1; u <= tiles; ++u) offsets[u] = pairs;
}

__global__ __launch_bounds__(256) void k_tile_counts(const uint32_t *__restrict__ offsets, uint32_t tiles,
                                                     uint32_t *__restrict__ counts) {
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < tiles) counts[t] = offsets[t + 1] - offsets[t];
}

static void binner_free(splat_binner *b) {
    if (b->counts) (void)hipFree(b->counts);
    if (b->offsets) (void)hipFree(b->offsets);
    if (b->hits) (void)hipFree(b->hits);
    if (b->ranges) (void)hipFree(b->ranges);
    b->counts = b->offsets = b->hits = nullptr;
    b->ranges = nullptr;
    b->tiles_cap = b->splats_cap = 0;
}

static void sorter_free_members(splat_sorter *s) {
    if (s->keys) (void)hipFree(s->keys);
    if (s->keys_b) (void)hipFree(s->keys_b);
    if (s->payload) (void)hipFree(s->payload);
    if (s->payload_b) (void)hipFree(s->payload_b);
    if (s->hist) (void)hipFree(s->hist);
    s->keys = s->keys_b = s->payload = s->payload_b = s->hist = nullptr;
    s->capacity = 0;
}

extern "C" {

int splat_bin_create(splat_ctx *ctx, uint32_t tile_size, splat_binner **out) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, out != nullptr);
    *out = nullptr;
    ARG_CHECK(ctx, tile_size >= 1 && tile_size <= 4096);
    splat_binner *b = new splat_binner();
    b->ctx = ctx;
    b->tile = tile_size;
    b->pairs.ctx = ctx;
    if (hipMalloc((void **)&b->d_total, 16) != hipSuccess) {
        delete b;
        return ctx_fail(ctx, SPLAT_ERR_OOM, "binner hipMalloc");
    }
    *out = b;
    return SPLAT_OK;
}

void splat_bin_destroy(splat_binner *b) {
    if (!b) return;
    (void)hipStreamSynchronize(b->ctx->stream);
    binner_free(b);
    sorter_free_members(&b->pairs);
    if (b->d_total) (void)hipFree(b->d_total);
    delete b;
}

int splat_bin_run(splat_binner *b, const void *projected, uint32_t n_splats, const void *sorted, uint32_t n_sorted,
                  uint32_t width, uint32_t height, uint32_t tile_row0, uint32_t tile_row1) {
    if (!b) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "binner is NULL");
    splat_ctx *ctx = b->ctx;
    ARG_CHECK(ctx, width >= 1 && height >= 1);
    ARG_CHECK(ctx, n_sorted == 0 || (projected && sorted));
    const uint32_t ntx = div_up(width, b->tile), nty = div_up(height, b->tile); // GPUTileBinner.ts:198-200
    ARG_CHECK(ctx, ntx <= 65535 && nty <= 65535 && (uint64_t)ntx * nty <= (1u << 24));
    const uint32_t tiles = ntx * nty;
    if (tile_row1 > nty) tile_row1 = nty;
    if (tile_row0 > tile_row1) tile_row0 = tile_row1;

    if (tiles > b->tiles_cap || n_sorted > b->splats_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        uint32_t tc = tiles > b->tiles_cap ? tiles : b->tiles_cap;
        uint32_t sc = n_sorted > b->splats_cap ? n_sorted : b->splats_cap;
        binner_free(b);
        if (hipMalloc((void **)&b->counts, (size_t)tc * 4 + 16) != hipSuccess ||
            hipMalloc((void **)&b->offsets, (size_t)tc * 4 + 16) != hipSuccess ||
            hipMalloc((void **)&b->hits, (size_t)sc * 4 + 16) != hipSuccess ||
            hipMalloc((void **)&b->ranges, (size_t)sc * 8 + 16) != hipSuccess) {
            binner_free(b);
            return ctx_fail(ctx, SPLAT_ERR_OOM, "binner hipMalloc");
        }
        b->tiles_cap = tc;
        b->splats_cap = sc;
    }
    b->ntx = ntx;
    b->nty = nty;
    b->ran = false;

    stage_begin(ctx, SPLAT_STAGE_BIN);
    uint32_t total32 = 0;
    if (n_sorted > 0) {
        hipLaunchKernelGGL(k_bin_count, dim3(div_up(n_sorted, 256)), dim3(256), 0, ctx->stream, (const float4 *)projected,
                           n_splats, (const uint32_t *)sorted, n_sorted, width, height, b->tile, ntx, nty, tile_row0,
                           tile_row1, b->hits, b->ranges);
        LAUNCH_CHECK(ctx, "k_bin_count");
        int rc = scan_exclusive_u32(ctx, b->hits, b->hits, n_sorted, b->d_total);
        if (rc != SPLAT_OK) return rc;
    }
    int rc = SPLAT_OK;
    if (n_sorted > 0) {
        // the one host round trip of the frame: the pair total sizes the fill (the reference reads
        // back all T counts here: GPUTileBinner.ts:244-263)
        rc = ctx_ensure_pinned(ctx, 16);
        if (rc != SPLAT_OK) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned, b->d_total, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        total32 = *(volatile uint32_t *)ctx->pinned;
    }
    b->total = total32;
    if (total32 > 0) {
        if (total32 > b->pairs.capacity) {
            uint64_t want = (uint64_t)total32 + total32 / 4 + 4096;
            if (want > 0xfffff000ull) want = 0xfffff000ull;
            rc = sorter_reserve(&b->pairs, (uint32_t)want);
            if (rc != SPLAT_OK) return rc;
        }
        hipLaunchKernelGGL(k_bin_expand, dim3(div_up(n_sorted, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)sorted,
                           n_sorted, b->hits, b->ranges, ntx, b->pairs.keys, b->pairs.payload);
        LAUNCH_CHECK(ctx, "k_bin_expand");
        uint32_t bits = 1;
        while ((1u << bits) < tiles) ++bits;
        rc = radix_sort_pairs(ctx, b->pairs.keys, b->pairs.payload, b->pairs.keys_b, b->pairs.payload_b, b->pairs.hist,
                              total32, 0, bits, &b->pairs.result_in_primary);
        if (rc != SPLAT_OK) return rc;
        const uint32_t *sorted_tiles = b->pairs.result_in_primary ? b->pairs.keys : b->pairs.keys_b;
        hipLaunchKernelGGL(k_tile_offsets, dim3(div_up(total32, 256)), dim3(256), 0, ctx->stream, sorted_tiles, total32, tiles,
                           b->offsets);
        LAUNCH_CHECK(ctx, "k_tile_offsets");
        hipLaunchKernelGGL(k_tile_counts, dim3(div_up(tiles, 256)), dim3(256), 0, ctx->stream, b->offsets, tiles, b->counts);
        LAUNCH_CHECK(ctx, "k_tile_counts");
    } else {
        HIP_TRY(ctx, hipMemsetAsync(b->counts, 0, (size_t)tiles * 4, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(b->offsets, 0, (size_t)(tiles + 1) * 4, ctx->stream));
        if (b->pairs.capacity == 0) {
            rc = sorter_reserve(&b->pairs, 1); // so getTileIndicesBuffer() has something to return ("at least 4 bytes" :288)
            if (rc != SPLAT_OK) return rc;
        }
    }
    stage_end(ctx, SPLAT_STAGE_BIN);
    b->ran = true;
    return SPLAT_OK;
}

uint32_t splat_bin_tile_size(const splat_binner *b) { return b ? b->tile : 0; }

int splat_bin_counts(splat_binner *b, void **dptr) {
    if (!b || !dptr) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "binner/dptr is NULL");
    if (!b->ran) return ctx_fail(b->ctx, SPLAT_ERR_STATE, "Tile counts buffer not initialized"); // GPUTileBinner.ts:354-359
    *dptr = b->counts;
    return SPLAT_OK;
}

int splat_bin_offsets(splat_binner *b, void **dptr) {
    if (!b || !dptr) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "binner/dptr is NULL");
    if (!b->ran) return ctx_fail(b->ctx, SPLAT_ERR_STATE, "Tile offsets buffer not initialized"); // :340-345
    *dptr = b->offsets;
    return SPLAT_OK;
}

int splat_bin_indices(splat_binner *b, void **dptr) {
    if (!b || !dptr) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "binner/dptr is NULL");
    if (!b->ran) return ctx_fail(b->ctx, SPLAT_ERR_STATE, "Tile indices buffer not initialized"); // :347-352
    *dptr = b->pairs.result_in_primary ? b->pairs.payload : b->pairs.payload_b;
    return SPLAT_OK;
}

int splat_bin_total(splat_binner *b, uint64_t *total_pairs) {
    if (!b || !total_pairs) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "binner/total is NULL");
    if (!b->ran) return ctx_fail(b->ctx, SPLAT_ERR_STATE, "binSplats has not run");
    *total_pairs = b->total;
    return SPLAT_OK;
}

int splat_bin_dims(splat_binner *b, uint32_t *ntx, uint32_t *nty) {
    if (!b) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "binner is NULL");
    if (!b->ran) return ctx_fail(b->ctx, SPLAT_ERR_STATE, "binSplats has not run");
    if (ntx) *ntx = b->ntx;
    if (nty) *nty = b->nty;
    return SPLAT_OK;
}

} // extern "C"
