// frame.hip — whole-frame convenience and the multi-GPU band filter.
//
// splat_render_frame strings the stages in the reference's intended order (from
// /root/reference/GPU_PIPELINE_PLAN.md:41-84): project(+keys) -> sort -> binSplats -> composite.
// splat_band_keys has no reference equivalent (the reference is single-device): it is the
// "keep splats whose tile-row range meets my band" step of SURVEY §8e, done as a stable
// compaction so that ties in depth still resolve by ascending global index.
#include "common.h"
#include "tile_range.h"

__device__ __forceinline__ uint32_t depth_key_of(float depth) {
    uint32_t bits = __float_as_uint(depth);
    uint32_t mask = ((bits >> 31) == 1u) ? 0xffffffffu : 0x80000000u; // extract-depth-keys.wgsl:57-58
    return bits ^ mask;
}

__global__ __launch_bounds__(256) void k_band_flag(const float4 *__restrict__ projected, uint32_t n, uint32_t width,
                                                   uint32_t height, uint32_t tile, uint32_t ntx, uint32_t nty, uint32_t row0,
                                                   uint32_t row1, uint32_t *__restrict__ flags) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t a, b, c, d;
    flags[i] = tile_range(projected[(size_t)i * 2], width, height, tile, ntx, nty, row0, row1, a, b, c, d) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_band_scatter(const float4 *__restrict__ projected, uint32_t n,
                                                      const uint32_t *__restrict__ slot, const uint32_t *__restrict__ total,
                                                      uint32_t *__restrict__ keys, uint32_t *__restrict__ payload) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    // slot[] is the exclusive scan of the flags: element i was kept iff slot[i+1] (or the total) is larger
    uint32_t s = slot[i];
    uint32_t next = (i + 1 < n) ? slot[i + 1] : *total;
    if (next == s) return;
    keys[s] = depth_key_of(reinterpret_cast<const float *>(projected)[(size_t)i * 8 + 4]);
    payload[s] = i;
}

// band filter, count left on the device in sorter->d_count (no host round trip)
static int band_keys_device(splat_ctx *ctx, splat_sorter *sorter, const void *projected, uint32_t n, uint32_t width,
                            uint32_t height, uint32_t tile_size, uint32_t tile_row0, uint32_t tile_row1) {
    if (n > sorter->capacity) return ctx_fail(ctx, SPLAT_ERR_CAPACITY, "splat_band_keys: n exceeds the sorter's capacity");
    const uint32_t ntx = div_up(width, tile_size), nty = div_up(height, tile_size);
    if (tile_row1 > nty) tile_row1 = nty;
    if (tile_row0 > tile_row1) tile_row0 = tile_row1;
    // flags live in the sorter's alternate key buffer, the kept count in its device counter
    uint32_t *flags = sorter->keys_b;
    uint32_t *d_total = sorter->d_count;
    stage_begin(ctx, SPLAT_STAGE_PROJECT);
    hipLaunchKernelGGL(k_band_flag, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, (const float4 *)projected, n, width,
                       height, tile_size, ntx, nty, tile_row0, tile_row1, flags);
    LAUNCH_CHECK(ctx, "k_band_flag");
    int rc = scan_exclusive_u32(ctx, flags, flags, n, d_total);
    if (rc != SPLAT_OK) return rc;
    hipLaunchKernelGGL(k_band_scatter, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, (const float4 *)projected, n, flags,
                       d_total, sorter->keys, sorter->payload);
    LAUNCH_CHECK(ctx, "k_band_scatter");
    stage_end(ctx, SPLAT_STAGE_PROJECT);
    return SPLAT_OK;
}

extern "C" {

int splat_band_keys(splat_ctx *ctx, splat_sorter *sorter, const void *projected, uint32_t n, uint32_t width, uint32_t height,
                    uint32_t tile_size, uint32_t tile_row0, uint32_t tile_row1, uint32_t *n_kept_host) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, sorter && n_kept_host && (n == 0 || projected));
    ARG_CHECK(ctx, tile_size >= 1 && width >= 1 && height >= 1);
    *n_kept_host = 0;
    if (n == 0) return SPLAT_OK;
    int rc = band_keys_device(ctx, sorter, projected, n, width, height, tile_size, tile_row0, tile_row1);
    if (rc != SPLAT_OK) return rc;
    rc = ctx_ensure_pinned(ctx, 16);
    if (rc != SPLAT_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned, sorter->d_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *n_kept_host = *(volatile uint32_t *)ctx->pinned;
    return SPLAT_OK;
}

int splat_band_frame(splat_ctx *ctx, splat_sorter *sorter, splat_binner *binner, const splat_composite_cfg *cfg,
                     const void *props, const void *normals, const void *records, uint32_t n_records, uint32_t width,
                     uint32_t height, void *out_rgba8, void *out_rgba32f, void *consumed_dptr) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, sorter && binner && cfg && props && normals && (n_records == 0 || records));
    ARG_CHECK(ctx, cfg->tile_size == splat_bin_tile_size(binner) && width >= 1 && height >= 1);
    const uint32_t tile = cfg->tile_size, nty = div_up(height, tile);
    uint32_t row0 = cfg->tile_row0, row1 = cfg->tile_row1 > nty ? nty : cfg->tile_row1;
    if (row0 > row1) row0 = row1;
    // keep -> sort -> bin with the kept count living on the device: no host round trip in here
    int rc = SPLAT_OK;
    if (n_records > 0) {
        rc = band_keys_device(ctx, sorter, records, n_records, width, height, tile, row0, row1);
        if (rc != SPLAT_OK) return rc;
        stage_begin(ctx, SPLAT_STAGE_SORT);
        rc = radix_sort_pairs(ctx, sorter->keys, sorter->payload, sorter->keys_b, sorter->payload_b, sorter->hist, n_records, 0, 32,
                              &sorter->result_in_primary, 0, sorter->d_count);
        stage_end(ctx, SPLAT_STAGE_SORT);
        if (rc != SPLAT_OK) return rc;
        sorter->ran = true;
    }
    rc = binner_run(binner, records, n_records, splat_sort_sorted_payload(sorter), n_records, width, height, row0, row1, nullptr,
                    n_records ? sorter->d_count : nullptr);
    if (rc != SPLAT_OK) return rc;
    void *indices = binner->pairs.result_in_primary ? binner->pairs.payload : binner->pairs.payload_b;
    splat_composite_cfg c2 = *cfg;
    c2.tile_row0 = row0;
    c2.tile_row1 = row1;
    return splat_composite(ctx, &c2, (const char *)props + 16, 2, normals, 1, records, indices, binner->counts, binner->offsets, width,
                           height, out_rgba8, out_rgba32f, consumed_dptr);
}

int splat_band_kept(splat_ctx *ctx, splat_sorter *sorter, uint32_t *n_kept_host) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, sorter && n_kept_host);
    int rc = ctx_ensure_pinned(ctx, 16);
    if (rc != SPLAT_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned, sorter->d_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *n_kept_host = *(volatile uint32_t *)ctx->pinned;
    return SPLAT_OK;
}

int splat_render_frame(splat_ctx *ctx, splat_sorter *sorter, splat_binner *binner, const splat_composite_cfg *cfg,
                       const float *uniforms, const void *props, const void *normals, uint32_t n, uint32_t width,
                       uint32_t height, void *projected, void *out_rgba8, void *out_rgba32f) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, sorter && binner && cfg && uniforms && props && normals && projected);
    ARG_CHECK(ctx, cfg->tile_size == splat_bin_tile_size(binner));
    if (n > splat_sort_capacity(sorter)) return ctx_fail(ctx, SPLAT_ERR_CAPACITY, "splat_render_frame: n exceeds the sorter's capacity");
    ARG_CHECK(ctx, width >= 1 && height >= 1);
    // SplatProjector.project + DepthKeyExtractor.extract fused; props is the interleaved buffer.
    // When tile coordinates fit 8 bits the projector also emits each splat's clamped tile range,
    // which turns the binner's 16-byte bounds gather (in sorted order) into a 4-byte one.
    const uint32_t tile = splat_bin_tile_size(binner);
    const uint32_t ntx = div_up(width, tile), nty = div_up(height, tile);
    uint32_t row0 = cfg->tile_row0, row1 = cfg->tile_row1 > nty ? nty : cfg->tile_row1;
    if (row0 > row1) row0 = row1;
    const bool fast = ntx <= 256 && nty <= 256 && n > 0;
    int rc = SPLAT_OK;
    uint32_t *range32 = nullptr;
    const BinParams bp = {width, height, tile, ntx, nty, row0, row1};
    if (fast) {
        rc = binner_reserve_range32(binner, n);
        if (rc != SPLAT_OK) return rc;
        range32 = binner->range32;
    }
    ARG_CHECK(ctx, (((uintptr_t)props | (uintptr_t)projected) & 15) == 0);
    // (the payload array is not written: payload = splat index, synthesised by the sort's first pass)
    rc = project_launch(ctx, uniforms, props, 2, n, 0, projected, splat_sort_keys(sorter), nullptr, n, range32, &bp);
    if (rc != SPLAT_OK) return rc;
    stage_begin(ctx, SPLAT_STAGE_SORT);
    rc = radix_sort_pairs(ctx, sorter->keys, sorter->payload, sorter->keys_b, sorter->payload_b, sorter->hist, n, 0, 32,
                          &sorter->result_in_primary, 0, nullptr, true); // RadixSorter.sort()
    stage_end(ctx, SPLAT_STAGE_SORT);
    if (rc != SPLAT_OK) return rc;
    sorter->ran = true;
    rc = binner_run(binner, projected, n, splat_sort_sorted_payload(sorter), n, width, height, row0, row1, range32);
    if (rc != SPLAT_OK) return rc;
    // (fields, not the public getters: those wait for a sync-free frame's pair total to come back)
    void *counts = binner->counts, *offsets = binner->offsets;
    void *indices = binner->pairs.result_in_primary ? binner->pairs.payload : binner->pairs.payload_b;
    const char *color = (const char *)props + 16; // second vec4 of each interleaved record
    return splat_composite(ctx, cfg, color, 2, normals, 1, projected, indices, counts, offsets, width, height, out_rgba8,
                           out_rgba32f, ctx->timing ? (void *)ctx->d_consumed : nullptr);
}

} // extern "C"
