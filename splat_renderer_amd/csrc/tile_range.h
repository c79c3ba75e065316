// tile_range.h — clamped tile range of one ProjectedSplat, shared by bin.hip and frame.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Tile range of one ProjectedSplat, TileBinner.ts:432-442 (f64 like the JS).  Returns false when
// the splat bins nowhere.
__device__ __forceinline__ bool tile_range(float4 bounds, uint32_t width, uint32_t height, uint32_t tile, uint32_t ntx,
                                           uint32_t nty, uint32_t row0, uint32_t row1, uint32_t &tx0, uint32_t &tx1,
                                           uint32_t &ty0, uint32_t &ty1) {
    if (isnan(bounds.x) || isnan(bounds.y) || isnan(bounds.z) || isnan(bounds.w)) return false;
    // max/min/compare of f32 values against integers < 2^24 are exact in f32
    float min_x = fmaxf(bounds.x, 0.0f), min_y = fmaxf(bounds.y, 0.0f);
    float max_x = fminf(bounds.z, (float)width), max_y = fminf(bounds.w, (float)height);
    if (min_x >= max_x || min_y >= max_y) return false;
    if ((tile & (tile - 1)) == 0) {
        // power-of-two tile: the clamped bounds are f32 values in [0, 2^20) and scaling by 1/tile is
        // exact in f32, so this is the same result as the f64 path below without its f64 divides
        // (which cost 100 us per 5M splats)
        const float inv = 1.0f / (float)tile;
        float a = floorf(min_x * inv), b = fminf(floorf(max_x * inv), (float)(ntx - 1u));
        float c = floorf(min_y * inv), d = fminf(floorf(max_y * inv), (float)(nty - 1u));
        if (a > b || c > d) return false;
        tx0 = (uint32_t)a; tx1 = (uint32_t)b; ty0 = (uint32_t)c; ty1 = (uint32_t)d;
    } else {
        double ts = (double)tile;
        double a = floor((double)min_x / ts), b = fmin(floor((double)max_x / ts), (double)ntx - 1.0);
        double c = floor((double)min_y / ts), d = fmin(floor((double)max_y / ts), (double)nty - 1.0);
        if (a > b || c > d) return false;
        tx0 = (uint32_t)a; tx1 = (uint32_t)b; ty0 = (uint32_t)c; ty1 = (uint32_t)d;
    }
    // multi-GPU band: keep only tile rows [row0, row1)
    if (ty0 < row0) ty0 = row0;
    if (row1 == 0) return false;
    if (ty1 > row1 - 1) ty1 = row1 - 1;
    return ty0 <= ty1;
}


// 8-bit packed tile range (frame path, ntx and nty <= 256): tx0 | tx1<<8 | ty0<<16 | ty1<<24.
// An empty range is encoded with tx0 = 1 > tx1 = 0.
__device__ __forceinline__ uint32_t pack_range32(bool ok, uint32_t tx0, uint32_t tx1, uint32_t ty0, uint32_t ty1) {
    return ok ? (tx0 | (tx1 << 8) | (ty0 << 16) | (ty1 << 24)) : 1u;
}

// First sort pass's histogram (tile_first.hip): one count per tile of the rectangle, keyed by the low
// tile-id digit, into a wave-private 256-counter LDS histogram.  Returns the number of tiles.
__device__ __forceinline__ uint32_t hist_add_rect(uint32_t *wave_hist, uint32_t tx0, uint32_t tx1, uint32_t ty0, uint32_t ty1,
                                                  uint32_t ntx, uint32_t mask) {
    for (uint32_t ty = ty0; ty <= ty1; ++ty)
        for (uint32_t tx = tx0; tx <= tx1; ++tx) atomicAdd(&wave_hist[(ty * ntx + tx) & mask], 1u);
    return (tx1 - tx0 + 1) * (ty1 - ty0 + 1);
}
