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
    double min_x = fmax((double)bounds.x, 0.0), min_y = fmax((double)bounds.y, 0.0);
    double max_x = fmin((double)bounds.z, (double)width), max_y = fmin((double)bounds.w, (double)height);
    if (min_x >= max_x || min_y >= max_y) return false;
    double ts = (double)tile;
    double a = floor(min_x / ts), b = fmin(floor(max_x / ts), (double)ntx - 1.0);
    double c = floor(min_y / ts), d = fmin(floor(max_y / ts), (double)nty - 1.0);
    if (a > b || c > d) return false;
    tx0 = (uint32_t)a; tx1 = (uint32_t)b; ty0 = (uint32_t)c; ty1 = (uint32_t)d;
    // multi-GPU band: keep only tile rows [row0, row1)
    if (ty0 < row0) ty0 = row0;
    if (row1 == 0) return false;
    if (ty1 > row1 - 1) ty1 = row1 - 1;
    return ty0 <= ty1;
}

