// shade.h — the reference's per-splat shading and the frame's 32-byte composite ("lit") record, shared by
// the projector (project.hip writes the record) and the composite (composite.hip reads it).
//
// Reference: /root/reference/src/ComputeShaderRenderer.ts:143-145 (shading), :117-147 (evaluateSplat: what the
// composite needs of a splat — its bounds, screen radius and colour), src/SplatProjector.ts:119-121 (bounds =
// centre -/+ 1.5 * radius).
//
// Lit record (SPLAT_RECORDS_LIT32), two float4 per splat:
//     {centre.x, centre.y, screen radius, depth}   {lit red, lit green, lit blue, opacity}
// The ProjectedSplat's bounds are a pure function of the first half (lit_bounds below, the projector's own
// operation order), so a frame that writes this record instead of the ProjectedSplat gives the composite
// everything evaluateSplat reads in ONE 32-byte gather per staged list entry instead of three lines
// (ProjectedSplat, colour, normal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The reference's shading of one splat (ComputeShaderRenderer.ts:143-145): colour scaled by
// kd = 0.85 + 0.15 * max(dot(normal, normalize(1,1,1)), 0).  Contraction is switched off for this function:
// one IEEE operation per operator, so the bits are the oracle's, and the same whether it runs per staged
// entry in the composite, once per splat in k_lit_colors or in the projector (composite.hip is compiled with
// contraction on, and call sites would otherwise be free to fuse differently).
__device__ __forceinline__ float4 lit_color(float4 c, float4 nrm) {
#pragma clang fp contract(off) // (HIP's __fmul_rn / __fadd_rn are plain operators and would be contracted like any other)
    const float k = 0.577350269189625764f; // normalize(vec3(1,1,1)) :143
    const float ndl = (nrm.x * k + nrm.y * k) + nrm.z * k;
    const float kd = 0.85f + 0.15f * fmaxf(ndl, 0.0f); // :144-145
    return make_float4(c.x * kd, c.y * kd, c.z * kd, c.w);
}

// Bounds of a splat from {centre x, y, radius}: exactly as the projector forms them (SplatProjector.ts:119-121),
// one rounding per operation.
__device__ __forceinline__ float4 lit_bounds(float4 c) {
#pragma clang fp contract(off)
    const float padded = c.z * 1.5f;
    return make_float4(c.x - padded, c.y - padded, c.x + padded, c.y + padded);
}

// Where the frame's projector finds colours / normals and leaves lit records (records == nullptr: it does not).
struct LitIO {
    const float4 *color;   // vec4(rgb, opacity), color_stride float4s apart; already lit when prelit
    const float4 *normals; // vec4(normal, scale), normal_stride float4s apart (unused when prelit)
    uint32_t color_stride, normal_stride, prelit;
    float4 *records;       // out: 2 x float4 per splat
};
