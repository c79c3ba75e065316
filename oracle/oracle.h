/*
 * oracle.h — CPU restatement of the ath92/splat-renderer tile-raster hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (libsplat_hip.so) never
 * links, loads or falls back to anything in this directory.
 *
 * PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures for this path
 *  and none of it is executable here (every class needs a WebGPU
 * GPUDevice).  The oracle is pinned only by (i) the single known answer the reference's
 * docs state (exclusive scan [1,2,3,4,5] -> [0,1,3,6,10], GPU_PIPELINE_PLAN.md:632-635),
 * (ii) the key-mapping definition (src/shaders/extract-depth-keys.wgsl:55-59) and
 * (iii) an independent NumPy restatement (oracle/np_oracle.py) that must agree with this
 * C one bit-for-bit on integers and on the projector's floats.
 *
 * All citations are file:line under /root/reference.
 */
#ifndef SPLAT_ORACLE_H
#define SPLAT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ProjectedSplat record: 8 x 4 bytes (src/SplatProjector.ts:47-54).
 * [0..1] screenBoundsMin.xy  [2..3] screenBoundsMax.xy  [4] depth  [5] screenRadius
 * [6] originalIndex (u32 bit pattern)  [7] padding = 0.0f */
#define ORC_PROJ_FLOATS 8

/* composite modes */
#define ORC_MODE_FRONT_TO_BACK 0 /* SURVEY.md §8a parity contract item 3 (north_star) */
#define ORC_MODE_REFERENCE_LITERAL 1 /* src/ComputeShaderRenderer.ts:175-190 exactly as written */

/* Camera.updateMatrices restated (src/Camera.ts:85-128) with gl-matrix 3.4.4 semantics:
 * f64 arithmetic, every store rounds to f32 (Float32Array). vp = column-major P*V. */
void orc_camera(const float target[3], double distance, double azimuth, double elevation,
                double fov_deg, double aspect, double near_, double far_,
                float vp_out[16], float eye_out[3]);

/* Frame uniforms (src/main.ts:126-144 + src/SplatProjector.ts:35-41):
 * u[0..15]=VP column-major, u[16..18]=eye, u[19]=time, u[20]=screenW, u[21]=screenH. */

/* SplatProjector main (src/SplatProjector.ts:64-132).  pos_radius points at the first
 * vec4(pos.xyz, radius); consecutive splats are `stride_floats` apart (8 for the reference's
 * interleaved buffer, 4 for a split plane). */
void orc_project(const float uniforms[22], const float *pos_radius, size_t stride_floats,
                 uint32_t n, float *projected);
/* multi-GPU exchange format: float4 {screen centre x, y, screen radius, depth} per splat, and its
 * expansion to the 32-byte record (originalIndex = index_base + i) */
void orc_project_compact(const float uniforms[22], const float *pos_radius, size_t stride_floats, uint32_t n,
                         float *records16);
void orc_expand_compact(const float *records16, uint32_t n, uint32_t index_base, float *projected);

/* extract-depth-keys main (src/shaders/extract-depth-keys.wgsl:37-63). */
void orc_extract_keys(const float *projected, uint32_t n, uint32_t n_padded,
                      uint32_t *keys, uint32_t *payload);

/* RadixSorter contract (src/RadixSorter.ts:39-100,263-271): stable ascending sort of
 * (key,payload) pairs; in place. */
void orc_sort_pairs(uint32_t *keys, uint32_t *payload, uint32_t n);

/* PrefixSumScanner contract (src/PrefixSumScanner.ts:150-155, prefix-sum.wgsl:28-96):
 * exclusive scan, out[0]=0. Returns the total. */
uint64_t orc_scan_exclusive(const uint32_t *in, uint32_t *out, uint32_t n);

/* TileBinner.binSorted (src/TileBinner.ts:426-495): counts, offsets (exclusive scan) and the
 * flat index list in sorted order.  Entries of `sorted` that are >= n_splats (the
 * 0xFFFFFFFF padding) are skipped, as the JS loop's NaN arithmetic does.
 * indices may be NULL (count-only).  Returns total = sum(counts); if indices != NULL and
 * total > cap nothing is written to indices and the total is still returned. */
uint64_t orc_bin_sorted(const float *projected, uint32_t n_splats, const uint32_t *sorted,
                        uint32_t n_sorted, uint32_t width, uint32_t height, uint32_t tile,
                        uint32_t *counts, uint32_t *offsets, uint32_t *indices, uint64_t cap);

/* count-tile-hits.wgsl:53-56 tile range (no cull, clamps into row/col 0 — SURVEY I5).
 * out[4] = minTx, minTy, maxTx, maxTy. Used to cross-check binSorted on on-screen splats. */
void orc_gpu_tile_range(const float *rec, uint32_t tile, uint32_t ntx, uint32_t nty,
                        uint32_t out[4]);

/* ComputeShaderRenderer main + evaluateSplat (src/ComputeShaderRenderer.ts:97-198).
 * color_opacity / normals: vec4 per splat, `*_stride` floats apart.
 * Rows [row0,row1) are rendered (row1 <= height); out_f32 is width*height*4 floats (only the
 * rows rendered are touched), out_u8 is rgba8unorm of the same, either may be NULL.
 * early_out = 0 disables the alpha >= 0.99 break (for tight tolerance tests).
 * Returns the number of list entries consumed summed over pixels (informational). */
uint64_t orc_composite(int mode, int early_out, const float *color_opacity, size_t color_stride,
                       const float *normals, size_t normal_stride, const float *projected,
                       const uint32_t *indices, const uint32_t *counts, const uint32_t *offsets,
                       uint32_t tile, uint32_t ntx, uint32_t width, uint32_t height,
                       uint32_t row0, uint32_t row1, float *out_f32, uint8_t *out_u8);
/* The same with two optional per-pixel outputs (width*height each): stop = list entries the pixel visited (the
 * index of its break + 1, or its tile's whole list) — the per-tile maximum is SURVEY §8d's P_used of that tile;
 * near = 1 where the pixel's alpha came within 2e-5 of the 0.99 threshold at some entry (such a pixel may stop
 * one entry earlier or later in another correct float evaluation). */
uint64_t orc_composite_ex(int mode, int early_out, const float *color_opacity, size_t color_stride,
                          const float *normals, size_t normal_stride, const float *projected,
                          const uint32_t *indices, const uint32_t *counts, const uint32_t *offsets,
                          uint32_t tile, uint32_t ntx, uint32_t width, uint32_t height,
                          uint32_t row0, uint32_t row1, float *out_f32, uint8_t *out_u8, uint32_t *stop, uint8_t *near);

/* SequentialRenderer (src/SequentialRenderer.ts:68-142,186-209,246-307), "model B":
 * one oriented quad per splat in `order[0..n_order)`, src-alpha blending over the clear colour.
 * The caller passes order = reverse(sorted) for back-to-front. Quads with any corner at
 * w <= 0 are skipped (no clipper; never happens in the synthetic scene). */
void orc_sequential(const float uniforms[22], const float *pos_radius, size_t pr_stride,
                    const float *color_opacity, size_t color_stride, const float *normals,
                    size_t normal_stride, const uint32_t *order, uint32_t n_order,
                    uint32_t width, uint32_t height, float *out_f32, uint8_t *out_u8);

/* Oriented-disc footprint: SequentialRenderer's quad + fragment (src/SequentialRenderer.ts:
 * 68-71, 91-142) evaluated per pixel through the inverse of the quad's plane-to-screen homography.
 * Disc record: 8 floats {centre.x, centre.y, B00, B01, B10, B11, q0, q1}: (u,v) = B*d / (1 - q.d), d = pixel -
 * centre; all zeros = culled.  orc_project_disc writes, per splat, the disc record and a ProjectedSplat whose
 * bounds are the disc's exact screen extent (orc_disc_bounds of the record), depth as orc_project, screenRadius =
 * half the larger extent. */
#define ORC_DISC_FLOATS 8
int orc_disc_bounds(const float *disc_record, float bounds_out[4]);
void orc_project_disc(const float uniforms[22], const float *pos_radius, size_t pr_stride, const float *normals,
                      size_t normal_stride, uint32_t n, float *projected, float *discs);
/* normals == NULL: lit_or_color already holds lit colours.  rim: optional width*height bytes, 1 where a
 * pixel lies within 1e-3 (in d2) of some disc's rim (see oracle.c). */
uint64_t orc_composite_disc(int early_out, const float *lit_or_color, size_t color_stride, const float *normals,
                            size_t normal_stride, const float *discs, const uint32_t *indices, const uint32_t *counts,
                            const uint32_t *offsets, uint32_t tile, uint32_t ntx, uint32_t width, uint32_t height,
                            uint32_t row0, uint32_t row1, float *out_f32, uint8_t *out_u8, uint8_t *rim);

/* SplatPropertyManager update kernel (src/SplatPropertyManager.ts:82-107). positions and
 * curvature are vec4 arrays; props is the interleaved 8-float record. */
void orc_update_props(const float *positions, const float *curvature, uint32_t n, float *props);

/* ---- SDF splat generation --------------------------------------------------------------------
 * src/sdf/CodeGenerator.ts:97-225 (sdgSphere/Box/Torus/Capsule, opUnion/Intersection/Subtraction/SmoothUnion),
 * :276-353 (sceneSDF: post-order walk), src/GradientSampler.ts (gradients[i] = sceneSDF(p_i)),
 * src/shaders/update-positions.wgsl:22-50, src/CurvatureSampler.ts:84-141.
 * The scene graph is a postfix program of {op, a[7]} instructions (op codes as in include/splat.h; restated here so that
 * the checker does not include the product's header): 0 sphere {c, r}, 1 box {c, half size}, 2 torus {c, major, minor},
 * 3 capsule {c, height, radius}, 16 union, 17 intersection, 18 subtraction, 19 smooth union {k}. */
typedef struct { uint32_t op; float a[7]; } orc_sdf_instr;
void orc_sdf_scene(const orc_sdf_instr *prog, uint32_t n_instr, const float p[3], float out[4]);
void orc_sdf_gradients(const orc_sdf_instr *prog, uint32_t n_instr, const float *positions, uint32_t n, float *gradients);
void orc_sdf_update_positions(const float *positions, const float *gradients, uint32_t n, float *next_positions);
void orc_sdf_scale_factors(const orc_sdf_instr *prog, uint32_t n_instr, const float *positions, uint32_t n, float *scale_factors);
void orc_sdf_curvature(const float *gradients, const float *scale_factors, uint32_t n, float *curvature);
void orc_sdf_seed_positions(const float *mn, const float *mx, uint32_t n, uint64_t seed, float *positions);

/* Whole frame for timing (project -> keys -> sort -> binSorted -> composite model A). Scratch is
 * allocated inside; returns 0 or -1 on allocation failure. P is written to *total_pairs. */
int orc_frame(int mode, int early_out, const float uniforms[22], const float *props,
              const float *normals, uint32_t n, uint32_t width, uint32_t height, uint32_t tile,
              int threads, float *out_f32, uint8_t *out_u8, uint64_t *total_pairs,
              double stage_ms[5]);

/* rgba8unorm conversion used by both composites (WebGPU texel store: clamp, *255, round half
 * away is not specified; round-to-nearest-even via lrintf is used and documented). */
uint8_t orc_unorm8(float v);

#ifdef __cplusplus
}
#endif
#endif
