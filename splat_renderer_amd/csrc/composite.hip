// composite.hip — the per-pixel alpha composite (ComputeShaderRenderer / TileRenderer front end).
//
// Reference: /root/reference/src/ComputeShaderRenderer.ts:97-198 (evaluateSplat + main, K11),
// dispatched 8x8 with every pixel re-gathering idx/props/normal/projected per list entry
// (:103-115).  CDNA4 design instead:
//   - one 256-thread workgroup per 16x16 tile; wave w owns the 8x8 pixel quadrant w (one pixel per
//     lane, so the 64-wide ballot/all of a wave is exactly "this quadrant")
//   - the tile's list is consumed in batches of 256 entries: thread t gathers entry t ONCE
//     (index, 32 B ProjectedSplat, colour vec4, normal vec4 — the reference's own layouts) and
//     does every per-entry computation there: centre, exp2 scale, lit colour, and for each of the
//     four quadrants the exact 64-bit mask of pixels inside the entry's box (the reference's four
//     float comparisons per pixel per entry become one mask per entry per quadrant)
//   - each wave reads the 64 masks of a chunk with one LDS read, ballots the non-empty ones and
//     walks only those (s_ff1); an entry whose covered pixels have all saturated is skipped on
//     the scalar unit; the survivors' 32 B of parameters come back as LDS broadcasts
//   - the set of pixels still accumulating is a wave-uniform 64-bit mask: a pixel leaves it at
//     alpha >= 0.99 exactly as :187-190, a wave with none left stops, the workgroup leaves when
//     all four waves have
//
// Roofline: HBM in the SURVEY §8d model — 68 B per consumed list entry (4 idx + 32 projected +
// 16 colour + 16 normal) + 4 B per pixel written.  The inner loop is VALU/LDS work, so the
// achieved fraction is reported honestly against that model.
//
// Compiled with -ffp-contract=fast; compared with the oracle within a stated tolerance.
#include "common.h"
#include "disc.h"
#include "shade.h"

#include <hip/hip_ext.h>

#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2))); // maps onto the packed FP32 instructions (v_pk_*_f32)

constexpr int CT = 16;        // tile edge (pixels)
constexpr int CBATCH = 256;   // list entries staged per round

struct CompositeParams {
    const float4 *color;  uint32_t color_stride;   // vec4(rgb, opacity)
    const float4 *normals; uint32_t normal_stride; // vec4(normal, scaleFactor)
    const float4 *projected;                       // 2 x float4 per splat (ProjectedSplat), or 1 x float4 (compact exchange record)
    uint32_t compact;
    uint32_t lit32;                                // projected holds lit composite records (shade.h): colour and normals are not read
    uint32_t disc;                                 // projected holds disc records (disc.h): the oriented-disc footprint
    uint32_t disc_stride;                          // float4s between disc records: 2 (projector's) or 3 (48-byte exchange records, lit disc records)
    uint32_t disc_lit;                             // the third float4 of a disc record is the splat's lit colour: colour and normals are not read
    uint32_t prelit;                               // color holds lit colours (k_lit_colors): normals are not read
    const uint32_t *indices, *counts, *offsets;
    uint32_t width, height, ntx, tile_row0;
    uint32_t *out_rgba8;
    float4 *out_rgba32f;
    unsigned long long *consumed; // per tile {entries staged, entries consumed}, accumulated (or NULL)
    // the frame's report (tile-first frames; NULL otherwise): this launch is the frame's last kernel, so its first
    // workgroup tells the host {pair total, flags incl. the per-tile sort's order check, sequence number}
    const uint32_t *frame_total;
    uint32_t *report;
    uint32_t report_seq;
    const uint32_t *tile_order; // k_composite_px: workgroup b works on tile tile_order[b] of the band (NULL: b)
    uint32_t *tile_cost;        // k_composite_px: chunks each tile's consumer walked (NULL: not kept)
    const uint32_t *order_src;  // k_composite_px, workgroup 0: the costs the PREVIOUS launch over this band left (NULL: none) ...
    uint32_t *order_dst;        // ... sorted into the order the NEXT launch takes its tiles in
    const uint32_t *cost_prev;  // k_composite_px: the same costs, read by every tile: how many chunks to build and gather ahead of need (NULL: all)
#ifdef PX_PROFILE
    uint32_t debug_cap;         // (measuring build only, SPLAT_PX_CAP: every list cut after this many entries — a WRONG image: what do the long tiles cost?)
#endif
};

__device__ __forceinline__ uint32_t unorm8(float v) {
    v = fminf(fmaxf(v, 0.0f), 1.0f); // fmaxf(NaN,0) = 0
    return (uint32_t)(v * 255.0f + 0.5f);
}

// 64-bit lane mask of one 8x8 quadrant from its 8-bit column mask xb and row mask yb: lane
// ly*8+lx is set iff bit lx of xb and bit ly of yb are.  (y & 15) * 0x00204081 drops bit i of y at
// bit 8i (the four shifted copies do not overlap), & 0x01010101 keeps those, * xb copies xb into
// every selected byte.
__device__ __forceinline__ uint2 quadrant_mask(uint32_t xb, uint32_t yb) {
    const uint32_t lo = (((yb & 15u) * 0x00204081u) & 0x01010101u) * xb;
    const uint32_t hi = (((yb >> 4) * 0x00204081u) & 0x01010101u) * xb;
    return make_uint2(lo, hi);
}

// Pixel columns j in [0,16) of a tile whose centres c0 + j lie inside [lo, hi]
// (ComputeShaderRenderer.ts:118-121 keeps a pixel iff !(p < min || p > max)).  c0 = tile origin +
// 0.5 >= 0.5.  For a result in [0,16) the subtraction is exact (lo >= c0 > 0 and the difference is a
// multiple of ulp(lo) no larger than lo), outside that range only its sign / being >= 16 matters
// and rounding is monotone (x - y == 0 only when x == y), so the mask is exactly the set the
// reference's comparisons select.
__device__ __forceinline__ uint32_t span_mask16(float lo, float hi, float c0) {
    const float a = fmaxf(ceilf(lo - c0), 0.0f), b = fminf(floorf(hi - c0), 15.0f);
    if (!(a <= b)) return 0u; // also NaN
    const uint32_t ia = (uint32_t)a, ib = (uint32_t)b;
    return ((2u << ib) - 1u) & ~((1u << ia) - 1u);
}

// Pins a wave-uniform 64-bit value into scalar registers (the compiler's divergence analysis gives
// up on loop-carried masks and would otherwise keep them, and every test on them, in VGPRs).
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
           (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// The lit colours of all splats as a plane: when the composite is given this plane (cfg->prelit) it
// gathers two lines per staged entry (record, lit colour) instead of three (record, colour, normal) —
// the gathers, not the arithmetic, are what a staged entry costs (108 -> 93 us at C2 for one line less).
__global__ __launch_bounds__(256) void k_lit_colors(const float4 *__restrict__ color, uint32_t color_stride,
                                                    const float4 *__restrict__ normals, uint32_t normal_stride, uint32_t n,
                                                    float4 *__restrict__ lit) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) lit[i] = lit_color(color[(size_t)i * color_stride], normals[(size_t)i * normal_stride]);
}

// bounds and screen radius of splat idx.  Compact exchange records (multi-GPU frame) carry {centre x,
// y, radius, depth}: the bounds are rebuilt exactly as the projector forms them (SplatProjector.ts:
// 119-121) — with contraction switched off for this function (the file is compiled with it on).
__device__ __forceinline__ void fetch_record(const CompositeParams &p, uint32_t idx, float4 &bounds, float &radius) {
    if (p.compact) {
        const float4 c = p.projected[idx];
        bounds = lit_bounds(c); // the bounds must be the projector's: one rounding per operation
        radius = c.z;
    } else {
        bounds = p.projected[(size_t)idx * 2];
        radius = reinterpret_cast<const float *>(p.projected)[(size_t)idx * 8 + 5];
    }
}

// The stop test of the nearest-first loop is (1 - T) >= 0.99 on the transmittance T (the reference's alpha >= 0.99,
// ComputeShaderRenderer.ts:187-190).  A correctly rounded 1 - T is monotone in T, so the test is EXACTLY T <= the
// largest binary32 T that passes it — 0x1.47ae4p-7 (found by stepping ulps; NaN fails both forms) — and the
// subtraction leaves the per-pixel loop.
constexpr float T_STOP = 0x1.47ae4p-7f;
static_assert((1.0f - T_STOP) >= 0.99f && !((1.0f - 0x1.47ae42p-7f) >= 0.99f), "T_STOP is the last transmittance that stops a pixel");

// exp(-0.5 * d2 / (0.4 * 0.4)) = exp2(d2 * this)   (SequentialRenderer.ts:132-133)
constexpr float DISC_EXP2_SCALE = -4.508422002777011f;

// DISC: the footprint is SequentialRenderer's oriented disc (disc.h) — per entry the 32-byte disc record and
// the lit colour are staged, a pixel is inside when u^2 + v^2 <= 1 with (u,v) = B*d / (1 - q.d); the
// coverage masks come from the disc's exact bounds, as the binner's tile ranges do.
// LIT32: `projected` holds the frame's lit composite records (shade.h) — ONE 32-byte gather per staged entry gives
// centre, radius and lit colour; colour and normal arrays are not touched.
template <int MODE, bool EARLY_OUT, bool DISC, bool LIT32>
__device__ __forceinline__ void fetch_entry(const CompositeParams &p, uint32_t idx, float4 &f_b, float4 &f_b2, float4 &f_c, float4 &f_n,
                                            float &f_r) {
    if constexpr (DISC) {
        f_b = p.projected[(size_t)idx * p.disc_stride];
        f_b2 = p.projected[(size_t)idx * p.disc_stride + 1];
        if (p.disc_lit) {
            f_c = p.projected[(size_t)idx * p.disc_stride + 2];
            return;
        }
    } else if constexpr (LIT32) {
        const float4 c = p.projected[(size_t)idx * 2];
        f_c = p.projected[(size_t)idx * 2 + 1];
        f_b = lit_bounds(c);
        f_r = c.z;
        return;
    } else {
        fetch_record(p, idx, f_b, f_r);
    }
    f_c = p.color[(size_t)idx * p.color_stride];
    if (!p.prelit) f_n = p.normals[(size_t)idx * p.normal_stride];
}

template <int MODE, bool EARLY_OUT, bool DISC, bool LIT32>
__global__ __launch_bounds__(256) void k_composite(CompositeParams p) {
    // per entry one 32-byte record {centre.x, centre.y, exp2 scale, lit blue | lit red, lit green, -, -}: both
    // halves are read off ONE address register (ds_read_b128 + ds_read_b64 offset:16), and forming an LDS
    // address from the scalar entry index costs a VALU move per register
    // (DISC: {centre.x, centre.y, -q0, -q1 | B00, B10, B01, B11 | lit red, green, blue, -}: B by columns, so that
    // (u,v) numerators are two packed operations on register pairs as they arrive)
    constexpr int PAR = DISC ? 3 : 2;
    __shared__ float4 s_par[CBATCH][PAR];
    __shared__ uint2 s_mask[4][CBATCH];  // per quadrant: which of its 64 pixels the entry's box covers
    __shared__ uint32_t s_wave_done[4];
    __shared__ uint32_t s_wave_consumed[4];

    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (p.report && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) tile_report(p.frame_total, p.report, p.report_seq);
    const uint32_t tx = blockIdx.x, ty = blockIdx.y + p.tile_row0;
    const uint32_t tile_idx = ty * p.ntx + tx; // ComputeShaderRenderer.ts:161-163
    const uint32_t count = p.counts[tile_idx], off = p.offsets[tile_idx];

    const uint32_t px = tx * CT + (w & 1) * 8 + (lane & 7), py = ty * CT + (w >> 1) * 8 + (lane >> 3);
    const bool pixel_ok = px < p.width && py < p.height;
    const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f; // :169
    const float tile_x0 = (float)(tx * CT), tile_y0 = (float)(ty * CT);
    const float tile_cx = tile_x0 + 0.5f, tile_cy = tile_y0 + 0.5f;
    const v2f p_local = {(float)((w & 1) * 8 + (lane & 7)) + 0.5f, (float)((w >> 1) * 8 + (lane >> 3)) + 0.5f}; // pixel centre in the tile

    float cr = 0.0f, cg = 0.0f, cb = 0.0f;
    float acc = (MODE == SPLAT_COMPOSITE_REFERENCE_LITERAL) ? 0.0f : 1.0f; // alpha (literal) or transmittance T
    // wave-uniform mask of the pixels still accumulating; a pixel leaves it when its alpha reaches
    // 0.99 (:187-190) and pixels outside the image never enter it
    unsigned long long live = uniform64(__ballot(pixel_ok));
    if (tid < 4) s_wave_done[tid] = 0;

    uint32_t staged = 0;
    // list entries this wave needed: the position after the entry at which its last pixel saturated, or the whole
    // list if some pixel never did (SURVEY §8d's P_used per tile = the largest of the four; = count with early-out off)
    uint32_t needed = 0;

    // ---- list-entry fetch, split from its use (issue early / write LDS late).  Almost every tile
    // saturates inside its first batch, but tiles on a silhouette keep some pixel open and walk
    // their whole list (thousands of entries): from their second batch on, the NEXT batch's gathers
    // are issued before the current batch is consumed, so the ~3 us dependent-load chain (index, then
    // record / colour / normal) overlaps the arithmetic instead of preceding it.
    uint32_t f_idx = 0xffffffffu;              // splat index of the entry this thread stages
    float4 f_b = make_float4(0, 0, 0, 0), f_c = f_b, f_n = f_b, f_b2 = f_b;
    float f_r = 0.0f;
    bool f_ready = false;                      // f_* already hold this thread's entry of the batch about to be staged
    uint32_t n_idx = 0xffffffffu;              // index of this thread's entry one batch further on (the gathers depend on it)
    bool n_idx_valid = false;

    for (uint32_t base = 0; base < count; base += CBATCH) {
        __syncthreads(); // previous batch fully consumed (and s_wave_done visible)
        if (EARLY_OUT) {
            // (readfirstlane: an LDS value is "divergent" to the compiler, which would then treat
            // this whole loop, and every mask carried through it, as per-lane)
            if (__builtin_amdgcn_readfirstlane((int)(s_wave_done[0] & s_wave_done[1] & s_wave_done[2] & s_wave_done[3]))) break;
        }
        // ---- stage: one entry per thread, everything per-entry is computed here, once per tile ----
        {
            const uint32_t e = base + tid;
            float4 geo = make_float4(0.0f, 0.0f, 0.0f, 0.0f), geo2 = geo;
            float2 col = make_float2(0.0f, 0.0f);
            float col_b = 0.0f;
            uint32_t xm = 0, ym = 0;
            if (!f_ready) { // the first three batches of a tile: fetch now
                f_idx = (tid < CBATCH && e < count) ? p.indices[off + e] : 0xffffffffu;
                if (f_idx != 0xffffffffu) fetch_entry<MODE, EARLY_OUT, DISC, LIT32>(p, f_idx, f_b, f_b2, f_c, f_n, f_r);
            }
            if (DISC && f_idx != 0xffffffffu) {
                const DiscRecord rec = {f_b, f_b2};
                float4 bnd;
                if (disc_bounds(rec, bnd)) { // (a culled splat's record is all zeros and is in no list anyway)
                    const float4 c = (p.prelit || p.disc_lit) ? f_c : lit_color(f_c, f_n);
                    col = make_float2(c.x, c.y);
                    col_b = c.z;
                    geo = make_float4(f_b.x, f_b.y, -f_b2.z, -f_b2.w);
                    geo2 = make_float4(f_b.z, f_b2.x, f_b.w, f_b2.y);
                    xm = span_mask16(bnd.x, bnd.z, tile_cx);
                    ym = span_mask16(bnd.y, bnd.w, tile_cy);
                }
            }
            if (!DISC && f_idx != 0xffffffffu) {
                const float4 b = f_b;
                const float r = f_r;
                if (!(r < 0.5f)) { // :127-129 "too small"
                    const float4 c = (LIT32 || p.prelit) ? f_c : lit_color(f_c, f_n);
                    col = make_float2(c.x, c.y);
                    // gaussian = exp(-0.5 nd^2 / 0.25), nd = dist / r  ->  exp2(-((dx k)^2 + (dy k)^2)), k = sqrt(2 log2 e) / r,
                    // evaluated per pixel as (p k - c k)^2 in TILE-LOCAL coordinates (|p|, |c| of the order of the tile, so the
                    // difference loses nothing that matters: < 1e-5 relative in the Gaussian for the smallest splat the
                    // reference draws): one packed multiply-add, one packed square and an add per (entry, quadrant) instead of
                    // two subtractions, two multiplies and a scale — the kernel is bound by vector-ALU issue slots
                    const float k = 1.6986436005760381f / r;                                     // sqrt(2.885390081777927)
                    const float lx = (b.x + b.z) * 0.5f - tile_x0, ly = (b.y + b.w) * 0.5f - tile_y0; // :124, then exact
                    geo = make_float4(lx * k, ly * k, k, c.z);
                    xm = span_mask16(b.x, b.z, tile_cx);
                    ym = span_mask16(b.y, b.w, tile_cy);
                }
            }
            if (tid < CBATCH) {
                s_par[tid][0] = geo;
                if constexpr (DISC) {
                    s_par[tid][1] = geo2;
                    s_par[tid][2] = make_float4(col.x, col.y, col_b, 0.0f);
                } else {
                    s_par[tid][1] = make_float4(col.x, col.y, 0.0f, 0.0f);
                }
                s_mask[0][tid] = quadrant_mask(xm & 0xffu, ym & 0xffu);
                s_mask[1][tid] = quadrant_mask(xm >> 8, ym & 0xffu);
                s_mask[2][tid] = quadrant_mask(xm & 0xffu, ym >> 8);
                s_mask[3][tid] = quadrant_mask(xm >> 8, ym >> 8);
            }
            // issue the fetches for later batches; nothing below waits for them until the next stage
            f_ready = false;
            if (base >= CBATCH) { // a tile that needed a second batch usually needs more
                if (n_idx_valid) { // index of batch k+1 arrived a batch ago: its gathers go out now
                    f_idx = n_idx;
                    if (f_idx != 0xffffffffu) fetch_entry<MODE, EARLY_OUT, DISC, LIT32>(p, f_idx, f_b, f_b2, f_c, f_n, f_r);
                    f_ready = true;
                }
                const uint32_t e2 = e + 2 * CBATCH; // batch k+2
                n_idx = (tid < CBATCH && e2 < count) ? p.indices[off + e2] : 0xffffffffu;
                n_idx_valid = true;
            }
        }
        staged = (count - base < CBATCH) ? count : base + CBATCH;
        __syncthreads();
        if (uniform64(live) == 0) { // nothing left to accumulate (or a quadrant wholly outside the image)
            if (EARLY_OUT && lane == 0) s_wave_done[w] = 1;
            continue;
        }
        // ---- consume: 4 chunks of 64 entries; lane j looks at entry c0+j's mask for this quadrant ---
        const uint32_t batch_n = (count - base < CBATCH) ? (count - base) : CBATCH;
        needed = base + batch_n; // unless the wave saturates inside this batch (below)
        for (uint32_t c0 = 0; c0 < batch_n && uniform64(live) != 0; c0 += 64) {
            const uint2 mm = s_mask[w][c0 + lane];
            // entries of this chunk that cover at least one pixel still accumulating
            const unsigned long long lv0 = uniform64(live);
            unsigned long long hits = uniform64(__ballot(((mm.x & (uint32_t)lv0) | (mm.y & (uint32_t)(lv0 >> 32))) != 0u));
            // two entries per trip: both parameter reads are in flight together, both Gaussians are
            // independent work, and the loop/branch overhead is paid once per pair; the second
            // entry's coverage is re-masked with the pixels the first one has just saturated, so the
            // per-pixel stop is exactly sequential
            bool saturated = false; // (one loop exit: the accumulators then stay in the registers they live in)
            while (hits && !saturated) {
                const uint32_t j0 = (uint32_t)__builtin_ctzll(hits);
                hits &= hits - 1;
                const bool two = hits != 0;
                const uint32_t j1 = two ? (uint32_t)__builtin_ctzll(hits) : j0;
                hits &= hits - 1; // (0 & anything stays 0)
                // (readlane returns int: go through uint32_t or the low word sign-extends into the high one)
                const unsigned long long cover0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mm.y, (int)j0) << 32) |
                                                  (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mm.x, (int)j0);
                unsigned long long cover1 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mm.y, (int)j1) << 32) |
                                            (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mm.x, (int)j1);
                if (!two) cover1 = 0;
                const float4 G0 = s_par[c0 + j0][0], G1 = s_par[c0 + j1][0]; // wave-uniform addresses: LDS broadcasts
                float2 C0, C1;
                float B0, B1, g0, g1; // blue, Gaussian
                if constexpr (DISC) {
                    const float4 M0 = s_par[c0 + j0][1], M1 = s_par[c0 + j1][1];
                    const float4 L0 = s_par[c0 + j0][2], L1 = s_par[c0 + j1][2];
                    C0 = make_float2(L0.x, L0.y); B0 = L0.z;
                    C1 = make_float2(L1.x, L1.y); B1 = L1.z;
                    const v2f pc = {pxf, pyf};
                    const v2f e0 = pc - (v2f){G0.x, G0.y}, e1 = pc - (v2f){G1.x, G1.y};
                    const float rd0 = __builtin_amdgcn_rcpf(__builtin_fmaf(G0.z, e0.x, __builtin_fmaf(G0.w, e0.y, 1.0f))); // 1 / (1 - q.d)
                    const float rd1 = __builtin_amdgcn_rcpf(__builtin_fmaf(G1.z, e1.x, __builtin_fmaf(G1.w, e1.y, 1.0f)));
                    const v2f uv0 = ((v2f){M0.x, M0.y} * e0.x + (v2f){M0.z, M0.w} * e0.y) * rd0; // B*d / (1 - q.d)
                    const v2f uv1 = ((v2f){M1.x, M1.y} * e1.x + (v2f){M1.z, M1.w} * e1.y) * rd1;
                    const float d0 = uv0.x * uv0.x + uv0.y * uv0.y, d1 = uv1.x * uv1.x + uv1.y * uv1.y; // :126
                    g0 = (d0 <= 1.0f) ? __builtin_amdgcn_exp2f(d0 * DISC_EXP2_SCALE) : 0.0f; // :128-133 (NaN: outside)
                    g1 = (d1 <= 1.0f) ? __builtin_amdgcn_exp2f(d1 * DISC_EXP2_SCALE) : 0.0f;
                } else {
                    C0 = *reinterpret_cast<const float2 *>(&s_par[c0 + j0][1]);
                    C1 = *reinterpret_cast<const float2 *>(&s_par[c0 + j1][1]);
                    B0 = G0.w;
                    B1 = G1.w;
                    const v2f t0 = p_local * (v2f){G0.z, G0.z} - (v2f){G0.x, G0.y}, t1 = p_local * (v2f){G1.z, G1.z} - (v2f){G1.x, G1.y};
                    const v2f q0 = t0 * t0, q1 = t1 * t1;
                    g0 = __builtin_amdgcn_exp2f(-(q0.x + q0.y));
                    g1 = __builtin_amdgcn_exp2f(-(q1.x + q1.y));
                }
                unsigned long long lv = uniform64(live); // pinned at the use: see uniform64()
                g0 = __builtin_amdgcn_inverse_ballot_w64(cover0 & lv) ? g0 : 0.0f;
                if (MODE == SPLAT_COMPOSITE_REFERENCE_LITERAL) { // :183-185 as written
                    const float om = 1.0f - g0;
                    cr = cr * om + C0.x * g0;
                    cg = cg * om + C0.y * g0;
                    cb = cb * om + B0 * g0;
                    acc = acc * om + g0;
                    if (EARLY_OUT) lv &= ~__ballot(acc >= 0.99f); // :187-190
                } else { // SURVEY §8a contract 3: nearest on top
                    const float wgt = acc * g0;
                    cr += C0.x * wgt;
                    cg += C0.y * wgt;
                    cb += B0 * wgt;
                    acc -= wgt; // T * (1 - g), with the product already in hand
                    if (EARLY_OUT) lv &= ~__ballot(acc <= T_STOP);
                }
                lv = uniform64(lv);
                const bool first_saturated = lv == 0; // (scalar; only read on the way out)
                g1 = __builtin_amdgcn_inverse_ballot_w64(cover1 & lv) ? g1 : 0.0f;
                if (MODE == SPLAT_COMPOSITE_REFERENCE_LITERAL) {
                    const float om = 1.0f - g1;
                    cr = cr * om + C1.x * g1;
                    cg = cg * om + C1.y * g1;
                    cb = cb * om + B1 * g1;
                    acc = acc * om + g1;
                    if (EARLY_OUT) lv &= ~__ballot(acc >= 0.99f);
                } else {
                    const float wgt = acc * g1;
                    cr += C1.x * wgt;
                    cg += C1.y * wgt;
                    cb += B1 * wgt;
                    acc -= wgt;
                    if (EARLY_OUT) lv &= ~__ballot(acc <= T_STOP);
                }
                live = lv;
                if (EARLY_OUT && uniform64(live) == 0) {
                    needed = base + c0 + (first_saturated ? j0 : j1) + 1;
                    saturated = true;
                }
            }
        }
        if (EARLY_OUT && uniform64(live) == 0 && lane == 0) s_wave_done[w] = 1;
    }

    // (per tile, no atomics: 8160 workgroups adding to ONE counter cost the kernel 30 us at C1 and 80 us
    // at C3 — the measurement was slowing down what it measured)
    if (p.consumed) { // (uniform branch; timed / diagnostic runs only)
        if (lane == 0) s_wave_consumed[w] = needed;
        __syncthreads();
        if (tid == 0 && staged) {
            const uint32_t used = max(max(s_wave_consumed[0], s_wave_consumed[1]), max(s_wave_consumed[2], s_wave_consumed[3]));
            p.consumed[(size_t)tile_idx * 2] += (unsigned long long)staged;
            p.consumed[(size_t)tile_idx * 2 + 1] += (unsigned long long)used;
        }
    }

    if (pixel_ok) {
        const float rem = (MODE == SPLAT_COMPOSITE_REFERENCE_LITERAL) ? (1.0f - acc) : acc;
        const float fr = cr + 0.05f * rem, fg = cg + 0.05f * rem, fb = cb + 0.1f * rem; // :193-195
        const size_t o = (size_t)py * p.width + px;
        if (p.out_rgba8) p.out_rgba8[o] = unorm8(fr) | (unorm8(fg) << 8) | (unorm8(fb) << 16) | (255u << 24);
        if (p.out_rgba32f) p.out_rgba32f[o] = make_float4(fr, fg, fb, 1.0f);
    }
}

// =====================================================================================================================
// k_composite_px — the lane-efficient composite (round 3; the default for the isotropic footprint, nearest on top).
//
// What k_composite spends its time on is lanes that get g = 0: one entry is evaluated by a whole 8x8-pixel wave although
// its box covers a quarter of it (18.5 live covered pixels per consumed entry at C2, 1.14 wave visits of 64 lanes).
// Here every lane walks ITS OWN entries, and the Gaussian is not evaluated per pixel at all:
//   * one WAVE per 16x16 tile (four tiles per workgroup, no workgroup barrier anywhere), lane = a 2x2-pixel block;
//   * the list is consumed in chunks of 32 entries.  Per chunk the wave stages, in its own 5 KB of LDS,
//       - the entry's Gaussian as two tables: gx[16] over the tile's pixel columns and gy[16] over its rows —
//         exp(-(dx^2+dy^2) k^2) = exp(-(dx k)^2) exp(-(dy k)^2), so 32 exponentials per entry serve all 256 pixels —
//         with the reference's box test (ComputeShaderRenderer.ts:118-121, exact: span_mask16) folded in as zeros,
//       - its lit colour;
//   * per lane a 32-bit queue: bit j = entry j's box touches this lane's 2x2 block (eight ballots transpose the
//     entries' column / row masks into per-column / per-row entry masks; a lane ANDs its column's with its row's);
//   * a trip = every lane with a non-empty queue takes its next entry: two 8-byte table reads (its two columns, its two
//     rows), one 16-byte colour read, then four pixels in packed arithmetic: g = gx*gy, w = T*g, C += c*w, T -= w.
//     Order per pixel is the list's order; pixels of different lanes work on different entries in the same instruction.
//   * the stop rule is per pixel (exactly :187-190): a pixel whose T reaches T_STOP takes its background term at once
//     and continues with T = 0 (it adds exact zeros from then on); a lane whose four pixels have stopped empties its
//     queue; the wave leaves when no pixel is left.
// Trips per consumed entry at C2: 0.30 (a wave instruction of up to 256 pixel-entries each) against 1.14 visits of a
// 64-pixel wave.  Reference semantics: /root/reference/src/ComputeShaderRenderer.ts:150-198.
// =====================================================================================================================
// One list entry as fetched, untouched (k_composite_px keeps a chunk of these in flight: nothing may be computed from
// them at issue).  LIT32: a = {centre.xy, radius, depth}, c = lit colour.  Otherwise a = bounds (or the compact record),
// c = colour, n = normal, r = radius.
struct PxRaw {
    float4 a, c, n;
    float r;
    float4 a2; // DISC: the disc record's second half
};
// DISC (the oriented-disc footprint): a, a2 = the 32-byte disc record (disc.h), c = colour, n = normal
__device__ __forceinline__ void px_fetch_disc(const CompositeParams &p, uint32_t idx, PxRaw &o) {
    o.a = p.projected[(size_t)idx * p.disc_stride];
    o.a2 = p.projected[(size_t)idx * p.disc_stride + 1];
    if (p.disc_lit) { // (lit disc records: one 48-byte gather is all)
        o.c = p.projected[(size_t)idx * p.disc_stride + 2];
        return;
    }
    o.c = p.color[(size_t)idx * p.color_stride];
    if (!p.prelit) o.n = p.normals[(size_t)idx * p.normal_stride];
}
template <bool LIT32>
__device__ __forceinline__ void px_fetch(const CompositeParams &p, uint32_t idx, PxRaw &o) {
    if constexpr (LIT32) {
        o.a = p.projected[(size_t)idx * 2];
        o.c = p.projected[(size_t)idx * 2 + 1];
    } else {
        if (p.compact) {
            o.a = p.projected[idx];
        } else {
            o.a = p.projected[(size_t)idx * 2];
            o.r = reinterpret_cast<const float *>(p.projected)[(size_t)idx * 8 + 5];
        }
        o.c = p.color[(size_t)idx * p.color_stride];
        if (!p.prelit) o.n = p.normals[(size_t)idx * p.normal_stride];
    }
}
// bounds, screen radius and lit colour of a fetched entry (fetch_entry's arithmetic, at the time of use)
template <bool LIT32>
__device__ __forceinline__ void px_unpack(const CompositeParams &p, const PxRaw &o, float4 &bounds, float &radius, float4 &lit) {
    if constexpr (LIT32) {
        bounds = lit_bounds(o.a);
        radius = o.a.z;
        lit = o.c;
    } else {
        if (p.compact) {
            bounds = lit_bounds(o.a); // the bounds must be the projector's: one rounding per operation
            radius = o.a.z;
        } else {
            bounds = o.a;
            radius = o.r;
        }
        lit = p.prelit ? o.c : lit_color(o.c, o.n);
    }
}

typedef uint32_t v2u __attribute__((ext_vector_type(2)));
constexpr int PXC = 32;     // entries per chunk = bits of a lane's queue
constexpr int PX_ROW = 35;  // float2 slots per table row: the idle slot, 32 entries, 2 of padding — rows 3 slots apart (mod 32),
                            // so the eight rows a wave reads for ONE entry fall into eight different bank pairs of a ds_read_b64
constexpr int PX_TROWS = 16;
// One chunk as the builder wave leaves it for the consumer wave.  Slot 0 of every row is the IDLE entry (zeros): what a
// lane with an empty queue "takes" — entry j lives at slot 1 + j, so ffbl's -1 for an empty queue addresses the idle
// slot without a select.
struct PxBuf {
    float2 t[PX_TROWS][PX_ROW]; // t[c][1 + j] = {gx_j(2c), gx_j(2c+1)}: entry j's factor at the tile's pixel columns 2c, 2c+1 (c < 8);
                                // t[8 + r][1 + j] = {gy_j(2r), gy_j(2r+1)} at its pixel rows; only the pairs entry j's box touches are
                                // written (no lane is ever sent to another)
    float4 col[PX_ROW];         // col[1 + j] = {lit r, g, b, 1}: ONE 16-byte read per trip (ds_read_b128, four LDS-array cycles) — as two
                                // 8-byte rows the compiler merged the pair into a ds_read2_b64: eight cycles on 32 banks
    uint4 q4[4];                // as uint2 q[8]: q[c].x: bit j = entry j's box meets pixel columns 2c, 2c+1; q[r].y: ... pixel rows 2r, 2r+1
};
static_assert(offsetof(PxBuf, col) % 16 == 0, "the colour entries are read 16 bytes at a time");
static_assert(sizeof(PxBuf) * 3 * 10 + 64 <= 160 * 1024, "ten tiles (twenty waves) per CU with three buffers");
// The same for the oriented-disc footprint (SequentialRenderer.ts:91-142; disc.h), which is not separable: no tables — per
// entry the inverse homography's eleven numbers, evaluated by the consumer for the four pixels of every lane the entry's
// box touches (k_composite evaluates it for all 64 pixels of every quadrant the box touches).
struct PxBufDisc {
    float4 par[PXC + 1][3]; // par[1 + j] = {c.x, c.y, -q0, -q1}, {B00, B10, B01, B11}, {lit r, g, b, 1}; par[0]: zeros (the idle entry)
    uint4 q4[4];            // the queue words, as PxBuf's
};
template <bool DISC> struct PxBufOf { typedef PxBuf type; };
template <> struct PxBufOf<true> { typedef PxBufDisc type; };

// Which tile each workgroup of k_composite_px takes: the tiles that took longest first.
// The kernel's duration is its longest tile's plus the time that tile spent sharing its SIMD before it was left alone:
// with every tile resident from the start (C2: 4969 workgroups with entries on 3840 slots) a silhouette tile that walks
// a thousand entries of pixels that never saturate finishes 60 us after tiles that need 150 — unless it is dispatched
// first: the hardware favours the oldest waves, and by the time the bulk has drained the long tiles are done as well
// (C2: 71 -> 57 us with the exact descending order, profiles/r03_c_px_tile_order_oracle_C2.txt).  Which tiles are long is
// not known before they are walked, but a frame resembles the ones before it: every launch leaves each tile's cost
// (chunks walked) behind, and ONE extra workgroup of the next launch — beside the tiles, costing the frame no launch —
// sorts them into nine classes, longest first, empty tiles last, for the launch after.  The order only says who goes
// first: any order (a stale one, one from another scene) gives the same image.
constexpr uint32_t PX_CLASSES = 9;
constexpr uint32_t PX_ORDER_SCRATCH = (PX_CLASSES * 128 + 2 * PX_CLASSES) * 4; // px_make_order's LDS, carved out of the table buffers
__device__ __forceinline__ uint32_t px_cost_class(uint32_t c) { // 0 = longest ... 8 = no entries
    return 8u - ((c >= 1) + (c >= 3) + (c >= 5) + (c >= 7) + (c >= 9) + (c >= 12) + (c >= 16) + (c >= 24));
}
__device__ __forceinline__ void px_make_order(const uint32_t *__restrict__ cost, uint32_t tiles, uint32_t *__restrict__ order, uint32_t *s_scratch) {
    // 128 threads; thread i owns the tiles [i * per, (i + 1) * per) — consecutive, so the order inside a class is the
    // row-major one — and reads their costs sixteen at a time (the arrays are padded to whole 16-byte reads:
    // px_order_prepare).  Its nine counters live in LDS (s_cnt[class][thread]: the class of a tile is a run-time index).
    // The workgroup shares its SIMDs with tile workgroups that would leave it an eighth of the issue slots: top priority.
    __builtin_amdgcn_s_setprio(3);
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t per = (((tiles + 127u) / 128u) + 15u) & ~15u, t0 = tid * per;
    uint32_t *s_cnt = s_scratch;                     // [PX_CLASSES][128]
    uint32_t *s_tot = s_scratch + PX_CLASSES * 128;  // [2][PX_CLASSES]
#pragma unroll
    for (uint32_t q = 0; q < PX_CLASSES; ++q) s_cnt[q * 128 + tid] = 0;
    for (uint32_t i = 0; i < per; i += 16) {
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = reinterpret_cast<const uint4 *>(cost + t0 + i)[j];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t c = reinterpret_cast<const uint32_t *>(v)[j];
            if (t0 + i + j < tiles) s_cnt[px_cost_class(c) * 128 + tid] += 1u; // (this thread's own words: no atomics)
        }
    }
    // exclusive scan over the threads, per class; then the classes one after the other
    uint32_t base[PX_CLASSES];
#pragma unroll
    for (uint32_t q = 0; q < PX_CLASSES; ++q) {
        const uint32_t mine = s_cnt[q * 128 + tid];
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t x = __shfl_up(incl, d);
            if ((int)lane >= d) incl += x;
        }
        if (lane == 63) s_tot[w * PX_CLASSES + q] = incl; // the wave's total
        base[q] = incl - mine;
    }
    __syncthreads();
    uint32_t run = 0;
#pragma unroll
    for (uint32_t q = 0; q < PX_CLASSES; ++q) {
        const uint32_t c0 = s_tot[q], c1 = s_tot[PX_CLASSES + q];
        s_cnt[q * 128 + tid] = base[q] + run + (w ? c0 : 0u); // where this thread's next tile of class q goes
        run += c0 + c1;
    }
    for (uint32_t i = 0; i < per; i += 16) {
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = reinterpret_cast<const uint4 *>(cost + t0 + i)[j];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t c = reinterpret_cast<const uint32_t *>(v)[j];
            const uint32_t t = t0 + i + j;
            if (t < tiles) {
                uint32_t *slot = &s_cnt[px_cost_class(c) * 128 + tid];
                const uint32_t pos = *slot;
                *slot = pos + 1u;
                order[pos] = t;
            }
        }
    }
}

#ifndef PX_WAVES
#define PX_WAVES 5 // (tuning knob of tools/build_variant.sh: minimum waves per SIMD the register allocation must leave room for; LDS allows 5)
#endif
// One workgroup of TWO waves per tile.  Wave 1, the BUILDER, gathers the list's entries (two chunks ahead of their use)
// and per chunk of 32 builds the tables, the colours and the queue words into one of AH + 1 LDS buffers; wave 0, the
// CONSUMER, owns the tile's 256 pixels (2x2 per lane) and walks them.  One s_barrier per chunk: at barrier k the consumer
// has finished chunk k - 1 and the builder chunk k + AH - 1.
//
// Round 4, what changed and why (profiles/r04_*; DESIGN.md "k_composite_px"):
//   * the tables by RECURRENCE.  Along an axis the Gaussian at pixel pairs p0, p0 + 1, ... is G(u + 2k) = G(u) R(u),
//     R(u + 2k) = R(u) D with R(u) = exp2(-4k(u + k)), D = exp2(-8k^2): five v_exp_f32 seed a lane's table at the FIRST
//     COVERED pair of its entry (inside the box the Gaussian is >= exp(-4.5): nothing underflows) and every further pair
//     costs two packed multiplies instead of two exponentials, a packed multiply-add and a packed square; only the covered
//     pairs are written (the consumer's queues never send a lane to an uncovered pair), and the two half-covered edge
//     pairs get their zero by a masked 4-byte store instead of an and-mask on all sixteen values.  In binary32 the
//     recurrence is no less accurate than the direct form it replaces (worst case over 4e5 random entries: 2.1e-6
//     absolute against 2.6e-6 — the direct form cancels in x k - c k —; tools/px_recurrence_error.py).
//   * AH = 2: the builder stays TWO chunks ahead and a lane whose queue for chunk k is empty goes on with its queue for
//     chunk k + 1 (per-pixel order is the list's order either way).  A chunk's trips are its LONGEST queue's length; lanes
//     that ran ahead shorten the next chunk's.
//   * nothing is built that the previous frame did not need, and one chunk more than that is gathered: every launch
//     leaves each tile's cost (chunks touched) behind, and the next launch over the same band builds only that many chunks
//     ahead of need (`lim`).  A tile that turns out to need more — the camera moved — finds the next chunk's records in
//     registers, pays its build at the first chunk past the prediction (both waves meet at one extra barrier) and runs
//     eagerly from there.
//   * both waves execute the same barriers by construction: the consumer's "every pixel has stopped" is latched into
//     s_done[(k + 1) & 1] BEFORE barrier k + 1 and read by the builder AFTER it — the word the consumer may write while
//     walking chunk k + 1 is the other one.
// a tile that has come this far is one of the long ones the kernel's end waits for: its waves go first (measured worth nothing
// either way once the tiles are dispatched longest first: profiles/r04_i_px_priority_variants_C2.txt; kept: it costs nothing)
#define PX_PRIORITY(N)                                         \
    do {                                                       \
        if ((N) == 3u) __builtin_amdgcn_s_setprio(1);          \
        else if ((N) == 6u) __builtin_amdgcn_s_setprio(2);     \
        else if ((N) == 10u) __builtin_amdgcn_s_setprio(3);    \
    } while (0)
// clamp(a b + c, 0, 1) per component in one fused operation (the clamp is the instruction's output modifier).  Packed: with
// composite.hip compiled WITHOUT the packed f32 instructions (-target-feature -packed-fp32-ops: 42 plain instead of 26 vector
// instructions per trip) a trip took 435 instead of 387 cycles, 348 instead of 285 for a wave alone on its SIMD
// (profiles/r04_o_trip_cost_C2.txt): a trip costs its instruction count.
__device__ __forceinline__ v2f fma_clamp01(v2f a, v2f b, v2f c) {
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <bool EARLY_OUT, bool LIT32, bool COUNT, int AH, bool DISC>
__global__ __launch_bounds__(128, PX_WAVES) void k_composite_px(CompositeParams p, uint32_t band_tiles) {
    constexpr uint32_t NB = AH + 1;
    typedef typename PxBufOf<DISC>::type Buf;
    constexpr uint32_t BUF_BYTES = NB * sizeof(Buf) > PX_ORDER_SCRATCH ? NB * sizeof(Buf) : PX_ORDER_SCRATCH;
    __shared__ __attribute__((aligned(16))) char s_raw[BUF_BYTES];
    Buf *const s_buf = reinterpret_cast<Buf *>(s_raw);
    __shared__ uint32_t s_done[2];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)); // 0 = consumer, 1 = builder (wave-uniform)
    if (blockIdx.x == 0) { // not a tile: the frame's report, and the next launch's tile order
        if (p.report && tid == 0) tile_report(p.frame_total, p.report, p.report_seq);
        if (p.order_dst) px_make_order(p.order_src, band_tiles, p.order_dst, reinterpret_cast<uint32_t *>(s_buf));
        return;
    }
    // (dealing the order's positions so that eight neighbouring tiles of a cost class — which gather many of the same records — run
    // behind ONE XCD's L2 changes nothing: C1 / C2 / C3 31.5 / 50.0 / 117.3 -> 30.8 / 49.3 / 117.6 us, profiles/r05_d_px_xcd_groups.txt)
    const uint32_t t_local = p.tile_order ? p.tile_order[blockIdx.x - 1u] : blockIdx.x - 1u;
    const uint32_t tx = t_local % p.ntx, ty = t_local / p.ntx + p.tile_row0;
    const uint32_t tile_idx = ty * p.ntx + tx; // ComputeShaderRenderer.ts:161-163
#ifdef PX_PROFILE
    const uint32_t count = min(p.counts[tile_idx], p.debug_cap), off = p.offsets[tile_idx];
#else
    const uint32_t count = p.counts[tile_idx], off = p.offsets[tile_idx];
#endif
    const uint32_t nchunks = (count + PXC - 1) / PXC;
    const float tile_x0 = (float)(tx * CT), tile_y0 = (float)(ty * CT);
    // chunks built and gathered ahead of need: what the previous launch over this band walked for this tile, at least one, at
    // most all.  Both waves derive the same value: it decides where they meet.  (Chunks of slack on top of it lose: every
    // tile then builds a chunk it does not walk — 54.6 against 47.2 us at C2, profiles/EXPERIMENTS.md §7.1.)
    uint32_t lim = nchunks;
    if (p.cost_prev) lim = min(max(p.cost_prev[t_local], 1u), nchunks);
    lim = (uint32_t)__builtin_amdgcn_readfirstlane((int)lim);

    if (role == 1) {
        // ================================================= builder =================================================
        if (count == 0) return; // (the consumer writes the background; no barrier is executed by either wave)
        const uint32_t e = lane & 31, h = lane >> 5; // lane (e, h) computes entry e's x (h = 0) or y (h = 1) table
        const float tile_c = (h ? tile_y0 : tile_x0) + 0.5f, tile_0 = h ? tile_y0 : tile_x0; // :169 pixel centres, this lane's axis
        if constexpr (DISC) {
            if (lane < 3 * NB) s_buf[lane / 3].par[0][lane % 3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); // the idle entry of every buffer
        } else if (lane <= PX_TROWS) { // the idle slots of every buffer
#pragma unroll
            for (uint32_t b = 0; b < NB; ++b) {
                if (lane < PX_TROWS) s_buf[b].t[lane][0] = make_float2(0.0f, 0.0f);
                else s_buf[b].col[0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
        // Entries are fetched two chunks ahead of their build and their indices three (the gather depends on the index).
        // Every load of this pipeline is UNCONDITIONAL — positions past the last one wanted re-read that one (one line for
        // the whole wave) and are ignored — and nothing is computed from a loaded value before its chunk comes up: a load
        // under a branch, or a use at issue, would make the compiler wait for everything in flight (s_waitcnt vmcnt(0))
        // where it can otherwise count.
        PxRaw ra = {}, rb = {}, rc = {};
        // last list position gathered without being asked for: one chunk beyond the last one BUILT ahead of need, so that a tile
        // which turns out to need one more chunk than the previous frame (the camera moved) finds its records in registers
        // already and pays a build, not a gather
        uint32_t fetch_lim = min(count, (lim + 1u) * PXC) - 1u;
        uint32_t idx_c = p.indices[off + min(2 * PXC + e, fetch_lim)];
#define PX_FETCH(IDX, DST) do { if constexpr (DISC) px_fetch_disc(p, IDX, DST); else px_fetch<LIT32>(p, IDX, DST); } while (0)
        PX_FETCH(p.indices[off + min(e, fetch_lim)], ra);
        PX_FETCH(p.indices[off + min(PXC + e, fetch_lim)], rb);
        uint32_t staged = min(2u * PXC, fetch_lim + 1u);
        uint32_t built = 0; // chunks built so far = the next one to build
#ifdef PX_PROFILE
        const unsigned long long pb0 = __builtin_amdgcn_s_memtime();
        unsigned long long pb_wait = 0, pb_chunks = 0;
#endif
        for (uint32_t k = 0;;) { // before barrier k: every chunk below min(k + AH, lim) is built
            const uint32_t due = min(k + (uint32_t)AH, lim);
            while (built < due) {
                const uint32_t m = built, cb0 = m * PXC;
                PX_PRIORITY(m);
                PX_FETCH(idx_c, rc); // chunk m + 2
                idx_c = p.indices[off + min(cb0 + 3 * PXC + e, fetch_lim)];
                if (COUNT) staged = max(staged, min(cb0 + 3 * PXC, fetch_lim + 1u));
                Buf &B = s_buf[m % NB];
                float4 b, colr;
                float rad;
                if constexpr (DISC) { // the disc's exact screen bounds (what the binner's tile ranges came from); a culled or degenerate record has none
                    const DiscRecord drec = {ra.a, ra.a2};
                    rad = disc_bounds(drec, b) ? 1.0f : 0.0f;
                    colr = (p.prelit || p.disc_lit) ? ra.c : lit_color(ra.c, ra.n);
                } else {
                    px_unpack<LIT32>(p, ra, b, rad, colr);
                }
                // this lane's axis only: its span of covered pixel columns (rows), span_mask16's arithmetic; the other axis's
                // comes from the partner lane (e, 1 - h) — v_permlane32_swap, one instruction — because an entry that misses
                // the tile on EITHER axis draws nothing in it (ComputeShaderRenderer.ts:118-121)
                const float lo = h ? b.y : b.x, hi = h ? b.w : b.z;
                const float fa = fmaxf(ceilf(lo - tile_c), 0.0f), fb = fminf(floorf(hi - tile_c), 15.0f);
                const uint32_t ia = (uint32_t)fminf(fa, 15.0f), ib = (uint32_t)fmaxf(fb, 0.0f); // (in range whatever the bounds: an empty span is rejected below)
                uint32_t own = ((2u << ib) - 1u) & ~((1u << ia) - 1u);
                if (!(fa <= fb) || !(cb0 + e < count) || rad < 0.5f) own = 0; // (NaN bounds;) past the list's end; :127-129 "too small"
                const v2u sw = __builtin_amdgcn_permlane32_swap(own, own, false, false);
                const uint32_t m16 = (h ? sw.x : sw.y) ? own : 0u; // covered columns (rows); zero unless the entry draws something in this tile
                if constexpr (DISC) {
                    // the inverse homography of the entry (disc.h: (u, v) = B d / (1 - q.d)) and its lit colour, as the consumer reads them
                    if (h == 0) {
                        B.par[1 + e][0] = make_float4(ra.a.x, ra.a.y, -ra.a2.z, -ra.a2.w);
                        B.par[1 + e][1] = make_float4(ra.a.z, ra.a2.x, ra.a.w, ra.a2.y); // B by columns
                    } else {
                        B.par[1 + e][2] = make_float4(colr.x, colr.y, colr.z, 1.0f);
                    }
                } else {
                const uint32_t p0 = ia >> 1, npairs = m16 ? (ib >> 1) - p0 + 1u : 0u;
                // gaussian = exp(-0.5 (dist / r)^2 / 0.25) = exp2(-((dx k)^2 + (dy k)^2)), k = sqrt(2 log2 e) / r (:133-140), one
                // axis per lane, in tile-local coordinates.  u = (pixel centre - splat centre) k at the first covered pair's two
                // pixels; v_rcp_f32, 1 ulp: the correctly rounded quotient costs ten instructions on this wave's path.
                const float k1 = 1.6986436005760381f * __builtin_amdgcn_rcpf(rad);
                const float cl = (lo + hi) * 0.5f - tile_0; // :124, then exact
                const float u0 = ((float)(2u * p0) + 0.5f - cl) * k1, u1 = u0 + k1, u2 = u1 + k1, k4 = 4.0f * k1;
                v2f G = {__builtin_amdgcn_exp2f(-(u0 * u0)), __builtin_amdgcn_exp2f(-(u1 * u1))};
                v2f R = {__builtin_amdgcn_exp2f(-(k4 * u1)), __builtin_amdgcn_exp2f(-(k4 * u2))}; // G(u + 2k) / G(u)
                const float Dd = __builtin_amdgcn_exp2f(-2.0f * (k4 * k1));                        // R(u + 2k) / R(u)
                const v2f D = {Dd, Dd};
                // (24-bit multiply: v_mul_lo_u32 is a quarter-rate instruction)
                float2 *row = reinterpret_cast<float2 *>(reinterpret_cast<char *>(&B.t[h * 8][1 + e]) + __umul24(p0, PX_ROW * 8u));
                // (nested: the set of lanes still writing only shrinks, and the loop ends with the wave's longest span)
#define PX_PAIR(I, REST)                                        \
    if (npairs > (I)) {                                         \
        row[(I) * PX_ROW] = make_float2(G.x, G.y);              \
        G *= R;                                                 \
        R *= D;                                                 \
        REST                                                    \
    }
                PX_PAIR(0, PX_PAIR(1, PX_PAIR(2, PX_PAIR(3, PX_PAIR(4, PX_PAIR(5, PX_PAIR(6, PX_PAIR(7, ))))))))
#undef PX_PAIR
                // the box test (ComputeShaderRenderer.ts:118-121, exact: span_mask16's set) on the two pairs the span may
                // cover by half: the uncovered pixel's factor is zero
                if (m16 && (ia & 1u)) row[0].x = 0.0f;
                if (m16 && !(ib & 1u)) reinterpret_cast<float2 *>(reinterpret_cast<char *>(row) + __umul24(npairs - 1u, PX_ROW * 8u))->y = 0.0f;
                // the lit colour beside the tables ({r, g | b, 1}: the 1 is the factor of T's update, PX_BLEND), a half per lane
                reinterpret_cast<float2 *>(&B.col[1 + e])[h] = h ? make_float2(colr.z, 1.0f) : make_float2(colr.x, colr.y);
                }
                // queue words: one ballot gives X[c] (lanes 0..31 test the x mask) and Y[c] (lanes 32..63 the y mask)
                const uint32_t mm = m16 | (m16 >> 1);
                unsigned long long bal[8];
#pragma unroll
                for (int c2 = 0; c2 < 8; ++c2) bal[c2] = __ballot((mm >> (2 * c2)) & 1u);
                if (lane == 0) {
#pragma unroll
                    for (int c2 = 0; c2 < 4; ++c2)
                        B.q4[c2] = make_uint4((uint32_t)bal[2 * c2], (uint32_t)(bal[2 * c2] >> 32), (uint32_t)bal[2 * c2 + 1], (uint32_t)(bal[2 * c2 + 1] >> 32));
                }
                ra = rb;
                rb = rc;
                ++built;
#ifdef PX_PROFILE
                pb_chunks++;
#endif
            }
#ifdef PX_PROFILE
            const unsigned long long pbw = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads(); // barrier k (or the extra one after a misprediction: the consumer waits for chunk k)
#ifdef PX_PROFILE
            pb_wait += __builtin_amdgcn_s_memtime() - pbw;
#endif
            if (EARLY_OUT && __builtin_amdgcn_readfirstlane((int)s_done[k & 1])) break; // latched before this barrier
            if (k >= nchunks) break;
            if (k >= lim) {
                // the tile needs more than was predicted: chunk k's records are here (ra: gathered one chunk beyond the bound);
                // chunk k + 1's gather goes out now, chunk k (.. k + AH - 1) is built at the top of the loop, the consumer is
                // met at the extra barrier, and the tile runs eagerly from here on
                lim = nchunks;
                fetch_lim = count - 1u;
                idx_c = p.indices[off + min((k + 2u) * PXC + e, fetch_lim)];
                PX_FETCH(p.indices[off + min((k + 1u) * PXC + e, fetch_lim)], rb);
                if (COUNT) staged = max(staged, min((k + 2u) * PXC, count));
                built = k;
                continue; // (k stays)
            }
            ++k;
        }
#ifdef PX_PROFILE
        if (p.consumed && lane == 0)
            p.consumed[(size_t)tile_idx * 2] = (((__builtin_amdgcn_s_memtime() - pb0) >> 4) & 0xffffull) | (((pb_wait >> 4) & 0xffffull) << 16) |
                                               ((pb_chunks & 0xffffull) << 48);
        return;
#endif
        if (COUNT && p.consumed && lane == 0) p.consumed[(size_t)tile_idx * 2] += (unsigned long long)staged;
        return;
    }

    // ================================================= consumer ====================================================
    const uint32_t bx = lane & 7, by = lane >> 3; // this lane's 2x2 block: pixel columns 2bx, 2bx+1, rows 2by, 2by+1 of the tile
    const uint32_t px0 = tx * CT + 2 * bx, py0 = ty * CT + 2 * by;
    const bool okx0 = px0 < p.width, okx1 = px0 + 1 < p.width, oky0 = py0 < p.height, oky1 = py0 + 1 < p.height;
    // pixel (row k, column i) of the block: component i of the k-th pair.  T = transmittance; a pixel accumulates while
    // T > T_STOP (:187-190, see T_STOP); pixels outside the image start at 0 and never do
    v2f cr[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, cg[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, cb[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    v2f T[2] = {{(okx0 && oky0) ? 1.0f : 0.0f, (okx1 && oky0) ? 1.0f : 0.0f}, {(okx0 && oky1) ? 1.0f : 0.0f, (okx1 && oky1) ? 1.0f : 0.0f}};
    uint32_t stop_pos = 0; // COUNT: list position after the last entry this lane took while one of its pixels still accumulated
    uint32_t walked = 0;   // chunks this tile needed
    constexpr uint32_t NONE = 0xffffffffu;
    // byte offsets inside a PxBuf of slot 1 (entry 0) of this lane's x row; its y row and the colour rows relative to that
    // (DISC: slot 1 of the parameter array, the same for every lane)
    const uint32_t lane_x = DISC ? 48u : (bx * PX_ROW + 1u) * 8u;
    const uint32_t d_xy = ((8u + by - bx) * PX_ROW) * 8u;
    const uint32_t d_col = DISC ? 0u : (uint32_t)offsetof(PxBuf, col) + 16u - lane_x; // from that slot to slot 1 of the colours
    // DISC: this lane's pixel centres (:169; global coordinates, as the disc records' centres are)
    const v2f pxc = {(float)px0 + 0.5f, (float)px0 + 1.5f}, pyc = {(float)py0 + 0.5f, (float)py0 + 1.5f};
    const char *const lds = reinterpret_cast<const char *>(s_buf);
    v2f k_huge = {0x1p40f, 0x1p40f}, k_stop = {-T_STOP * 0x1p40f, -T_STOP * 0x1p40f}; // PX_BLEND's stop factor
    asm volatile("" : "+v"(k_huge), "+v"(k_stop)); // (kept in vector registers: as known constants they are moved there from scalar ones every trip)
    // PX_DISC_ROW's inside factor, (1 - d2) 2^23 + 1: 1 at d2 = 1, 0 at the next binary32 number above it, exact in one fused
    // operation (2^23 + 1 is the largest odd number a binary32 holds next to 2^23)
    v2f k_in_a = {-0x1p23f, -0x1p23f}, k_in_b = {0x1p23f + 1.0f, 0x1p23f + 1.0f};
    static_assert(0x1p23f + 1.0f == 8388609.0f, "2^23 + 1 is a binary32 number");
    if (DISC) asm volatile("" : "+v"(k_in_a), "+v"(k_in_b));
#ifdef PX_PROFILE
    const unsigned long long pc0 = __builtin_amdgcn_s_memtime();
    unsigned long long pc_wait = 0, pc_trips = 0, pc_ntrips = 0;
#endif
    if (count) {
        if (lane < 2) s_done[lane] = 0;
        bool finished = false;      // every pixel has stopped (latched in s_done for the builder)
        bool carried_ok = false;    // `carried` is what is left of this chunk's queues (the lanes ran ahead into it)
        uint32_t carried = 0;
        for (uint32_t k = 0;; ++k) {
#ifdef PX_PROFILE
            { const unsigned long long w0_ = __builtin_amdgcn_s_memtime(); __syncthreads(); pc_wait += __builtin_amdgcn_s_memtime() - w0_; }
#else
            __syncthreads(); // barrier k: chunks below min(k + AH, lim) are built; the buffer of chunk k - 1 is the builder's again
#endif
            if (EARLY_OUT && finished) break;
            if (k >= nchunks) break;
            if (k >= lim) { // chunk k was not predicted: the builder gathers and builds it now
                lim = nchunks;
                __syncthreads();
            }
            const uint32_t o0 = (k % NB) * (uint32_t)sizeof(Buf), o1 = ((k + 1u) % NB) * (uint32_t)sizeof(Buf);
            const uint32_t cb0 = k * PXC;
            walked = k + 1;
            PX_PRIORITY(k);
            const bool ahead = AH >= 2 && k + 1u < lim; // chunk k + 1 is built: lanes may run ahead into it
            // a lane whose four pixels have all stopped takes no more entries
            const bool lane_live = !EARLY_OUT || fmaxf(fmaxf(T[0].x, T[0].y), fmaxf(T[1].x, T[1].y)) > T_STOP;
            uint32_t mine, nxt = 0;
            {
                const uint2 *q0 = reinterpret_cast<const uint2 *>(lds + o0 + offsetof(Buf, q4));
                mine = carried_ok ? carried : (q0[bx].x & q0[by].y);
                if (ahead) {
                    const uint2 *q1 = reinterpret_cast<const uint2 *>(lds + o1 + offsetof(Buf, q4));
                    nxt = q1[bx].x & q1[by].y;
                }
                if (!lane_live) mine = nxt = 0;
                carried_ok = ahead;
            }
            const uint32_t nxt0 = nxt;
            const uint32_t ax0 = o0 + lane_x, ax1 = o1 + lane_x;
            // ---- trips.  Every lane takes its next entry — of chunk k while it has one, then of chunk k + 1, the idle one when
            // both queues are empty: no divergence, an idle lane adds exact zeros; the NEXT entry's table and colour reads are
            // issued before the current one is blended, so a trip costs its arithmetic, not an LDS round trip.
#define PX_POP(J, R, A, S)                                                                       \
    do {                                                                                         \
        uint32_t mc_ = mine;                                                                     \
        asm("" : "+v"(mc_)); /* (an opaque copy: left visible, the compiler derives `mine != 0` from the borrow of mine - 1 below — three instructions for one compare) */ \
        const bool sel_ = mc_ != 0;                                                              \
        R = __ballot(sel_); /* lanes that still have an entry of chunk k: none = the chunk is finished, this trip is not made */ \
        const uint32_t q_ = (AH >= 2) ? (sel_ ? mine : nxt) : mine;                              \
        asm("v_ffbl_b32 %0, %1" : "=v"(J) : "v"(q_)); /* -1 for 0 (the idle slot): no select; __builtin_ctz(0) is undefined */ \
        const uint32_t q2_ = q_ & (q_ - 1u);                                                     \
        if (AH >= 2) {                                                                           \
            mine = sel_ ? q2_ : 0u;                                                              \
            /* an entry of chunk k + 1 leaves a lane's queue only in a trip that is made (R != 0) and only while one of the lane's \
               pixels still accumulates (as of the blend before last: the walk's cost counts the chunks LIVE lanes touched);   \
               the mask is scalar work */                                                        \
            nxt = __builtin_amdgcn_inverse_ballot_w64((R ? R : ~0ull) | ~alive_m) ? nxt : q2_;   \
            A = sel_ ? ax0 : ax1;                                                                \
            if (COUNT) S = sel_ ? 0u : (uint32_t)PXC;                                            \
        } else {                                                                                 \
            mine = q2_;                                                                          \
            A = ax0;                                                                             \
        }                                                                                        \
    } while (0)
#define PX_LOAD(J, A, E)                                                               \
    do {                                                                               \
        if constexpr (DISC) {                                                          \
            const float4 *a_ = reinterpret_cast<const float4 *>(lds + (A + (uint32_t)((int)J * 48))); \
            E.p0 = a_[0];                                                              \
            E.p1 = a_[1];                                                              \
            E.p2 = a_[2];                                                              \
        } else {                                                                       \
            const char *a_ = lds + (A + (uint32_t)((int)J * 8));                       \
            E.gx = *reinterpret_cast<const float2 *>(a_);                              \
            E.gy = *reinterpret_cast<const float2 *>(a_ + d_xy);                       \
            const float4 c_ = *reinterpret_cast<const float4 *>(lds + (A + d_col + (uint32_t)((int)J * 16))); \
            E.c0 = make_float2(c_.x, c_.y);                                            \
            E.c1 = make_float2(c_.z, c_.w);                                            \
        }                                                                              \
    } while (0)
// The per-pixel stop (:187-190: a pixel that has reached alpha >= 0.99, i.e. T <= T_STOP, takes nothing more) as a FACTOR:
// m = clamp((T - T_STOP) 2^40, 0, 1) is exactly 1 while T > T_STOP (the smallest positive difference of two binary32 numbers
// near T_STOP is 2^-30) and exactly 0 from then on — one v_pk_fma_f32 with the clamp modifier per pixel PAIR and one packed
// multiply (w = T g m: the same bits as T g where m = 1), where a compare and a select per PIXEL stood before.
// CHECK (every other trip; every trip when the consumed entries are counted): which lanes still have a pixel accumulating —
// a lane with none gives up the rest of its queue at once, not at the next chunk, and when there is none at all the chunk's
// remaining trips are not made.  Neither changes a pixel: the factor does the stopping.
// DISC: the footprint of SequentialRenderer.ts:91-142 per pixel (disc.h): (u, v) = B d / (1 - q.d) with d = pixel centre -
// disc centre, inside when u^2 + v^2 <= 1 (:128-130), gaussian exp(-0.5 d2 / 0.16) (:132-133).  Per row of the lane's 2x2
// block the two columns go through packed arithmetic; the row's own terms (q1 dy, B01 dy, B11 dy) are scalars of it.
#define PX_DISC_ROW(DY, G)                                                                                                    \
    do {                                                                                                                      \
        const float s_ = __builtin_fmaf(E_.p0.w, DY, 1.0f);                         /* 1 - q1 dy */                            \
        const v2f den_ = (v2f){E_.p0.z, E_.p0.z} * dx_ + (v2f){s_, s_};             /* 1 - q.d   */                            \
        const v2f rd_ = {__builtin_amdgcn_rcpf(den_.x), __builtin_amdgcn_rcpf(den_.y)};                                       \
        const v2f by_ = (v2f){E_.p1.z, E_.p1.w} * (v2f){DY, DY};                    /* B01 dy, B11 dy */                       \
        const v2f nu_ = (v2f){E_.p1.x, E_.p1.x} * dx_ + (v2f){by_.x, by_.x};        /* (B d) */                                \
        const v2f nv_ = (v2f){E_.p1.y, E_.p1.y} * dx_ + (v2f){by_.y, by_.y};                                                  \
        /* :126 u^2 + v^2 = |B d|^2 / (1 - q.d)^2 — in this form no 0 * inf can arise (|B d| = 0 only at the centre, where    \
           1 - q.d = 1): a pixel on the disc plane's horizon line gets inf (outside), never NaN */                            \
        const v2f d2_ = (nu_ * nu_ + nv_ * nv_) * (rd_ * rd_);                                                                \
        const v2f ar_ = d2_ * (v2f){DISC_EXP2_SCALE, DISC_EXP2_SCALE};                                                        \
        /* :128-133 discard outside the unit circle, as a factor: clamp((1 - d2) 2^23 + 1, 0, 1) is exactly 1 for d2 <= 1 and  \
           exactly 0 from the next binary32 number above 1 on (and for NaN): one packed multiply-add for two compares + selects */ \
        v2f in_;                                                                                                              \
        in_ = fma_clamp01(d2_, k_in_a, k_in_b);                                                                               \
        G = (v2f){__builtin_amdgcn_exp2f(ar_.x), __builtin_amdgcn_exp2f(ar_.y)} * in_;                                        \
    } while (0)
#define PX_BLEND(J, R, S, E, CHECK)                                                                                           \
    do {                                                                                                                      \
        const auto &E_ = E;                                                                                                   \
        v2f w0, w1;                                                                                                           \
        float2 C0, C1;                                                                                                        \
        if constexpr (DISC) {                                                                                                 \
            const v2f dx_ = pxc - (v2f){E_.p0.x, E_.p0.x};                                                                    \
            v2f g0_, g1_;                                                                                                     \
            PX_DISC_ROW(pyc.x - E_.p0.y, g0_);                                                                                \
            PX_DISC_ROW(pyc.y - E_.p0.y, g1_);                                                                                \
            w0 = T[0] * g0_;                                                                                                  \
            w1 = T[1] * g1_;                                                                                                  \
            C0 = make_float2(E_.p2.x, E_.p2.y);                                                                               \
            C1 = make_float2(E_.p2.z, E_.p2.w);                                                                               \
        } else {                                                                                                              \
            const v2f gxx = {E_.gx.x, E_.gx.y};                                                                               \
            w0 = T[0] * (gxx * (v2f){E_.gy.x, E_.gy.x});                            /* rows 2by, 2by+1: w = T g */             \
            w1 = T[1] * (gxx * (v2f){E_.gy.y, E_.gy.y});                                                                      \
            C0 = E_.c0;                                                                                                       \
            C1 = E_.c1;                                                                                                       \
        }                                                                                                                     \
        if (EARLY_OUT) {                                                                                                      \
            v2f m0, m1;                                                                                                       \
            m0 = fma_clamp01(T[0], k_huge, k_stop);                                                                          \
            m1 = fma_clamp01(T[1], k_huge, k_stop);                                                                          \
            if (CHECK) {                                                                                                      \
                const v2f ms_ = m0 + m1; /* the four factors are 0 or 1: their sum says whether any pixel of the lane still accumulates */ \
                const unsigned long long al_ = __ballot(ms_.x + ms_.y > 0.0f);                                               \
                /* COUNT: the last entry a lane takes while one of its pixels still accumulates is the one that stops its    \
                   last pixel (if they all stop) */                                                                           \
                if (COUNT) jlast = (__builtin_amdgcn_inverse_ballot_w64(al_) && J != NONE) ? J + S : jlast;                   \
                all_stopped = al_ == 0;                                                                                       \
                mine = __builtin_amdgcn_inverse_ballot_w64(al_) ? mine : 0u;                                                  \
                alive_m = al_;                                                                                                \
            }                                                                                                                 \
            w0 *= m0;                                                                                                         \
            w1 *= m1;                                                                                                         \
        }                                                                                                                     \
        /* SURVEY §8a contract 3: nearest on top.  C += c w in ONE rounding (v_pk_fma_f32; composite.hip is the one file built     \
           with contraction on: the sum is within the composite's stated tolerance either way) */                             \
        cr[0] = __builtin_elementwise_fma((v2f){C0.x, C0.x}, w0, cr[0]); cr[1] = __builtin_elementwise_fma((v2f){C0.x, C0.x}, w1, cr[1]); \
        cg[0] = __builtin_elementwise_fma((v2f){C0.y, C0.y}, w0, cg[0]); cg[1] = __builtin_elementwise_fma((v2f){C0.y, C0.y}, w1, cg[1]); \
        cb[0] = __builtin_elementwise_fma((v2f){C1.x, C1.x}, w0, cb[0]); cb[1] = __builtin_elementwise_fma((v2f){C1.x, C1.x}, w1, cb[1]); \
        /* T (1 - g) with the product in hand; C1.y is 1.0 for an entry, 0 for the idle one: the product is exact, so this is \
           the same bits as T - w */                                                                                          \
        T[0] = __builtin_elementwise_fma((v2f){-C1.y, -C1.y}, w0, T[0]); T[1] = __builtin_elementwise_fma((v2f){-C1.y, -C1.y}, w1, T[1]); \
    } while (0)
#ifdef PX_PROFILE
            const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#endif
            uint32_t ja, jb, jlast = NONE, aa = 0, ab = 0, sa = 0, sb = 0;
            unsigned long long alive_m = __ballot(lane_live); // lanes with a pixel still accumulating (as of the last blend)
            bool all_stopped = false; // every pixel of the tile had stopped before the entry just blended: the rest of the chunk is zeros
            unsigned long long ra_, rb_;
            struct PxEnt { // what a trip reads of its entry: table values and colour, or (DISC) the three parameter vectors
                float2 gx, gy, c0, c1;
                float4 p0, p1, p2;
            } ea, eb;
            PX_POP(ja, ra_, aa, sa);
            PX_LOAD(ja, aa, ea);
            for (;;) {
                if (ra_ == 0) break;
#ifdef PX_PROFILE
                pc_ntrips++;
#endif
                PX_POP(jb, rb_, ab, sb);
                PX_LOAD(jb, ab, eb);
                __builtin_amdgcn_sched_barrier(0); // (the scheduler would sink the reads to their use, one trip later: the point is lost)
                PX_BLEND(ja, ra_, sa, ea, true);
                if (EARLY_OUT && all_stopped) break;
                if (rb_ == 0) break;
#ifdef PX_PROFILE
                pc_ntrips++;
#endif
                PX_POP(ja, ra_, aa, sa);
                PX_LOAD(ja, aa, ea);
                __builtin_amdgcn_sched_barrier(0);
                PX_BLEND(jb, rb_, sb, eb, COUNT);
                if (EARLY_OUT && all_stopped) break;
            }
#undef PX_POP
#undef PX_LOAD
#undef PX_BLEND
#undef PX_DISC_ROW
#ifdef PX_PROFILE
            if (__ballot(T[0].x > 1e30f) == 0) pc_trips += __builtin_amdgcn_s_memtime() - pt0;
#endif
            carried = nxt;
            // the tile's cost = the chunks it touched: if this turns out to be its last chunk and some lane took an entry of the next
            // one in passing, the next launch must find that one built as well (or the tile would need it and not have it every
            // other frame)
            if (AH >= 2) walked += __ballot(nxt != nxt0) != 0 ? 1u : 0u;
            if (COUNT && EARLY_OUT && jlast != NONE) stop_pos = cb0 + jlast + 1; // (a lane's entries come in list order: later ones overwrite)
            if (EARLY_OUT) {
                // the tile is finished when every pixel has stopped: both waves leave after the next barrier
                const bool live = fmaxf(fmaxf(T[0].x, T[0].y), fmaxf(T[1].x, T[1].y)) > T_STOP;
                finished = __ballot(live) == 0;
                if (finished && lane == 0) s_done[(k + 1u) & 1u] = 1;
            }
        }
    }

    if (p.tile_cost && lane == 0) p.tile_cost[t_local] = walked; // (what the next launch over this band is ordered and sized by)
#ifdef PX_PROFILE
    if (p.consumed && lane == 0)
        p.consumed[(size_t)tile_idx * 2 + 1] = (((__builtin_amdgcn_s_memtime() - pc0) >> 4) & 0xffffull) | (((pc_wait >> 4) & 0xffffull) << 16) |
                                               (((pc_trips >> 4) & 0xffffull) << 32) | ((pc_ntrips & 0xffffull) << 48);
#else
    if (COUNT && p.consumed && count) { // (uniform branch; timed / diagnostic runs only)
        uint32_t used = stop_pos;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) used = max(used, (uint32_t)__shfl_xor((int)used, d));
        const bool px_live = fmaxf(fmaxf(T[0].x, T[0].y), fmaxf(T[1].x, T[1].y)) > T_STOP;
        if (!EARLY_OUT || __ballot(px_live) != 0) used = count;
        if (lane == 0) p.consumed[(size_t)tile_idx * 2 + 1] += (unsigned long long)used;
    }
#endif

    // :193-197
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t py = py0 + r;
        if (py >= p.height || !okx0) continue;
        const float r0 = cr[r].x + 0.05f * T[r].x, g0 = cg[r].x + 0.05f * T[r].x, b0 = cb[r].x + 0.1f * T[r].x;
        const float r1 = cr[r].y + 0.05f * T[r].y, g1 = cg[r].y + 0.05f * T[r].y, b1 = cb[r].y + 0.1f * T[r].y;
        const size_t o = (size_t)py * p.width + px0;
        const uint32_t q0 = unorm8(r0) | (unorm8(g0) << 8) | (unorm8(b0) << 16) | (255u << 24);
        const uint32_t q1 = unorm8(r1) | (unorm8(g1) << 8) | (unorm8(b1) << 16) | (255u << 24);
        if (p.out_rgba8) {
            if (okx1 && (o & 1) == 0) *reinterpret_cast<uint2 *>(p.out_rgba8 + o) = make_uint2(q0, q1);
            else {
                p.out_rgba8[o] = q0;
                if (okx1) p.out_rgba8[o + 1] = q1;
            }
        }
        if (p.out_rgba32f) {
            p.out_rgba32f[o] = make_float4(r0, g0, b0, 1.0f);
            if (okx1) p.out_rgba32f[o + 1] = make_float4(r1, g1, b1, 1.0f);
        }
    }
}

static bool g_px_order_on = true;   // SPLAT_TILE_ORDER=0: workgroups take their tiles row-major (read once per process)
static bool g_px_predict_on = true; // SPLAT_PX_PREDICT=0: every tile builds and gathers ahead of need without a bound
static int g_px_ahead = 0;          // SPLAT_PX_AHEAD=1 | 2: chunks the builder stays ahead (2: lanes run ahead too); unset: 1 with the early-out, 2 without

// The history of one band of one binner's lists: two cost arrays and two order arrays, alternating.  Launch k over the band
// writes cost[k & 1]; it sizes every tile's look-ahead by cost[(k + 1) & 1] (launch k - 1's), takes its tiles in order[k & 1] —
// which launch k - 1's ordering workgroup derived from cost[k & 1] as launch k - 2 had left it — and derives
// order[(k + 1) & 1] from cost[(k + 1) & 1].  All of it is a hint: any order and any bound give the same image.
// A context keeps PX_HISTORIES of them, found by `key` (the band, the screen, whose lists), least recently used evicted: two
// frames in flight on one context, a band frame next to the whole frame, virtual ranks on one context each keep theirs
// (with ONE slot they reset each other at every launch, and neither the order nor the bound ever engaged: ADVICE r4).
static int px_order_prepare(splat_ctx *ctx, uint32_t band_tiles, uint64_t key, CompositeParams &p) {
    p.tile_order = nullptr;
    p.tile_cost = nullptr;
    p.order_src = nullptr;
    p.order_dst = nullptr;
    p.cost_prev = nullptr;
    const bool predict_on = ctx->opt_px_predict >= 0 ? ctx->opt_px_predict != 0 : g_px_predict_on;
    if (!g_px_order_on && !predict_on) return SPLAT_OK;
    PxHistory *h = nullptr, *lru = &ctx->px_hist[0];
    for (auto &e : ctx->px_hist) {
        if (e.key == key) h = &e;
        if (e.last_use < lru->last_use) lru = &e;
    }
    if (!h) { // a band this context has not composited lately: the least recently used slot starts over for it
        h = lru;
        h->key = key;
        h->streak = 0;
    }
    h->last_use = ++ctx->px_clock;
    if (band_tiles > h->cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // (a launch in flight may still write the old arrays)
        if (h->mem) (void)hipFree(h->mem);
        h->mem = nullptr;
        h->cap = 0;
        h->streak = 0;
        // (each array padded so that px_make_order's 128 threads read whole 64-byte groups of costs past the last tile)
        const uint32_t cap = (band_tiles + 128u * 16u + 1023u) & ~1023u;
        if (hipMalloc((void **)&h->mem, (size_t)cap * 16) != hipSuccess) {
            h->key = 0;
            return ctx_fail(ctx, SPLAT_ERR_OOM, "composite tile order hipMalloc");
        }
        (void)hipMemsetAsync(h->mem, 0, (size_t)cap * 16, ctx->stream);
        h->cap = cap;
    }
    uint32_t *cost[2] = {h->mem, h->mem + h->cap}, *order[2] = {h->mem + 2 * (size_t)h->cap, h->mem + 3 * (size_t)h->cap};
    const uint32_t q = h->parity & 1u;
    p.tile_cost = cost[q];
    if (g_px_order_on && h->streak >= 2) p.tile_order = order[q]; // written by the previous launch from the costs of the one before it
    if (h->streak >= 1) {                                          // the previous launch left its costs
        if (predict_on) p.cost_prev = cost[q ^ 1u];
        if (g_px_order_on) { // sort them for the next launch
            p.order_src = cost[q ^ 1u];
            p.order_dst = order[q ^ 1u];
        }
    }
    h->parity ^= 1u;
    if (h->streak < 2) h->streak++;
#ifdef PX_PROFILE
    { // (measuring build only, SPLAT_PX_FREEZE=1: the order found by the first launches is kept and no launch computes another: is the ordering workgroup free?)
        static int freeze = -1;
        static uint32_t *frozen = nullptr;
        if (freeze < 0) freeze = getenv("SPLAT_PX_FREEZE") ? 1 : 0;
        if (freeze) {
            if (frozen && p.tile_order) { p.tile_order = frozen; p.order_src = nullptr; p.order_dst = nullptr; }
            else if (p.tile_order) frozen = const_cast<uint32_t *>(p.tile_order);
        }
    }
#endif
    return SPLAT_OK;
}

extern "C" int splat_lit_colors(splat_ctx *ctx, const void *color_opacity, uint32_t color_stride_vec4, const void *normals,
                                uint32_t normal_stride_vec4, uint32_t n, void *lit) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, n == 0 || (color_opacity && normals && lit));
    ARG_CHECK(ctx, color_stride_vec4 >= 1 && normal_stride_vec4 >= 1);
    ARG_CHECK(ctx, (((uintptr_t)color_opacity | (uintptr_t)normals | (uintptr_t)lit) & 15) == 0);
    if (n == 0) return SPLAT_OK;
    hipLaunchKernelGGL(k_lit_colors, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, (const float4 *)color_opacity, color_stride_vec4,
                       (const float4 *)normals, normal_stride_vec4, n, (float4 *)lit);
    LAUNCH_CHECK(ctx, "k_lit_colors");
    return SPLAT_OK;
}

__global__ void k_frame_report(const uint32_t *frame_total, uint32_t *report, uint32_t seq) { tile_report(frame_total, report, seq); }

extern "C" int splat_composite(splat_ctx *ctx, const splat_composite_cfg *cfg, const void *color_opacity,
                               uint32_t color_stride_vec4, const void *normals, uint32_t normal_stride_vec4,
                               const void *projected, const void *tile_indices, const void *tile_counts,
                               const void *tile_offsets, uint32_t width, uint32_t height, void *out_rgba8, void *out_rgba32f,
                               void *consumed_dptr) {
    // lit DISC records (48 bytes: the lit colour behind each disc record) exist only inside a frame's binner; a public caller's
    // disc records are the projector's 32-byte ones (splat_project_disc) or 48-byte exchange records (taken at its
    // word this pair would read idx * 48 out of an n * 32 allocation)
    if (ctx && cfg && cfg->footprint == SPLAT_FOOTPRINT_DISC && cfg->record_format == SPLAT_RECORDS_LIT32)
        return ctx_fail(ctx, SPLAT_ERR_INVALID, "splat_composite: the oriented-disc footprint composites from SPLAT_RECORDS_PROJECTED (32-byte disc "
                                                "records) or SPLAT_RECORDS_DISC48; lit disc records are internal to splat_render_frame");
    return composite_launch(ctx, cfg, color_opacity, color_stride_vec4, normals, normal_stride_vec4, projected, tile_indices, tile_counts,
                            tile_offsets, width, height, out_rgba8, out_rgba32f, consumed_dptr, nullptr, nullptr, 0u);
}

static int composite_launch_checked(splat_ctx *ctx, const splat_composite_cfg *cfg, const void *color_opacity, uint32_t color_stride_vec4,
                                    const void *normals, uint32_t normal_stride_vec4, const void *projected, const void *tile_indices,
                                    const void *tile_counts, const void *tile_offsets, uint32_t width, uint32_t height, void *out_rgba8,
                                    void *out_rgba32f, void *consumed_dptr, const uint32_t *frame_total, uint32_t *report,
                                    uint32_t report_seq, bool *launched);

// The composite with the frame's report attached.  Whatever happens to the launch, a report that was promised is sent
// (the host waits for it at its next call).
int composite_launch(splat_ctx *ctx, const splat_composite_cfg *cfg, const void *color_opacity, uint32_t color_stride_vec4, const void *normals,
                     uint32_t normal_stride_vec4, const void *projected, const void *tile_indices, const void *tile_counts,
                     const void *tile_offsets, uint32_t width, uint32_t height, void *out_rgba8, void *out_rgba32f, void *consumed_dptr,
                     const uint32_t *frame_total, uint32_t *report, uint32_t report_seq) {
    bool launched = false;
    const int rc = composite_launch_checked(ctx, cfg, color_opacity, color_stride_vec4, normals, normal_stride_vec4, projected, tile_indices,
                                            tile_counts, tile_offsets, width, height, out_rgba8, out_rgba32f, consumed_dptr, frame_total, report,
                                            report_seq, &launched);
    if (ctx && report && !launched) { // an empty band of tile rows, or a rejected argument: the report goes out on its own
        hipLaunchKernelGGL(k_frame_report, dim3(1), dim3(1), 0, ctx->stream, frame_total, report, report_seq);
        (void)hipGetLastError();
    }
    return rc;
}

static int composite_launch_checked(splat_ctx *ctx, const splat_composite_cfg *cfg, const void *color_opacity, uint32_t color_stride_vec4,
                                    const void *normals, uint32_t normal_stride_vec4, const void *projected, const void *tile_indices,
                                    const void *tile_counts, const void *tile_offsets, uint32_t width, uint32_t height, void *out_rgba8,
                                    void *out_rgba32f, void *consumed_dptr, const uint32_t *frame_total, uint32_t *report,
                                    uint32_t report_seq, bool *launched) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, cfg != nullptr);
    ARG_CHECK(ctx, cfg->tile_size == CT); // the kernel's quadrant mapping is built for 16x16 tiles
    ARG_CHECK(ctx, cfg->mode == SPLAT_COMPOSITE_FRONT_TO_BACK || cfg->mode == SPLAT_COMPOSITE_REFERENCE_LITERAL);
    ARG_CHECK(ctx, cfg->record_format <= SPLAT_RECORDS_LIT32);
    ARG_CHECK(ctx, width >= 1 && height >= 1 && width <= 65535u * CT && height <= 65535u * CT);
    // the records carry the lit colour (isotropic: 32-byte lit composite records; disc: the colour behind each disc record): no colour / normal arrays
    const bool lit32 = cfg->record_format == SPLAT_RECORDS_LIT32;
    ARG_CHECK(ctx, lit32 || (color_opacity && (normals || cfg->prelit)));
    ARG_CHECK(ctx, projected && tile_indices && tile_counts && tile_offsets);
    ARG_CHECK(ctx, color_stride_vec4 >= 1 && normal_stride_vec4 >= 1);
    ARG_CHECK(ctx, out_rgba8 || out_rgba32f);
    ARG_CHECK(ctx, (((uintptr_t)color_opacity | (uintptr_t)normals | (uintptr_t)projected | (uintptr_t)out_rgba32f) & 15) == 0);
    ARG_CHECK(ctx, cfg->prelit <= 1 && cfg->footprint <= SPLAT_FOOTPRINT_DISC);
    // the oriented disc is SequentialRenderer's footprint: nearest-on-top "over" is its only blend, and its
    // records are the projector's 32-byte disc records
    ARG_CHECK(ctx, cfg->footprint != SPLAT_FOOTPRINT_DISC ||
                       (cfg->mode == SPLAT_COMPOSITE_FRONT_TO_BACK && cfg->record_format != SPLAT_RECORDS_COMPACT));
    ARG_CHECK(ctx, cfg->record_format != SPLAT_RECORDS_DISC48 || cfg->footprint == SPLAT_FOOTPRINT_DISC);
    const uint32_t ntx = div_up(width, CT), nty = div_up(height, CT);
    uint32_t r0 = cfg->tile_row0, r1 = cfg->tile_row1 > nty ? nty : cfg->tile_row1;
    if (r0 >= r1) return SPLAT_OK;
    CompositeParams p;
    p.color = (const float4 *)color_opacity;
    p.color_stride = color_stride_vec4;
    p.normals = (const float4 *)normals;
    p.normal_stride = normal_stride_vec4;
    p.projected = (const float4 *)projected;
    p.compact = cfg->record_format == SPLAT_RECORDS_COMPACT;
    p.lit32 = cfg->record_format == SPLAT_RECORDS_LIT32;
    p.prelit = cfg->prelit != 0;
    p.disc = cfg->footprint == SPLAT_FOOTPRINT_DISC;
    p.disc_stride = (cfg->record_format == SPLAT_RECORDS_DISC48 || (p.disc && lit32)) ? 3u : 2u;
    p.disc_lit = p.disc && lit32;
    p.indices = (const uint32_t *)tile_indices;
    p.counts = (const uint32_t *)tile_counts;
    p.offsets = (const uint32_t *)tile_offsets;
    p.width = width;
    p.height = height;
    p.ntx = ntx;
    p.tile_row0 = r0;
    p.out_rgba8 = (uint32_t *)out_rgba8;
    p.out_rgba32f = (float4 *)out_rgba32f;
    p.consumed = (unsigned long long *)consumed_dptr;
    p.frame_total = frame_total;
    p.report = report;
    p.report_seq = report_seq;
    p.tile_order = nullptr;
    p.tile_cost = nullptr;
    p.order_src = nullptr;
    p.order_dst = nullptr;
    p.cost_prev = nullptr;
#ifdef PX_PROFILE
    p.debug_cap = getenv("SPLAT_PX_CAP") ? (uint32_t)strtoul(getenv("SPLAT_PX_CAP"), nullptr, 10) : 0xffffffffu;
#endif
    dim3 grid(ntx, r1 - r0), block(256);
    const bool eo = cfg->early_out != 0;
    // timed runs attach the event pair to the launch itself (no marker packets around the kernel).  The pair is taken right
    // before the launch, after everything that can fail: a pair that was handed out and never recorded would be read by
    // splat_stage_time_stats
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
#define SPLAT_COMPOSITE_LAUNCH(MODE, EO, DISC, LIT)                                                                       \
    do {                                                                                                                  \
        if (timed) hipExtLaunchKernelGGL((k_composite<MODE, EO, DISC, LIT>), grid, block, 0, ctx->stream, ev0, ev1, 0, p); \
        else hipLaunchKernelGGL((k_composite<MODE, EO, DISC, LIT>), grid, block, 0, ctx->stream, p);                     \
    } while (0)
    // Which kernel composites the isotropic footprint nearest-on-top (the frame's default): k_composite_px — one wave per
    // tile, every lane walking its own entries — unless SPLAT_COMPOSITE=quadrant asks for round 2's k_composite (one
    // 8x8-pixel wave per quadrant visiting every entry of the tile), which also serves the oriented disc (its footprint is
    // not separable) and the reference-literal blend.
    // Which kernel composites the isotropic footprint nearest-on-top (the frame's default): k_composite_px — a builder and
    // a consumer wave per tile, every lane walking its own entries — on screens of at least PX_MIN_TILES tiles (the whole
    // screen decides, not the band of tile rows this call renders: the ranks of a multi-GPU frame must run the same
    // arithmetic for their bands to stitch into the single-GPU image bit for bit); below that (C0's 256 tiles: 18 against
    // 22 us) and for the oriented disc (its footprint is not separable) and the reference-
    // literal blend, round 2's k_composite (four waves per tile, each visiting every entry that touches its quadrant).
    // SPLAT_COMPOSITE=pixel | quadrant forces one of them.
    constexpr uint32_t PX_MIN_TILES = 2048;
    static int s_px = -2;
    if (s_px == -2) {
        const char *e = getenv("SPLAT_COMPOSITE");
        s_px = !e ? -1 : (e[0] == 'p' || e[0] == 'P') ? 1 : (e[0] == 'q' || e[0] == 'Q') ? 0 : -1;
        const char *o = getenv("SPLAT_TILE_ORDER"); // =0: workgroups take tiles row-major
        g_px_order_on = !(o && o[0] == '0');
        const char *pr = getenv("SPLAT_PX_PREDICT"); // =0: no look-ahead bound from the previous launch's costs
        g_px_predict_on = !(pr && pr[0] == '0');
        const char *ah = getenv("SPLAT_PX_AHEAD");
        g_px_ahead = !ah ? 0 : ah[0] == '1' ? 1 : 2;
    }
    const int opt_kernel = ctx->opt_composite_kernel >= 0 ? ctx->opt_composite_kernel : s_px;
    // Two chunks ahead with lanes running ahead makes 27 % fewer trips, each four instructions longer, on five waves per
    // SIMD instead of eight (three table buffers): a gain where every list is walked to its end (early-out off: C2 319 ->
    // 298 us), a loss where most tiles stop after a few chunks and a tile's life is mostly latency (50.6 -> 52.8 us):
    // profiles/r04_b_px_ab_C2.txt
    // (the oriented disc keeps eleven numbers per entry instead of tables: a third buffer costs it no resident workgroup)
    const int opt_ahead = ctx->opt_px_ahead > 0 ? ctx->opt_px_ahead : g_px_ahead > 0 ? g_px_ahead : ((cfg->early_out && !p.disc) ? 1 : 2);
    const uint32_t band_tiles = ntx * (r1 - r0);
    const bool use_px = cfg->mode == SPLAT_COMPOSITE_FRONT_TO_BACK && (opt_kernel == 1 || (opt_kernel == -1 && ntx * nty >= PX_MIN_TILES));
    if (use_px) {
        // (the band, the screen, and whose lists these are: two binners on one context do not share a history)
        const uint64_t key = (((uint64_t)ntx << 40) ^ ((uint64_t)r0 << 20) ^ (uint64_t)r1 ^ ((uint64_t)width << 50) ^
                              ((uint64_t)(uintptr_t)tile_counts * 0x9E3779B97F4A7C15ull)) | 1u;
        int orc = px_order_prepare(ctx, band_tiles, key, p);
        if (orc != SPLAT_OK) return orc;
#ifdef SPLAT_TEST_HOOKS
        if (ctx->debug_tile_order) p.tile_order = ctx->debug_tile_order;
#endif
        timed = stage_event_pair(ctx, SPLAT_STAGE_COMPOSITE, &ev0, &ev1);
        // one workgroup of two waves (consumer, builder) per tile, behind workgroup 0 (report, next launch's tile order)
        const dim3 pgrid(band_tiles + 1u), pblock(128);
#define SPLAT_COMPOSITE_PX_LAUNCH2(EO, LIT, CNT, AH, DISC)                                                                       \
    do {                                                                                                                        \
        if (timed) hipExtLaunchKernelGGL((k_composite_px<EO, LIT, CNT, AH, DISC>), pgrid, pblock, 0, ctx->stream, ev0, ev1, 0, p, band_tiles); \
        else hipLaunchKernelGGL((k_composite_px<EO, LIT, CNT, AH, DISC>), pgrid, pblock, 0, ctx->stream, p, band_tiles);                       \
    } while (0)
#define SPLAT_COMPOSITE_PX_LAUNCH(EO, LIT, DISC)                                       \
    do {                                                                               \
        if (opt_ahead == 2) {                                                          \
            if (p.consumed) SPLAT_COMPOSITE_PX_LAUNCH2(EO, LIT, true, 2, DISC);        \
            else            SPLAT_COMPOSITE_PX_LAUNCH2(EO, LIT, false, 2, DISC);       \
        } else {                                                                       \
            if (p.consumed) SPLAT_COMPOSITE_PX_LAUNCH2(EO, LIT, true, 1, DISC);        \
            else            SPLAT_COMPOSITE_PX_LAUNCH2(EO, LIT, false, 1, DISC);       \
        }                                                                              \
    } while (0)
        if (p.disc) { // the oriented disc: per-lane queues as well, the footprint evaluated per queued pixel
            if (eo) SPLAT_COMPOSITE_PX_LAUNCH(true, false, true);
            else    SPLAT_COMPOSITE_PX_LAUNCH(false, false, true);
        } else if (lit32) {
            if (eo) SPLAT_COMPOSITE_PX_LAUNCH(true, true, false);
            else    SPLAT_COMPOSITE_PX_LAUNCH(false, true, false);
        } else {
            if (eo) SPLAT_COMPOSITE_PX_LAUNCH(true, false, false);
            else    SPLAT_COMPOSITE_PX_LAUNCH(false, false, false);
        }
#undef SPLAT_COMPOSITE_PX_LAUNCH2
#undef SPLAT_COMPOSITE_PX_LAUNCH
        *launched = hipPeekAtLastError() == hipSuccess; // (only a launch that went out carries the frame's report: composite_launch sends it otherwise)
        LAUNCH_CHECK(ctx, "k_composite_px");
        return SPLAT_OK;
    }
    timed = stage_event_pair(ctx, SPLAT_STAGE_COMPOSITE, &ev0, &ev1);
    if (p.disc) {
        if (eo) SPLAT_COMPOSITE_LAUNCH(SPLAT_COMPOSITE_FRONT_TO_BACK, true, true, false);
        else    SPLAT_COMPOSITE_LAUNCH(SPLAT_COMPOSITE_FRONT_TO_BACK, false, true, false);
    } else if (cfg->mode == SPLAT_COMPOSITE_FRONT_TO_BACK) {
        if (lit32) {
            if (eo) SPLAT_COMPOSITE_LAUNCH(SPLAT_COMPOSITE_FRONT_TO_BACK, true, false, true);
            else    SPLAT_COMPOSITE_LAUNCH(SPLAT_COMPOSITE_FRONT_TO_BACK, false, false, true);
        } else {
            if (eo) SPLAT_COMPOSITE_LAUNCH(SPLAT_COMPOSITE_FRONT_TO_BACK, true, false, false);
            else    SPLAT_COMPOSITE_LAUNCH(SPLAT_COMPOSITE_FRONT_TO_BACK, false, false, false);
        }
    } else {
        if (lit32) {
            if (eo) SPLAT_COMPOSITE_LAUNCH(SPLAT_COMPOSITE_REFERENCE_LITERAL, true, false, true);
            else    SPLAT_COMPOSITE_LAUNCH(SPLAT_COMPOSITE_REFERENCE_LITERAL, false, false, true);
        } else {
            if (eo) SPLAT_COMPOSITE_LAUNCH(SPLAT_COMPOSITE_REFERENCE_LITERAL, true, false, false);
            else    SPLAT_COMPOSITE_LAUNCH(SPLAT_COMPOSITE_REFERENCE_LITERAL, false, false, false);
        }
    }
#undef SPLAT_COMPOSITE_LAUNCH
    *launched = hipPeekAtLastError() == hipSuccess;
    LAUNCH_CHECK(ctx, "k_composite");
    return SPLAT_OK;
}
