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
    uint32_t disc_stride;                          // float4s between disc records: 2 (projector's) or 3 (48-byte exchange records)
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
                    const float4 c = p.prelit ? f_c : lit_color(f_c, f_n);
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
constexpr int PXC = 32;     // entries per chunk = bits of a lane's queue
constexpr int PX_ROW = 36;  // float2 slots per table row: 32 entries + 4 — rows 4 slots apart (mod 32) so that the eight
                            // rows of one entry fall into eight different bank pairs of a ds_read_b64
struct PxShared {
    float2 tx[8][PX_ROW]; // tx[c][j] = {gx_j(2c), gx_j(2c+1)}: entry j's factor at the tile's pixel columns 2c, 2c+1
    float2 ty[8][PX_ROW]; // ty[r][j] = {gy_j(2r), gy_j(2r+1)}
    float4 col[PXC];      // lit colour of entry j
};
static_assert(sizeof(PxShared) == 5120, "four waves x 5 KB: eight workgroups fill a CU's 160 KB");

template <bool EARLY_OUT, bool LIT32>
__global__ __launch_bounds__(256) void k_composite_px(CompositeParams p, uint32_t band_tiles) {
    __shared__ PxShared s_all[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t t_local = blockIdx.x * 4u + w;
    if (p.report && blockIdx.x == 0 && tid == 0) tile_report(p.frame_total, p.report, p.report_seq);
    if (t_local >= band_tiles) return; // (whole waves: nothing in this kernel waits for another wave)
    PxShared &sh = s_all[w];
    const uint32_t tx = t_local % p.ntx, ty = t_local / p.ntx + p.tile_row0;
    const uint32_t tile_idx = ty * p.ntx + tx; // ComputeShaderRenderer.ts:161-163
    const uint32_t count = p.counts[tile_idx], off = p.offsets[tile_idx];
    const uint32_t bx = lane & 7, by = lane >> 3; // this lane's 2x2 block: pixel columns 2bx, 2bx+1, rows 2by, 2by+1 of the tile
    const uint32_t px0 = tx * CT + 2 * bx, py0 = ty * CT + 2 * by;
    const bool okx0 = px0 < p.width, okx1 = px0 + 1 < p.width, oky0 = py0 < p.height, oky1 = py0 + 1 < p.height;
    const float tile_x0 = (float)(tx * CT), tile_y0 = (float)(ty * CT);
    const float tile_cx = tile_x0 + 0.5f, tile_cy = tile_y0 + 0.5f; // :169 pixel centres

    // pixel (row k, column i) of the block: component i of the k-th pair.  T = transmittance while the pixel accumulates,
    // 0 once it has stopped (its background term is then already in C) and for pixels outside the image
    v2f cr[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, cg[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, cb[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    v2f T[2] = {{(okx0 && oky0) ? 1.0f : 0.0f, (okx1 && oky0) ? 1.0f : 0.0f}, {(okx0 && oky1) ? 1.0f : 0.0f, (okx1 && oky1) ? 1.0f : 0.0f}};
    // wave-uniform masks of the lanes whose pixel (k, i) still accumulates
    unsigned long long al00 = uniform64(__ballot(okx0 && oky0)), al01 = uniform64(__ballot(okx1 && oky0));
    unsigned long long al10 = uniform64(__ballot(okx0 && oky1)), al11 = uniform64(__ballot(okx1 && oky1));
    uint32_t stop_pos = 0; // the largest list position (+1) at which one of this lane's pixels stopped
    uint32_t staged = 0;

    const uint32_t e = lane & 31, h = lane >> 5; // table building: lane (e, h) computes entry e's x (h = 0) or y (h = 1) table
    constexpr uint32_t NONE = 0xffffffffu;
    // entries are fetched a chunk ahead of their use and their indices two chunks ahead (the gather depends on the index)
    uint32_t idx_cur = e < count ? p.indices[off + e] : NONE;
    uint32_t idx_nxt = PXC + e < count ? p.indices[off + PXC + e] : NONE;
    float4 c_b = make_float4(0, 0, 0, 0), c_b2 = c_b, c_c = c_b, c_n = c_b;
    float c_r = 0.0f;
    if (idx_cur != NONE) fetch_entry<SPLAT_COMPOSITE_FRONT_TO_BACK, EARLY_OUT, false, LIT32>(p, idx_cur, c_b, c_b2, c_c, c_n, c_r);

    for (uint32_t cb0 = 0; cb0 < count; cb0 += PXC) {
        if (EARLY_OUT && (al00 | al01 | al10 | al11) == 0) break; // every pixel of the tile has stopped
        staged = min(cb0 + 2 * PXC, count); // entries gathered so far: this chunk and the one fetched ahead
        // ---- next chunk's gathers and the index load behind them go out before this chunk is touched
        float4 n_b = make_float4(0, 0, 0, 0), n_b2 = n_b, n_c = n_b, n_n = n_b;
        float n_r = 0.0f;
        if (idx_nxt != NONE) fetch_entry<SPLAT_COMPOSITE_FRONT_TO_BACK, EARLY_OUT, false, LIT32>(p, idx_nxt, n_b, n_b2, n_c, n_n, n_r);
        const uint32_t idx_nn = cb0 + 2 * PXC + e < count ? p.indices[off + cb0 + 2 * PXC + e] : NONE;

        // ---- stage: both halves of the wave hold entry e; half h builds the table of its axis
        float k = 0.0f, ck = 0.0f;
        uint32_t m16 = 0; // this axis's mask of covered pixel columns / rows (zero unless the entry draws something in this tile)
        if (idx_cur != NONE && !(c_r < 0.5f)) { // :127-129 "too small"
            const float4 b = c_b;
            const uint32_t xm = span_mask16(b.x, b.z, tile_cx), ym = span_mask16(b.y, b.w, tile_cy);
            if (xm != 0 && ym != 0) {
                // gaussian = exp(-0.5 (dist / r)^2 / 0.25) = exp2(-((dx k)^2 + (dy k)^2)), k = sqrt(2 log2 e) / r (:133-140),
                // in tile-local coordinates (as k_composite)
                k = 1.6986436005760381f / c_r;
                const float lx = (b.x + b.z) * 0.5f - tile_x0, ly = (b.y + b.w) * 0.5f - tile_y0; // :124, then exact
                ck = (h ? ly : lx) * k;
                m16 = h ? ym : xm;
            }
        }
        if (h == 0) {
            const float4 c = (LIT32 || p.prelit) ? c_c : lit_color(c_c, c_n);
            sh.col[e] = make_float4(c.x, c.y, c.z, 1.0f); // (.w = 1: read back as the factor of T's update, which makes the read one ds_read_b128)
        }
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
            const v2f pc = {(float)(2 * c2) + 0.5f, (float)(2 * c2) + 1.5f};
            const v2f t = pc * (v2f){k, k} - (v2f){ck, ck};
            const v2f q = t * t;
            // (the box test as a bit mask on the value: v_bfe_i32 spreads the coverage bit over the word)
            const uint32_t g0 = __float_as_uint(__builtin_amdgcn_exp2f(-q.x)) & (uint32_t)__builtin_amdgcn_sbfe((int)m16, 2 * c2, 1);
            const uint32_t g1 = __float_as_uint(__builtin_amdgcn_exp2f(-q.y)) & (uint32_t)__builtin_amdgcn_sbfe((int)m16, 2 * c2 + 1, 1);
            (h ? sh.ty : sh.tx)[c2][e] = make_float2(__uint_as_float(g0), __uint_as_float(g1));
        }
        // ---- queues: bit j of X[c] = entry j's box meets pixel columns 2c, 2c+1; of Y[r] likewise for rows.  One ballot
        // gives X[c] (lanes 0..31 test the x mask) and Y[c] (lanes 32..63 test the y mask) together.
        const uint32_t mm = m16 | (m16 >> 1);
        uint32_t qx = 0, qy = 0;
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
            const unsigned long long bal = __ballot((mm >> (2 * c2)) & 1u);
            qx = __builtin_amdgcn_inverse_ballot_w64(0x0101010101010101ull << c2) ? (uint32_t)bal : qx;          // lanes with bx == c2
            qy = __builtin_amdgcn_inverse_ballot_w64(0xffull << (8 * c2)) ? (uint32_t)(bal >> 32) : qy;           // lanes with by == c2
        }
        uint32_t mine = qx & qy;
        if (EARLY_OUT && !__builtin_amdgcn_inverse_ballot_w64(al00 | al01 | al10 | al11)) mine = 0; // this lane's block has stopped
        // the wave's own LDS stores are read back by other lanes: DS operations of one wave execute in order; the compiler
        // must not move the reads up
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- trips
        for (;;) {
            const bool act = mine != 0;
            if (!__builtin_amdgcn_readfirstlane((int)(__ballot(act) != 0))) break;
            uint32_t j = 0;
            if (act) {
                j = (uint32_t)__builtin_ctz(mine);
                mine &= mine - 1;
                const float2 gx = sh.tx[bx][j], gy = sh.ty[by][j];
                const float4 c = sh.col[j];
                const v2f gxx = {gx.x, gx.y};
                const v2f g0 = gxx * (v2f){gy.x, gy.x}, g1 = gxx * (v2f){gy.y, gy.y}; // rows 2by, 2by+1
                const v2f w0 = T[0] * g0, w1 = T[1] * g1;                                // SURVEY §8a contract 3: nearest on top
                cr[0] += (v2f){c.x, c.x} * w0; cr[1] += (v2f){c.x, c.x} * w1;
                cg[0] += (v2f){c.y, c.y} * w0; cg[1] += (v2f){c.y, c.y} * w1;
                cb[0] += (v2f){c.z, c.z} * w0; cb[1] += (v2f){c.z, c.z} * w1;
                // T (1 - g) with the product in hand; c.w is 1.0 (exact: the same bits as T - w) — its use keeps the colour
                // read a single 16-byte ds_read_b128 (4 LDS cycles; the compiler would shrink it to a ds_read_b96: 8)
                T[0] -= (v2f){c.w, c.w} * w0; T[1] -= (v2f){c.w, c.w} * w1;
            }
            if (EARLY_OUT) {
                // pixels that have just reached alpha >= 0.99 (:187-190): T <= T_STOP on a pixel that was accumulating (a stopped
                // pixel carries T = 0 and is masked by its `al` bit)
                // (an accumulating pixel has T > T_STOP until the trip that stops it, so no test of `act` is needed)
                const unsigned long long f00 = uniform64(__ballot(T[0].x <= T_STOP) & al00);
                const unsigned long long f01 = uniform64(__ballot(T[0].y <= T_STOP) & al01);
                const unsigned long long f10 = uniform64(__ballot(T[1].x <= T_STOP) & al10);
                const unsigned long long f11 = uniform64(__ballot(T[1].y <= T_STOP) & al11);
                if ((f00 | f01 | f10 | f11) != 0) {
                    // the background term of :193-195 now, T = 0 from here on
                    if (__builtin_amdgcn_inverse_ballot_w64(f00)) { cr[0].x += 0.05f * T[0].x; cg[0].x += 0.05f * T[0].x; cb[0].x += 0.1f * T[0].x; T[0].x = 0.0f; }
                    if (__builtin_amdgcn_inverse_ballot_w64(f01)) { cr[0].y += 0.05f * T[0].y; cg[0].y += 0.05f * T[0].y; cb[0].y += 0.1f * T[0].y; T[0].y = 0.0f; }
                    if (__builtin_amdgcn_inverse_ballot_w64(f10)) { cr[1].x += 0.05f * T[1].x; cg[1].x += 0.05f * T[1].x; cb[1].x += 0.1f * T[1].x; T[1].x = 0.0f; }
                    if (__builtin_amdgcn_inverse_ballot_w64(f11)) { cr[1].y += 0.05f * T[1].y; cg[1].y += 0.05f * T[1].y; cb[1].y += 0.1f * T[1].y; T[1].y = 0.0f; }
                    if (__builtin_amdgcn_inverse_ballot_w64(f00 | f01 | f10 | f11)) stop_pos = max(stop_pos, cb0 + j + 1);
                    al00 &= ~f00; al01 &= ~f01; al10 &= ~f10; al11 &= ~f11;
                    if (!__builtin_amdgcn_inverse_ballot_w64(al00 | al01 | al10 | al11)) mine = 0;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the next chunk's table stores stay behind this chunk's reads
        __builtin_amdgcn_wave_barrier();
        idx_cur = idx_nxt; idx_nxt = idx_nn;
        c_b = n_b; c_b2 = n_b2; c_c = n_c; c_n = n_n; c_r = n_r;
    }

    if (p.consumed) { // (uniform branch; timed / diagnostic runs only)
        // entries this tile needed: the position at which its last pixel stopped, or the whole list if one never did
        // (SURVEY §8d's P_used per tile = count with early-out off)
        uint32_t used = stop_pos;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) used = max(used, (uint32_t)__shfl_xor((int)used, d));
        if (!EARLY_OUT || (al00 | al01 | al10 | al11) != 0) used = count;
        if (lane == 0 && staged) {
            p.consumed[(size_t)tile_idx * 2] += (unsigned long long)staged;
            p.consumed[(size_t)tile_idx * 2 + 1] += (unsigned long long)used;
        }
    }

    // :193-197 (a stopped pixel has T = 0 here: its background term went in when it stopped)
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
    const bool lit32 = cfg->record_format == SPLAT_RECORDS_LIT32; // the records carry the lit colour: no colour / normal arrays
    ARG_CHECK(ctx, lit32 || (color_opacity && (normals || cfg->prelit)));
    ARG_CHECK(ctx, projected && tile_indices && tile_counts && tile_offsets);
    ARG_CHECK(ctx, color_stride_vec4 >= 1 && normal_stride_vec4 >= 1);
    ARG_CHECK(ctx, out_rgba8 || out_rgba32f);
    ARG_CHECK(ctx, (((uintptr_t)color_opacity | (uintptr_t)normals | (uintptr_t)projected | (uintptr_t)out_rgba32f) & 15) == 0);
    ARG_CHECK(ctx, cfg->prelit <= 1 && cfg->footprint <= SPLAT_FOOTPRINT_DISC);
    // the oriented disc is SequentialRenderer's footprint: nearest-on-top "over" is its only blend, and its
    // records are the projector's 32-byte disc records
    ARG_CHECK(ctx, cfg->footprint != SPLAT_FOOTPRINT_DISC ||
                       (cfg->mode == SPLAT_COMPOSITE_FRONT_TO_BACK && cfg->record_format != SPLAT_RECORDS_COMPACT && !lit32));
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
    p.disc_stride = cfg->record_format == SPLAT_RECORDS_DISC48 ? 3u : 2u;
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
    *launched = true;
    dim3 grid(ntx, r1 - r0), block(256);
    const bool eo = cfg->early_out != 0;
    // timed runs attach the event pair to the launch itself (no marker packets around the kernel)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const bool timed = stage_event_pair(ctx, SPLAT_STAGE_COMPOSITE, &ev0, &ev1);
#define SPLAT_COMPOSITE_LAUNCH(MODE, EO, DISC, LIT)                                                                       \
    do {                                                                                                                  \
        if (timed) hipExtLaunchKernelGGL((k_composite<MODE, EO, DISC, LIT>), grid, block, 0, ctx->stream, ev0, ev1, 0, p); \
        else hipLaunchKernelGGL((k_composite<MODE, EO, DISC, LIT>), grid, block, 0, ctx->stream, p);                     \
    } while (0)
    // Which kernel composites the isotropic footprint nearest-on-top (the frame's default): k_composite_px — one wave per
    // tile, every lane walking its own entries — unless SPLAT_COMPOSITE=quadrant asks for round 2's k_composite (one
    // 8x8-pixel wave per quadrant visiting every entry of the tile), which also serves the oriented disc (its footprint is
    // not separable) and the reference-literal blend.
    static int s_px = -1;
    if (s_px < 0) {
        const char *e = getenv("SPLAT_COMPOSITE");
        s_px = (e && (e[0] == 'p' || e[0] == 'P')) ? 1 : 0; // (opt-in until it measures faster: SPLAT_COMPOSITE=pixel)
    }
    if (s_px && !p.disc && cfg->mode == SPLAT_COMPOSITE_FRONT_TO_BACK) {
        const uint32_t band_tiles = ntx * (r1 - r0);
        const dim3 pgrid(div_up(band_tiles, 4));
#define SPLAT_COMPOSITE_PX_LAUNCH(EO, LIT)                                                                                   \
    do {                                                                                                                     \
        if (timed) hipExtLaunchKernelGGL((k_composite_px<EO, LIT>), pgrid, block, 0, ctx->stream, ev0, ev1, 0, p, band_tiles); \
        else hipLaunchKernelGGL((k_composite_px<EO, LIT>), pgrid, block, 0, ctx->stream, p, band_tiles);                     \
    } while (0)
        if (lit32) {
            if (eo) SPLAT_COMPOSITE_PX_LAUNCH(true, true);
            else    SPLAT_COMPOSITE_PX_LAUNCH(false, true);
        } else {
            if (eo) SPLAT_COMPOSITE_PX_LAUNCH(true, false);
            else    SPLAT_COMPOSITE_PX_LAUNCH(false, false);
        }
#undef SPLAT_COMPOSITE_PX_LAUNCH
        LAUNCH_CHECK(ctx, "k_composite_px");
        return SPLAT_OK;
    }
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
    LAUNCH_CHECK(ctx, "k_composite");
    return SPLAT_OK;
}
