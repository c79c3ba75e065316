// project.hip — SplatProjector + DepthKeyExtractor + SplatPropertyManager update as CDNA4 kernels.
//
// Reference: /root/reference/src/SplatProjector.ts:64-132 (K1), src/shaders/extract-depth-keys.wgsl:37-63
// (K2), src/SplatPropertyManager.ts:82-107 (K12).
//
// Roofline: HBM.  Algorithmic bytes per splat: 16 (pos,radius) in + 32 (ProjectedSplat) + 4 (key)
// + 4 (payload) out = 56 B.  One lane per splat, 16-byte vector load, two 16-byte stores per lane
// that together fill a 32-byte record, 4-byte coalesced key/payload stores.
//
// This file is compiled with -ffp-contract=off: every float op below is one IEEE binary32 op in
// the order written, identical to oracle/oracle.c, so records and keys are bit-exact.
#include "common.h"
#include "tile_range.h"
#include "disc.h"
#include "shade.h"

struct FrameUniforms {
    float m[16];   // VP, column-major
    float eye[3];
    float time;
    float w, h;
};

__device__ __forceinline__ void to_screen(const FrameUniforms &u, float x, float y, float z, float &sx, float &sy) {
    float cx = ((u.m[0] * x + u.m[4] * y) + u.m[8] * z) + u.m[12];
    float cy = ((u.m[1] * x + u.m[5] * y) + u.m[9] * z) + u.m[13];
    float cw = ((u.m[3] * x + u.m[7] * y) + u.m[11] * z) + u.m[15];
    float nx = cx / cw, ny = cy / cw;
    sx = ((nx + 1.0f) * 0.5f) * u.w;
    sy = ((1.0f - ny) * 0.5f) * u.h;
}

__device__ __forceinline__ uint32_t depth_key(float depth) {
    uint32_t bits = __float_as_uint(depth);
    uint32_t mask = ((bits >> 31) == 1u) ? 0xffffffffu : 0x80000000u; // extract-depth-keys.wgsl:57-58
    return bits ^ mask;
}

// The projector up to the point where the record is formed: screen centre, screen radius, depth.
__device__ __forceinline__ float4 project_centre(const FrameUniforms &u, float4 pr) {
    float x = pr.x, y = pr.y, z = pr.z, radius = pr.w;
    float dx = x - u.eye[0], dy = y - u.eye[1], dz = z - u.eye[2];
    float depth = sqrtf((dx * dx + dy * dy) + dz * dz); // SplatProjector.ts:77
    float scx, scy;
    to_screen(u, x, y, z, scx, scy);
    // max_k sqrt(d2_k) == sqrt(max_k d2_k) bit for bit: a correctly rounded square root is monotone, d2 >= +0,
    // and fmaxf drops a NaN operand either way — one square root per splat instead of six (the kernel is
    // VALU-bound: SQ counters show the vector ALUs 89 % busy, IEEE divides and square roots being most of it)
    float max_d2 = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) { // :93-113, same offset order as the shader
        float ox = (k == 0) ? radius : (k == 1) ? -radius : 0.0f;
        float oy = (k == 2) ? radius : (k == 3) ? -radius : 0.0f;
        float oz = (k == 4) ? radius : (k == 5) ? -radius : 0.0f;
        float sx, sy;
        to_screen(u, x + ox, y + oy, z + oz, sx, sy);
        float ex = scx - sx, ey = scy - sy;
        max_d2 = fmaxf(max_d2, ex * ex + ey * ey);
    }
    const float max_r = sqrtf(max_d2);
    return make_float4(scx, scy, max_r, depth);
}

// The oriented-disc projector's second input and output (disc.h): normals in, 32-byte disc records out.
struct DiscIO {
    const float4 *normals;
    uint32_t normal_stride;
    float4 *discs;
    uint32_t disc_stride; // float4s per record: 2, or 3 when the frame asks for LIT disc records (the splat's lit colour behind the record)
};

// Band frames without an exchange (every rank projects all splats): most splats cannot reach the rank's
// tile rows, and the six offset projections that give the screen radius (12 IEEE divides) are what the projector
// costs.  This is a cheap, provably conservative test: with a_i = radius * VP[:, i] the screen displacement of the
// offset along axis i is W/2 * (a_x - ndc_x * a_w) / (c_w + a_w) (and likewise in y), so
//     |dx| <= W/2 * (max_i |a_x| + |ndc_x| * max_i |a_w|) / (c_w - max_i |a_w|)
// bounds the radius, hence the padded box; 0.1 % and one pixel of slack cover the rounding of the bound itself, and
// two pixels more on either side the rounding of the centre (below).  A splat it rejects has an empty clamped
// tile range in the exact path too (its record is not written: no list of the band can contain it); anything
// doubtful (w <= 0, NaN) goes through the exact path.
// BALL (the oriented disc): the offsets are not along the axes but anywhere in a ball of radius `r` (the disc p +
// r (t u + b v), |t| = 1, |b| = |n|, lies in the ball of radius r * max(1, |n|)), so the row NORMS of VP bound the
// clip-space displacement instead of the largest entry of each row.
template <bool BALL>
__device__ __forceinline__ bool cannot_reach_band(const FrameUniforms &u, float4 pr, float r, const BinParams &bp) {
    const float *m = u.m;
    const float cx = ((m[0] * pr.x + m[4] * pr.y) + m[8] * pr.z) + m[12];
    const float cy = ((m[1] * pr.x + m[5] * pr.y) + m[9] * pr.z) + m[13];
    const float cw = ((m[3] * pr.x + m[7] * pr.y) + m[11] * pr.z) + m[15];
    const float ax = r * (BALL ? 1.001f * __builtin_amdgcn_sqrtf((m[0] * m[0] + m[4] * m[4]) + m[8] * m[8])
                               : fmaxf(fmaxf(fabsf(m[0]), fabsf(m[4])), fabsf(m[8])));
    const float ay = r * (BALL ? 1.001f * __builtin_amdgcn_sqrtf((m[1] * m[1] + m[5] * m[5]) + m[9] * m[9])
                               : fmaxf(fmaxf(fabsf(m[1]), fabsf(m[5])), fabsf(m[9])));
    const float aw = r * (BALL ? 1.001f * __builtin_amdgcn_sqrtf((m[3] * m[3] + m[7] * m[7]) + m[11] * m[11])
                               : fmaxf(fmaxf(fabsf(m[3]), fabsf(m[7])), fabsf(m[11])));
    const float den = cw - aw;
    if (!(cw > 0.0f) || !(den > 0.0f)) return false;
    // (hardware reciprocal and square root, 1 ulp: the slack below is a thousand times that, and this test runs for
    // every splat of the scene on every rank)
    const float icw = __builtin_amdgcn_rcpf(cw), iden = __builtin_amdgcn_rcpf(den);
    const float ndx = cx * icw, ndy = cy * icw;
    const float bx = (0.5f * u.w) * (ax + fabsf(ndx) * aw) * iden, by = (0.5f * u.h) * (ay + fabsf(ndy) * aw) * iden;
    const float reach = __builtin_amdgcn_sqrtf(bx * bx + by * by) * (1.5f * 1.001f) + 1.0f; // >= the padded radius of SplatProjector.ts:119
    const float scy = ((1.0f - ndy) * 0.5f) * u.h;
    const float ts = (float)bp.tile;
    // The exact path bins the splat into the band iff max_y >= 16 row0 and min_y < 16 row1 (tile_range: floor, clamp): with reach
    // >= the padded radius + 1 px and this centre within 1e-3 px of the exact one, two more pixels on either side are slack
    // enough — a whole tile row (rounds 2-4) let 40 % more splats through to the exact projection than the band keeps.
    return (scy + reach < (float)bp.row0 * ts - 2.0f) || (scy - reach > (float)bp.row1 * ts + 2.0f); // (NaN: false)
}

// One splat: record, key, payload, packed tile range.  Returns the packed range (1 = empty).
// DISC: the footprint is SequentialRenderer's oriented disc — the ProjectedSplat's bounds are the disc's exact
// screen extent, screenRadius half the larger one, and the disc record goes to dio.discs.
// LIT (isotropic frames): the 32-byte composite record {centre, radius, depth | lit colour} goes to lio.records
// (shade.h) — everything the composite reads of a splat, in one line; the ProjectedSplat is then optional.
// Everything the projector reads of one splat.  Kernels that project several splats per thread load all of them
// before the first one's arithmetic (fourteen IEEE divides and two square roots deep) and stores: left to the
// compiler the loads stay behind the previous splat's stores — the output pointers may alias the inputs for all it
// knows — and every wave sits out two memory round trips per splat.
struct SplatIn {
    float4 pr, col, nrm;
};
template <bool DISC, bool LIT>
__device__ __forceinline__ SplatIn load_splat(const float4 *__restrict__ pos_radius, uint32_t stride_vec4, uint32_t i, const DiscIO &dio,
                                              const LitIO &lio) {
    SplatIn s;
    s.pr = pos_radius[(size_t)i * stride_vec4];
    s.col = s.nrm = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (DISC) s.nrm = dio.normals[(size_t)i * dio.normal_stride];
    if (LIT) {
        s.col = lio.color[(size_t)i * lio.color_stride];
        if (!lio.prelit) s.nrm = lio.normals[(size_t)i * lio.normal_stride];
    }
    return s;
}

template <bool WITH_KEYS, bool WITH_RANGE, bool DISC, bool LIT = false>
__device__ __forceinline__ uint32_t project_one(const FrameUniforms &u, const SplatIn &in, uint32_t i, uint32_t index_base,
                                                float4 *__restrict__ projected, uint32_t *__restrict__ keys,
                                                uint32_t *__restrict__ payload, uint32_t *__restrict__ range32, const BinParams &bp,
                                                const DiscIO &dio, const LitIO &lio = LitIO{}, uint32_t kslot = 0xffffffffu) {
    const uint32_t ks = kslot == 0xffffffffu ? i : kslot; // where the key and the packed range go (k_project_hist_bandc: a compacted slot)
    float4 a, b;
    float depth;
    if (DISC) {
        const float4 pr = in.pr;
        const DiscRecord d = disc_record(u.m, u.w, u.h, pr, in.nrm);
        dio.discs[(size_t)i * dio.disc_stride] = d.a;
        dio.discs[(size_t)i * dio.disc_stride + 1] = d.b;
        // LIT: the lit colour rides behind the record (48 bytes per splat): the disc frame's composite then gathers this one
        // record per staged list entry instead of record + colour + normal — the gathers are what a staged entry costs
        // (C2: 168 -> 124 us for one line less, profiles/r04_d_disc_composite_C2.txt)
        if (LIT) dio.discs[(size_t)i * dio.disc_stride + 2] = lio.prelit ? in.col : lit_color(in.col, in.nrm);
        disc_bounds(d, a);
        const float dx = pr.x - u.eye[0], dy = pr.y - u.eye[1], dz = pr.z - u.eye[2];
        depth = sqrtf((dx * dx + dy * dy) + dz * dz); // SplatProjector.ts:77: the sort key does not depend on the footprint
        b = make_float4(depth, 0.5f * fmaxf(a.z - a.x, a.w - a.y), __uint_as_float(index_base + i), 0.0f);
    } else {
        const float4 c = project_centre(u, in.pr);
        const float scx = c.x, scy = c.y, max_r = c.z;
        depth = c.w;
        float padded = max_r * 1.5f; // :119
        a = make_float4(scx - padded, scy - padded, scx + padded, scy + padded);
        b = make_float4(depth, max_r, __uint_as_float(index_base + i), 0.0f); // :128 originalIndex
        if (LIT) {
            lio.records[(size_t)i * 2] = c;
            lio.records[(size_t)i * 2 + 1] = lio.prelit ? in.col : lit_color(in.col, in.nrm);
        }
    }
    if ((!DISC && !LIT) || projected) { // (a disc or lit frame's composite reads its own records: the ProjectedSplat is optional there)
        projected[(size_t)i * 2] = a;
        projected[(size_t)i * 2 + 1] = b;
    }
    if (WITH_KEYS) {
        keys[ks] = depth_key(depth);
        if (payload) payload[ks] = index_base + i; // (frame path: the sort's first pass synthesises it)
    }
    uint32_t packed = 1u;
    if (WITH_RANGE) { // the binner's clamped tile range while the bounds are still in registers
        uint32_t tx0, tx1, ty0, ty1;
        const bool ok = tile_range(a, bp.width, bp.height, bp.tile, bp.ntx, bp.nty, bp.row0, bp.row1, tx0, tx1, ty0, ty1);
        packed = pack_range32(ok, tx0, tx1, ty0, ty1);
        range32[ks] = packed;
    }
    return packed;
}

// Multi-GPU exchange records (no reference counterpart): 16 bytes per splat, float4 {screen
// centre x, y, screen radius, depth}.  The 32-byte ProjectedSplat is a pure function of it (bounds =
// centre -/+ radius * 1.5 in this file's operation order, originalIndex = position in the gathered
// array), so the ranks exchange half the bytes and rebuild records bit-exactly where they need them.
__global__ __launch_bounds__(256) void k_project_compact(FrameUniforms u, const float4 *__restrict__ pos_radius, uint32_t stride_vec4,
                                                         uint32_t n, float4 *__restrict__ records16) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) records16[i] = project_centre(u, pos_radius[(size_t)i * stride_vec4]);
}

// The same for the oriented-disc footprint: 48 bytes per splat {disc record, depth, 0, 0, 0} (the disc's bounds are a
// pure function of its record: disc_bounds).
__global__ __launch_bounds__(256) void k_project_disc48(FrameUniforms u, const float4 *__restrict__ pos_radius, uint32_t stride_vec4,
                                                        const float4 *__restrict__ normals, uint32_t normal_stride, uint32_t n,
                                                        float4 *__restrict__ records48) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 pr = pos_radius[(size_t)i * stride_vec4];
    const DiscRecord d = disc_record(u.m, u.w, u.h, pr, normals[(size_t)i * normal_stride]);
    const float dx = pr.x - u.eye[0], dy = pr.y - u.eye[1], dz = pr.z - u.eye[2];
    records48[(size_t)i * 3] = d.a;
    records48[(size_t)i * 3 + 1] = d.b;
    records48[(size_t)i * 3 + 2] = make_float4(sqrtf((dx * dx + dy * dy) + dz * dz), 0.0f, 0.0f, 0.0f); // SplatProjector.ts:77
}

__global__ __launch_bounds__(256) void k_expand_compact(const float4 *__restrict__ records16, uint32_t n, uint32_t index_base,
                                                        float4 *__restrict__ projected) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 c = records16[i];
    const float padded = c.z * 1.5f;
    projected[(size_t)i * 2] = make_float4(c.x - padded, c.y - padded, c.x + padded, c.y + padded);
    projected[(size_t)i * 2 + 1] = make_float4(c.w, c.z, __uint_as_float(index_base + i), 0.0f);
}

template <bool WITH_KEYS, bool WITH_RANGE, bool DISC, bool LIT = false>
__global__ __launch_bounds__(256) void k_project(FrameUniforms u, const float4 *__restrict__ pos_radius,
                                                 uint32_t stride_vec4, uint32_t n, uint32_t n_padded, uint32_t index_base,
                                                 float4 *__restrict__ projected, uint32_t *__restrict__ keys,
                                                 uint32_t *__restrict__ payload, uint32_t *__restrict__ range32, BinParams bp,
                                                 DiscIO dio, LitIO lio) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) {
        if (WITH_KEYS && i < n_padded) { // extract-depth-keys.wgsl:46-50
            keys[i] = 0xffffffffu;
            if (payload) payload[i] = 0xffffffffu;
        }
        return;
    }
    const SplatIn in = load_splat<DISC, LIT>(pos_radius, stride_vec4, i, dio, lio);
    project_one<WITH_KEYS, WITH_RANGE, DISC, LIT>(u, in, i, index_base, projected, keys, payload, range32, bp, dio, lio);
}

// Tile-first frame path: 1024 splats per workgroup (the binner's block), and while each splat's tile
// rectangle is in registers the block's pairs are counted per low tile-id digit — the histogram the
// first pass of the tile-id sort needs (tile_first.hip; k_band_prepare_tf in frame.hip does the same for
// the gathered records of a multi-GPU band).  The kernel is HBM-bound; the LDS counting hides under the stores.
template <bool DISC, bool LIT, uint32_t PER, uint32_t AHEAD = 1>
__global__ __launch_bounds__(256) void k_project_hist(FrameUniforms u, const float4 *__restrict__ pos_radius, uint32_t stride_vec4,
                                                      uint32_t n, uint32_t n_padded, float4 *__restrict__ projected,
                                                      uint32_t *__restrict__ keys, uint32_t *__restrict__ range32, BinParams bp,
                                                      TfHistOut ho, DiscIO dio, LitIO lio) {
    __shared__ uint32_t lh[4][256];
    __shared__ uint32_t wsum[4];
    const uint32_t tid = threadIdx.x, w = tid >> 6;
    const uint32_t blk = xcd_block_of(blockIdx.x, ho.xcd_per); // (blocks past the grid's last: nothing to project, no key to pad)
    if (blk * (PER * 256u) >= n_padded) return;
    SplatIn in[PER]; // (unrolled: a splat's registers are live from its load to its last use only)
#pragma unroll
    for (uint32_t k = 0; k < AHEAD; ++k) { // AHEAD splats' loads in flight before the first dependent instruction
        const uint32_t i = blk * (PER * 256u) + k * 256u + tid;
        if (i < n) in[k] = load_splat<DISC, LIT>(pos_radius, stride_vec4, i, dio, lio);
    }
    if (blk == 0 && tid == 0) *ho.overflow_flag = 0;
    for (uint32_t j = tid; j < 4 * 256; j += 256) (&lh[0][0])[j] = 0;
    __syncthreads();
    uint32_t local = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) { // (PER * 256 splats per workgroup: the binner's block)
        if (k + AHEAD < PER) { // ... and AHEAD of them in flight from then on
            const uint32_t i = blk * (PER * 256u) + (k + AHEAD) * 256u + tid;
            if (i < n) in[k + AHEAD] = load_splat<DISC, LIT>(pos_radius, stride_vec4, i, dio, lio);
        }
        const uint32_t i = blk * (PER * 256u) + k * 256u + tid;
        if (i >= n) {
            if (i < n_padded) keys[i] = 0xffffffffu;
            continue;
        }
        const uint32_t r = project_one<true, true, DISC, LIT>(u, in[k], i, 0, projected, keys, nullptr, range32, bp, dio, lio);
        const uint32_t tx0 = r & 0xffu, tx1 = (r >> 8) & 0xffu, ty0 = (r >> 16) & 0xffu, ty1 = r >> 24;
        if (tx0 > tx1 || ty0 > ty1) continue;
        local += hist_add_rect(lh[w], tx0, tx1, ty0, ty1, bp.ntx, ho.mask);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d);
    if ((tid & 63) == 0) wsum[w] = local;
    __syncthreads();
    if (blk < ho.num_parts) { // (blocks that only pad keys past n have no histogram column)
        if (tid <= ho.mask) ho.hist[(size_t)tid * ho.num_parts + blk] = lh[0][tid] + lh[1][tid] + lh[2][tid] + lh[3][tid];
        if (tid == 0) ho.blocksums[blk] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

// The same kernel for a strict band of tile rows (bp.skip_outside): a cheap conservative test first, for all 1024
// splats of the block; the survivors — about 1/G of them, scattered at random over the lanes — are compacted through
// LDS so that the full projection runs on dense waves (left in place, every wave would still execute it for its few
// surviving lanes).  Same records, keys, ranges and histogram for every splat that can reach the band; the others get
// an all-ones key, an empty range and no record.
template <bool DISC, bool LIT>
__global__ __launch_bounds__(256) void k_project_hist_band(FrameUniforms u, const float4 *__restrict__ pos_radius, uint32_t stride_vec4,
                                                           uint32_t n, uint32_t n_padded, float4 *__restrict__ projected,
                                                           uint32_t *__restrict__ keys, uint32_t *__restrict__ range32, BinParams bp,
                                                           TfHistOut ho, DiscIO dio, LitIO lio) {
    __shared__ uint32_t lh[4][256];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t s_list[1024];
    __shared__ uint32_t s_count;
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    if (blockIdx.x == 0 && tid == 0) *ho.overflow_flag = 0;
    if (tid == 0) s_count = 0;
    for (uint32_t j = tid; j < 4 * 256; j += 256) (&lh[0][0])[j] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t i = blockIdx.x * 1024u + k * 256u + tid;
        bool keep = false;
        if (i < n) {
            const float4 pr = pos_radius[(size_t)i * stride_vec4];
            float r = fabsf(pr.w);
            if (DISC) {
                const float4 nr = dio.normals[(size_t)i * dio.normal_stride];
                r *= fmaxf(1.0f, 1.001f * __builtin_amdgcn_sqrtf((nr.x * nr.x + nr.y * nr.y) + nr.z * nr.z));
            }
            keep = !cannot_reach_band<DISC>(u, pr, r, bp); // (NaN radius or normal: kept, the exact path decides)
            if (!keep) {
                keys[i] = 0xffffffffu; // (never read: a splat without pairs contributes no key; written so that the array is defined)
                range32[i] = 1u;       // pack_range32's empty range
            }
        } else if (i < n_padded) {
            keys[i] = 0xffffffffu;
        }
        const unsigned long long m = __ballot(keep);
        uint32_t base = 0;
        if (lane == 0 && m) base = atomicAdd(&s_count, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, 0);
        if (keep) s_list[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = i;
    }
    __syncthreads();
    const uint32_t kept = s_count;
    uint32_t local = 0;
    for (uint32_t j = tid; j < kept; j += 256) {
        const uint32_t i = s_list[j];
        const SplatIn in = load_splat<DISC, LIT>(pos_radius, stride_vec4, i, dio, lio);
        const uint32_t r = project_one<true, true, DISC, LIT>(u, in, i, 0, projected, keys, nullptr, range32, bp, dio, lio);
        const uint32_t tx0 = r & 0xffu, tx1 = (r >> 8) & 0xffu, ty0 = (r >> 16) & 0xffu, ty1 = r >> 24;
        if (tx0 > tx1 || ty0 > ty1) continue;
        local += hist_add_rect(lh[w], tx0, tx1, ty0, ty1, bp.ntx, ho.mask);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d);
    if (lane == 0) wsum[w] = local;
    __syncthreads();
    if (blockIdx.x < ho.num_parts) {
        if (tid <= ho.mask) ho.hist[(size_t)tid * ho.num_parts + blockIdx.x] = lh[0][tid] + lh[1][tid] + lh[2][tid] + lh[3][tid];
        if (tid == 0) ho.blocksums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

// The strict band's projector for bands of a fraction of the screen (the exchange-free multi-GPU cut: every rank projects all
// splats for its own tile rows): groups of 4096 splats, the conservative test for all of them (every position load of a thread in
// flight at once), the survivors compacted through LDS, their full projection on DENSE lanes, and what the binner's first pass
// reads of them — packed tile range, depth key, splat index — left compacted at the front of the group's segment
// (ho.cidx / ho.kept_groups): k_tf_scatter<COMPACTED> then runs one workgroup per group over the survivors only, as it does after
// k_band_prepare_tfc for gathered records.  Same records, keys, ranges, histogram and lists as k_project_hist_band.
// (256 threads x 16 splats — every group of a 5 M-splat frame resident at once instead of two rounds — measured the same: 51 us at
// C2 on eight ranks.  20 us of that are the test of all splats, at the copy rate; the survivors' exact projection — ~550
// instructions a splat with its fourteen IEEE divides, behind a gather — is the rest: VALU 0.34 busy, waves waiting 0.55 of their
// cycles: profiles/r05_r_exchange_free_band_projector.txt)
constexpr uint32_t PBC_THREADS = 512, PBC_WAVES = PBC_THREADS / 64, PBC_PER = 8, PBC_GROUP = PBC_THREADS * PBC_PER;
static_assert(PBC_GROUP == 4096 && PBC_PER * PBC_WAVES == 64, "a group is 4096 splats = 64 (row, wave) cells of 64 splats");
template <bool DISC, bool LIT>
__global__ __launch_bounds__(PBC_THREADS) void k_project_hist_bandc(FrameUniforms u, const float4 *__restrict__ pos_radius, uint32_t stride_vec4,
                                                                    uint32_t n, float4 *__restrict__ projected, uint32_t *__restrict__ keys_c,
                                                                    uint32_t *__restrict__ range_c, BinParams bp, TfHistOut ho, DiscIO dio, LitIO lio) {
    __shared__ uint32_t lh[PBC_WAVES][256];
    __shared__ uint32_t rowcnt[64];
    __shared__ uint32_t wsum[PBC_WAVES];
    __shared__ uint32_t s_list[PBC_GROUP];
    __shared__ uint32_t s_kept;
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const uint32_t blk = xcd_block_of(blockIdx.x, ho.xcd_per);
    if (blk >= ho.num_parts) return;
    if (blk == 0 && tid == 0) *ho.overflow_flag = 0;
    for (uint32_t j = tid; j < PBC_WAVES * 256; j += PBC_THREADS) (&lh[0][0])[j] = 0;
    const uint32_t g0 = blk * PBC_GROUP;
    uint32_t okbits = 0;
#pragma unroll
    for (uint32_t k0 = 0; k0 < PBC_PER; k0 += 8) {
        float4 pr[8], nr[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t i = g0 + (k0 + k) * PBC_THREADS + tid;
            pr[k] = nr[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (i < n) {
                pr[k] = pos_radius[(size_t)i * stride_vec4];
                if (DISC) nr[k] = dio.normals[(size_t)i * dio.normal_stride];
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t i = g0 + (k0 + k) * PBC_THREADS + tid;
            bool keep = false;
            if (i < n) {
                float r = fabsf(pr[k].w);
                if (DISC) r *= fmaxf(1.0f, 1.001f * __builtin_amdgcn_sqrtf((nr[k].x * nr[k].x + nr[k].y * nr[k].y) + nr[k].z * nr[k].z));
                keep = !cannot_reach_band<DISC>(u, pr[k], r, bp); // (NaN radius or normal: kept, the exact path decides)
            }
            const unsigned long long m = __ballot(keep);
            okbits |= keep ? (1u << (k0 + k)) : 0u;
            if (lane == 0) rowcnt[(k0 + k) * PBC_WAVES + w] = (uint32_t)__popcll(m);
        }
    }
    __syncthreads();
    if (w == 0) { // the cells in (row, wave) order = ascending splat index
        const uint32_t mine = rowcnt[lane];
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t x = __shfl_up(incl, d);
            if ((int)lane >= d) incl += x;
        }
        rowcnt[lane] = incl - mine;
        if (lane == 63) {
            ho.kept_groups[blk] = incl;
            s_kept = incl;
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < PBC_PER; ++k) {
        const bool keep = (okbits >> k) & 1u;
        const unsigned long long m = __ballot(keep);
        if (keep) s_list[rowcnt[k * PBC_WAVES + w] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = g0 + k * PBC_THREADS + tid;
    }
    __syncthreads();
    const uint32_t kept = s_kept;
    uint32_t local = 0;
    for (uint32_t j = tid; j < kept; j += PBC_THREADS) {
        const uint32_t i = s_list[j];
        const SplatIn in = load_splat<DISC, LIT>(pos_radius, stride_vec4, i, dio, lio);
        // (the record goes to the splat's own index — the composite gathers it by that —, key and range to the compacted slot)
        const uint32_t r = project_one<true, true, DISC, LIT>(u, in, i, 0, projected, keys_c, nullptr, range_c, bp, dio, lio, g0 + j);
        ho.cidx[g0 + j] = i;
        const uint32_t tx0 = r & 0xffu, tx1 = (r >> 8) & 0xffu, ty0 = (r >> 16) & 0xffu, ty1 = r >> 24;
        if (tx0 > tx1 || ty0 > ty1) continue; // (passed the conservative test, reaches no tile of the band: an empty range in the list)
        local += hist_add_rect(lh[w], tx0, tx1, ty0, ty1, bp.ntx, ho.mask);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d);
    if (lane == 0) wsum[w] = local;
    __syncthreads();
    if (tid <= ho.mask) {
        uint32_t hs = 0;
#pragma unroll
        for (uint32_t v = 0; v < PBC_WAVES; ++v) hs += lh[v][tid];
        ho.hist[(size_t)tid * ho.num_parts + blk] = hs;
    }
    if (tid == 0) {
        uint32_t ps = 0;
#pragma unroll
        for (uint32_t v = 0; v < PBC_WAVES; ++v) ps += wsum[v];
        ho.blocksums[blk] = ps;
    }
}

__global__ __launch_bounds__(256) void k_extract_keys(const float4 *__restrict__ projected, uint32_t n, uint32_t n_padded,
                                                      uint32_t *__restrict__ keys, uint32_t *__restrict__ payload) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_padded) return;
    if (i >= n) {
        keys[i] = 0xffffffffu;
        payload[i] = 0xffffffffu;
        return;
    }
    float depth = reinterpret_cast<const float *>(projected)[(size_t)i * 8 + 4];
    keys[i] = depth_key(depth);
    payload[i] = i;
}

__global__ __launch_bounds__(256) void k_update_props(const float4 *__restrict__ positions, const float4 *__restrict__ curvature,
                                                      uint32_t n, float4 *__restrict__ props) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 p = positions[i], c = curvature[i];
    props[(size_t)i * 2] = make_float4(p.x, p.y, p.z, 0.04f);                                                   // :94
    props[(size_t)i * 2 + 1] = make_float4(fabsf(c.x) * 0.8f + 0.2f, fabsf(c.y) * 0.8f + 0.2f, fabsf(c.z) * 0.8f + 0.2f, 1.0f); // :97-101
}

// the same update into two planes (the projector reads the first, the composite gathers from the second)
__global__ __launch_bounds__(256) void k_update_props_planes(const float4 *__restrict__ positions,
                                                             const float4 *__restrict__ curvature, uint32_t n,
                                                             float4 *__restrict__ pos_radius, float4 *__restrict__ color_opacity) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 p = positions[i], c = curvature[i];
    pos_radius[i] = make_float4(p.x, p.y, p.z, 0.04f);                                                                    // :94
    color_opacity[i] = make_float4(fabsf(c.x) * 0.8f + 0.2f, fabsf(c.y) * 0.8f + 0.2f, fabsf(c.z) * 0.8f + 0.2f, 1.0f); // :97-101
}

__global__ __launch_bounds__(256) void k_props_to_planes(const float4 *__restrict__ props, uint32_t n,
                                                         float4 *__restrict__ pos_radius, float4 *__restrict__ color_opacity) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    pos_radius[i] = props[(size_t)i * 2];
    color_opacity[i] = props[(size_t)i * 2 + 1];
}

static void load_uniforms(FrameUniforms &u, const float *uniforms) {
    for (int i = 0; i < 16; ++i) u.m[i] = uniforms[i];
    u.eye[0] = uniforms[16]; u.eye[1] = uniforms[17]; u.eye[2] = uniforms[18];
    u.time = uniforms[19]; u.w = uniforms[20]; u.h = uniforms[21];
}

int project_launch(splat_ctx *ctx, const float *uniforms, const void *pos_radius, uint32_t pr_stride_vec4, uint32_t n,
                   uint32_t index_base, void *projected, void *keys, void *payload, uint32_t n_padded, uint32_t *range32,
                   const BinParams *bp, const TfHistOut *hist_out, const void *normals, uint32_t normal_stride_vec4, void *discs,
                   const LitIO *lit) {
    FrameUniforms u;
    load_uniforms(u, uniforms);
    const uint32_t work = keys ? n_padded : n;
    if (work == 0) return SPLAT_OK;
    BinParams none = {0, 0, 1, 0, 0, 0, 0};
    const float4 *src = (const float4 *)pos_radius + (size_t)index_base * pr_stride_vec4;
    const bool disc = discs != nullptr; // the oriented-disc footprint (disc.h): normals in, disc records out
    // lit composite records (shade.h) of an isotropic frame, written next to the keys and tile ranges; a disc frame's lit
    // colours go behind its disc records (lit->records = discs there: 48-byte records)
    const bool with_lit = lit && lit->records && !disc, disc_lit = lit && lit->records && disc;
    if ((with_lit || disc_lit) && !(keys && range32 && !payload && index_base == 0))
        return ctx_fail(ctx, SPLAT_ERR_INVALID, "project_launch: lit records are written by the frame's projector only");
    const DiscIO dio = {disc ? (const float4 *)normals + (size_t)index_base * normal_stride_vec4 : nullptr, normal_stride_vec4,
                        (float4 *)discs, disc_lit ? 3u : 2u};
    const LitIO lio = (with_lit || disc_lit) ? *lit : LitIO{};
    stage_begin(ctx, SPLAT_STAGE_PROJECT);
    dim3 grid(div_up(work, 256)), block(256);
#define SPLAT_PROJECT_LAUNCH(K, R, D, L, RANGE, BP)                                                                                \
    hipLaunchKernelGGL((k_project<K, R, D, L>), grid, block, 0, ctx->stream, u, src, pr_stride_vec4, n, keys ? n_padded : n, index_base, \
                       (float4 *)projected, (uint32_t *)keys, (uint32_t *)payload, RANGE, BP, dio, lio)
#define SPLAT_PROJECT_HIST_LAUNCH(KERNEL, D, L)                                                                               \
    hipLaunchKernelGGL((KERNEL<D, L>), dim3(div_up(work, 1024)), block, 0, ctx->stream, u, src, pr_stride_vec4, n, n_padded, \
                       (float4 *)projected, (uint32_t *)keys, range32, *bp, *hist_out, dio, lio)
#define SPLAT_PROJECT_HIST_LAUNCH_PER(D, L)                                                                                                \
    do {                                                                                                                                   \
        TfHistOut ho_ = *hist_out;                                                                                                         \
        const uint32_t blocks_ = div_up(work, hist_out->block == TF_BLOCK_SMALL ? 256u : 1024u);                                           \
        ho_.xcd_per = blocks_ >= 64u ? div_up(blocks_, 8u) : 0u; /* the blocks dealt as k_tf_scatter's are (common.h: xcd_block_of) */      \
        const dim3 grid_(ho_.xcd_per ? 8u * ho_.xcd_per : blocks_);                                                                        \
        if (hist_out->block == TF_BLOCK_SMALL)                                                                                             \
            hipLaunchKernelGGL((k_project_hist<D, L, 1>), grid_, block, 0, ctx->stream, u, src, pr_stride_vec4, n, n_padded,               \
                               (float4 *)projected, (uint32_t *)keys, range32, *bp, ho_, dio, lio);                                        \
        else                                                                                                                               \
            hipLaunchKernelGGL((k_project_hist<D, L, 4>), grid_, block, 0, ctx->stream, u, src, pr_stride_vec4, n, n_padded,               \
                               (float4 *)projected, (uint32_t *)keys, range32, *bp, ho_, dio, lio);                                        \
    } while (0)
    if (hist_out && hist_out->cidx && bp->skip_outside && keys && range32 && !payload && index_base == 0) {
        // a strict band of a fraction of the screen: groups of 4096 splats, the survivors left compacted (hist_out->num_parts = groups)
        TfHistOut ho_ = *hist_out;
        ho_.xcd_per = ho_.num_parts >= 64u ? div_up(ho_.num_parts, 8u) : 0u;
        const dim3 grid_(ho_.xcd_per ? 8u * ho_.xcd_per : ho_.num_parts);
#define SPLAT_PROJECT_BANDC(D, L)                                                                                                      \
    hipLaunchKernelGGL((k_project_hist_bandc<D, L>), grid_, dim3(PBC_THREADS), 0, ctx->stream, u, src, pr_stride_vec4, n, (float4 *)projected, \
                       (uint32_t *)keys, range32, *bp, ho_, dio, lio)
        if (disc_lit) SPLAT_PROJECT_BANDC(true, true);
        else if (disc) SPLAT_PROJECT_BANDC(true, false);
        else if (with_lit) SPLAT_PROJECT_BANDC(false, true);
        else SPLAT_PROJECT_BANDC(false, false);
#undef SPLAT_PROJECT_BANDC
    } else if (hist_out && keys && range32 && !payload && index_base == 0) {
        // (a strict band's kernel works in 1024-splat blocks only: the caller keeps hist_out->block at TF_BLOCK_LARGE for it)
        if (disc_lit && bp->skip_outside) SPLAT_PROJECT_HIST_LAUNCH(k_project_hist_band, true, true);
        else if (disc_lit) SPLAT_PROJECT_HIST_LAUNCH_PER(true, true);
        else if (disc && bp->skip_outside) SPLAT_PROJECT_HIST_LAUNCH(k_project_hist_band, true, false);
        else if (disc) SPLAT_PROJECT_HIST_LAUNCH_PER(true, false);
        else if (bp->skip_outside && with_lit) SPLAT_PROJECT_HIST_LAUNCH(k_project_hist_band, false, true);
        else if (bp->skip_outside) SPLAT_PROJECT_HIST_LAUNCH(k_project_hist_band, false, false);
        else if (with_lit) SPLAT_PROJECT_HIST_LAUNCH_PER(false, true);
        else SPLAT_PROJECT_HIST_LAUNCH_PER(false, false);
    } else if (keys && range32) {
        if (disc_lit) SPLAT_PROJECT_LAUNCH(true, true, true, true, range32, *bp);
        else if (disc) SPLAT_PROJECT_LAUNCH(true, true, true, false, range32, *bp);
        else if (with_lit) SPLAT_PROJECT_LAUNCH(true, true, false, true, range32, *bp);
        else SPLAT_PROJECT_LAUNCH(true, true, false, false, range32, *bp);
    } else if (keys) {
        if (disc) SPLAT_PROJECT_LAUNCH(true, false, true, false, nullptr, none);
        else SPLAT_PROJECT_LAUNCH(true, false, false, false, nullptr, none);
    } else {
        if (disc) SPLAT_PROJECT_LAUNCH(false, false, true, false, nullptr, none);
        else SPLAT_PROJECT_LAUNCH(false, false, false, false, nullptr, none);
    }
#undef SPLAT_PROJECT_LAUNCH
#undef SPLAT_PROJECT_HIST_LAUNCH
#undef SPLAT_PROJECT_HIST_LAUNCH_PER
    LAUNCH_CHECK(ctx, "k_project");
    stage_end(ctx, SPLAT_STAGE_PROJECT);
    return SPLAT_OK;
}

extern "C" {

int splat_project(splat_ctx *ctx, const float *uniforms, const void *pos_radius, uint32_t pr_stride_vec4, uint32_t n,
                  void *projected, void *keys, void *payload, uint32_t n_padded) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, uniforms && (n == 0 || (pos_radius && projected)));
    ARG_CHECK(ctx, pr_stride_vec4 >= 1);
    ARG_CHECK(ctx, (keys == nullptr) == (payload == nullptr));
    ARG_CHECK(ctx, keys == nullptr || n_padded >= n);
    ARG_CHECK(ctx, (((uintptr_t)pos_radius | (uintptr_t)projected) & 15) == 0);
    return project_launch(ctx, uniforms, pos_radius, pr_stride_vec4, n, 0, projected, keys, payload, n_padded, nullptr, nullptr, nullptr);
}

int splat_project_disc(splat_ctx *ctx, const float *uniforms, const void *pos_radius, uint32_t pr_stride_vec4, const void *normals,
                       uint32_t normal_stride_vec4, uint32_t n, void *projected, void *discs, void *keys, void *payload,
                       uint32_t n_padded) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, uniforms && (n == 0 || (pos_radius && normals && projected && discs)));
    ARG_CHECK(ctx, pr_stride_vec4 >= 1 && normal_stride_vec4 >= 1);
    ARG_CHECK(ctx, (keys == nullptr) == (payload == nullptr));
    ARG_CHECK(ctx, keys == nullptr || n_padded >= n);
    ARG_CHECK(ctx, (((uintptr_t)pos_radius | (uintptr_t)normals | (uintptr_t)projected | (uintptr_t)discs) & 15) == 0);
    // (n == 0 with keys only pads them, which either footprint's kernel does)
    return project_launch(ctx, uniforms, pos_radius, pr_stride_vec4, n, 0, projected, keys, payload, n_padded, nullptr, nullptr, nullptr,
                          normals, normal_stride_vec4, n ? discs : nullptr);
}

int splat_project_slice(splat_ctx *ctx, const float *uniforms, const void *pos_radius, uint32_t pr_stride_vec4, uint32_t first,
                        uint32_t count, void *projected_slice) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, uniforms && (count == 0 || (pos_radius && projected_slice)) && pr_stride_vec4 >= 1);
    ARG_CHECK(ctx, (((uintptr_t)pos_radius | (uintptr_t)projected_slice) & 15) == 0);
    return project_launch(ctx, uniforms, pos_radius, pr_stride_vec4, count, first, projected_slice, nullptr, nullptr, 0, nullptr,
                          nullptr, nullptr);
}


int splat_project_slice_compact(splat_ctx *ctx, const float *uniforms, const void *pos_radius, uint32_t pr_stride_vec4,
                                uint32_t first, uint32_t count, void *records16_slice) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, uniforms && (count == 0 || (pos_radius && records16_slice)) && pr_stride_vec4 >= 1);
    ARG_CHECK(ctx, (((uintptr_t)pos_radius | (uintptr_t)records16_slice) & 15) == 0);
    if (count == 0) return SPLAT_OK;
    FrameUniforms u;
    load_uniforms(u, uniforms);
    stage_begin(ctx, SPLAT_STAGE_PROJECT);
    hipLaunchKernelGGL(k_project_compact, dim3(div_up(count, 256)), dim3(256), 0, ctx->stream, u,
                       (const float4 *)pos_radius + (size_t)first * pr_stride_vec4, pr_stride_vec4, count, (float4 *)records16_slice);
    LAUNCH_CHECK(ctx, "k_project_compact");
    stage_end(ctx, SPLAT_STAGE_PROJECT);
    return SPLAT_OK;
}

int splat_project_slice_disc(splat_ctx *ctx, const float *uniforms, const void *pos_radius, uint32_t pr_stride_vec4, const void *normals,
                             uint32_t normal_stride_vec4, uint32_t first, uint32_t count, void *records48_slice) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, uniforms && (count == 0 || (pos_radius && normals && records48_slice)) && pr_stride_vec4 >= 1 && normal_stride_vec4 >= 1);
    ARG_CHECK(ctx, (((uintptr_t)pos_radius | (uintptr_t)normals | (uintptr_t)records48_slice) & 15) == 0);
    if (count == 0) return SPLAT_OK;
    FrameUniforms u;
    load_uniforms(u, uniforms);
    stage_begin(ctx, SPLAT_STAGE_PROJECT);
    hipLaunchKernelGGL(k_project_disc48, dim3(div_up(count, 256)), dim3(256), 0, ctx->stream, u,
                       (const float4 *)pos_radius + (size_t)first * pr_stride_vec4, pr_stride_vec4,
                       (const float4 *)normals + (size_t)first * normal_stride_vec4, normal_stride_vec4, count, (float4 *)records48_slice);
    LAUNCH_CHECK(ctx, "k_project_disc48");
    stage_end(ctx, SPLAT_STAGE_PROJECT);
    return SPLAT_OK;
}

int splat_expand_compact(splat_ctx *ctx, const void *records16, uint32_t n, uint32_t index_base, void *projected) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, n == 0 || (records16 && projected));
    ARG_CHECK(ctx, (((uintptr_t)records16 | (uintptr_t)projected) & 15) == 0);
    if (n == 0) return SPLAT_OK;
    hipLaunchKernelGGL(k_expand_compact, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, (const float4 *)records16, n, index_base,
                       (float4 *)projected);
    LAUNCH_CHECK(ctx, "k_expand_compact");
    return SPLAT_OK;
}

int splat_extract_keys(splat_ctx *ctx, const void *projected, uint32_t n, uint32_t n_padded, void *keys, void *payload) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, keys && payload && (n == 0 || projected) && n_padded >= n);
    if (n_padded == 0) return SPLAT_OK;
    hipLaunchKernelGGL(k_extract_keys, dim3(div_up(n_padded, 256)), dim3(256), 0, ctx->stream, (const float4 *)projected, n,
                       n_padded, (uint32_t *)keys, (uint32_t *)payload);
    LAUNCH_CHECK(ctx, "k_extract_keys");
    return SPLAT_OK;
}

int splat_update_props(splat_ctx *ctx, const void *positions, const void *curvature, uint32_t n, void *props) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, n == 0 || (positions && curvature && props));
    if (n == 0) return SPLAT_OK;
    hipLaunchKernelGGL(k_update_props, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, (const float4 *)positions,
                       (const float4 *)curvature, n, (float4 *)props);
    LAUNCH_CHECK(ctx, "k_update_props");
    return SPLAT_OK;
}

int splat_update_props_planes(splat_ctx *ctx, const void *positions, const void *curvature, uint32_t n, void *pos_radius,
                              void *color_opacity) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, n == 0 || (positions && curvature && pos_radius && color_opacity));
    if (n == 0) return SPLAT_OK;
    hipLaunchKernelGGL(k_update_props_planes, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, (const float4 *)positions,
                       (const float4 *)curvature, n, (float4 *)pos_radius, (float4 *)color_opacity);
    LAUNCH_CHECK(ctx, "k_update_props_planes");
    return SPLAT_OK;
}

int splat_props_to_planes(splat_ctx *ctx, const void *props, uint32_t n, void *pos_radius, void *color_opacity) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, n == 0 || (props && pos_radius && color_opacity));
    ARG_CHECK(ctx, (((uintptr_t)props | (uintptr_t)pos_radius | (uintptr_t)color_opacity) & 15) == 0);
    if (n == 0) return SPLAT_OK;
    hipLaunchKernelGGL(k_props_to_planes, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, (const float4 *)props, n, (float4 *)pos_radius,
                       (float4 *)color_opacity);
    LAUNCH_CHECK(ctx, "k_props_to_planes");
    return SPLAT_OK;
}

} // extern "C"
