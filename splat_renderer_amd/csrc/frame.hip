// frame.hip — whole-frame convenience and the multi-GPU band filter.
//
// splat_render_frame strings the stages in the reference's intended order (from
// /root/reference/GPU_PIPELINE_PLAN.md:41-84): project(+keys) -> sort -> binSplats -> composite.
// splat_band_keys has no reference equivalent (the reference is single-device): it is the
// "keep splats whose tile-row range meets my band" step of SURVEY §8e, done as a stable
// compaction so that ties in depth still resolve by ascending global index.
#include "common.h"
#include "tile_range.h"
#include "disc.h"
#include "shade.h"

#include <cstdlib>

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

// ---- fused band filter for splat_band_frame (tile coordinates fit 8 bits) -------------------------
// prepare: one coalesced pass over the gathered 32-byte records: per record the depth key and the
// band-clamped tile range (in index order), per 512-record block the number kept.
constexpr uint32_t BAND_THREADS = 256, BAND_PER_THREAD = 2, BAND_BLOCK = BAND_THREADS * BAND_PER_THREAD;

__global__ __launch_bounds__(BAND_THREADS) void k_band_prepare(const float4 *__restrict__ records, uint32_t n, BinParams bp,
                                                               uint32_t *__restrict__ keys_by_idx,
                                                               uint32_t *__restrict__ range32, uint32_t *__restrict__ blocksums) {
    __shared__ uint32_t wsum[4];
    uint32_t kept = 0;
#pragma unroll
    for (uint32_t k = 0; k < BAND_PER_THREAD; ++k) {
        const uint32_t i = blockIdx.x * BAND_BLOCK + k * BAND_THREADS + threadIdx.x;
        if (i < n) {
            const float4 a = records[(size_t)i * 2], b = records[(size_t)i * 2 + 1];
            uint32_t tx0, tx1, ty0, ty1;
            const bool ok = tile_range(a, bp.width, bp.height, bp.tile, bp.ntx, bp.nty, bp.row0, bp.row1, tx0, tx1, ty0, ty1);
            range32[i] = pack_range32(ok, tx0, tx1, ty0, ty1);
            keys_by_idx[i] = depth_key_of(b.x);
            kept += ok ? 1u : 0u;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) kept += __shfl_xor(kept, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = kept;
    __syncthreads();
    if (threadIdx.x == 0) blocksums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Tile-first band frame: the same pass, 1024 records per block (the binner's block), which also counts
// the block's pairs per low tile-id digit like k_project_hist; a splat outside the band simply has an
// empty range (no compaction, no sort of the kept splats).
constexpr uint32_t BTF_PER_THREAD = 4, BTF_BLOCK = BAND_THREADS * BTF_PER_THREAD;

// FORMAT = cfg->record_format.  COMPACT: 16-byte exchange records, the bounds are rebuilt as the projector forms
// them.  DISC48: 48-byte oriented-disc exchange records, the bounds are the disc's (disc_bounds of the record).
// (Round 4 also let this pass write 32-byte lit composite records for the splats the band keeps, so that the band's composite
// gathers one line per staged entry instead of three: 20 us per rank SLOWER with eight ranks — the kept splats are an eighth
// of all, scattered, every record a line of its own; profiles/r04_c_band_lit_records_C2.txt.  Removed in round 5.)
template <int FORMAT>
__global__ __launch_bounds__(BAND_THREADS) void k_band_prepare_tf(const float4 *__restrict__ records, uint32_t n, BinParams bp,
                                                                  uint32_t *__restrict__ keys_by_idx, uint32_t *__restrict__ range32,
                                                                  uint32_t *__restrict__ kept_blocks, TfHistOut ho) {
    __shared__ uint32_t lh[4][256]; // the block's pairs per low tile-id digit, as k_project_hist counts them
    __shared__ uint32_t wsum[2][4];
    const uint32_t tid = threadIdx.x, w = tid >> 6;
    if (blockIdx.x == 0 && tid == 0) *ho.overflow_flag = 0;
    for (uint32_t j = tid; j < 4 * 256; j += BAND_THREADS) (&lh[0][0])[j] = 0;
    __syncthreads();
    uint32_t kept = 0, pairs = 0;
#pragma unroll
    for (uint32_t k = 0; k < BTF_PER_THREAD; ++k) {
        const uint32_t i = blockIdx.x * BTF_BLOCK + k * BAND_THREADS + tid;
        if (i < n) {
            float4 a;
            float depth;
            if (FORMAT == SPLAT_RECORDS_DISC48) {
                const DiscRecord d = {records[(size_t)i * 3], records[(size_t)i * 3 + 1]};
                disc_bounds(d, a); // (NaN padding records: not finite -> all zero -> bins nowhere)
                depth = records[(size_t)i * 3 + 2].x;
            } else if (FORMAT == SPLAT_RECORDS_COMPACT) {
                const float4 c = records[i];
                const float padded = c.z * 1.5f; // SplatProjector.ts:119-121 (this file is compiled with -ffp-contract=off)
                a = make_float4(c.x - padded, c.y - padded, c.x + padded, c.y + padded);
                depth = c.w;
            } else {
                a = records[(size_t)i * 2];
                depth = records[(size_t)i * 2 + 1].x;
            }
            uint32_t tx0, tx1, ty0, ty1;
            const bool ok = tile_range(a, bp.width, bp.height, bp.tile, bp.ntx, bp.nty, bp.row0, bp.row1, tx0, tx1, ty0, ty1);
            range32[i] = pack_range32(ok, tx0, tx1, ty0, ty1);
            keys_by_idx[i] = depth_key_of(depth);
            if (ok) {
                kept += 1u;
                pairs += hist_add_rect(lh[w], tx0, tx1, ty0, ty1, bp.ntx, ho.mask);
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        kept += __shfl_xor(kept, d);
        pairs += __shfl_xor(pairs, d);
    }
    if ((tid & 63) == 0) {
        wsum[0][w] = kept;
        wsum[1][w] = pairs;
    }
    __syncthreads();
    if (tid <= ho.mask) ho.hist[(size_t)tid * ho.num_parts + blockIdx.x] = lh[0][tid] + lh[1][tid] + lh[2][tid] + lh[3][tid];
    if (tid == 0) {
        kept_blocks[blockIdx.x] = wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
        ho.blocksums[blockIdx.x] = wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
    }
}

// The same pass for bands that keep a FRACTION of the records (every band of a multi-GPU frame): a workgroup takes a group of
// BTC_GROUP = 4096 consecutive records and leaves the splats the band keeps COMPACTED, in index order, at the start of the
// group's segment of the range / key arrays, their indices beside them, their number in kept_groups[group]; the first-pass
// histogram and the pair count are per group.  k_tf_scatter then runs one workgroup per group over the kept splats only
// (tile_first.hip, COMPACTED): a band of an eighth of the screen keeps an eighth of the splats, and what the scatter costs
// is its workgroups (C2, eight virtual ranks: scatter 34 -> 17 us per rank, this pass 26 -> 33 us, the rank's frame 0.154 ->
// 0.144 ms; profiles/r04_j_band_compaction_C2.txt).
constexpr uint32_t BTC_THREADS = 512, BTC_WAVES = BTC_THREADS / 64, BTC_PER_THREAD = 8, BTC_GROUP = BTC_THREADS * BTC_PER_THREAD;
static_assert(BTC_GROUP == 4096 && BTC_PER_THREAD * BTC_WAVES == 64, "a group is 4096 records = 64 (row, wave) cells of 64 records");
template <int FORMAT>
__global__ __launch_bounds__(BTC_THREADS) __attribute__((amdgpu_waves_per_eu(FORMAT == SPLAT_RECORDS_COMPACT ? 6 : 4, 8))) void k_band_prepare_tfc(const float4 *__restrict__ records, uint32_t n, BinParams bp,
                                                                  uint32_t *__restrict__ keys_c, uint32_t *__restrict__ range_c,
                                                                  uint32_t *__restrict__ idx_c, uint32_t *__restrict__ kept_groups,
                                                                  TfHistOut ho) {
    __shared__ uint32_t lh[BTC_WAVES][256];
    __shared__ uint32_t rowcnt[64]; // kept per (row of 512 records, wave), then its exclusive scan: the cells in index order
    __shared__ uint32_t wsum[BTC_WAVES];
    __shared__ uint32_t s_rng[BTC_GROUP]; // the kept splats' packed tile ranges, compacted: the histogram is counted over DENSE lanes
    __shared__ uint32_t s_kept;
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const uint32_t blk = xcd_block_of(blockIdx.x, ho.xcd_per); // (each XCD a contiguous eighth of the groups: common.h)
    if (blk >= ho.num_parts) return;
    if (blk == 0 && tid == 0) *ho.overflow_flag = 0;
    for (uint32_t j = tid; j < BTC_WAVES * 256; j += BTC_THREADS) (&lh[0][0])[j] = 0;
    const uint32_t g0 = blk * BTC_GROUP;
    // every record of the thread is in flight before the first is looked at (one memory round trip per workgroup)
    float4 ra[BTC_PER_THREAD], rb[BTC_PER_THREAD], rc[BTC_PER_THREAD];
#pragma unroll
    for (uint32_t k = 0; k < BTC_PER_THREAD; ++k) {
        const uint32_t i = g0 + k * BTC_THREADS + tid;
        ra[k] = rb[k] = rc[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (i < n) {
            if (FORMAT == SPLAT_RECORDS_DISC48) {
                ra[k] = records[(size_t)i * 3];
                rb[k] = records[(size_t)i * 3 + 1];
                rc[k] = records[(size_t)i * 3 + 2];
            } else if (FORMAT == SPLAT_RECORDS_COMPACT) {
                ra[k] = records[i];
            } else {
                ra[k] = records[(size_t)i * 2];
                rb[k] = records[(size_t)i * 2 + 1];
            }
        }
    }
    uint32_t rng[BTC_PER_THREAD], key[BTC_PER_THREAD];
    uint32_t okbits = 0;
#pragma unroll
    for (uint32_t k = 0; k < BTC_PER_THREAD; ++k) {
        const uint32_t i = g0 + k * BTC_THREADS + tid;
        bool ok = false;
        rng[k] = 1u; // (the empty range)
        key[k] = 0u;
        if (i < n) {
            float4 a;
            float depth;
            if (FORMAT == SPLAT_RECORDS_DISC48) {
                const DiscRecord d = {ra[k], rb[k]};
                disc_bounds(d, a); // (NaN padding records: not finite -> all zero -> bins nowhere)
                depth = rc[k].x;
            } else if (FORMAT == SPLAT_RECORDS_COMPACT) {
                const float4 c = ra[k];
                const float padded = c.z * 1.5f; // SplatProjector.ts:119-121 (this file is compiled with -ffp-contract=off)
                a = make_float4(c.x - padded, c.y - padded, c.x + padded, c.y + padded);
                depth = c.w;
            } else {
                a = ra[k];
                depth = rb[k].x;
            }
            uint32_t tx0, tx1, ty0, ty1;
            ok = tile_range(a, bp.width, bp.height, bp.tile, bp.ntx, bp.nty, bp.row0, bp.row1, tx0, tx1, ty0, ty1);
            if (ok) {
                rng[k] = pack_range32(true, tx0, tx1, ty0, ty1);
                key[k] = depth_key_of(depth);
            }
        }
        const unsigned long long m = __ballot(ok);
        okbits |= ok ? (1u << k) : 0u;
        if (lane == 0) rowcnt[k * BTC_WAVES + w] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    // where each (row, wave) cell's kept splats go: an exclusive scan of the 64 counts in (row, wave) order = ascending index
    if (w == 0) {
        const uint32_t mine = rowcnt[lane];
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t x = __shfl_up(incl, d);
            if ((int)lane >= d) incl += x;
        }
        rowcnt[lane] = incl - mine;
        if (lane == 63) {
            kept_groups[blk] = incl;
            s_kept = incl;
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < BTC_PER_THREAD; ++k) {
        const bool ok = (okbits >> k) & 1u;
        const unsigned long long m = __ballot(ok);
        if (ok) {
            const uint32_t local = rowcnt[k * BTC_WAVES + w] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            range_c[g0 + local] = rng[k];
            keys_c[g0 + local] = key[k];
            idx_c[g0 + local] = g0 + k * BTC_THREADS + tid;
            s_rng[local] = rng[k];
        }
    }
    __syncthreads();
    // the first pass's histogram over the kept splats, one per lane (a band of an eighth of the screen keeps an eighth: counted
    // where they stood, every wave ran the rectangle loops eight times over for its few kept lanes)
    const uint32_t kept = s_kept;
    uint32_t pairs = 0;
    for (uint32_t j = tid; j < kept; j += BTC_THREADS) {
        const uint32_t r = s_rng[j];
        pairs += hist_add_rect(lh[w], r & 0xffu, (r >> 8) & 0xffu, (r >> 16) & 0xffu, r >> 24, bp.ntx, ho.mask);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) pairs += __shfl_xor(pairs, d);
    if (lane == 0) wsum[w] = pairs;
    __syncthreads();
    if (tid <= ho.mask) {
        uint32_t hs = 0;
#pragma unroll
        for (uint32_t v = 0; v < BTC_WAVES; ++v) hs += lh[v][tid];
        ho.hist[(size_t)tid * ho.num_parts + blk] = hs;
    }
    if (tid == 0) {
        uint32_t ps = 0;
#pragma unroll
        for (uint32_t v = 0; v < BTC_WAVES; ++v) ps += wsum[v];
        ho.blocksums[blk] = ps;
    }
}

// compact: the kept (key, global index) pairs of block b go to [base[b], ...) in ascending index order
__global__ __launch_bounds__(BAND_THREADS) void k_band_compact(const uint32_t *__restrict__ keys_by_idx,
                                                               const uint32_t *__restrict__ range32, uint32_t n,
                                                               const uint32_t *__restrict__ block_base,
                                                               uint32_t *__restrict__ keys, uint32_t *__restrict__ payload) {
    __shared__ uint32_t wsum[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint32_t carry = block_base[blockIdx.x];
#pragma unroll
    for (uint32_t k = 0; k < BAND_PER_THREAD; ++k) {
        const uint32_t i = blockIdx.x * BAND_BLOCK + k * BAND_THREADS + tid;
        const bool keep = (i < n) && ((range32[i] & 0xffu) <= ((range32[i] >> 8) & 0xffu)); // tx0 <= tx1: not the empty code
        const unsigned long long m = __ballot(keep);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
        if (lane == 0) wsum[w] = (uint32_t)__popcll(m);
        __syncthreads();
        const uint32_t s0 = wsum[0], s1 = wsum[1], s2 = wsum[2], s3 = wsum[3];
        __syncthreads();
        if (keep) {
            const uint32_t o = carry + (w > 0 ? s0 : 0u) + (w > 1 ? s1 : 0u) + (w > 2 ? s2 : 0u) + below;
            keys[o] = keys_by_idx[i];
            payload[o] = i;
        }
        carry += s0 + s1 + s2 + s3;
    }
}

// Resolves the previous splat_band_frame's kept-count readback.  If that frame kept more splats than
// the bound its grids were sized for, it was rendered from a truncated set: say so (once).
static int band_settle_count(splat_ctx *ctx, splat_sorter *sorter) {
    if (!sorter->count_pending) return SPLAT_OK;
    sorter->count_pending = false;
    HIP_TRY(ctx, hipEventSynchronize(sorter->count_event));
    const uint32_t kept = *(volatile uint32_t *)sorter->pinned_count;
    sorter->last_count = kept;
    sorter->have_last_count = true;
    if (kept > sorter->count_bound) {
        sorter->have_last_count = false; // next frame: full-size grids, then learn again
        return ctx_fail(ctx, SPLAT_ERR_CAPACITY,
                        "the previous band frame kept more splats than its sync-free bound (1.125x the frame before it): it "
                        "was rendered from a truncated set; render that frame again");
    }
    return SPLAT_OK;
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

// which order of work splat_render_frame uses for this binner (see splat_bin_set_frame_order)
static int g_frame_order = -2;
static int frame_order(const splat_binner *b) {
    if (b->frame_order >= 0) return b->frame_order;
    if (g_frame_order == -2) {
        // tile-first measures faster at every bench size (C2: 0.507 vs 0.600 ms/frame)
        g_frame_order = SPLAT_FRAME_TILE_FIRST;
        if (const char *e = getenv("SPLAT_FRAME_ORDER")) g_frame_order = (e[0] == 's' || e[0] == '0') ? SPLAT_FRAME_SORT_FIRST : SPLAT_FRAME_TILE_FIRST;
    }
    return g_frame_order;
}

// A band that is a fraction of the screen keeps a fraction of the splats: its first pass compacts them per group of 4096 and the
// scatter runs over the kept splats only.  Measured with virtual ranks at C2: a gain up to a third of the rows and still at 23 of
// 68 (the tallest of four pair-balanced bands: 0.200 -> 0.185 ms without an exchange, 0.184 -> 0.178 with), a loss at half of
// them (two ranks: +3..8 us).  The rule: bands of at most two fifths of the rows.  (SPLAT_BAND_COMPACT=0 | 1 forces one of them.)
static bool band_compacting(uint32_t row0, uint32_t row1, uint32_t nty) {
    static int s_compact = -2;
    if (s_compact == -2) {
        const char *e = getenv("SPLAT_BAND_COMPACT");
        s_compact = !e ? -1 : (e[0] == '0' ? 0 : 1);
    }
    return s_compact == 1 || (s_compact == -1 && 5u * (row1 - row0) <= 2u * nty);
}

// (the compacted splats' indices: one slot per record, rounded up to whole groups)
static int binner_reserve_band_idx(splat_ctx *ctx, splat_binner *binner, uint32_t n_records) {
    if (n_records <= binner->band_idx_cap) return SPLAT_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (binner->band_idx) (void)hipFree(binner->band_idx);
    binner->band_idx = nullptr;
    binner->band_idx_cap = 0;
    const size_t slots = (size_t)div_up(n_records, BTC_GROUP) * BTC_GROUP;
    if (hipMalloc((void **)&binner->band_idx, slots * 4 + 256) != hipSuccess) return ctx_fail(ctx, SPLAT_ERR_OOM, "band index hipMalloc");
    binner->band_idx_cap = n_records;
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
    ARG_CHECK(ctx, sorter && binner && cfg && props && (normals || cfg->prelit) && (n_records == 0 || records));
    ARG_CHECK(ctx, cfg->tile_size == splat_bin_tile_size(binner) && width >= 1 && height >= 1);
    // (what the gathered records are: ProjectedSplat, 16-byte exchange records, 48-byte disc exchange records)
    ARG_CHECK(ctx, cfg->footprint <= SPLAT_FOOTPRINT_DISC && cfg->record_format <= SPLAT_RECORDS_DISC48);
    // the oriented disc travels as its own 48-byte records, and only those carry it
    ARG_CHECK(ctx, (cfg->footprint == SPLAT_FOOTPRINT_DISC) == (cfg->record_format == SPLAT_RECORDS_DISC48));
    // colours: the second vec4 of the reference's interleaved records, or (cfg->prelit) `props` IS the plane of lit colours
    const void *band_color = cfg->prelit ? props : (const void *)((const char *)props + 16);
    const uint32_t band_color_stride = cfg->prelit ? 1u : 2u;
    const uint32_t tile = cfg->tile_size, nty = div_up(height, tile);
    uint32_t row0 = cfg->tile_row0, row1 = cfg->tile_row1 > nty ? nty : cfg->tile_row1;
    if (row0 > row1) row0 = row1;
    const uint32_t ntx = div_up(width, tile);
    const bool fast = ntx <= 256 && nty <= 256 && n_records > 0;
    const bool compact = cfg->record_format == SPLAT_RECORDS_COMPACT;
    const bool disc = cfg->record_format == SPLAT_RECORDS_DISC48;
    if (disc && n_records > 0 && !(fast && frame_order(binner) == SPLAT_FRAME_TILE_FIRST))
        return ctx_fail(ctx, SPLAT_ERR_INVALID,
                        "splat_band_frame: oriented-disc records need the tile-first frame order and a screen of at most 256 x 256 tiles");
    ARG_CHECK(ctx, ((uintptr_t)records & 15) == 0);
    if (compact && n_records > 0 && !(fast && frame_order(binner) == SPLAT_FRAME_TILE_FIRST)) {
        // the other orders of work read ProjectedSplat records: rebuild them once (bit-exact) and go on
        if (n_records > binner->expanded_cap) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (binner->expanded) (void)hipFree(binner->expanded);
            binner->expanded = nullptr;
            binner->expanded_cap = 0;
            if (hipMalloc(&binner->expanded, (size_t)n_records * 32 + 256) != hipSuccess) return ctx_fail(ctx, SPLAT_ERR_OOM, "band frame hipMalloc");
            binner->expanded_cap = n_records;
        }
        int rc = splat_expand_compact(ctx, records, n_records, 0, binner->expanded);
        if (rc != SPLAT_OK) return rc;
        splat_composite_cfg c2 = *cfg;
        c2.record_format = SPLAT_RECORDS_PROJECTED;
        return splat_band_frame(ctx, sorter, binner, &c2, props, normals, binner->expanded, n_records, width, height, out_rgba8,
                                out_rgba32f, consumed_dptr);
    }
    if (fast && frame_order(binner) == SPLAT_FRAME_TILE_FIRST) {
        // tile-first: one pass over the records (keys, band-clamped ranges, pair counts), then the
        // binner in index order — a splat outside the band has an empty range and costs nothing more
        if (n_records > sorter->capacity) return ctx_fail(ctx, SPLAT_ERR_CAPACITY, "splat_band_frame: n_records exceeds the sorter's capacity");
        int rc = binner_reserve_range32(binner, n_records);
        if (rc != SPLAT_OK) return rc;
        rc = binner_reserve(binner, ntx * nty, n_records);
        if (rc != SPLAT_OK) return rc;
        const BinParams bp = {width, height, tile, ntx, nty, row0, row1};
        const bool compacting = band_compacting(row0, row1, nty);
        const uint32_t blocks = div_up(n_records, compacting ? BTC_GROUP : BTF_BLOCK);
        if (compacting) {
            rc = binner_reserve_band_idx(ctx, binner, n_records);
            if (rc != SPLAT_OK) return rc;
        }
        TfHistOut ho = {binner->tf_hist, binner->blocksums, binner->d_total + 1, (1u << tile_id_low_bits(ntx * nty)) - 1u, blocks};
        stage_begin(ctx, SPLAT_STAGE_PROJECT);
        if (compacting) {
            ho.xcd_per = blocks >= 64u ? div_up(blocks, 8u) : 0u; // (the groups dealt as k_tf_scatter<COMPACTED> deals them)
            const uint32_t grid_blocks = ho.xcd_per ? 8u * ho.xcd_per : blocks;
#define SPLAT_BAND_PREPARE_C(FORMAT)                                                                                                  \
    hipLaunchKernelGGL(k_band_prepare_tfc<FORMAT>, dim3(grid_blocks), dim3(BTC_THREADS), 0, ctx->stream, (const float4 *)records, n_records, bp, \
                       sorter->keys, binner->range32, binner->band_idx, sorter->hist, ho)
            if (disc) SPLAT_BAND_PREPARE_C(SPLAT_RECORDS_DISC48);
            else if (compact) SPLAT_BAND_PREPARE_C(SPLAT_RECORDS_COMPACT);
            else SPLAT_BAND_PREPARE_C(SPLAT_RECORDS_PROJECTED);
#undef SPLAT_BAND_PREPARE_C
        } else if (disc)
            hipLaunchKernelGGL(k_band_prepare_tf<SPLAT_RECORDS_DISC48>, dim3(blocks), dim3(BAND_THREADS), 0, ctx->stream,
                               (const float4 *)records, n_records, bp, sorter->keys, binner->range32, sorter->hist, ho);
        else if (compact)
            hipLaunchKernelGGL(k_band_prepare_tf<SPLAT_RECORDS_COMPACT>, dim3(blocks), dim3(BAND_THREADS), 0, ctx->stream,
                               (const float4 *)records, n_records, bp, sorter->keys, binner->range32, sorter->hist, ho);
        else
            hipLaunchKernelGGL(k_band_prepare_tf<SPLAT_RECORDS_PROJECTED>, dim3(blocks), dim3(BAND_THREADS), 0, ctx->stream,
                               (const float4 *)records, n_records, bp, sorter->keys, binner->range32, sorter->hist, ho);
        LAUNCH_CHECK(ctx, "k_band_prepare_tf");
        stage_end(ctx, SPLAT_STAGE_PROJECT);
        binner->tf_hist_ready = true;
        binner->tf_block = compacting ? BTC_GROUP : BTF_BLOCK;
        binner->tf_cidx = compacting ? binner->band_idx : nullptr;
        binner->tf_kept = compacting ? sorter->hist : nullptr;
        sorter->ran = false;
        sorter->count_pending = false;
        sorter->kept_blocks = blocks; // the kept count is summed on demand (splat_band_kept / splat_band_settle)
        rc = binner_run(binner, records, n_records, nullptr, n_records, width, height, row0, row1, binner->range32, nullptr, sorter->keys);
        if (rc != SPLAT_OK) return rc;
        splat_composite_cfg c2 = *cfg;
        c2.tile_row0 = row0;
        c2.tile_row1 = row1;
        uint32_t *report = binner->report_for_composite; // (the frame's last kernel reports it: common.h)
        binner->report_for_composite = nullptr;
        return composite_launch(ctx, &c2, band_color, band_color_stride, normals, 1, records, binner->pairs.payload, binner->counts, binner->offsets, width, height, out_rgba8, out_rgba32f, consumed_dptr,
                                binner->d_total, report, binner->report_seq);
    }
    sorter->kept_blocks = 0;
    // keep -> sort -> bin with the kept count living on the device: no host round trip in here
    int rc = band_settle_count(ctx, sorter); // the previous frame's kept count (async readback)
    if (rc != SPLAT_OK) {
        binner->have_last = false; // its pair total came from the truncated set: do not size the next frame from it
        return rc;
    }
    uint32_t *range32 = nullptr;
    // grids of the sort and of the binner's count/expand are sized for `bound` kept splats: all
    // records on a first frame, 1.125x the previous frame's kept count afterwards
    uint32_t bound = n_records;
    if (sorter->have_last_count) {
        const uint64_t b = (uint64_t)sorter->last_count + sorter->last_count / 8 + 4096;
        if (b < bound) bound = (uint32_t)b;
    }
    if (n_records > 0) {
        if (n_records > sorter->capacity) return ctx_fail(ctx, SPLAT_ERR_CAPACITY, "splat_band_frame: n_records exceeds the sorter's capacity");
        if (fast) {
            rc = binner_reserve_range32(binner, n_records);
            if (rc != SPLAT_OK) return rc;
            range32 = binner->range32;
            const BinParams bp = {width, height, tile, ntx, nty, row0, row1};
            const uint32_t blocks = div_up(n_records, BAND_BLOCK);
            uint32_t *keys_by_idx = sorter->keys_b, *blocksums = sorter->hist; // both free until the sort starts
            stage_begin(ctx, SPLAT_STAGE_PROJECT);
            hipLaunchKernelGGL(k_band_prepare, dim3(blocks), dim3(BAND_THREADS), 0, ctx->stream, (const float4 *)records, n_records,
                               bp, keys_by_idx, range32, blocksums);
            LAUNCH_CHECK(ctx, "k_band_prepare");
            rc = scan_exclusive_u32(ctx, blocksums, blocksums, blocks, sorter->d_count);
            if (rc != SPLAT_OK) return rc;
            hipLaunchKernelGGL(k_band_compact, dim3(blocks), dim3(BAND_THREADS), 0, ctx->stream, keys_by_idx, range32, n_records,
                               blocksums, sorter->keys, sorter->payload);
            LAUNCH_CHECK(ctx, "k_band_compact");
            stage_end(ctx, SPLAT_STAGE_PROJECT);
        } else {
            rc = band_keys_device(ctx, sorter, records, n_records, width, height, tile, row0, row1);
            if (rc != SPLAT_OK) return rc;
        }
        // kept count -> host, without stalling the stream
        if (!sorter->pinned_count) {
            if (hipHostMalloc((void **)&sorter->pinned_count, 16, hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&sorter->count_event, hipEventDisableTiming) != hipSuccess)
                return ctx_fail(ctx, SPLAT_ERR_OOM, "band frame readback allocation");
        }
        HIP_TRY(ctx, hipMemcpyAsync(sorter->pinned_count, sorter->d_count, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(sorter->count_event, ctx->stream));
        sorter->count_pending = true;
        sorter->count_bound = bound;
        stage_begin(ctx, SPLAT_STAGE_SORT);
        rc = radix_sort_pairs(ctx, sorter->keys, sorter->payload, sorter->keys_b, sorter->payload_b, sorter->hist, bound, 0, 32,
                              &sorter->result_in_primary, 0, sorter->d_count);
        stage_end(ctx, SPLAT_STAGE_SORT);
        if (rc != SPLAT_OK) return rc;
        sorter->ran = true;
    }
    rc = binner_run(binner, records, n_records, splat_sort_sorted_payload(sorter), bound, width, height, row0, row1, range32,
                    n_records ? sorter->d_count : nullptr);
    if (rc != SPLAT_OK) return rc;
    void *indices = binner->pairs.result_in_primary ? binner->pairs.payload : binner->pairs.payload_b;
    splat_composite_cfg c2 = *cfg;
    c2.tile_row0 = row0;
    c2.tile_row1 = row1;
    return splat_composite(ctx, &c2, band_color, band_color_stride, normals, 1, records, indices, binner->counts, binner->offsets, width,
                           height, out_rgba8, out_rgba32f, consumed_dptr);
}

int splat_band_settle(splat_ctx *ctx, splat_sorter *sorter, splat_binner *binner, uint32_t *n_kept_host, uint64_t *pairs_host) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, sorter && binner);
    int rc = band_settle_count(ctx, sorter);
    if (rc != SPLAT_OK) {
        binner->have_last = false;
        (void)binner_settle(binner); // drop that frame's pair readback as well
        return rc;
    }
    rc = binner_settle(binner);
    if (rc != SPLAT_OK) return rc;
    if (sorter->kept_blocks) { // tile-first band frame: no kept-count bound to settle, the count is summed on demand
        if (pairs_host) *pairs_host = binner->total;
        return n_kept_host ? splat_band_kept(ctx, sorter, n_kept_host) : SPLAT_OK;
    }
    if (n_kept_host) *n_kept_host = sorter->last_count;
    if (pairs_host) *pairs_host = binner->total;
    return SPLAT_OK;
}

int splat_band_kept(splat_ctx *ctx, splat_sorter *sorter, uint32_t *n_kept_host) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, sorter && n_kept_host);
    int rc = ctx_ensure_pinned(ctx, 16);
    if (rc != SPLAT_OK) return rc;
    if (sorter->kept_blocks) { // tile-first band frame: sum the per-block kept counts now
        rc = scan_exclusive_u32(ctx, sorter->hist, sorter->hist, sorter->kept_blocks, sorter->d_count);
        if (rc != SPLAT_OK) return rc;
        sorter->kept_blocks = 0;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned, sorter->d_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *n_kept_host = *(volatile uint32_t *)ctx->pinned;
    return SPLAT_OK;
}

} // extern "C"

// pos_radius / color_opacity: vec4 per splat, *_stride float4s apart (2 and 2 with color = props + 16 bytes for the
// reference's interleaved records; 1 and 1 for separate planes)
static int render_frame_impl(splat_ctx *ctx, splat_sorter *sorter, splat_binner *binner, const splat_composite_cfg *cfg,
                             const float *uniforms, const void *props, uint32_t pos_stride, const void *color,
                             uint32_t color_stride, const void *normals, uint32_t n, uint32_t width, uint32_t height,
                             void *projected, void *out_rgba8, void *out_rgba32f) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, sorter && binner && cfg && uniforms && props && color && (normals || cfg->prelit));
    ARG_CHECK(ctx, cfg->tile_size == splat_bin_tile_size(binner));
    ARG_CHECK(ctx, cfg->footprint <= SPLAT_FOOTPRINT_DISC);
    // the ProjectedSplat records are the isotropic composite's input; a disc frame reads its disc records instead and
    // may leave them out (160 MB of stores per 5M splats that nothing reads)
    ARG_CHECK(ctx, projected || cfg->footprint == SPLAT_FOOTPRINT_DISC);
    // the oriented disc (SequentialRenderer's footprint): its projector needs the normals, its records live with the binner
    const bool disc = cfg->footprint == SPLAT_FOOTPRINT_DISC;
    // cfg->record_format says what the frame leaves in `projected` and composites from: the reference's ProjectedSplat
    // records, or the lit composite records (shade.h) — one gathered line per staged list entry instead of three
    // (a disc frame with SPLAT_RECORDS_LIT32: the lit colour rides behind each 32-byte disc record — 48-byte records, owned by the
    // binner like the plain disc records — and the composite gathers that one record per staged entry)
    ARG_CHECK(ctx, cfg->record_format == SPLAT_RECORDS_PROJECTED || cfg->record_format == SPLAT_RECORDS_LIT32);
    const bool lit = cfg->record_format == SPLAT_RECORDS_LIT32 && !disc, disc_lit = cfg->record_format == SPLAT_RECORDS_LIT32 && disc;
    ARG_CHECK(ctx, !disc || (normals && (((uintptr_t)normals) & 15) == 0));
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
    if ((!projected || lit || disc_lit) && n > 0 && !fast)
        return ctx_fail(ctx, SPLAT_ERR_INVALID, "splat_render_frame: screens beyond 256 x 256 tiles bin from ProjectedSplat records: pass a "
                                                "buffer and cfg->record_format = SPLAT_RECORDS_PROJECTED");
    int rc = SPLAT_OK;
    uint32_t *range32 = nullptr;
    // (a strict band: the projector skips what provably cannot reach it; those splats' records are then not written)
    const BinParams bp = {width, height, tile, ntx, nty, row0, row1, (row0 > 0 || row1 < nty) ? 1u : 0u};
    if (fast) {
        rc = binner_reserve_range32(binner, n);
        if (rc != SPLAT_OK) return rc;
        range32 = binner->range32;
    }
    ARG_CHECK(ctx, (((uintptr_t)props | (uintptr_t)projected) & 15) == 0);
    const bool tile_first = fast && frame_order(binner) == SPLAT_FRAME_TILE_FIRST;
    TfHistOut ho = {};
    // a strict band of a fraction of the screen (the exchange-free multi-GPU cut): the projector leaves the splats that can reach
    // it compacted per group of 4096 (project.hip: k_project_hist_bandc) and the scatter runs over those only
    const bool band_compact = tile_first && bp.skip_outside && band_compacting(row0, row1, nty);
    if (tile_first) { // the projector also counts each 1024-splat block's pairs per low tile-id digit
        rc = binner_reserve(binner, ntx * nty, n);
        if (rc != SPLAT_OK) return rc;
        // small frames: 256-splat blocks, so that the first pass is more than a handful of workgroups
        const uint32_t block = band_compact ? BTC_GROUP : (n <= TF_SMALL_FRAME_SPLATS && !bp.skip_outside) ? TF_BLOCK_SMALL : TF_BLOCK_LARGE;
        ho = {binner->tf_hist, binner->blocksums, binner->d_total + 1, (1u << tile_id_low_bits(ntx * nty)) - 1u, div_up(n, block), block};
        binner->tf_block = block;
        if (band_compact) {
            rc = binner_reserve_band_idx(ctx, binner, n);
            if (rc != SPLAT_OK) return rc;
            ho.cidx = binner->band_idx;
            ho.kept_groups = sorter->hist; // (free in the tile-first order: the sorter's own passes do not run)
        }
    }
    if (disc && n > binner->discs_cap) { // (48 bytes per splat: room for the lit records)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (binner->discs) (void)hipFree(binner->discs);
        binner->discs = nullptr;
        binner->discs_cap = 0;
        if (hipMalloc(&binner->discs, (size_t)n * 48 + 256) != hipSuccess) return ctx_fail(ctx, SPLAT_ERR_OOM, "disc records hipMalloc");
        binner->discs_cap = n;
    }
    if (ctx->timing) { // (before binner_run promises a report: nothing that can fail may stand between that promise and the composite_launch that keeps it)
        rc = ctx_ensure_consumed(ctx, ntx * nty);
        if (rc != SPLAT_OK) return rc;
    }
    // (the payload array is not written: payload = splat index)
    const LitIO lio = {(const float4 *)color, (const float4 *)normals, color_stride, 1u, cfg->prelit,
                       lit ? (float4 *)projected : disc_lit ? (float4 *)binner->discs : nullptr};
    rc = project_launch(ctx, uniforms, props, pos_stride, n, 0, lit ? nullptr : projected, splat_sort_keys(sorter), nullptr, n, range32, &bp,
                        tile_first ? &ho : nullptr, normals, 1, disc ? binner->discs : nullptr, &lio);
    if (rc != SPLAT_OK) return rc;
    binner->tf_hist_ready = tile_first;
    binner->tf_cidx = band_compact ? binner->band_idx : nullptr;
    binner->tf_kept = band_compact ? sorter->hist : nullptr;
    // (with per-index tile ranges the binner never reads the records; it only wants a non-null pointer)
    const void *bin_records = projected ? projected : (const void *)binner->discs;
    if (tile_first) {
        // bin in index order, depth-sort per tile: no global sort, no gather (tile_first.hip)
        sorter->ran = false; // the sorter holds this frame's unsorted depth keys
        rc = binner_run(binner, bin_records, n, nullptr, n, width, height, row0, row1, range32, nullptr, sorter->keys);
        if (rc != SPLAT_OK) return rc;
    } else {
        stage_begin(ctx, SPLAT_STAGE_SORT);
        rc = radix_sort_pairs(ctx, sorter->keys, sorter->payload, sorter->keys_b, sorter->payload_b, sorter->hist, n, 0, 32,
                              &sorter->result_in_primary, 0, nullptr, true); // RadixSorter.sort()
        stage_end(ctx, SPLAT_STAGE_SORT);
        if (rc != SPLAT_OK) return rc;
        sorter->ran = true;
        rc = binner_run(binner, bin_records, n, splat_sort_sorted_payload(sorter), n, width, height, row0, row1, range32);
        if (rc != SPLAT_OK) return rc;
    }
    // (fields, not the public getters: those wait for a sync-free frame's pair total to come back)
    void *counts = binner->counts, *offsets = binner->offsets;
    void *indices = binner->pairs.result_in_primary ? binner->pairs.payload : binner->pairs.payload_b;
    // (n == 0: every list is empty and no record is read; the composite only wants a non-null pointer)
    const void *records = disc ? (n ? (const void *)binner->discs : (projected ? projected : (const void *)counts)) : projected;
    uint32_t *report = binner->report_for_composite; // (tile-first frames: the frame's last kernel reports it: common.h)
    binner->report_for_composite = nullptr;
    return composite_launch(ctx, cfg, color, color_stride, normals, 1, records, indices, counts, offsets, width, height, out_rgba8,
                            out_rgba32f, (ctx->timing && (ctx->timing_mask & SPLAT_TIMING_COUNT_ENTRIES)) ? (void *)ctx->d_consumed : nullptr, binner->d_total,
                            report, binner->report_seq);
}

extern "C" {

int splat_render_frame(splat_ctx *ctx, splat_sorter *sorter, splat_binner *binner, const splat_composite_cfg *cfg,
                       const float *uniforms, const void *props, const void *normals, uint32_t n, uint32_t width,
                       uint32_t height, void *projected, void *out_rgba8, void *out_rgba32f) {
    // the reference's interleaved records: colour is the second vec4 of each
    return render_frame_impl(ctx, sorter, binner, cfg, uniforms, props, 2, props ? (const char *)props + 16 : nullptr, 2, normals, n,
                             width, height, projected, out_rgba8, out_rgba32f);
}

int splat_render_frame_planes(splat_ctx *ctx, splat_sorter *sorter, splat_binner *binner, const splat_composite_cfg *cfg,
                              const float *uniforms, const void *pos_radius, const void *color_opacity, const void *normals,
                              uint32_t n, uint32_t width, uint32_t height, void *projected, void *out_rgba8, void *out_rgba32f) {
    return render_frame_impl(ctx, sorter, binner, cfg, uniforms, pos_radius, 1, color_opacity, 1, normals, n, width, height, projected,
                             out_rgba8, out_rgba32f);
}

} // extern "C"
