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
#include <cstring>
#include "tile_range.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sched.h>

constexpr uint32_t BIN_THREADS = 256;
constexpr uint32_t BIN_PER_THREAD = 4;
constexpr uint32_t BIN_BLOCK = BIN_THREADS * BIN_PER_THREAD; // sorted positions per workgroup
constexpr uint32_t BIN_STAGE_PAIRS = 4096;                   // pairs staged in LDS for coalesced stores (32 KB)

__device__ __forceinline__ uint32_t range_hits(uint2 r) {
    const uint32_t tx0 = r.x & 0xffffu, tx1 = r.x >> 16, ty0 = r.y & 0xffffu, ty1 = r.y >> 16;
    return (tx0 > tx1 || ty0 > ty1) ? 0u : (tx1 - tx0 + 1) * (ty1 - ty0 + 1);
}

__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *wsum) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    return wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// count: every sorted position gets its clamped tile range (gathered per splat: 16 B of bounds, or
// the 4 B range32 the projector wrote on the frame path) and every 1024-position block its pair count.
// The gather is the cost: 5M random lines at ~64 B per request through each CU's ~10 B/clk L1 fill
// path = 52 us at best, 80 us measured; 2 or 4 positions per thread, or streaming the 20 MB table
// into the memory-side cache just before, change nothing.
template <bool FROM_RANGE32>
__global__ __launch_bounds__(BIN_THREADS) void k_bin_count(const float4 *__restrict__ projected,
                                                           const uint32_t *__restrict__ range32, uint32_t n_splats,
                                                           const uint32_t *__restrict__ sorted, uint32_t n_sorted_host,
                                                           const uint32_t *__restrict__ n_sorted_dev, BinParams bp,
                                                           uint2 *__restrict__ ranges, uint32_t *__restrict__ blocksums,
                                                           uint32_t *__restrict__ overflow_flag) {
    __shared__ uint32_t wsum[4];
    uint32_t local = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *overflow_flag = 0; // set by k_bin_expand of this frame if it clips
    uint32_t n_sorted = n_sorted_host;
    if (n_sorted_dev) n_sorted = min(*n_sorted_dev, n_sorted_host); // sync-free callers: count on the device
#pragma unroll
    for (uint32_t k = 0; k < BIN_PER_THREAD; ++k) {
        const uint32_t i = blockIdx.x * BIN_BLOCK + k * BIN_THREADS + threadIdx.x;
        if (i < n_sorted) {
            const uint32_t s = sorted[i];
            uint32_t tx0 = 1, tx1 = 0, ty0 = 1, ty1 = 0;
            if (s < n_splats) { // 0xFFFFFFFF padding bins nowhere
                if (FROM_RANGE32) {
                    const uint32_t r = range32[s];
                    tx0 = r & 0xffu; tx1 = (r >> 8) & 0xffu; ty0 = (r >> 16) & 0xffu; ty1 = r >> 24;
                } else {
                    const float4 b = projected[(size_t)s * 2];
                    if (!tile_range(b, bp.width, bp.height, bp.tile, bp.ntx, bp.nty, bp.row0, bp.row1, tx0, tx1, ty0, ty1)) {
                        tx0 = 1; tx1 = 0; ty0 = 1; ty1 = 0;
                    }
                }
            }
            const uint2 r = make_uint2(tx0 | (tx1 << 16), ty0 | (ty1 << 16));
            ranges[i] = r;
            local += range_hits(r);
        } else if (i < n_sorted_host) {
            ranges[i] = make_uint2(1u, 1u); // past the device-side count: empty
        }
    }
    const uint32_t total = block_sum(local, wsum);
    if (threadIdx.x == 0) blocksums[blockIdx.x] = total;
}

// expand: the (tile id, splat idx) pairs of a block land in [block_base, block_base + block_total) in
// sorted order: position-major (k * 256 + thread), then row-major over the splat's tile rectangle.
// Small blocks are staged in LDS so that the global stores are contiguous runs.
__global__ __launch_bounds__(BIN_THREADS) void k_bin_expand(const uint32_t *__restrict__ sorted, uint32_t n_sorted,
                                                            const uint2 *__restrict__ ranges,
                                                            const uint32_t *__restrict__ block_base, uint32_t ntx,
                                                            uint32_t pair_limit, uint32_t *__restrict__ overflow,
                                                            uint32_t *__restrict__ pair_tile, uint32_t *__restrict__ pair_splat) {
    __shared__ uint32_t wsum[4];
    __shared__ uint2 stage[BIN_STAGE_PAIRS];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint2 r[BIN_PER_THREAD];
    uint32_t s[BIN_PER_THREAD], h[BIN_PER_THREAD];
#pragma unroll
    for (uint32_t k = 0; k < BIN_PER_THREAD; ++k) {
        const uint32_t i = blockIdx.x * BIN_BLOCK + k * BIN_THREADS + tid;
        r[k] = make_uint2(1u, 1u); // empty
        s[k] = 0;
        if (i < n_sorted) {
            r[k] = ranges[i];
            s[k] = sorted[i];
        }
        h[k] = range_hits(r[k]);
    }
    // exclusive offsets inside the block, position-major: all k = 0 positions, then all k = 1 ...
    uint32_t off[BIN_PER_THREAD];
    uint32_t carry = 0;
#pragma unroll
    for (uint32_t k = 0; k < BIN_PER_THREAD; ++k) {
        uint32_t incl = h[k];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if ((int)lane >= d) incl += t;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        const uint32_t s0 = wsum[0], s1 = wsum[1], s2 = wsum[2], s3 = wsum[3];
        __syncthreads();
        off[k] = carry + (w > 0 ? s0 : 0u) + (w > 1 ? s1 : 0u) + (w > 2 ? s2 : 0u) + incl - h[k];
        carry += s0 + s1 + s2 + s3;
    }
    const uint32_t total = carry, base = block_base[blockIdx.x];
    if (total == 0) return;
    // sync-free frames: pairs at or past the limit are dropped (and flagged).  Everything below the
    // limit is still written, so the pair buffers never hold stale tile ids / splat indices that a
    // later kernel could index with.
    const bool clipped = base + total > pair_limit || base + total < base;
    if (clipped && tid == 0) atomicOr(overflow, 1u);
    if (base >= pair_limit) return;
    const bool staged = total <= BIN_STAGE_PAIRS;
#pragma unroll
    for (uint32_t k = 0; k < BIN_PER_THREAD; ++k) {
        if (h[k] == 0) continue;
        const uint32_t tx0 = r[k].x & 0xffffu, tx1 = r[k].x >> 16, ty0 = r[k].y & 0xffffu, ty1 = r[k].y >> 16;
        uint32_t o = off[k];
        for (uint32_t ty = ty0; ty <= ty1; ++ty)
            for (uint32_t tx = tx0; tx <= tx1; ++tx) {
                if (staged) {
                    stage[o] = make_uint2(ty * ntx + tx, s[k]);
                } else if (!clipped || base + o < pair_limit) {
                    pair_tile[base + o] = ty * ntx + tx;
                    pair_splat[base + o] = s[k];
                }
                ++o;
            }
    }
    if (staged) {
        __syncthreads();
        const uint32_t keep = clipped ? pair_limit - base : total;
        for (uint32_t o = tid; o < keep && o < total; o += BIN_THREADS) {
            const uint2 p = stage[o];
            pair_tile[base + o] = p.x;
            pair_splat[base + o] = p.y;
        }
    }
}

// Tile offsets and counts from the tile-sorted pair keys: offsets[t] = index of the first pair whose
// tile id is >= t (the same values as the exclusive scan of the counts, TileBinner.ts:452-459).
// One WAVE per tile does a 65-ary search: every step the 64 lanes probe 64 evenly spaced positions of
// the remaining range and a ballot picks the sub-range, so 11M pairs take 4 dependent loads instead
// of the 24 of a binary search.  No atomics (the first version's 11M global atomic increments cost
// 1.03 ms at C2) and no serial gap filling (a band of a multi-GPU frame has thousands of empty tiles
// before its first pair).
__global__ __launch_bounds__(256) void k_tile_offsets(const uint32_t *__restrict__ sorted_tiles, uint32_t pairs_host,
                                                      const uint32_t *__restrict__ pairs_dev, uint32_t tiles,
                                                      uint32_t *__restrict__ offsets, const uint32_t *__restrict__ d_total,
                                                      uint32_t *report, uint32_t seq) {
    const uint32_t t = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (report && blockIdx.x == 0 && threadIdx.x == 0) tile_report(d_total, report, seq);
    if (t > tiles) return;
    const uint32_t pairs = pairs_dev ? min(*pairs_dev, pairs_host) : pairs_host;
    uint32_t lo = 0, hi = pairs; // invariant: every pair before lo has id < t, every pair from hi on has id >= t
    while (hi - lo > 64) {
        const uint32_t step = (hi - lo + 64) / 65; // 64 probes cut the range into 65 parts of at most `step`
        const uint32_t pos = lo + (lane + 1) * step;
        const bool less = pos < hi && sorted_tiles[pos] < t;
        const uint32_t k = (uint32_t)__popcll(__ballot(less)); // probes are ordered: the first k are "less"
        const uint32_t nlo = (k == 0) ? lo : lo + k * step + 1;
        const uint32_t nhi = (lo + (k + 1) * step < hi) ? lo + (k + 1) * step : hi;
        lo = nlo;
        hi = nhi;
    }
    const uint32_t pos = lo + lane;
    const bool less = pos < hi && sorted_tiles[pos] < t;
    const uint32_t first = lo + (uint32_t)__popcll(__ballot(less));
    if (lane == 0) offsets[t] = first; // offsets[tiles] = pairs
}

__global__ __launch_bounds__(256) void k_tile_counts(const uint32_t *__restrict__ offsets, uint32_t tiles,
                                                     uint32_t *__restrict__ counts) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < tiles) counts[t] = offsets[t + 1] - offsets[t];
}

// PerTileSorter's job, as a check instead of a sort (src/PerTileSorter.ts:66-122 re-sorts every tile's
// list by depth; here the lists leave the binner already in (depth key, index) order): pair i and
// its successor, when they belong to the same tile, must be strictly increasing in (key, index).
__global__ __launch_bounds__(256) void k_validate_tile_order(const float4 *__restrict__ projected,
                                                             const uint32_t *__restrict__ offsets, uint32_t tiles,
                                                             const uint32_t *__restrict__ indices, uint32_t pairs,
                                                             unsigned long long *__restrict__ violations) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i + 1 >= pairs) return;
    // tile of pair i: the last t with offsets[t] <= i (empty tiles share an offset with their successor)
    uint32_t lo = 0, hi = tiles; // offsets[tiles] = pairs > i
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    if (i + 1 >= offsets[lo + 1]) return; // i is the last entry of its tile
    const uint32_t a = indices[i], b = indices[i + 1];
    const uint32_t ka = __float_as_uint(reinterpret_cast<const float *>(projected)[(size_t)a * 8 + 4]);
    const uint32_t kb = __float_as_uint(reinterpret_cast<const float *>(projected)[(size_t)b * 8 + 4]);
    const uint32_t sa = ka ^ (((ka >> 31) == 1u) ? 0xffffffffu : 0x80000000u), sb = kb ^ (((kb >> 31) == 1u) ? 0xffffffffu : 0x80000000u);
    if (sa > sb || (sa == sb && a >= b)) atomicAdd(violations, 1ull);
}

static void binner_free(splat_binner *b) {
    if (b->counts) (void)hipFree(b->counts);
    if (b->offsets) (void)hipFree(b->offsets);
    if (b->blocksums) (void)hipFree(b->blocksums);
    if (b->ranges) (void)hipFree(b->ranges);
    if (b->tf_hist) (void)hipFree(b->tf_hist);
    b->tf_hist = nullptr;
    b->counts = b->offsets = b->blocksums = nullptr;
    b->ranges = nullptr;
    b->tiles_cap = b->splats_cap = 0;
}

static void sorter_free_members(splat_sorter *s) {
    if (s->keys) (void)hipFree(s->keys); // payload / payload_b live inside the keys / keys_b allocations
    if (s->keys_b) (void)hipFree(s->keys_b);
    if (s->hist) (void)hipFree(s->hist);
    if (s->d_count) (void)hipFree(s->d_count);
    s->keys = s->keys_b = s->payload = s->payload_b = s->hist = s->d_count = nullptr;
    s->capacity = 0;
}

int binner_reserve_range32(splat_binner *b, uint32_t n_splats) {
    splat_ctx *ctx = b->ctx;
    if (n_splats <= b->range32_cap) return SPLAT_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (b->range32) (void)hipFree(b->range32);
    b->range32 = nullptr;
    b->range32_cap = 0;
    if (hipMalloc((void **)&b->range32, (size_t)n_splats * 4 + 16) != hipSuccess) return ctx_fail(ctx, SPLAT_ERR_OOM, "binner hipMalloc");
    b->range32_cap = n_splats;
    return SPLAT_OK;
}

static void binner_free_wide(splat_binner *b) {
    if (b->wide_a) (void)hipFree(b->wide_a);
    if (b->wide_b) (void)hipFree(b->wide_b);
    if (b->tf_hi) (void)hipFree(b->tf_hi);
    if (b->tf2_hist) (void)hipFree(b->tf2_hist);
    if (b->tf_runs_mem) (void)hipFree(b->tf_runs_mem);
    b->wide_a = b->wide_b = nullptr;
    b->tf_hi = nullptr;
    b->tf2_hist = b->tf_runs_mem = nullptr;
    b->tf_runs = TfRuns{};
    b->wide_cap = 0;
}

// the tile-first path's pair buffers, sized with the pair capacity: the first pass's output (values + one byte of
// tile id per pair, with room for the padding that aligns its runs), the second pass's output and workspace
static int binner_reserve_wide(splat_binner *b) {
    splat_ctx *ctx = b->ctx;
    if (b->pairs.capacity <= b->wide_cap) return SPLAT_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    binner_free_wide(b);
    const size_t cap = b->pairs.capacity;
    const size_t parts = (size_t)div_up(b->pairs.capacity, TF_RUN_ALIGN) + 256; // >= tf2_parts_bound(any pair count, any digit split)
    const size_t runs_bytes = (256 + 256 + 4) * 4 + parts + 16;
    if (hipMalloc((void **)&b->wide_a, (cap + TF_RUN_SLACK) * 8 + 256) != hipSuccess || hipMalloc((void **)&b->wide_b, cap * 8 + 256) != hipSuccess ||
        hipMalloc((void **)&b->tf_hi, cap + TF_RUN_SLACK + 256) != hipSuccess ||
        hipMalloc((void **)&b->tf2_hist, (256 * parts + 256) * 4) != hipSuccess || hipMalloc((void **)&b->tf_runs_mem, runs_bytes) != hipSuccess) {
        binner_free_wide(b);
        return ctx_fail(ctx, SPLAT_ERR_OOM, "binner hipMalloc (pair values)");
    }
    b->tf_runs = {b->tf_runs_mem, b->tf_runs_mem + 256, reinterpret_cast<uint8_t *>(b->tf_runs_mem + 516), b->tf_runs_mem + 512};
    b->wide_cap = b->pairs.capacity;
    return SPLAT_OK;
}

int binner_reserve(splat_binner *b, uint32_t tiles, uint32_t n_sorted) {
    splat_ctx *ctx = b->ctx;
    if (tiles <= b->tiles_cap && n_sorted <= b->splats_cap) return SPLAT_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t tc = tiles > b->tiles_cap ? tiles : b->tiles_cap;
    uint32_t sc = n_sorted > b->splats_cap ? n_sorted : b->splats_cap;
    binner_free(b);
    if (hipMalloc((void **)&b->counts, (size_t)tc * 4 + 16) != hipSuccess ||
        hipMalloc((void **)&b->offsets, (size_t)tc * 4 + 16) != hipSuccess ||
        hipMalloc((void **)&b->blocksums, ((size_t)div_up(sc, BIN_BLOCK) + TF_SMALL_FRAME_SPLATS / TF_BLOCK_SMALL) * 4 + 16) != hipSuccess ||
        hipMalloc((void **)&b->ranges, (size_t)sc * 8 + 16) != hipSuccess ||
        // (first-pass histograms: one column per 1024-splat block, or per 256-splat block of a small frame)
        hipMalloc((void **)&b->tf_hist, ((size_t)256 * (div_up(sc, BIN_BLOCK) + TF_SMALL_FRAME_SPLATS / TF_BLOCK_SMALL) + 256) * 4) != hipSuccess) {
        binner_free(b);
        return ctx_fail(ctx, SPLAT_ERR_OOM, "binner hipMalloc");
    }
    b->tiles_cap = tc;
    b->splats_cap = sc;
    return SPLAT_OK;
}

// binSplats.  range32 (optional): per splat index, the packed range the projector computed with the
// same BinParams (frame path); otherwise ranges are derived from the projected bounds here.
// depth_keys (frame path, with range32): per splat index, its depth key — selects the tile-first
// order of work (tile_first.hip): `sorted` is then unused, the lists are depth-sorted per tile.
int binner_run(splat_binner *b, const void *projected, uint32_t n_splats, const void *sorted, uint32_t n_sorted, uint32_t width,
               uint32_t height, uint32_t tile_row0, uint32_t tile_row1, const uint32_t *range32, const uint32_t *n_sorted_dev,
               const uint32_t *depth_keys) {
    splat_ctx *ctx = b->ctx;
    const bool tile_first = depth_keys != nullptr;
    // what the caller prepared for THIS run (its projector's histogram; a band's compacted splats) is taken and cleared before
    // anything can return: a call that fails below must not leave it for another frame function's run
    const bool hist_ready = b->tf_hist_ready;
    const uint32_t *tf_cidx = b->tf_cidx, *tf_kept = b->tf_kept;
    b->tf_hist_ready = false;
    b->tf_cidx = nullptr;
    b->tf_kept = nullptr;
    ARG_CHECK(ctx, width >= 1 && height >= 1);
    ARG_CHECK(ctx, n_sorted == 0 || (projected && (sorted || tile_first)));
    ARG_CHECK(ctx, !tile_first || (range32 && n_sorted == n_splats && !n_sorted_dev));
    const uint32_t ntx = div_up(width, b->tile), nty = div_up(height, b->tile); // GPUTileBinner.ts:198-200
    ARG_CHECK(ctx, ntx <= 65535 && nty <= 65535 && (uint64_t)ntx * nty <= (1u << 24));
    const uint32_t tiles = ntx * nty;
    if (tile_row1 > nty) tile_row1 = nty;
    if (tile_row0 > tile_row1) tile_row0 = tile_row1;
    const uint32_t blocks = div_up(n_sorted, BIN_BLOCK);

    int rc = binner_reserve(b, tiles, n_sorted);
    if (rc != SPLAT_OK) return rc;
    b->ntx = ntx;
    b->nty = nty;
    b->ran = false;
    const BinParams bp = {width, height, b->tile, ntx, nty, tile_row0, tile_row1};

    // the previous frame's async readback (if any): learn its pair total, detect overflow
    rc = binner_settle(b);
    if (rc != SPLAT_OK) return rc;

    // tile ids up to 16 bits, sorted in two passes with the bits split evenly (13 bits -> 6 + 7) rather
    // than 8 + 5: a pass scatters in digit runs, and 64 + 128 bins give longer runs than 256 + 32
    const uint32_t tf_bits = tile_id_bits(tiles), tf_lo_bits = tile_id_low_bits(tiles);
    ARG_CHECK(ctx, !tile_first || hist_ready); // the caller's projector / band prepare has counted the first pass's histogram

    stage_begin(ctx, SPLAT_STAGE_BIN);
    uint32_t total32 = 0;
    bool async = false;
    if (n_sorted > 0) {
        if (tile_first) {
            // (tf_hist: per 1024-splat block its pairs per low tile-id digit, the first sort pass's histogram)
            rc = radix_rowscan_launch(ctx, b->tf_hist, div_up(n_splats, b->tf_block), 1u << tf_lo_bits);
            if (rc != SPLAT_OK) return rc;
        } else if (range32)
            hipLaunchKernelGGL(k_bin_count<true>, dim3(blocks), dim3(BIN_THREADS), 0, ctx->stream, (const float4 *)projected, range32,
                               n_splats, (const uint32_t *)sorted, n_sorted, n_sorted_dev, bp, b->ranges, b->blocksums, b->d_total + 1);
        else
            hipLaunchKernelGGL(k_bin_count<false>, dim3(blocks), dim3(BIN_THREADS), 0, ctx->stream, (const float4 *)projected,
                               nullptr, n_splats, (const uint32_t *)sorted, n_sorted, n_sorted_dev, bp, b->ranges, b->blocksums, b->d_total + 1);
        LAUNCH_CHECK(ctx, "k_bin_count");
        const uint64_t want_async = (uint64_t)b->last_total + b->last_total / 8 + 4096; // 12.5 % frame-to-frame growth
        async = b->allow_async && b->have_last && b->last_total > 0 && want_async <= b->pairs.capacity;
        // block bases + pair total (PrefixSumScanner.scan :296-303).  A sync-free tile-first frame needs
        // neither: positions come from the digit histogram and k_tf_scatter sums the total itself.
        if (!(tile_first && async)) {
            // (tile-first: one count per block of the projector's histogram, which is 256 splats for small frames)
            rc = scan_exclusive_u32(ctx, b->blocksums, b->blocksums, tile_first ? div_up(n_splats, b->tf_block) : blocks, b->d_total);
            if (rc != SPLAT_OK) return rc;
        }
        if (async) {
            b->pair_limit = (uint32_t)want_async;
            total32 = b->pair_limit; // grid bound; the real count is read on the device
        } else {
            // the one host round trip of a (first / growing) frame: the pair total sizes the fill
            // (the reference reads back all T counts here: GPUTileBinner.ts:244-263)
            HIP_TRY(ctx, hipMemcpyAsync(b->pinned, b->d_total, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            total32 = *(volatile uint32_t *)b->pinned;
            b->last_total = total32;
            b->have_last = true;
            if (total32 > b->pairs.capacity) {
                uint64_t want = (uint64_t)total32 + total32 / 2 + 8192; // headroom for sync-free frames to come
                if (want > 0x3ffff000ull) want = 0x3ffff000ull;
                if (total32 > want) return ctx_fail(ctx, SPLAT_ERR_CAPACITY, "binSplats: more than 2^30 tile-splat pairs");
                rc = sorter_reserve(&b->pairs, (uint32_t)want);
                if (rc != SPLAT_OK) return rc;
            }
            b->pair_limit = total32;
        }
    }
    b->total = async ? b->last_total : total32;
    if (total32 > 0 && tile_first) {
        rc = binner_reserve_wide(b);
        if (rc != SPLAT_OK) return rc;
        // first pass of the tile-id sort, fused with the expansion (tile_first.hip); later kernels take
        // their pair count from d_total[2], which k_tf_scatter sets (0 if the pairs do not fit)
        const uint32_t tf_hi_bits = tf_bits - tf_lo_bits;
        // Every tile-first frame is reported — {pair total, flags, sequence number} into host-mapped words — by its LAST
        // kernel, the composite (report_for_composite): the flags then include the per-tile sort's order check, which is
        // what allows these kernels to rank with returning LDS atomics (common.h: rank_atomic_ok).  The report is examined
        // at the next call (binner_settle).
        stage_begin(ctx, SPLAT_STAGE_BIN_SCATTER); // (the row scan in front of it ran above: a few us not in this interval)
        rc = tf_scatter_launch(ctx, range32, depth_keys, n_splats, ntx, (1u << tf_lo_bits) - 1u, b->tf_hist, b->d_total, b->pair_limit,
                               b->d_total + 1, b->tf_hi, b->wide_a, b->tf_block, tf_lo_bits, tf_hi_bits > 0, &b->tf_runs, b->offsets, tiles,
                               nullptr, 0u, tf_cidx, tf_kept);
        stage_end(ctx, SPLAT_STAGE_BIN_SCATTER);
        if (rc != SPLAT_OK) return rc;
        // second pass (high digit) into wide_b, and the tile offsets out of its histogram; a screen of at most 256
        // tiles is sorted by the first pass alone (which then writes the offsets too)
        const bool primary = tf_hi_bits == 0;
        stage_begin(ctx, SPLAT_STAGE_BIN_PASS2);
        rc = tf_second_pass_launch(ctx, b->tf_hi, b->wide_a, b->wide_b, &b->tf_runs, total32, tiles, tf_lo_bits, tf_hi_bits, b->tf2_hist,
                                   b->offsets, b->d_total);
        stage_end(ctx, SPLAT_STAGE_BIN_PASS2);
        if (rc != SPLAT_OK) return rc;
        // PerTileSorter: depth order inside every tile; the index lists land in the primary payload array
        // (its first launch also writes the tile counts); every list's order is checked there
        // (the frame's mean list length picks the short class's size — on a band of a few thousand tiles, a multi-GPU rank's,
        //  every class is one round of workgroups and the smallest short class is the fastest: measured at G = 2, 4, 8,
        //  profiles/r04_u_tile_sort_classes.txt)
        const uint32_t band_tiles = tile_row1 > tile_row0 ? (tile_row1 - tile_row0) * ntx : 1u;
        const uint64_t band_key = ((uint64_t)tile_row0 << 48) ^ ((uint64_t)tile_row1 << 32) ^ tiles; // (same screen, same band)
        stage_begin(ctx, SPLAT_STAGE_BIN_TILE_SORT);
        rc = tile_sort_launch(ctx, b->offsets, tiles, primary ? b->wide_a : b->wide_b, primary ? b->wide_b : b->wide_a, b->pairs.payload,
                              b->counts, b->d_total + 1, band_tiles >= 6144u ? (uint32_t)(b->total / band_tiles) : 0u, band_tiles,
                              (async && b->last_band_key == band_key) ? b->last_long_tiles : 0xffffffffu, &b->last_short_class);
        stage_end(ctx, SPLAT_STAGE_BIN_TILE_SORT);
        if (rc != SPLAT_OK) return rc;
        b->pairs.result_in_primary = true;
        b->pending_tile_first = true;
        b->last_band_key = band_key;
        b->report_for_composite = b->pinned_dev;
        b->report_seq = ++b->seq;
        b->pending = true;
    } else if (total32 > 0) {
        hipLaunchKernelGGL(k_bin_expand, dim3(blocks), dim3(BIN_THREADS), 0, ctx->stream, (const uint32_t *)sorted, n_sorted,
                           b->ranges, b->blocksums, ntx, b->pair_limit, b->d_total + 1, b->pairs.keys, b->pairs.payload);
        LAUNCH_CHECK(ctx, "k_bin_expand");
        uint32_t bits = 1;
        while ((1u << bits) < tiles) ++bits;
        const uint32_t *p_dev = async ? b->d_total : nullptr;
        // tile ids up to 16 bits: two passes with the bits split evenly (13 bits -> 6 + 7) rather than
        // 8 + 5: a pass scatters in digit runs, and 64 + 128 bins give longer runs than 256 + 32
        // measured at C2 (13 bits): 5+8 0.627, 6+7 0.619, 7+6 0.629, 8+5 0.649 ms/frame
        const uint32_t lo_bits = bits <= 8 ? bits : bits / 2;
        rc = radix_sort_pairs(ctx, b->pairs.keys, b->pairs.payload, b->pairs.keys_b, b->pairs.payload_b, b->pairs.hist, total32, 0,
                              bits, &b->pairs.result_in_primary, 0, p_dev, false, lo_bits);
        if (rc != SPLAT_OK) return rc;
        const uint32_t *sorted_tiles = b->pairs.result_in_primary ? b->pairs.keys : b->pairs.keys_b;
        hipLaunchKernelGGL(k_tile_offsets, dim3(div_up(tiles + 1, 4)), dim3(256), 0, ctx->stream, sorted_tiles, total32, p_dev,
                           tiles, b->offsets, b->d_total, async ? b->pinned_dev : nullptr, async ? ++b->seq : 0u);
        LAUNCH_CHECK(ctx, "k_tile_offsets");
        hipLaunchKernelGGL(k_tile_counts, dim3(div_up(tiles, 256)), dim3(256), 0, ctx->stream, b->offsets, tiles, b->counts);
        LAUNCH_CHECK(ctx, "k_tile_counts");
        if (async) b->pending = true; // k_tile_offsets reported {total, overflow, seq} into b->pinned; examined at the next call
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

int binner_settle(splat_binner *b) {
    if (!b->pending) return SPLAT_OK;
    splat_ctx *ctx = b->ctx;
    b->pending = false;
    // wait for that frame's report word (normally long there: it was launched a frame ago)
    volatile uint32_t *rep = (volatile uint32_t *)b->pinned;
    // The stream is only queried (is it in error? has everything finished without the report?) after
    // 5 ms of waiting and then every millisecond: hipStreamQuery puts a marker into the queue, which
    // showed up as ~6 us of idle GPU per frame when it ran on every wait.
    auto next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(5);
    for (uint32_t spins = 0; rep[2] != b->seq; ++spins) {
        if ((spins & 63u) != 63u) continue;
        if (std::chrono::steady_clock::now() < next_query) {
            sched_yield();
            continue;
        }
        next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(1);
        const hipError_t q = hipStreamQuery(ctx->stream);
        if (q == hipSuccess) { // everything launched has finished
            if (rep[2] != b->seq) return ctx_fail(ctx, SPLAT_ERR_STATE, "binner: the pair-total report of the previous frame never arrived");
            break;
        }
        if (q != hipErrorNotReady) { // the stream is in error: the report will never come, do not spin on it
            b->have_last = false;
            return ctx_fail(ctx, SPLAT_ERR_HIP, "binner: waiting for the previous frame's pair total", q);
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const uint32_t total = ((volatile uint32_t *)b->pinned)[0], flags = ((volatile uint32_t *)b->pinned)[1];
    // (tile-first frames: how many tiles that frame's short size class could not hold — what the next frame's per-tile sort
    // launches by: tile_sort_launch)
    b->last_long_tiles = b->pending_tile_first ? ((volatile uint32_t *)b->pinned)[3] : 0xffffffffu;
    b->pending_tile_first = false;
    b->last_total = total;
    b->have_last = true;
    b->total = total;
    if (flags & FRAME_FLAG_ORDER) {
        // a tile's list failed k_tile_sort's order check: some pass ranked equal digits out of lane order.  That frame's
        // lists (and its image) are wrong; from here on this context ranks with ballots, which assume nothing.
        b->ran = false;
        ctx->order_faults++;
        if (ctx->rank_policy != RANK_BALLOT) // said ONCE, where an operator sees it: from here on every frame costs the ballot path's +12..19 %
            fprintf(stderr, "[splat] a frame's tile lists failed the per-tile order check (tile lists are re-rendered): this context ranks with "
                            "ballots from now on (splat_rank_status: policy ballot, orderFaults %u)\n", ctx->order_faults);
        ctx->rank_policy = RANK_BALLOT;
        return ctx_fail(ctx, SPLAT_ERR_RETRY,
                        "the previous frame's tile lists failed the order check (ranking with returning LDS atomics was not in lane "
                        "order): the context now ranks with ballots; render that frame again");
    }
    if ((flags & FRAME_FLAG_OVERFLOW) || total > b->pair_limit) {
        // that frame's lists (and anything composited from them) are incomplete: make room, tell the caller
        uint64_t want = (uint64_t)total + total / 2 + 8192;
        if (want > 0x3ffff000ull) want = 0x3ffff000ull;
        b->ran = false;
        int rc = sorter_reserve(&b->pairs, (uint32_t)want);
        if (rc != SPLAT_OK) return rc;
        return ctx_fail(ctx, SPLAT_ERR_CAPACITY,
                        "the previous frame produced more tile-splat pairs than its sync-free limit (sized from the frame "
                        "before it): its tile lists are incomplete; capacity has been raised, render that frame again");
    }
    return SPLAT_OK;
}

extern "C" {

#ifdef SPLAT_TEST_HOOKS
// EXPERIMENT HOOK (tools/overlap_probe.py): the per-tile sort of the binner's last tile-first frame once more, on `ctx`'s stream
// (any context of the device), from the second pass's output as it stands.  Timing only: tiles beyond the LDS path's size
// were sorted through both pair arrays and are sorted again from whatever that left.
int splat_debug_rerun_tile_sort(splat_ctx *ctx, splat_binner *b) {
    if (!ctx || !b) return ctx_fail(ctx, SPLAT_ERR_INVALID, "ctx/binner is NULL");
    ARG_CHECK(ctx, b->ran && b->wide_a && b->wide_b && b->ntx && b->nty);
    const uint32_t tiles = b->ntx * b->nty;
    const bool primary = tile_id_bits(tiles) - tile_id_low_bits(tiles) == 0;
    return tile_sort_launch(ctx, b->offsets, tiles, primary ? b->wide_a : b->wide_b, primary ? b->wide_b : b->wide_a, b->pairs.payload_b,
                            nullptr, b->d_total + 4, tiles >= 6144u ? (uint32_t)(b->total / tiles) : 0u); // (flags and counters of its own: words 4..6)
}
#endif

int splat_bin_create(splat_ctx *ctx, uint32_t tile_size, splat_binner **out) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, out != nullptr);
    *out = nullptr;
    ARG_CHECK(ctx, tile_size >= 1 && tile_size <= 4096);
    splat_binner *b = new splat_binner();
    b->ctx = ctx;
    b->tile = tile_size;
    b->pairs.ctx = ctx;
    if (hipMalloc((void **)&b->d_total, 32) != hipSuccess || hipHostMalloc((void **)&b->pinned, 16, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void **)&b->pinned_dev, b->pinned, 0) != hipSuccess ||
        hipEventCreateWithFlags(&b->readback_done, hipEventDisableTiming) != hipSuccess) {
        if (b->d_total) (void)hipFree(b->d_total);
        if (b->pinned) (void)hipHostFree(b->pinned);
        delete b;
        return ctx_fail(ctx, SPLAT_ERR_OOM, "binner allocation");
    }
    // the report words start at zero: binner_settle waits for word 2 to become the frame's sequence number (from 1 up), and
    // recycled pinned memory may well hold an earlier binner's "1"
    memset(b->pinned, 0, 16);
    if (const char *e = getenv("SPLAT_BIN_SYNC")) b->allow_async = !(e[0] == '1');
    *out = b;
    return SPLAT_OK;
}

void splat_bin_destroy(splat_binner *b) {
    if (!b) return;
    (void)hipStreamSynchronize(b->ctx->stream);
    binner_free(b);
    sorter_free_members(&b->pairs);
    if (b->d_total) (void)hipFree(b->d_total);
    if (b->range32) (void)hipFree(b->range32);
    binner_free_wide(b);
    if (b->expanded) (void)hipFree(b->expanded);
    if (b->discs) (void)hipFree(b->discs);
    if (b->band_idx) (void)hipFree(b->band_idx);
    if (b->pinned) (void)hipHostFree(b->pinned);
    if (b->readback_done) (void)hipEventDestroy(b->readback_done);
    delete b;
}

int splat_bin_run(splat_binner *b, const void *projected, uint32_t n_splats, const void *sorted, uint32_t n_sorted,
                  uint32_t width, uint32_t height, uint32_t tile_row0, uint32_t tile_row1) {
    if (!b) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "binner is NULL");
    return binner_run(b, projected, n_splats, sorted, n_sorted, width, height, tile_row0, tile_row1, nullptr);
}

uint32_t splat_bin_tile_size(const splat_binner *b) { return b ? b->tile : 0; }

int splat_bin_set_frame_order(splat_binner *b, int order) {
    if (!b) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "binner is NULL");
    ARG_CHECK(b->ctx, order >= SPLAT_FRAME_ORDER_DEFAULT && order <= SPLAT_FRAME_TILE_FIRST);
    b->frame_order = order;
    return SPLAT_OK;
}

int splat_bin_counts(splat_binner *b, void **dptr) {
    if (!b || !dptr) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "binner/dptr is NULL");
    if (int rc = binner_settle(b)) return rc;
    if (!b->ran) return ctx_fail(b->ctx, SPLAT_ERR_STATE, "Tile counts buffer not initialized"); // GPUTileBinner.ts:354-359
    *dptr = b->counts;
    return SPLAT_OK;
}

int splat_bin_offsets(splat_binner *b, void **dptr) {
    if (!b || !dptr) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "binner/dptr is NULL");
    if (int rc = binner_settle(b)) return rc;
    if (!b->ran) return ctx_fail(b->ctx, SPLAT_ERR_STATE, "Tile offsets buffer not initialized"); // :340-345
    *dptr = b->offsets;
    return SPLAT_OK;
}

int splat_bin_indices(splat_binner *b, void **dptr) {
    if (!b || !dptr) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "binner/dptr is NULL");
    if (int rc = binner_settle(b)) return rc;
    if (!b->ran) return ctx_fail(b->ctx, SPLAT_ERR_STATE, "Tile indices buffer not initialized"); // :347-352
    *dptr = b->pairs.result_in_primary ? b->pairs.payload : b->pairs.payload_b;
    return SPLAT_OK;
}

int splat_bin_total(splat_binner *b, uint64_t *total_pairs) {
    if (!b || !total_pairs) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "binner/total is NULL");
    if (int rc = binner_settle(b)) return rc;
    if (!b->ran) return ctx_fail(b->ctx, SPLAT_ERR_STATE, "binSplats has not run");
    *total_pairs = b->total;
    return SPLAT_OK;
}

int splat_validate_tile_order(splat_ctx *ctx, const void *projected, const void *tile_offsets, uint32_t num_tiles,
                              const void *tile_indices, uint64_t total_pairs, uint64_t *violations_host) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, violations_host && (total_pairs == 0 || (projected && tile_offsets && tile_indices)) && num_tiles >= 1);
    ARG_CHECK(ctx, total_pairs < (1ull << 32));
    *violations_host = 0;
    if (total_pairs < 2) return SPLAT_OK;
    int rc = ctx_ensure_scan_ws(ctx, 256);
    if (rc != SPLAT_OK) return rc;
    unsigned long long *d = (unsigned long long *)ctx->scan_ws;
    HIP_TRY(ctx, hipMemsetAsync(d, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_validate_tile_order, dim3(div_up((uint32_t)total_pairs, 256)), dim3(256), 0, ctx->stream,
                       (const float4 *)projected, (const uint32_t *)tile_offsets, num_tiles, (const uint32_t *)tile_indices,
                       (uint32_t)total_pairs, d);
    LAUNCH_CHECK(ctx, "k_validate_tile_order");
    unsigned long long v = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&v, d, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *violations_host = v;
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
