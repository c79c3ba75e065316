// common.h — shared by every translation unit of libsplat_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/splat.h"

constexpr int kWave = 64; // CDNA wavefront

#include <vector>
// One begin/end event pair per timed launch of a stage; pairs are pooled and reused after
// splat_set_timing(ctx, 1) resets the cursor, so K frames give K samples per stage.
struct StageTimer {
    std::vector<hipEvent_t> beg, end;
    size_t used = 0;
    uint32_t tick = 0; // launches of this stage seen while timing is on (splat_set_timing_sampling: every n-th carries a pair)
};
// k_composite_px's history for ONE band of one binner's lists (composite.hip, px_order_prepare): two arrays of per-tile costs
// and two of tile orders (cap entries each), alternating between launches.  A context keeps a few (two frames in flight on
// one context, virtual ranks, a band next to the whole frame: each has its own), the least recently used one gives way.
struct PxHistory {
    uint64_t key = 0; // 0 = free
    uint32_t *mem = nullptr;
    uint32_t cap = 0, parity = 0, streak = 0;
    uint64_t last_use = 0;
};
constexpr int PX_HISTORIES = 4;

struct splat_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    bool timing = false;
    int lds_atomic_ordered = -1; // -1 unknown, else the result of radix_probe_lds_atomic_order on this device
    // Which kernels may rank with returning LDS atomics (stable only if colliding lanes complete in lane order: measured
    // on gfx950, not an ISA promise).  RANK_CHECKED (default): only where a COMPLETE check of the result follows in the
    // same frame — the tile-first frame path, whose per-tile sort verifies every list's strict (depth key, index) order
    // and whose host side renders a frame that failed again with ballots (rank_atomic_ok) — and ballots everywhere else;
    // RANK_ATOMIC (SPLAT_RANK=atomic): wherever the probe allows; RANK_BALLOT (SPLAT_RANK=ballot, or after a failed
    // check): nowhere.
    int rank_policy = 0;
    uint32_t order_faults = 0;      // frames whose lists failed the order check (each was reported and rendered again)
#ifdef SPLAT_TEST_HOOKS // (libsplat_hip_hooks.so, the test build: the shipped library carries none of this)
    const uint32_t *debug_tile_order = nullptr; // experiment hook (splat_debug_set_tile_order)
    const uint32_t *debug_sort_order = nullptr; // experiment hook (splat_debug_set_tile_sort_order)
    uint32_t inject_order_fault = 0; // test hook (splat_debug_inject_order_fault): tile + 1 whose list the next tile sort swaps ...
    uint32_t inject_order_position = 0; // ... at entries position, position + 1
    uint32_t tile_sort_launches = 0; // test hook (splat_debug_tile_sort_launches): k_tile_sort launches of the last per-tile sort
#endif
    // splat_composite_options (-1 / 0 = the process default, i.e. the environment's): which kernel composites nearest-on-top
    // isotropic frames (0 quadrant, 1 pixel), and k_composite_px's schedule
    int opt_composite_kernel = -1, opt_px_ahead = 0, opt_px_predict = -1;
    PxHistory px_hist[PX_HISTORIES]; // k_composite_px's per-band histories
    uint64_t px_clock = 0;
    uint32_t timing_mask = 0xffffffffu; // which stages record events while timing is on
    uint32_t timing_every = 1; // one-kernel stages (the composite): an event pair on every timing_every-th launch (StageTimer::tick)
    StageTimer timers[SPLAT_STAGE_COUNT];
    // scratch for the generic scan (block sums) and for small device scalars
    void *scan_ws = nullptr;
    size_t scan_ws_bytes = 0;
    unsigned long long *d_consumed = nullptr; // per tile: list entries staged by the composite while timing is on
    uint32_t consumed_tiles = 0;              // entries allocated in d_consumed
    // pinned host staging for uploads / tiny readbacks
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
};

int ctx_fail(splat_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess);
int ctx_ensure_scan_ws(splat_ctx *ctx, size_t bytes);
int ctx_ensure_pinned(splat_ctx *ctx, size_t bytes);
int ctx_ensure_consumed(splat_ctx *ctx, uint32_t tiles); // per-tile counters for timed frames (zeroed when (re)allocated)
void stage_begin(splat_ctx *ctx, int stage);
void stage_end(splat_ctx *ctx, int stage);
bool stage_event_pair(splat_ctx *ctx, int stage, hipEvent_t *start, hipEvent_t *stop);

#define HIP_TRY(ctx, expr)                                                   \
    do {                                                                     \
        hipError_t e__ = (expr);                                             \
        if (e__ != hipSuccess) return ctx_fail((ctx), SPLAT_ERR_HIP, #expr, e__); \
    } while (0)

#define LAUNCH_CHECK(ctx, name)                                                       \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) return ctx_fail((ctx), SPLAT_ERR_HIP, "launch " name, e__); \
    } while (0)

#define ARG_CHECK(ctx, cond)                                               \
    do {                                                                   \
        if (!(cond)) return ctx_fail((ctx), SPLAT_ERR_INVALID, "argument check failed: " #cond); \
    } while (0)

static inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
static inline uint64_t div_up64(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// ---- internal launchers shared between translation units ------------------------------------
// scan.hip
int scan_exclusive_u32(splat_ctx *ctx, const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total);
// radix_sort.hip: sorts n pairs from (k0,p0) using (k1,p1) as the ping-pong partner; returns in
// *result_in_primary whether the result ended in (k0,p0). hist is a workspace of
// 256 * div_up(n, RADIX_PART) + 256 u32.  n_dev (optional, rowscan mode): the number of pairs is
// min(*n_dev, n) read on the device, n only sizes the grid — no host round trip to learn it.
constexpr uint32_t RADIX_PART = 4096; // keys per workgroup partition
int radix_sort_pairs(splat_ctx *ctx, uint32_t *k0, uint32_t *p0, uint32_t *k1, uint32_t *p1,
                     uint32_t *hist, uint32_t n, uint32_t bit_begin, uint32_t bit_end,
                     bool *result_in_primary, int mode = -1, const uint32_t *n_dev = nullptr,
                     bool iota_payload = false, uint32_t first_bits = 8);

int radix_probe_lds_atomic_order(splat_ctx *ctx, uint64_t *mismatches_host);
int ctx_resolve_rank_mode(splat_ctx *ctx); // sets ctx->lds_atomic_ordered (probe) and ctx->rank_policy (SPLAT_RANK)
constexpr int RANK_CHECKED = 0, RANK_ATOMIC = 1, RANK_BALLOT = 2;
// may this launch rank with returning LDS atomics?  checked: its result is verified completely later in the frame
static inline bool rank_atomic_ok(const splat_ctx *ctx, bool checked) {
    if (ctx->lds_atomic_ordered != 1 || ctx->rank_policy == RANK_BALLOT) return false;
    return ctx->rank_policy == RANK_ATOMIC || checked;
}
// frame flags word (binner d_total[1]): bit 0 = the pairs outgrew the sync-free limit, bit 1 = a tile's list failed the
// order check of k_tile_sort
constexpr uint32_t FRAME_FLAG_OVERFLOW = 1u, FRAME_FLAG_ORDER = 2u;

struct splat_sorter {
    splat_ctx *ctx = nullptr;
    uint32_t capacity = 0;
    uint32_t *keys = nullptr, *keys_b = nullptr, *payload = nullptr, *payload_b = nullptr;
    uint32_t *hist = nullptr;
    uint32_t *d_count = nullptr; // device-side element count for sync-free callers (splat_band_keys)
    // splat_band_frame: the kept count of the previous frame bounds this frame's sort/bin grids
    // (1.125x), its own count comes back asynchronously and is examined at the next call
    bool count_pending = false, have_last_count = false;
    uint32_t last_count = 0, count_bound = 0;
    uint32_t *pinned_count = nullptr; // 4 u32 host-pinned
    hipEvent_t count_event = nullptr;
    uint32_t kept_blocks = 0; // tile-first band frame: hist[0..kept_blocks) holds per-block kept counts, not yet summed
    bool result_in_primary = true;
    bool ran = false;
    int mode = -1; // -1 / 0 = rank as the context's policy says, 2 = always with ballots
};

// grows the sorter's buffers (contents are NOT preserved)
int sorter_reserve(splat_sorter *s, uint32_t capacity);

// The tile-first binner's first sort pass (k_tf_scatter) leaves every low digit's run of pairs starting on a boundary
// of the second pass's partitions (TF_RUN_ALIGN pairs; tile_first.hip says why), and describes the layout here:
struct TfRuns {
    uint32_t *start, *total; // per low digit (256 entries each): first slot of its run, pairs in it
    uint8_t *part_digit;     // per partition of the second pass: the low digit whose run it lies in
    uint32_t *parts;         // number of partitions in use
};
constexpr uint32_t TF_RUN_ALIGN = 4096;
constexpr uint32_t TF_RUN_SLACK = 256 * TF_RUN_ALIGN; // slots the padding can add to the first pass's output, at most
static inline uint32_t tf2_parts_bound(uint32_t pairs, uint32_t lo_bits) { return div_up(pairs, TF_RUN_ALIGN) + (1u << lo_bits); }

struct splat_binner {
    splat_ctx *ctx = nullptr;
    uint32_t tile = 16;
    uint32_t ntx = 0, nty = 0;
    uint32_t tiles_cap = 0, splats_cap = 0, range32_cap = 0;
    uint32_t *counts = nullptr, *offsets = nullptr; // per tile
    uint32_t *blocksums = nullptr;                  // pairs per 512-position block, then their exclusive scan
    uint2 *ranges = nullptr;                        // per sorted position: packed clamped tile range
    uint32_t *range32 = nullptr;                    // per splat INDEX: 8-bit packed range written by the projector (frame path)
    uint32_t *d_total = nullptr;                    // [0] pair total of the last run, [1] overflow flag
    splat_sorter pairs;                             // (tileId, splatIdx) ping-pong buffers
    uint2 *wide_a = nullptr, *wide_b = nullptr;     // tile-first path: (depth key, splat idx) per pair, ping-pong
    uint32_t wide_cap = 0;
    uint8_t *tf_hi = nullptr;                       // tile-first path: high tile-id digit per pair of the first pass's output
    uint32_t *tf2_hist = nullptr;                   // ... the second pass's histogram (256 rows of partitions + 256 totals)
    uint32_t *tf_runs_mem = nullptr;                // ... TfRuns storage: start[256], total[256], parts, then part_digit bytes
    TfRuns tf_runs = {};
    uint32_t *tf_hist = nullptr;                    // tile-first path: per 1024-splat block digit histograms (first sort pass)
    int frame_order = -1;                           // splat_bin_set_frame_order
    void *expanded = nullptr;                       // band frame: ProjectedSplat records rebuilt from compact exchange records
    uint32_t expanded_cap = 0;
    void *discs = nullptr;                          // frame path, oriented-disc footprint: the projector's 32-byte disc records
    uint32_t discs_cap = 0;
    bool tf_hist_ready = false;                     // the projector already filled tf_hist / blocksums for the next tile-first run
    uint32_t tf_block = 1024;                       // splats per block of that histogram (TF_BLOCK_SMALL for small frames)
    // a multi-GPU band frame whose prepare pass compacted the kept splats per group of 4096 records (k_band_prepare_tfc): their
    // indices and the groups' kept counts, for the next tile-first run only (binner_run clears them)
    const uint32_t *tf_cidx = nullptr, *tf_kept = nullptr;
    uint32_t *band_idx = nullptr;                   // storage of tf_cidx (one index per gathered record)
    uint32_t band_idx_cap = 0;
    uint64_t total = 0;
    bool ran = false;
    // sync-free operation: when the previous frame's pair total is known and 1.125x of it fits the
    // pair buffers, the next frame does not wait for its own total: kernels read it on the device,
    // grids are sized for `pair_limit` = 1.125x the previous total, and the total comes back through an
    // async copy that is examined at the next call.  A frame that outgrew its limit is reported then
    // (SPLAT_ERR_CAPACITY).
    bool allow_async = true;
    bool pending = false;      // an async {total, overflow} readback is in flight
    bool have_last = false;
    uint32_t last_total = 0;
    // the previous tile-first frame's count of tiles beyond its short size class (0xffffffff: not known), of which band
    uint32_t last_long_tiles = 0xffffffffu;
    uint64_t last_band_key = ~0ull;
    uint32_t last_short_class = 0; // (the short class that frame's count is relative to: tile_sort_launch)
    bool pending_tile_first = false;
    uint32_t pair_limit = 0;   // pairs this frame's grids / stores are bounded by
    uint32_t *pinned = nullptr; // 4 u32, host-pinned, mapped into the device's address space as pinned_dev
    uint32_t *pinned_dev = nullptr;
    uint32_t seq = 0;          // sequence number of the last reported frame; its report carries it
    // tile-first frames: the report {pair total, flags, seq} is sent by the frame's LAST kernel, the composite (its flags
    // include the per-tile sort's order check); the frame function hands these to the composite launch and clears them
    uint32_t *report_for_composite = nullptr;
    uint32_t report_seq = 0;
    hipEvent_t readback_done = nullptr;
};

// Tile-range parameters handed to the projector so that it can emit range32[] (frame path)
struct BinParams {
    uint32_t width, height, tile, ntx, nty, row0, row1;
    uint32_t skip_outside; // frame of a strict band of tile rows: splats that provably cannot reach it are not projected in full
};

// Splats per block of the tile-first binner's first pass: 1024, or 256 for small frames — a 10 000-splat frame in
// 1024-splat blocks is a ten-workgroup kernel expanding 14 pairs per splat in four rounds on an otherwise idle device
// (64 of that frame's 120 us).
// (frames of up to 2^20 splats: 256-splat blocks — at C1, 1 M splats with 4.6 pairs each, a 1024-splat block holds more pairs than
//  the scatter stages in one round, and 977 blocks are less than one round of workgroups: 0.1554 -> 0.1515 ms per frame; at C2, 5 M
//  splats, the scatter's 19 500 small blocks cost it 78 instead of 48 us: profiles/r04_v_first_pass_block_size.txt)
constexpr uint32_t TF_BLOCK_LARGE = 1024, TF_BLOCK_SMALL = 256, TF_SMALL_FRAME_SPLATS = 1u << 20;
// Where the frame path's projector leaves the first sort pass's histogram (tile_first.hip): per
// block of splats the pairs per low tile-id digit (digit-major, num_parts columns) and in total.
struct TfHistOut {
    uint32_t *hist, *blocksums, *overflow_flag;
    uint32_t mask, num_parts;
    uint32_t block = TF_BLOCK_LARGE; // splats per block
    uint32_t xcd_per = 0;            // (set by project_launch: xcd_block_of's second argument)
    // a strict band's projector may leave the splats that can reach the band COMPACTED per group of 4096 (range and key at the
    // front of the group's segment of range32 / keys, their indices in cidx, their number in kept_groups[group]; the histogram
    // then has one column per group): k_tf_scatter<COMPACTED> runs over those only (frame.hip, k_band_prepare_tfc, does the same
    // for gathered records)
    uint32_t *cidx = nullptr, *kept_groups = nullptr;
};
// Which logical block (partition, 1024-splat block) workgroup i of a grid takes.  xcd_per == 0: block i.  Otherwise (xcd_per =
// ceil(blocks / 8), grid = 8 * xcd_per): the hardware deals consecutive workgroups to the eight XCDs in turn, so workgroup i
// runs on XCD i % 8; it takes block (i % 8) * xcd_per + i / 8 — each XCD's L2 then sees a CONTIGUOUS eighth of the blocks:
// the 4-byte column entries that neighbouring blocks write into one histogram row, the 128-byte lines of the scanned rows they
// read back, and the neighbouring fragments of one digit run in a pass's output (16 pairs = 128 bytes per partition and digit
// at C3) are merged / fetched once in one L2 instead of partially in eight (C3: k_tf_upsweep2 33.6 -> 19.8 us,
// k_tf_downsweep2 158.9 -> 147.4; nothing either way at C2: profiles/r05_b_second_pass_xcd_C3_C2.txt).  A block past the
// last one is skipped by the caller.
__device__ __forceinline__ uint32_t xcd_block_of(uint32_t i, uint32_t xcd_per) {
    return xcd_per ? (i & 7u) * xcd_per + (i >> 3) : i;
}
// tile-id bits and their split over the two sort passes (13 bits -> 6 + 7: longer digit runs than 8 + 5)
static inline uint32_t tile_id_bits(uint32_t tiles) {
    uint32_t bits = 1;
    while ((1u << bits) < tiles) ++bits;
    return bits;
}
static inline uint32_t tile_id_low_bits(uint32_t tiles) {
    const uint32_t bits = tile_id_bits(tiles);
    return bits <= 8 ? bits : bits / 2;
}

// report: host-mapped pinned words that receive {pair total, overflow flag, frame sequence number} —
// a sync-free frame's readback with no copy and no event in the stream (hipMemcpyAsync +
// hipEventRecord left the GPU idle for ~14 us per frame between the binner and the composite).
__device__ __forceinline__ void tile_report(const uint32_t *d_total, uint32_t *report, uint32_t seq) {
    report[0] = d_total[0];
    report[1] = d_total[1];
    report[3] = d_total[3]; // (tile-first frames: tiles beyond the per-tile sort's short class, counted by its first launch)
    __threadfence_system();
    __hip_atomic_store(&report[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); // the host polls this word
}

// bin.hip internals used by frame.hip
int binner_reserve_range32(splat_binner *b, uint32_t n_splats);
int binner_run(splat_binner *b, const void *projected, uint32_t n_splats, const void *sorted, uint32_t n_sorted, uint32_t width,
               uint32_t height, uint32_t tile_row0, uint32_t tile_row1, const uint32_t *range32,
               const uint32_t *n_sorted_dev = nullptr, const uint32_t *depth_keys = nullptr);
int binner_reserve(splat_binner *b, uint32_t tiles, uint32_t n_sorted); // per-tile and per-position buffers
// tile_first.hip (the frame path's bin-then-sort-per-tile kernels) and the wide-payload radix sort
int tf_scatter_launch(splat_ctx *ctx, const uint32_t *range32, const uint32_t *depth_keys, uint32_t n, uint32_t ntx, uint32_t mask,
                      const uint32_t *hist, uint32_t *d_total, uint32_t pair_limit, uint32_t *overflow, uint8_t *out_hi,
                      uint2 *out_val, uint32_t block_splats, uint32_t lo_bits, bool second_pass, const TfRuns *runs,
                      uint32_t *offsets_if_final, uint32_t tiles, uint32_t *report, uint32_t seq, const uint32_t *cidx = nullptr,
                      const uint32_t *kept = nullptr); // cidx / kept: a band's kept splats compacted per group of 4096 records (frame.hip)
int tf_second_pass_launch(splat_ctx *ctx, const uint8_t *hi, const uint2 *val_in, uint2 *val_out, const TfRuns *runs, uint32_t pairs_bound,
                          uint32_t tiles, uint32_t lo_bits, uint32_t hi_bits, uint32_t *hist, uint32_t *offsets, const uint32_t *d_total);
int radix_rowscan_launch(splat_ctx *ctx, uint32_t *hist, uint32_t parts, uint32_t rows); // rows -> exclusive prefixes, totals at hist + 256*parts
int tile_sort_launch(splat_ctx *ctx, const uint32_t *offsets, uint32_t tiles, uint2 *vals, uint2 *scratch, uint32_t *out_idx,
                     uint32_t *counts, uint32_t *frame_flags, uint32_t mean_list, uint32_t band_tiles = 0,
                     uint32_t long_tiles_hint = 0xffffffffu, uint32_t *short_class_io = nullptr); // frame_flags: FRAME_FLAG_ORDER is raised if a list fails the order check
int binner_settle(splat_binner *b); // resolves a pending report; SPLAT_ERR_CAPACITY if that frame overflowed, SPLAT_ERR_RETRY if its lists failed the order check
// composite.hip: splat_composite with the frame's report attached (report != NULL: the launch's first workgroup stores
// {frame_total[0], frame_total[1], seq} into the host-mapped report words; see tile_report)
int composite_launch(splat_ctx *ctx, const splat_composite_cfg *cfg, const void *color_opacity, uint32_t color_stride_vec4, const void *normals,
                     uint32_t normal_stride_vec4, const void *projected, const void *tile_indices, const void *tile_counts,
                     const void *tile_offsets, uint32_t width, uint32_t height, void *out_rgba8, void *out_rgba32f, void *consumed_dptr,
                     const uint32_t *frame_total, uint32_t *report, uint32_t report_seq);
// project.hip internal: the projector with the optional per-index tile range output
int project_launch(splat_ctx *ctx, const float *uniforms, const void *pos_radius, uint32_t pr_stride_vec4, uint32_t n,
                   uint32_t index_base, void *projected, void *keys, void *payload, uint32_t n_padded, uint32_t *range32,
                   const BinParams *bp, const TfHistOut *hist_out = nullptr, const void *normals = nullptr,
                   uint32_t normal_stride_vec4 = 1, void *discs = nullptr, // discs != NULL: the oriented-disc footprint (disc.h)
                   const struct LitIO *lit = nullptr);                     // lit->records != NULL: also write lit composite records (shade.h)
