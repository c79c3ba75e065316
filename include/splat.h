/*
 * splat.h — C ABI of libsplat_hip.so: the MI355X (gfx950) tile-raster hot path of
 * ath92/splat-renderer behind plain pointers and sizes.
 *
 * The reference has no FFI: its boundary is the TypeScript class surface of the stage classes
 * (SURVEY.md §8b).  Each entry point below replaces the GPU work of one reference verb; the
 * reference file:line it replaces is cited on every declaration (paths under /root/reference).
 * The bindings a maintainer adds on the reference side (N-API stub + host classes) are shown in
 * INTEGRATION.md; the Python mirror used by tests/bench is splat_renderer_amd/host.py.
 *
 * Conventions
 *  - every function returns SPLAT_OK (0) or a negative SPLAT_ERR_*; the message is available
 *    from splat_last_error(ctx).  No C++ exception crosses this boundary.
 *  - one ctx = one HIP device + one stream.  All work is enqueued on that stream in call order
 *    (the reference's "queue submission order").  A ctx is not thread-safe; distinct ctxs are
 *    independent.  Functions do not synchronise with the host unless documented.
 *  - "dptr" arguments are device pointers (from splat_buf_alloc, hipMalloc, or a
 *    torch.Tensor.data_ptr()) and must be 16-byte aligned.
 *  - there is NO CPU fallback anywhere behind this ABI.
 */
#ifndef SPLAT_H
#define SPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPLAT_ABI_VERSION 3 /* 3 (round 5): splat_composite_options lost its slack argument; splat_sort_lookback_timeouts and sort mode 1 are gone; the splat_debug_* hooks moved behind SPLAT_TEST_HOOKS */

#define SPLAT_OK 0
#define SPLAT_ERR_INVALID (-1)  /* bad argument */
#define SPLAT_ERR_HIP (-2)      /* a HIP runtime call failed */
#define SPLAT_ERR_OOM (-3)      /* device allocation failed */
#define SPLAT_ERR_CAPACITY (-4) /* a fixed-capacity object is too small for this call */
#define SPLAT_ERR_STATE (-5)    /* getter called before the verb that produces its result */
#define SPLAT_ERR_NO_DEVICE (-6)
#define SPLAT_ERR_COMM (-7)     /* RCCL call failed */
#define SPLAT_ERR_RETRY (-8)    /* the previous frame's tile lists failed the per-tile sort's order check (see
                                 * splat_rank_status): the context has switched to ballot ranking; render that frame again.
                                 * Returned where SPLAT_ERR_CAPACITY is for an overflowed sync-free frame, and handled the
                                 * same way by a caller: call the frame function again with the same arguments. */

typedef struct splat_ctx splat_ctx;
typedef struct splat_sorter splat_sorter;
typedef struct splat_binner splat_binner;
typedef struct splat_comm splat_comm;

/* Sizes of the reference's records (bytes). */
#define SPLAT_PROPS_BYTES 32     /* vec4(pos,radius), vec4(rgb,opacity): src/SplatPropertyManager.ts:1-5 */
#define SPLAT_PROJECTED_BYTES 32 /* ProjectedSplat: src/SplatProjector.ts:47-54 */
#define SPLAT_SORT_BLOCK 3840    /* key-buffer padding quantum: src/RadixSorter.ts:12-19,46-52 */

/* stage ids for splat_stage_time_ms */
enum {
    SPLAT_STAGE_PROJECT = 0, /* project + depth keys */
    SPLAT_STAGE_SORT = 1,
    SPLAT_STAGE_BIN = 2,     /* count + scan + fill */
    SPLAT_STAGE_COMPOSITE = 3,
    SPLAT_STAGE_EXCHANGE = 4, /* multi-GPU all-gather */
    /* parts of SPLAT_STAGE_BIN in the tile-first frame order (each inside the BIN interval): */
    SPLAT_STAGE_BIN_SCATTER = 5,   /* first pass of the tile-id sort fused with the pair expansion (+ its row scan) */
    SPLAT_STAGE_BIN_PASS2 = 6,     /* second pass (upsweep, row scan, downsweep) + tile offsets */
    SPLAT_STAGE_BIN_TILE_SORT = 7, /* PerTileSorter: depth order inside every tile */
    SPLAT_STAGE_COUNT = 8
};

/* ---- context ---------------------------------------------------------------------------- */
/* GPUDevice + queue equivalent (src/main.ts:16-36). */
int splat_ctx_create(int device_ordinal, splat_ctx **out);
/* Same, but enqueue on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
int splat_ctx_create_on_stream(int device_ordinal, void *hip_stream, splat_ctx **out);
void splat_ctx_destroy(splat_ctx *ctx);
/* Last error text of this ctx (or of the calling thread when ctx is NULL). Never NULL. */
const char *splat_last_error(splat_ctx *ctx);
/* device.queue.onSubmittedWorkDone equivalent: wait for everything enqueued so far. */
int splat_sync(splat_ctx *ctx);
int splat_abi_version(void);
/* Enable per-stage hipEvent timing (off by default: events cost a few us per stage). */
int splat_set_timing(splat_ctx *ctx, int enabled);
/* Restrict event recording to the stages whose bit (1 << stage id) is set (default: all).  Each
 * recorded stage costs ~10 us of stream idle per frame, so a throughput measurement that only needs
 * one kernel's duration enables only that stage.  Bit 31 (SPLAT_TIMING_COUNT_ENTRIES, set by default):
 * a timed whole-frame call also counts, per tile, the list entries its composite staged and consumed
 * (splat_timing_consumed) — the counting instantiation of the kernel is a few per cent slower than the
 * one every other frame runs, so a measurement of the production kernel clears the bit and takes the
 * counts from a frame outside its timed region (they are a property of the input). */
#define SPLAT_TIMING_COUNT_ENTRIES 0x80000000u
int splat_set_timing_stages(splat_ctx *ctx, uint32_t stage_mask);
/* A stage that is ONE kernel (the composite) carries its event pair on the launch itself; even so a timed launch costs the
 * stream ~5 us (C2: 0.326 ms per frame with a pair on every composite, 0.321 with none).  every = n: only every n-th such
 * launch while timing is on is timed (the first one is); splat_stage_time_stats then averages over those.  Default 1. */
int splat_set_timing_sampling(splat_ctx *ctx, uint32_t every);
/* Duration of the most recent run of `stage`; synchronises on that stage's end event. */
int splat_stage_time_ms(splat_ctx *ctx, int stage, float *ms);
/* Every timed run of `stage` since timing was last enabled: number of samples and their summed
 * duration (synchronises).  Enabling timing again starts a new sample set. */
int splat_stage_time_stats(splat_ctx *ctx, int stage, uint32_t *samples, double *total_ms);
/* List entries of splat_render_frame's composite since timing was last enabled, summed over tiles and timed
 * frames; synchronises.  *consumed = P_used of SURVEY §8d: per tile, the entries its pixels visited before the
 * last of them reached alpha >= 0.99 (the whole list if one never did; = the pair total with early-out off).
 * *staged = what the kernel actually gathered: the same, rounded up to its 256-entry batches. */
int splat_timing_consumed(splat_ctx *ctx, uint64_t *staged, uint64_t *consumed);

/* ---- buffers (GPUBuffer equivalent: device.createBuffer / queue.writeBuffer / mapAsync) --- */
int splat_buf_alloc(splat_ctx *ctx, size_t bytes, void **dptr);
int splat_buf_free(splat_ctx *ctx, void *dptr);
int splat_buf_upload(splat_ctx *ctx, void *dst_dptr, const void *src_host, size_t bytes);   /* async on the stream, src is staged */
int splat_buf_download(splat_ctx *ctx, void *dst_host, const void *src_dptr, size_t bytes); /* synchronous */
int splat_buf_zero(splat_ctx *ctx, void *dptr, size_t bytes);
int splat_buf_copy(splat_ctx *ctx, void *dst_dptr, const void *src_dptr, size_t bytes); /* copyBufferToBuffer: device to device, async on the stream */

/* ---- SplatPropertyManager.updateFromCurvature  (src/SplatPropertyManager.ts:82-107,153-173) */
/* positions, curvature: vec4 per splat; props: 32-byte interleaved records. */
int splat_update_props(splat_ctx *ctx, const void *positions, const void *curvature, uint32_t n,
                       void *props);
/* The same update writing two planes (vec4(pos, radius) | vec4(rgb, opacity)) — SURVEY §8f row 1 — and the
 * one-off conversion of interleaved records into planes. */
int splat_update_props_planes(splat_ctx *ctx, const void *positions, const void *curvature, uint32_t n,
                              void *pos_radius, void *color_opacity);
int splat_props_to_planes(splat_ctx *ctx, const void *props, uint32_t n, void *pos_radius, void *color_opacity);

/* ---- SplatProjector.project  (src/SplatProjector.ts:64-132,174-194) ----------------------- */
/* uniforms: 22 host floats = VP column-major [0..15], eye [16..18], time [19], screenW [20],
 * screenH [21] (src/main.ts:126-144, src/SplatProjector.ts:35-41).
 * pos_radius: first vec4(pos,radius); consecutive splats are pr_stride_vec4 float4s apart
 * (2 = the reference's interleaved property buffer, 1 = a split plane).
 * projected: n 32-byte ProjectedSplat records (bit-exact vs oracle/orc_project).
 * keys/payload: if non-NULL the DepthKeyExtractor pass is fused in (n_padded entries, tail =
 * 0xFFFFFFFF); pass NULL to run the reference's unfused sequence. */
int splat_project(splat_ctx *ctx, const float *uniforms, const void *pos_radius,
                  uint32_t pr_stride_vec4, uint32_t n, void *projected, void *keys, void *payload,
                  uint32_t n_padded);

/* ---- DepthKeyExtractor.extract  (src/DepthKeyExtractor.ts:71-109, extract-depth-keys.wgsl:37-63) */
int splat_extract_keys(splat_ctx *ctx, const void *projected, uint32_t n, uint32_t n_padded,
                       void *keys, void *payload);

/* ---- RadixSorter  (src/RadixSorter.ts:39-100,197-271) ------------------------------------- */
/* Owns keys/keys_b/payload_a/payload_b for `capacity` pairs (capacity is rounded up to a
 * multiple of SPLAT_SORT_BLOCK like the reference's paddedSize). */
int splat_sort_create(splat_ctx *ctx, uint32_t capacity, splat_sorter **out);
void splat_sort_destroy(splat_sorter *s);
uint32_t splat_sort_capacity(const splat_sorter *s);
void *splat_sort_keys(splat_sorter *s);    /* getKeysBuffer():    input keys (u32)    */
void *splat_sort_payload(splat_sorter *s); /* getPayloadBuffer(): input payload (u32) */
/* sort(): stable ascending LSD sort of the first n pairs on key bits [bit_begin, bit_end).
 * The reference always sorts all 32 bits (4 x 8-bit passes). */
int splat_sort_run(splat_sorter *s, uint32_t n, uint32_t bit_begin, uint32_t bit_end);
/* getSortedIndicesBuffer(): payload in sorted order (valid after splat_sort_run). */
void *splat_sort_sorted_payload(splat_sorter *s);
void *splat_sort_sorted_keys(splat_sorter *s);
/* Hardware probe (diagnostic): runs ~2M wave instructions of returning LDS atomics with colliding
 * addresses and counts those whose return values were NOT in ascending lane order.  Synchronises. */
int splat_probe_lds_atomic_order(splat_ctx *ctx, uint64_t *mismatches);
/* How this sorter ranks equal digits: 0 = as the context's policy says (per-pass histogram + row scan + scatter; returning
 * LDS atomics only where the policy allows them, see the NOTE), 2 = always with ballots, -1 = library default (0).  (Round
 * 1's onesweep mode with decoupled look-back — the reference's structure, slower on MI355X — was removed in round 5.)
 * NOTE on ranking (mode 0, and the frame path's binning kernels).  Ranking a wave's keys with returning LDS atomics is
 * stable only if the lanes of one instruction that collide on an address complete in ascending lane order.  That is what
 * gfx950 does (splat_probe_lds_atomic_order: 0 mismatches in 8.4 M colliding instructions) but it is not an ISA guarantee,
 * and index lists are bit-exact work, so nothing rests on it unverified:
 *   - default: atomics are used ONLY by the tile-first frame path (splat_render_frame*, splat_band_frame), whose per-tile
 *     sort checks every tile's final list for strictly increasing (depth key, splat index) order — the contract itself,
 *     hence a complete check of every ranking pass that produced the list.  A frame that fails is reported like an
 *     overflowed sync-free frame (SPLAT_ERR_RETRY at the next call), the context ranks with ballots from then on, and the
 *     caller renders the frame again.  Every other sort (splat_sort_run, splat_bin_run, the sort-first frame order), whose
 *     result nothing checks, ranks with ballots: lane order by construction (8 ballots + mbcnt per key).
 *     WHAT IS GUARANTEED, precisely: frame N's image and lists are PROVISIONAL until frame N's report has been examined —
 *     by the next frame call on the same binner, by splat_bin_total / splat_bin_get_* / splat_band_settle, or by anything the
 *     host classes read results through (Renderer.finish / readPixels / readPixelsFloat call splat_bin_total first and
 *     render the frame again on SPLAT_ERR_RETRY / SPLAT_ERR_CAPACITY).  The output buffer of a frame that fails the check
 *     HAS been written from the wrong lists: a caller that maps it without settling the frame reads that image.  The
 *     context prints one line to stderr when it switches to ballots, and splat_rank_status counts the frames.
 *   - SPLAT_RANK=atomic: atomics wherever the start-up probe passes (the unchecked sorts too).
 *   - SPLAT_RANK=ballot: ballots everywhere.
 * The price of the guaranteed ranking on the frame path, measured on one MI355X (profiles/r03_a_rank_ab.txt): C0 0.048 ->
 * 0.054 ms, C1 0.177 -> 0.210, C2 0.359 -> 0.425, C3 0.878 -> 0.986 (+12..19 %); the price of the check: see DESIGN.md. */
int splat_sort_set_mode(splat_sorter *s, int mode);
/* Ranking status of this context: *policy = 0 checked default / 1 SPLAT_RANK=atomic / 2 ballots (SPLAT_RANK=ballot, or
 * after a failed order check); *atomics_ordered = the start-up probe's verdict (1 = in lane order, 0 = not or not asked);
 * *order_faults = frames of this context whose tile lists failed the order check (each was reported with
 * SPLAT_ERR_RETRY).  Runs the probe if it has not run yet (synchronises then). */
int splat_rank_status(splat_ctx *ctx, int *policy, int *atomics_ordered, uint32_t *order_faults);
#ifdef SPLAT_TEST_HOOKS
/* ---- test and experiment hooks: compiled only into libsplat_hip_hooks.so (-DSPLAT_TEST_HOOKS, built beside the library for
 * tests/ and tools/); the shipped library neither exports them nor carries their kernel parameters ---------------------- */
/* TEST HOOK: the next per-tile sort of this context swaps entries `position` and `position + 1` of tile `tile`'s finished
 * list just before its order check, as an out-of-lane-order ranking would have left them: the check must raise the
 * frame's flag, the next call return SPLAT_ERR_RETRY, and the frame rendered again be right.  One shot. */
int splat_debug_inject_order_fault(splat_ctx *ctx, uint32_t tile, uint32_t position);
/* EXPERIMENT HOOK: the order in which the lane-efficient composite's workgroups take the tiles of the rendered band
 * (a permutation of 0 .. tiles - 1 as u32 on the device; NULL = row-major).  Any order gives the same image. */
int splat_debug_set_tile_order(splat_ctx *ctx, const void *order_dptr);
/* EXPERIMENT HOOK: the same for the per-tile sort's workgroups (a permutation of ALL the screen's tiles; NULL = row-major). */
int splat_debug_set_tile_sort_order(splat_ctx *ctx, const void *order_dptr);
/* EXPERIMENT HOOK (tools/overlap_probe.py): the per-tile sort of the binner's last tile-first frame once more, on ctx's stream. */
int splat_debug_rerun_tile_sort(splat_ctx *ctx, splat_binner *binner);
/* TEST HOOK: how many k_tile_sort launches this context's last per-tile sort made (2: a short and a long size class; 1: a band of
 * few tiles, or a sync-free frame after one that had no tile beyond the short class). */
int splat_debug_tile_sort_launches(splat_ctx *ctx, uint32_t *launches);
/* EXPERIMENT HOOK (tools/lds_atomic_rate.py): milliseconds of a kernel that does iters x 4 LDS instructions per wave at random
 * counters of the wave's own 256-entry table, workgroups_per_cu four-wave workgroups per CU: kind 0 returning atomic adds (the
 * sort kernels' ranking instruction), 1 plain reads, 2 non-returning atomic adds. */
int splat_debug_lds_rate(splat_ctx *ctx, int kind, uint32_t workgroups_per_cu, uint32_t iters, float *ms);
#endif
/* The lane-efficient composite keeps, per context, for each of the last few (four) bands of tile rows it composited (a band
 * = these rows of this binner's lists), what its previous launch over that band cost per tile (chunks of
 * 32 list entries walked): the next launch over the same band takes its tiles longest-first and builds / gathers for each
 * tile only what that launch needed ahead of need (a tile that needs more pays one exposed gather).  Both are hints — any
 * history, stale or from another scene, gives the same image.  This forgets the history (the next two launches run
 * row-major and without a bound, as a context's first do): for measurements and tests that want the kernel's first-frame
 * behaviour. */
int splat_composite_forget_history(splat_ctx *ctx);
/* Per-context choices the environment otherwise makes for the whole process (environment table).  EVERY call
 * sets all three: a value of -1 (kernel, predict) or 0 (ahead) means "the process default", i.e. what the environment
 * variable named beside it says — not "leave as it is"; a per-context choice takes precedence over the environment:
 *   kernel  -1 default (SPLAT_COMPOSITE: lane-efficient k_composite_px on screens of >= 2048 tiles), 0 k_composite ("quadrant"),
 *            1 k_composite_px ("pixel") — for nearest-on-top frames of either footprint; the reference-literal blend always
 *            takes k_composite;
 *   ahead    0 default (SPLAT_PX_AHEAD: 1 for the isotropic footprint with the early-out, 2 otherwise), 1 or 2: chunks
 *            k_composite_px's builder wave stays ahead of its consumer wave (2: lanes whose queue for a chunk is empty go on
 *            with the next chunk's);
 *   predict -1 default (SPLAT_PX_PREDICT, on), 0 / 1: bound each tile's look-ahead by what the previous launch walked.
 * ahead and predict change the schedule only: the same bytes.  The two kernels evaluate the Gaussian differently
 * (k_composite_px builds an entry's table by a recurrence from five exponentials) and agree within the composite's stated
 * tolerance, 2e-5 per float channel, <= 1 LSB on rgba8 (tests/test_gpu_stages.py runs the oracle comparisons over all of
 * them).  Forgets the composite's history. */
int splat_composite_options(splat_ctx *ctx, int kernel, int ahead, int predict);

/* ---- PrefixSumScanner.scan  (src/PrefixSumScanner.ts:74-87, prefix-sum.wgsl:28-96) -------- */
/* Exclusive scan of n u32 (out[0] = 0); in may equal out.  total_dptr (optional) receives the
 * sum of all inputs as one u32.  Entirely on the device for any n (the reference falls back to
 * a CPU loop above 512 elements: src/PrefixSumScanner.ts:131-162). */
int splat_scan_u32(splat_ctx *ctx, const void *in, void *out, uint32_t n, void *total_dptr);

/* ---- GPUTileBinner  (src/GPUTileBinner.ts:35-50,190-377) ---------------------------------- */
/* Lists are exactly TileBinner.binSorted's (src/TileBinner.ts:426-495): splats fully
 * off-screen are culled and every tile's list is in `sorted` order. */
int splat_bin_create(splat_ctx *ctx, uint32_t tile_size, splat_binner **out);
void splat_bin_destroy(splat_binner *b);
/* binSplats(): sorted = n_sorted u32 splat indices (entries >= n_splats are padding and are
 * skipped).  Only tile rows [tile_row0, tile_row1) are binned (0, UINT32_MAX = all rows) —
 * the multi-GPU band.
 * Host round trips: the FIRST run (and any run whose pair buffers must grow) reads the 4-byte pair
 * total back to size the fill.  After that, while the previous run's total leaves 50 % headroom,
 * runs are sync-free: the total is read on the device, grids are sized from the previous total, and
 * it returns through an async copy examined by the next splat_bin_* call on this binner.  If a
 * frame's pairs outgrew that limit (more than 1.5x the previous frame's), its lists are incomplete
 * and that next call returns SPLAT_ERR_CAPACITY after raising the capacity: render the frame
 * again.  The getters below wait for the async copy. */
int splat_bin_run(splat_binner *b, const void *projected, uint32_t n_splats, const void *sorted,
                  uint32_t n_sorted, uint32_t width, uint32_t height, uint32_t tile_row0,
                  uint32_t tile_row1);
uint32_t splat_bin_tile_size(const splat_binner *b);        /* getTileSize() */
int splat_bin_counts(splat_binner *b, void **dptr);          /* getTileCountsBuffer()  u32[numTiles] */
int splat_bin_offsets(splat_binner *b, void **dptr);         /* getTileOffsetsBuffer() u32[numTiles] (+1: [numTiles] = total) */
int splat_bin_indices(splat_binner *b, void **dptr);         /* getTileIndicesBuffer() u32[total]    */
int splat_bin_total(splat_binner *b, uint64_t *total_pairs); /* sum of counts of the last run */
/* Order of work inside splat_render_frame (results are identical, tests hold both to the same lists):
 *   SPLAT_FRAME_SORT_FIRST  global depth sort of the splats, then bin in sorted order (the staged API's order);
 *   SPLAT_FRAME_TILE_FIRST  bin in index order, then depth-sort every tile's list (PerTileSorter,
 *                           src/PerTileSorter.ts:66-122) — needs tile coordinates that fit 8 bits;
 *   SPLAT_FRAME_ORDER_DEFAULT  the library's choice (environment SPLAT_FRAME_ORDER=sortfirst|tilefirst overrides). */
#define SPLAT_FRAME_ORDER_DEFAULT (-1)
#define SPLAT_FRAME_SORT_FIRST 0
#define SPLAT_FRAME_TILE_FIRST 1
int splat_bin_set_frame_order(splat_binner *b, int order);
int splat_bin_dims(splat_binner *b, uint32_t *ntx, uint32_t *nty);

/* ---- PerTileSorter.sort  (src/PerTileSorter.ts:66-122,174-213) --------------------------------- */
/* The reference re-sorts every tile's list by depth in LDS (racy, capped at 2048: SURVEY I3).
 * splat_bin_run bins an already sorted order, so its lists leave it in (depth key, index) order and
 * for the staged API the stage is a CHECK: counts adjacent pairs of one tile that are not strictly
 * increasing in (depth key, splat index).  (Inside splat_render_frame's tile-first order the per-tile
 * sort is real — any list length, stable — see splat_bin_set_frame_order.)
 * tile_offsets must have num_tiles + 1 entries (as splat_bin_offsets returns).  Synchronises. */
int splat_validate_tile_order(splat_ctx *ctx, const void *projected, const void *tile_offsets,
                              uint32_t num_tiles, const void *tile_indices, uint64_t total_pairs,
                              uint64_t *violations_host);

/* ---- ComputeShaderRenderer.render / TileRenderer.render  (src/ComputeShaderRenderer.ts:97-198,362-422) */
#define SPLAT_COMPOSITE_FRONT_TO_BACK 0     /* SURVEY §8a contract 3: nearest on top (default) */
#define SPLAT_COMPOSITE_REFERENCE_LITERAL 1 /* src/ComputeShaderRenderer.ts:175-190 as written */
#define SPLAT_RECORDS_PROJECTED 0
#define SPLAT_RECORDS_COMPACT 1
#define SPLAT_RECORDS_DISC48 2 /* oriented-disc exchange records, see splat_project_slice_disc (footprint DISC only) */
#define SPLAT_RECORDS_LIT32 3  /* lit composite records: float4 {centre x, y, screen radius, depth}, float4 {lit r, g, b, opacity} per
                                * splat — everything evaluateSplat (src/ComputeShaderRenderer.ts:117-147) reads of a splat in ONE
                                * 32-byte line: the ProjectedSplat's bounds are centre -/+ radius * 1.5 in the projector's operation
                                * order (src/SplatProjector.ts:119-121), the colour already carries the shading of :143-145.
                                * splat_render_frame* writes them in place of the ProjectedSplat records when asked to (isotropic
                                * footprint); the composite then gathers one line per staged list entry instead of three. */
#define SPLAT_FOOTPRINT_ISOTROPIC 0
#define SPLAT_FOOTPRINT_DISC 1
typedef struct splat_composite_cfg {
    uint32_t mode;       /* SPLAT_COMPOSITE_* */
    uint32_t early_out;  /* 1 = stop a pixel at alpha >= 0.99 (reference :187-190) */
    uint32_t tile_size;  /* must equal the binner's; only 16 is implemented */
    uint32_t tile_row0;  /* render tile rows [tile_row0, tile_row1) (multi-GPU band) */
    uint32_t tile_row1;  /* UINT32_MAX = to the last row */
    uint32_t record_format; /* what `projected` / `records` point at: SPLAT_RECORDS_PROJECTED (32-byte
                             * ProjectedSplat, the reference's struct), SPLAT_RECORDS_COMPACT (16-byte
                             * exchange records, see splat_project_slice_compact) or SPLAT_RECORDS_LIT32 (color_opacity
                             * and normals are then not read and may be NULL).  In splat_render_frame*: the format the
                             * frame's projector leaves in `projected` — PROJECTED or LIT32 */
    uint32_t prelit;        /* 1 = color_opacity holds LIT colours (splat_lit_colors): the composite then gathers two
                             * lines per staged entry instead of three and does not read normals (may be NULL) */
    uint32_t footprint;     /* SPLAT_FOOTPRINT_ISOTROPIC: ComputeShaderRenderer's screen-space Gaussian (default).
                             * SPLAT_FOOTPRINT_DISC: SequentialRenderer's / TileRenderer's oriented disc —
                             * `projected` then points at the 32-byte DISC records of splat_project_disc, mode must
                             * be FRONT_TO_BACK and record_format PROJECTED; in splat_render_frame* the projector
                             * used is splat_project_disc (normals are required even when prelit) and `projected`
                             * may be NULL (the ProjectedSplat records are then not written: a disc frame's composite
                             * reads the disc records) */
} splat_composite_cfg;
/* color_opacity / normals: vec4 per splat, *_stride_vec4 float4s apart.  out_rgba8 (W*H*4 bytes,
 * rgba8unorm, may be NULL) and out_rgba32f (W*H*16 bytes, may be NULL) are full-frame images;
 * only pixels of the rendered tile rows are written.  consumed_dptr (optional): u64[2 * ceil(W/16) *
 * ceil(H/16)], two counters per tile; a rendered tile's counters are incremented by {entries staged,
 * entries consumed} (see splat_timing_consumed; the sum of the second over tiles is P_used). */
/* The reference shades every splat with kd = 0.85 + 0.15 * max(dot(normal, (1,1,1)/sqrt 3), 0)
 * (src/ComputeShaderRenderer.ts:143-145).  lit[i] = vec4(rgb * kd, opacity): computed once per property
 * update instead of once per staged entry; the same bits either way (explicitly rounded operations). */
int splat_lit_colors(splat_ctx *ctx, const void *color_opacity, uint32_t color_stride_vec4,
                     const void *normals, uint32_t normal_stride_vec4, uint32_t n, void *lit);
int splat_composite(splat_ctx *ctx, const splat_composite_cfg *cfg, const void *color_opacity,
                    uint32_t color_stride_vec4, const void *normals, uint32_t normal_stride_vec4,
                    const void *projected, const void *tile_indices, const void *tile_counts,
                    const void *tile_offsets, uint32_t width, uint32_t height, void *out_rgba8,
                    void *out_rgba32f, void *consumed_dptr);

/* ---- whole frame: project -> keys -> sort -> bin -> composite ----------------
 * With cfg->tile_row0/1 set to a strict band of tile rows (multi-GPU, no exchange: every rank renders its band from
 * its own copy of the splats) only that band's pixels, lists and counts are produced, and `projected` holds records
 * only for splats that may reach the band (the projector skips the others after a conservative test). */
int splat_render_frame(splat_ctx *ctx, splat_sorter *sorter, splat_binner *binner,
                       const splat_composite_cfg *cfg, const float *uniforms, const void *props,
                       const void *normals, uint32_t n, uint32_t width, uint32_t height,
                       void *projected, void *out_rgba8, void *out_rgba32f);
/* The same frame from the MI355X-native property layout: two planes of vec4 per splat instead of the
 * reference's interleaved 32-byte records (the projector then reads 16 useful bytes per 16 fetched
 * instead of per 32).  splat_update_props_planes / splat_props_to_planes produce them. */
int splat_render_frame_planes(splat_ctx *ctx, splat_sorter *sorter, splat_binner *binner,
                              const splat_composite_cfg *cfg, const float *uniforms,
                              const void *pos_radius, const void *color_opacity, const void *normals,
                              uint32_t n, uint32_t width, uint32_t height, void *projected,
                              void *out_rgba8, void *out_rgba32f);

/* ---- multi-GPU band path (no reference equivalent — the reference is single-device) */
/* The oriented-disc projector (src/SequentialRenderer.ts:68-71,91-112): the splat is the disc
 * p + r*(t*u + b*v), u^2+v^2 <= 1, in the tangent plane of its normal (t = normalize(cross(up, n)), b =
 * cross(n, t)).  Per splat it writes
 *   discs[i]     = 8 floats {c.x, c.y, B00, B01, B10, B11, q0, q1}: the inverse of the disc's plane-to-screen
 *                  homography about its screen centre c — (u,v) = B*d / (1 - q.d), d = pixel centre - c —
 *                  which is what the rasteriser's perspective-correct interpolation evaluates; all zeros when
 *                  the quad has a corner at w <= 0 or is seen edge-on;
 *   projected[i] = a ProjectedSplat whose bounds are the disc's exact screen extent (a pure function of the
 *                  disc record), depth as splat_project, screenRadius = half the larger extent;
 * and keys / payload exactly as splat_project.  Sort and bin as usual with `projected`; composite with
 * cfg.footprint = SPLAT_FOOTPRINT_DISC and `discs` as the records.  Bit-exact against the oracle. */
int splat_project_disc(splat_ctx *ctx, const float *uniforms, const void *pos_radius, uint32_t pr_stride_vec4,
                       const void *normals, uint32_t normal_stride_vec4, uint32_t n, void *projected, void *discs,
                       void *keys, void *payload, uint32_t n_padded);
/* Projects splats [first, first+count) of the scene into projected_slice[0..count) with
 * originalIndex = global index: the per-rank share of the projector before the all-gather. */
int splat_project_slice(splat_ctx *ctx, const float *uniforms, const void *pos_radius,
                        uint32_t pr_stride_vec4, uint32_t first, uint32_t count,
                        void *projected_slice);
/* The multi-GPU exchange format: 16 bytes per splat, float4 {screen centre x, y, screen radius, depth}.
 * The 32-byte ProjectedSplat is a pure function of it — bounds = centre -/+ radius * 1.5 in the
 * projector's own operation order (src/SplatProjector.ts:119-121), originalIndex = position in the
 * gathered array — so ranks all-gather half the bytes and splat_band_frame / splat_composite rebuild
 * what they need bit-exactly (cfg->record_format = SPLAT_RECORDS_COMPACT).  splat_expand_compact
 * materialises ProjectedSplat records (originalIndex = index_base + i) for callers that want them. */
int splat_project_slice_compact(splat_ctx *ctx, const float *uniforms, const void *pos_radius,
                                uint32_t pr_stride_vec4, uint32_t first, uint32_t count,
                                void *records16_slice);
/* The oriented-disc footprint's exchange records: 48 bytes per splat, 3 x float4 {disc record (8 floats, see
 * splat_project_disc), depth, 0, 0, 0}.  The disc's bounds are a pure function of the record and the index is the
 * position in the gathered array, so splat_band_frame (cfg->footprint = SPLAT_FOOTPRINT_DISC, cfg->record_format =
 * SPLAT_RECORDS_DISC48; tile-first order, screens up to 256 x 256 tiles) needs nothing else. */
int splat_project_slice_disc(splat_ctx *ctx, const float *uniforms, const void *pos_radius, uint32_t pr_stride_vec4,
                             const void *normals, uint32_t normal_stride_vec4, uint32_t first, uint32_t count,
                             void *records48_slice);
int splat_expand_compact(splat_ctx *ctx, const void *records16, uint32_t n, uint32_t index_base,
                         void *projected);
/* Stable compaction of the splats whose clamped tile-row range meets [tile_row0, tile_row1):
 * writes (depth key, global index) pairs in ascending index order into the sorter's input
 * buffers and the number kept to *n_kept_host (synchronises). */
int splat_band_keys(splat_ctx *ctx, splat_sorter *sorter, const void *projected, uint32_t n,
                    uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_row0,
                    uint32_t tile_row1, uint32_t *n_kept_host);

/* One rank's frame after the exchange, without a host round trip: tile rows [cfg->tile_row0,
 * cfg->tile_row1) binned, depth-sorted and composited from the gathered records.  Tile-first order
 * (default): one pass over the records gives depth keys and tile ranges clamped to the band — a splat
 * outside it has an empty range — then the frame's binner and per-tile sort; sort-first order: band
 * filter (kept count on the device) -> depth sort -> bin.  records: n_records records in
 * cfg->record_format whose position is the global splat index (the all-gathered shards);
 * props/normals: the full scene in the reference's layouts (props = interleaved records); with
 * cfg->prelit, props is the plane of lit colours (splat_lit_colors) and normals may be NULL.
 * (cfg->record_format = SPLAT_RECORDS_LIT32 is not a band frame's format: round 4's variant that built lit composite records
 * for the splats a band keeps measured 20 us per rank slower on eight ranks and was removed.) */
int splat_band_frame(splat_ctx *ctx, splat_sorter *sorter, splat_binner *binner,
                     const splat_composite_cfg *cfg, const void *props, const void *normals,
                     const void *records, uint32_t n_records, uint32_t width, uint32_t height,
                     void *out_rgba8, void *out_rgba32f, void *consumed_dptr);
/* After the first frame splat_band_frame sizes its grids from the PREVIOUS frame's pair total (and, in
 * the sort-first order, kept count), x1.125, and learns its own asynchronously; a frame that outgrew
 * those bounds is reported by the next splat_band_frame / splat_band_settle call with
 * SPLAT_ERR_CAPACITY (render it again; the bounds have been raised).  splat_band_settle waits for the
 * last frame's readback, so on SPLAT_OK that frame's image is final; it returns the number of splats
 * with a tile in the band and the pair total. */
int splat_band_settle(splat_ctx *ctx, splat_sorter *sorter, splat_binner *binner, uint32_t *n_kept_host,
                      uint64_t *pairs_host);
/* Number of splats the last splat_band_frame / splat_band_keys kept (synchronises). */
int splat_band_kept(splat_ctx *ctx, splat_sorter *sorter, uint32_t *n_kept_host);

/* ---- SDF splat generation: the producer of the positions and normals this path consumes ----------
 * src/sdf/CodeGenerator.ts:97-225,276-353 (primitive / operation library, sceneSDF), src/GradientSampler.ts,
 * src/shaders/update-positions.wgsl:22-50, src/CurvatureSampler.ts:84-141.  The reference generates one WGSL function
 * per scene graph; here the graph is DATA: a postfix program (children first, then their operation — the order
 * CodeGenerator's traverse() emits) of at most SPLAT_SDF_MAX_INSTR instructions, handed over with every call (host
 * memory; it travels in the kernel arguments).  A value is vec4(distance, gradient).  Bit-exact against oracle/oracle.c. */
#define SPLAT_SDF_SPHERE 0u        /* a = {center.xyz, radius}                    sdgSphere  :100-106 */
#define SPLAT_SDF_BOX 1u           /* a = {center.xyz, half size.xyz}             sdgBox     :109-133 */
#define SPLAT_SDF_TORUS 2u         /* a = {center.xyz, major radius, minor radius} sdgTorus   :136-157 */
#define SPLAT_SDF_CAPSULE 3u       /* a = {center.xyz, height, radius}            sdgCapsule :160-176 */
#define SPLAT_SDF_UNION 16u        /* pops b, a; pushes the nearer               opUnion        :181-187 */
#define SPLAT_SDF_INTERSECTION 17u /*                                             opIntersection :190-196 */
#define SPLAT_SDF_SUBTRACTION 18u  /*                                             opSubtraction  :199-202 */
#define SPLAT_SDF_SMOOTH_UNION 19u /* a = {k}                                     opSmoothUnion  :206-224 */
#define SPLAT_SDF_MAX_INSTR 32u
typedef struct splat_sdf_instr {
    uint32_t op;
    float a[7];
} splat_sdf_instr;
/* GradientSampler.evaluateGradients: gradients[i] = sceneSDF(positions[i].xyz) (vec4 in, vec4 out). */
int splat_sdf_gradients(splat_ctx *ctx, const splat_sdf_instr *program, uint32_t n_instr, const void *positions, uint32_t n,
                        void *gradients);
/* PositionUpdater.updatePositions: next = vec4(pos - normalize(gradient) * distance, 0) (pos where the gradient vanishes). */
int splat_sdf_update_positions(splat_ctx *ctx, const void *positions, const void *gradients, uint32_t n, void *next_positions);
/* CurvatureSampler.computeScaleFactors: one f32 per point from the normals at six offsets of 0.02. */
int splat_sdf_scale_factors(splat_ctx *ctx, const splat_sdf_instr *program, uint32_t n_instr, const void *positions, uint32_t n,
                            void *scale_factors);
/* vec4(normalize(gradient), scale factor): the "curvatureData" layout splat_update_props reads
 * (src/SplatPropertyManager.ts:70-72; the reference's samplers write the two halves to separate buffers). */
int splat_sdf_curvature(splat_ctx *ctx, const void *gradients, const void *scale_factors, uint32_t n, void *curvature);
/* PointManager.generateRandomPositions (src/PointManager.ts:96-189) on the device: n points on the faces of the box
 * [aabb_min, aabb_max] (the caller's scaled global AABB of the scene: host floats), a face chosen with probability
 * proportional to its area, uniform on the face, w = 0.  The reference draws them from an unseeded Math.random on the
 * CPU and uploads them every frame (:220-231); here point i of a cloud is a pure function of (seed, i) — a 64-bit
 * counter hash (splitmix64 of seed * 0x9E3779B97F4A7C15 + 2 i and + 2 i + 1), 24 bits per uniform — so a frame's fresh
 * cloud costs one small kernel and no transfer, and the oracle restates it bit for bit (orc_sdf_seed_positions). */
int splat_sdf_seed_positions(splat_ctx *ctx, const float *aabb_min3, const float *aabb_max3, uint32_t n, uint64_t seed,
                             void *positions);
/* The producer half of the reference's frame (src/main.ts:146-180) in one launch: per point its fresh position
 * (aabb_min3 / aabb_max3 != NULL: drawn as splat_sdf_seed_positions draws point i of cloud `seed`; NULL: read from
 * positions_in), `steps` >= 1 rounds of {splat_sdf_gradients, splat_sdf_update_positions} (main.ts runs five), then
 * splat_sdf_scale_factors at the final position, splat_sdf_curvature with the gradient of the LAST evaluation (one step
 * behind the position, as main.ts:186 hands it on) and splat_update_props.  Outputs: final positions, that gradient
 * (optional), vec4(normal, scale), the 32-byte property records (optional) — bit for bit what the stage-by-stage calls
 * give (tests/test_gpu_sdf.py), without their thirteen launches. */
int splat_sdf_generate(splat_ctx *ctx, const splat_sdf_instr *program, uint32_t n_instr, const float *aabb_min3, const float *aabb_max3,
                       uint64_t seed, const void *positions_in, uint32_t n, uint32_t steps, void *positions_out, void *gradients_out,
                       void *curvature_out, void *props_out);

/* ---- the multi-GPU frame's one exchange (no reference equivalent): RCCL over xGMI ------------------
 * One process per GPU.  Rank 0 makes a unique id (splat_comm_unique_id) and hands its SPLAT_COMM_ID_BYTES to the
 * other ranks by any channel the host has (a file, a socket, MPI, torch.distributed.broadcast); every rank then
 * calls splat_comm_init (collective: returns when all `world` ranks have joined).  A frame is then
 *     splat_project_slice_compact(my slice) -> splat_allgather_records(shard, gathered) -> splat_band_frame(gathered)
 * all enqueued on the ctx stream: no host synchronisation in between.  librccl is bound at run time, on first use:
 * when it cannot be loaded these return SPLAT_ERR_COMM (nothing falls back). */
#define SPLAT_COMM_ID_BYTES 128
int splat_comm_unique_id(void *id_out /* SPLAT_COMM_ID_BYTES host bytes */);
int splat_comm_init(splat_ctx *ctx, int rank, int world, const void *unique_id, splat_comm **out);
void splat_comm_destroy(splat_comm *comm);
int splat_comm_rank(const splat_comm *comm, int *rank, int *world);
/* What the communicator ITSELF reports (ncclCommCount / ncclCommUserRank), as opposed to what splat_comm_init was told:
 * a frame loop that records these can show that the ranks it timed really formed one communicator. */
int splat_comm_count(const splat_comm *comm, int *rccl_ranks, int *rccl_rank);
/* All-gather of equal shards: rank r's bytes_per_rank bytes at `shard` land at gathered + r * bytes_per_rank on every
 * rank (in place when shard == gathered + rank * bytes_per_rank).  Asynchronous on the ctx stream (any ctx of the
 * communicator's device); timed as SPLAT_STAGE_EXCHANGE. */
int splat_allgather_records(splat_ctx *ctx, splat_comm *comm, const void *shard, void *gathered, size_t bytes_per_rank);

#ifdef __cplusplus
}
#endif
#endif
