// tile_first.hip — the frame path's binner: bin FIRST (in index order), depth-sort each tile's list after.
//
// Reference stages: GPUTileBinner.binSplats (/root/reference/src/GPUTileBinner.ts:190-338) followed by
// PerTileSorter.sort (/root/reference/src/PerTileSorter.ts:66-122, src/shaders/sort-tile-splats.wgsl) — the
// reference's own plan bins and then re-sorts every tile's list by depth.  The result contract is
// unchanged: tile t's list is the stable depth order (key, then splat index) restricted to t, i.e.
// TileBinner.binSorted (src/TileBinner.ts:426-495) applied to the stably sorted order.
//
// Why this order on MI355X: the sort-first path pays a 4-pass global sort of N keys and then gathers
// each splat's tile range in sorted (= random) order, which is bound by every CU's L1 fill rate
// (80 us for 5M splats).  Here nothing is gathered: pairs are expanded from the projector's per-index
// arrays with coalesced reads, each pair carrying its own (depth key, index) as an 8-byte payload
// through the 2-pass tile-id sort, and the depth order is established inside each tile by one
// workgroup sorting a few thousand elements in LDS.
//
//   (projector)     per 1024-splat block: pairs per low tile-id digit (first-pass histogram)
//   k_radix_rowscan digit rows -> block bases, digit totals
//   k_tf_scatter    expansion fused with the first sort pass      N*8 B read, P*9 B written
//   k_tf_upsweep2 / k_radix_rowscan / k_tf_downsweep2
//                   second (high digit) pass, 8-byte payload      P*(1 + 9 + 8) B
//                   + tile offsets from the second pass's scanned histogram (no search, no sorted tile ids): the
//                   downsweep launch's last workgroups (tf_offsets_block)
//   k_tile_sort     per tile: LSD radix on (key - tile min key)    P*8 B read, P*4 B written
#include "common.h"
#include "tile_range.h"

constexpr uint32_t TF_THREADS = 256;
constexpr uint32_t TF_STAGE = 4096; // pairs staged per block for contiguous stores
static_assert(TF_BLOCK_LARGE == 1024 && TF_BLOCK_SMALL == 256, "the stage word packs the block-local slot in 10 bits; blocks are 256 x {4, 1}");

__device__ __forceinline__ uint32_t range32_hits(uint32_t r) {
    const uint32_t tx0 = r & 0xffu, tx1 = (r >> 8) & 0xffu, ty0 = (r >> 16) & 0xffu, ty1 = r >> 24;
    return (tx0 > tx1 || ty0 > ty1) ? 0u : (tx1 - tx0 + 1) * (ty1 - ty0 + 1);
}

// ---------------------------------------------------------------------------------------------
// PerTileSorter: one workgroup per tile sorts the tile's (depth key, splat index) elements by key,
// stable, and writes the tile's index list.
//
// LSD radix, 8 bits per pass, on (key - smallest key of the tile): a tile's keys span a narrow depth
// range (<= 24 bits of difference at the bench sizes), so 3 passes instead of 4, and none at all for
// a tile whose keys are equal.  Lists up to 5888 elements are sorted inside LDS — every thread
// holds its elements in registers across the pass's barrier, so one LDS array is read and rewritten
// in place.  Longer lists take the same passes through global memory (this tile's segment of the two
// pair-value buffers as ping-pong), 4096 elements at a time.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t TS_THREADS = 256, TS_WAVES = 4;
// Two size classes, one launch each over all tiles (a workgroup whose tile belongs to the other class
// exits at once): the kernel is latency-bound (six barriers per pass, dependent LDS round trips), so
// what matters is resident workgroups per CU, and most lists are short.
//   short  <= 2048 / 3072 / 4096 elements: 8 / 12 / 16 per thread, 16 / 24 / 32 KiB staged -> six / five / four workgroups per
//           CU.  Which of the three: by the frame's mean list length (tile_sort_launch) — the denser the lists, the more of
//           them are worth taking out of the long class at the price of fewer resident workgroups in the short one
//   long    the rest: up to 24 per thread, 46 KiB staged -> three workgroups per CU; beyond 5888 elements the passes go
//           through global memory
constexpr uint32_t TS_LONG_ITEMS = 24; // (the short class holds 8, 12 or 16 elements per thread: tile_sort_launch picks per frame)
constexpr uint32_t TS_LDS_ELEMS = 5888; // 46 KiB + 5 KiB of counters: three workgroups in a CU's 160 KiB
constexpr uint32_t TS_CHUNK_ITEMS = 16; // pairs per thread and chunk of the global-memory passes (at most: a class of fewer holds fewer)

struct TileSortShared {
    uint32_t wave_hist[TS_WAVES][256];
    uint32_t digit_base[256];
    uint32_t wave_sums[TS_WAVES];
    uint32_t kmin, kmax;
};

// rank of this lane's element among the elements of the same digit seen so far by this wave
// (earlier instructions, then lower lanes), adding it to the wave's digit counter
template <bool RANK_ATOMIC>
__device__ __forceinline__ uint32_t wave_rank(uint32_t *wave_hist, uint32_t d) {
    if (RANK_ATOMIC) return atomicAdd(&wave_hist[d], 1u); // lane-ordered returning LDS atomic (probed per context)
    const uint64_t active = __ballot(true); // callers rank under `if (p < n)`: peers are active lanes only
    uint32_t plo = (uint32_t)active, phi = (uint32_t)(active >> 32);
#pragma unroll
    for (uint32_t b = 0; b < 8; ++b) {
        const uint32_t m = (uint32_t)(((int32_t)(d << (31 - b))) >> 31);
        const uint64_t bal = __ballot(m != 0);
        plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)bal, m, 0x90);
        phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(bal >> 32), m, 0x90);
    }
    const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0));
    const uint32_t leader = plo ? (uint32_t)__builtin_ctz(plo) : 32u + (uint32_t)__builtin_ctz(phi);
    uint32_t prev = 0;
    if (below == 0) prev = atomicAdd(&wave_hist[d], (uint32_t)(__popc(plo) + __popc(phi)));
    return (uint32_t)__shfl((int)prev, (int)leader) + below;
}

// thread d: turns the four waves' counts of digit d into exclusive wave prefixes (in place) and returns
// the total; then digit_base[d] = exclusive scan of the totals over the digits (+ nothing else)
__device__ __forceinline__ uint32_t ts_wave_prefixes(uint32_t (*wave_hist)[256], uint32_t tid) {
    const uint32_t c0 = wave_hist[0][tid], c1 = wave_hist[1][tid], c2 = wave_hist[2][tid], c3 = wave_hist[3][tid];
    wave_hist[0][tid] = 0;
    wave_hist[1][tid] = c0;
    wave_hist[2][tid] = c0 + c1;
    wave_hist[3][tid] = c0 + c1 + c2;
    return c0 + c1 + c2 + c3;
}

// exclusive scan of one value per thread over the 256 threads (two barriers)
__device__ __forceinline__ uint32_t ts_scan256(uint32_t *wave_sums, uint32_t v, uint32_t tid) {
    const uint32_t lane = tid & 63, w = tid >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t t = __shfl_up(incl, s);
        if ((int)lane >= s) incl += t;
    }
    if (lane == 63) wave_sums[w] = incl;
    __syncthreads();
    const uint32_t wprefix = (w > 0 ? wave_sums[0] : 0u) + (w > 1 ? wave_sums[1] : 0u) + (w > 2 ? wave_sums[2] : 0u);
    __syncthreads();
    return wprefix + incl - v;
}

// ---------------------------------------------------------------------------------------------
// The pairs are never written in expansion order: a 1024-splat block IS a partition of the tile-id
// sort's first pass.  The projector (project.hip: k_project_hist; frame.hip: k_band_prepare_tf for a
// multi-GPU band) counts the block's pairs per low tile-id digit while the tile rectangle is in
// registers (the upsweep's histogram); k_tf_scatter expands the block into LDS, ranks the pairs by
// that digit and scatters (tile id | depth key, index) to their first-pass positions (the downsweep).
// Saved per frame: the expanded array's write and two reads (12 + 4 + 12 B per pair).
//
// Order inside a block: ascending splat index, then row-major over the splat's tile rectangle; the
// stable passes keep it inside each tile, so equal depth keys still resolve by ascending index.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t TFS_ITEMS = TF_STAGE / TF_THREADS; // 16 staged pairs per thread and round

struct TfScatterShared {
    uint32_t wave_hist[TS_WAVES][256];
    uint32_t digit_base[256];  // round-local start of each digit's run
    uint32_t global_base[256]; // where this block's pairs of each digit start in the output (advances per round)
    uint32_t wave_sums[TS_WAVES];
};

// d_total (out): [0] = pair total of this frame (the sum of the digit totals); [2] = the count the
// later kernels work on: the total, or 0 when it exceeds pair_limit (then nothing is written and
// the overflow flag is raised: the frame is rendered again with room).
// TF_PER_THREAD splats per thread: 4 (1024-splat blocks) or 1 (256-splat blocks of small frames, common.h).
#ifndef TF_SCATTER_WAVES
#define TF_SCATTER_WAVES 5 // (tuning knob of tools/build_variant.sh: 4 and 6 measured, profiles/r03_q_small_knobs_C2.txt)
#endif
// COMPACTED (a multi-GPU band's frame, frame.hip: k_band_prepare_tfc): a workgroup's input is not 1024 consecutive splat
// indices but the splats its band KEEPS of a group of TF_GROUP consecutive records, compacted in index order to the start of
// the group's segment of range32 / depth_keys, their indices beside them (`cidx`), their number in kept[group] — a band of
// an eighth of the screen keeps an eighth of the splats, and what this kernel costs is its workgroups, not its pairs.  The
// histogram has one column per group.  A group that keeps more than 1024 splats goes in rounds of 1024.
constexpr uint32_t TF_GROUP = 4096;
template <bool RANK_ATOMIC, uint32_t TF_PER_THREAD, bool COMPACTED = false>
__global__ __launch_bounds__(TF_THREADS, TF_SCATTER_WAVES) void k_tf_scatter(const uint32_t *__restrict__ range32,
                                                           const uint32_t *__restrict__ depth_keys, uint32_t n, uint32_t ntx,
                                                           uint32_t mask, uint32_t num_parts,
                                                           const uint32_t *__restrict__ scanned_hist,
                                                           const uint32_t *__restrict__ totals, uint32_t *__restrict__ d_total,
                                                           uint32_t pair_limit, uint32_t *__restrict__ overflow,
                                                           uint8_t *__restrict__ out_hi, uint2 *__restrict__ out_val,
                                                           uint32_t lo_bits, uint32_t align_m1, TfRuns runs,
                                                           uint32_t *__restrict__ offsets_out, uint32_t tiles, uint32_t *report,
                                                           uint32_t seq, const uint32_t *__restrict__ cidx,
                                                           const uint32_t *__restrict__ kept, uint32_t xcd_per) {
    static_assert(!COMPACTED || TF_PER_THREAD == 4, "compacted groups are read four splats per thread");
    const uint32_t blk = xcd_block_of(blockIdx.x, xcd_per); // this workgroup's block of splats = its histogram column
    if (blk >= num_parts) return;
    __shared__ TfScatterShared sh;
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t stage[TF_STAGE]; // tile id << 10 | block-local slot
    __shared__ uint32_t s_key[TF_THREADS * TF_PER_THREAD];
    __shared__ uint32_t s_idx[COMPACTED ? TF_THREADS * TF_PER_THREAD : 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint32_t first = blk * (COMPACTED ? TF_GROUP : TF_THREADS * TF_PER_THREAD);
    // (COMPACTED: n = this group's kept splats end at first + kept[group]; the loads below stop there)
    if (COMPACTED) n = first + min(kept[blk], TF_GROUP);
    // every global load of the prologue is issued before anything waits on one: the block's ranges and
    // keys, the digit totals, and this block's row of scanned histogram (measured per phase, a workgroup
    // spent 15 % of its life on the totals alone when they were loaded, scanned and waited for first)
    static_assert(TF_PER_THREAD == 4 || TF_PER_THREAD == 1, "uint4 or scalar loads");
    uint32_t i0 = first + tid * TF_PER_THREAD; // thread t owns splats first + 4t .. 4t+3 (one 16-byte load each of ranges and keys)
    uint4 rr = make_uint4(1u, 1u, 1u, 1u), kk = make_uint4(0, 0, 0, 0); // 1 = empty range
    if (TF_PER_THREAD == 1) {
        if (i0 < n) { rr.x = range32[i0]; kk.x = depth_keys[i0]; }
    } else if (i0 + 3 < n) {
        rr = reinterpret_cast<const uint4 *>(range32)[i0 >> 2];
        kk = reinterpret_cast<const uint4 *>(depth_keys)[i0 >> 2];
    } else {
        if (i0 < n) { rr.x = range32[i0]; kk.x = depth_keys[i0]; }
        if (i0 + 1 < n) { rr.y = range32[i0 + 1]; kk.y = depth_keys[i0 + 1]; }
        if (i0 + 2 < n) { rr.z = range32[i0 + 2]; kk.z = depth_keys[i0 + 2]; }
    }
    const uint32_t digit_total = totals[tid];
    const uint32_t row_prefix = tid <= mask ? scanned_hist[(size_t)tid * num_parts + blk] : 0u;

    // digit starts = exclusive scan of the digit totals, each rounded up to whole partitions of the second pass when
    // there is one (align_m1 = TF2_PART - 1: every partition then holds pairs of ONE low digit, see k_tf_downsweep2);
    // the sum of the totals themselves is this frame's pair total
    const uint32_t digit_room = (digit_total + align_m1) & ~align_m1;
    // (the frame's pair total in 64 bits: a sum that wrapped would pass the `fits` test below with a small value and let
    // the scatter write out of bounds; k_radix_rowscan saturates each digit's total at 2^30)
    uint32_t gincl = digit_total, aincl = digit_room;
    unsigned long long g64 = digit_total;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t t = __shfl_up(gincl, s), a = __shfl_up(aincl, s);
        const unsigned long long t64 = __shfl_up(g64, s);
        if ((int)lane >= s) {
            gincl += t;
            aincl += a;
            g64 += t64;
        }
    }
    __shared__ unsigned long long wsum64[4];
    if (lane == 63) {
        sh.wave_sums[w] = gincl;
        wsum[w] = aincl;
        wsum64[w] = g64;
    }
    __syncthreads();
    const uint32_t aprefix = (w > 0 ? wsum[0] : 0u) + (w > 1 ? wsum[1] : 0u) + (w > 2 ? wsum[2] : 0u);
    const unsigned long long all_pairs64 = wsum64[0] + wsum64[1] + wsum64[2] + wsum64[3];
    const uint32_t all_pairs = all_pairs64 > 0xffffffffull ? 0xffffffffu : (uint32_t)all_pairs64; // saturated: it can only fail `fits`
    const uint32_t run_start = aprefix + aincl - digit_room; // where digit tid's pairs start in the output
    if (blk == 0) {
        const bool fits = all_pairs <= pair_limit;
        if (tid == 0) {
            d_total[0] = all_pairs;
            d_total[2] = fits ? all_pairs : 0u;
            d_total[3] = 0u; // (k_tile_sort's first launch counts the tiles beyond its short class here)
            if (!fits) atomicOr(overflow, 1u);
        }
        // the second pass's view of this output: per low digit where its run starts and how many pairs it holds,
        // per partition the digit it belongs to, and the number of partitions (all zero when the pairs do not fit)
        runs.start[tid] = fits ? run_start : 0u;
        runs.total[tid] = fits ? digit_total : 0u;
        if (align_m1 && fits)
            for (uint32_t j = 0, first_part = run_start / (align_m1 + 1u); j < digit_room / (align_m1 + 1u); ++j)
                runs.part_digit[first_part + j] = (uint8_t)tid;
        if (tid == 255) *runs.parts = (align_m1 && fits) ? (run_start + digit_room) / (align_m1 + 1u) : 0u;
        // a screen of at most 256 tiles is sorted by this pass alone and its (dense) runs ARE the tiles: the tile
        // offsets go out from here (tf_offsets_block has nothing to read them from)
        if (offsets_out) {
            if (tid < tiles) offsets_out[tid] = fits ? run_start : 0u;
            if (tid == 0) {
                offsets_out[tiles] = fits ? all_pairs : 0u;
                if (report) tile_report(d_total, report, seq);
            }
        }
    }
    if (all_pairs > pair_limit) return;
    // where this block's pairs of digit tid start: digit start + the earlier blocks' share
    sh.global_base[tid] = run_start + row_prefix;
    for (;;) { // (COMPACTED: rounds of 1024 kept splats; otherwise once)
    __syncthreads(); // wave_sums and wsum are reused below (and the round before has left stage, s_key, s_idx)

    uint32_t r[TF_PER_THREAD], h[TF_PER_THREAD];
    if constexpr (TF_PER_THREAD == 4) {
        reinterpret_cast<uint4 *>(s_key)[tid] = kk;
        r[0] = rr.x; r[1] = rr.y; r[2] = rr.z; r[3] = rr.w;
        if constexpr (COMPACTED) { // the kept splats' own indices (what a pair carries)
            uint4 ii = make_uint4(0, 0, 0, 0);
            if (i0 + 3 < n) ii = reinterpret_cast<const uint4 *>(cidx)[i0 >> 2];
            else {
                if (i0 < n) ii.x = cidx[i0];
                if (i0 + 1 < n) ii.y = cidx[i0 + 1];
                if (i0 + 2 < n) ii.z = cidx[i0 + 2];
            }
            reinterpret_cast<uint4 *>(s_idx)[tid] = ii;
        }
    } else {
        s_key[tid] = kk.x;
        r[0] = rr.x;
    }
#pragma unroll
    for (uint32_t k = 0; k < TF_PER_THREAD; ++k) h[k] = range32_hits(r[k]);
    // offsets of every splat's pairs inside the block, in ascending splat index: one block scan of the
    // per-thread totals, then the thread's own running sum
    uint32_t off[TF_PER_THREAD];
    uint32_t carry;
    {
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < TF_PER_THREAD; ++k) mine += h[k];
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if ((int)lane >= d) incl += t;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        const uint32_t s0 = wsum[0], s1 = wsum[1], s2 = wsum[2], s3 = wsum[3];
        off[0] = (w > 0 ? s0 : 0u) + (w > 1 ? s1 : 0u) + (w > 2 ? s2 : 0u) + incl - mine;
#pragma unroll
        for (uint32_t k = 1; k < TF_PER_THREAD; ++k) off[k] = off[k - 1] + h[k - 1];
        carry = s0 + s1 + s2 + s3;
    }
    const uint32_t total = carry;
    const bool more_splats = COMPACTED && first + TF_THREADS * TF_PER_THREAD < n; // (uniform) another round of this group's kept splats follows
    if (total == 0 && !more_splats) return;
    // rounds of TF_STAGE pairs (one round unless the block's splats are unusually large)
    for (uint32_t c0 = 0; c0 < total; c0 += TF_STAGE) {
        const uint32_t cnt = (total - c0 < TF_STAGE) ? total - c0 : TF_STAGE;
        for (uint32_t i = tid; i < TS_WAVES * 256; i += TF_THREADS) (&sh.wave_hist[0][0])[i] = 0;
#pragma unroll
        for (uint32_t k = 0; k < TF_PER_THREAD; ++k) {
            if (h[k] == 0 || off[k] >= c0 + TF_STAGE || off[k] + h[k] <= c0) continue;
            const uint32_t tx0 = r[k] & 0xffu, tx1 = (r[k] >> 8) & 0xffu, ty0 = (r[k] >> 16) & 0xffu, ty1 = r[k] >> 24;
            const uint32_t slot = tid * TF_PER_THREAD + k;
            uint32_t o = off[k] - c0; // may wrap below zero for pairs of an earlier round: the range check rejects them
            for (uint32_t ty = ty0; ty <= ty1; ++ty)
                for (uint32_t tx = tx0; tx <= tx1; ++tx) {
                    if (o < TF_STAGE) stage[o] = ((ty * ntx + tx) << 10) | slot;
                    ++o;
                }
        }
        __syncthreads();
        // rank by digit, wave-striped positions (wave, item, lane), items in groups of four
        const uint32_t items = ((cnt + TF_THREADS - 1) / TF_THREADS + 3u) & ~3u;
        const uint32_t wbase = w * items * 64 + lane;
        uint32_t el[TFS_ITEMS], rank[TFS_ITEMS];
#pragma unroll
        for (uint32_t g = 0; g < TFS_ITEMS; g += 4) {
            if (g < items) {
#pragma unroll
                for (uint32_t i = g; i < g + 4; ++i) {
                    const uint32_t p = wbase + i * 64;
                    el[i] = stage[p < cnt ? p : cnt - 1];
                }
#pragma unroll
                for (uint32_t i = g; i < g + 4; ++i) {
                    rank[i] = 0;
                    if (wbase + i * 64 < cnt) rank[i] = wave_rank<RANK_ATOMIC>(sh.wave_hist[w], (el[i] >> 10) & mask);
                }
            }
        }
        __syncthreads();
        const uint32_t dcount = ts_wave_prefixes(sh.wave_hist, tid);
        const uint32_t excl = ts_scan256(sh.wave_sums, dcount, tid);
        sh.digit_base[tid] = excl;
        __syncthreads();
        // reorder in place (every thread holds its elements in registers): digit runs become contiguous
#pragma unroll
        for (uint32_t g = 0; g < TFS_ITEMS; g += 4) {
            if (g < items) {
#pragma unroll
                for (uint32_t i = g; i < g + 4; ++i) {
                    const uint32_t d = (el[i] >> 10) & mask;
                    if (wbase + i * 64 < cnt) stage[sh.digit_base[d] + sh.wave_hist[w][d] + rank[i]] = el[i];
                }
            }
        }
        __syncthreads();
        for (uint32_t pos = tid; pos < cnt; pos += TF_THREADS) {
            const uint32_t e = stage[pos], slot = e & 1023u, d = (e >> 10) & mask;
            const uint32_t g = sh.global_base[d] + (pos - sh.digit_base[d]);
            out_hi[g] = (uint8_t)(e >> (10u + lo_bits)); // (the low digit is the run the pair sits in)
            out_val[g] = make_uint2(s_key[slot], COMPACTED ? s_idx[slot] : first + slot);
        }
        if (c0 + TF_STAGE >= total && !more_splats) break; // (the usual case: one round)
        __syncthreads();
        sh.global_base[tid] += dcount; // the next round's pairs of digit tid follow this round's
        __syncthreads();
    }
    if (!more_splats) return;
    // the group's next 1024 kept splats
    first += TF_THREADS * TF_PER_THREAD;
    {
        const uint32_t j0 = first + tid * TF_PER_THREAD;
        rr = make_uint4(1u, 1u, 1u, 1u);
        kk = make_uint4(0, 0, 0, 0);
        if (j0 + 3 < n) {
            rr = reinterpret_cast<const uint4 *>(range32)[j0 >> 2];
            kk = reinterpret_cast<const uint4 *>(depth_keys)[j0 >> 2];
        } else {
            if (j0 < n) { rr.x = range32[j0]; kk.x = depth_keys[j0]; }
            if (j0 + 1 < n) { rr.y = range32[j0 + 1]; kk.y = depth_keys[j0 + 1]; }
            if (j0 + 2 < n) { rr.z = range32[j0 + 2]; kk.z = depth_keys[j0 + 2]; }
        }
        i0 = j0;
    }
    } // (rounds of kept splats)
}

// The end of the in-LDS sort: the tile's index list goes out, and THE ORDER CHECK.  The tile's list must be in strictly
// increasing (depth key, splat index) order — that IS the contract (TileBinner.binSorted applied to the stable depth
// order), so verifying it here verifies every pass that led to it, in this kernel and in the two passes of the tile-id
// sort before it, whichever way they ranked: an unstable rank anywhere leaves equal digits out of their earlier order,
// i.e. keys or tied indices out of order in the final list.  A violation raises the frame's flag, and the host renders
// the frame again with ballot ranking (binner_settle).
// A wave takes 63 elements per step and reads 64: an element's successor sits in the next lane (wave_shl:1 — no LDS
// instruction: the kernel is bound by those), lane 63 holds the successor of lane 62 and writes nothing — the next step's
// lane 0 has that element.
__device__ __forceinline__ void tile_list_out(uint2 *s_el, uint32_t n, uint32_t *__restrict__ out, uint32_t *frame_flags, bool inject,
                                              uint32_t inject_pos, uint32_t tid) {
    const uint32_t lane = tid & 63, w = tid >> 6;
    if (inject) { // test hook (splat_debug_inject_order_fault): the check below must see this
        if (tid == 0 && n >= inject_pos + 2u) {
            const uint2 a = s_el[inject_pos];
            s_el[inject_pos] = s_el[inject_pos + 1];
            s_el[inject_pos + 1] = a;
        }
        __syncthreads();
    }
    bool bad = false;
    for (uint32_t s0 = 0; s0 * 63u < n; s0 += 4 * TS_WAVES) { // (uniform trip count: the lane shifts below want whole waves)
        uint2 a[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) { // four reads in flight, as in the passes
            const uint32_t p = (s0 + i * TS_WAVES + w) * 63u + lane;
            a[i] = s_el[p < n ? p : n - 1];
        }
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            const uint32_t p = (s0 + i * TS_WAVES + w) * 63u + lane;
            if (p < n && lane < 63) out[p] = a[i].y;
            const uint32_t bx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a[i].x, 0x130, 0xf, 0xf, false);
            const uint32_t by = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a[i].y, 0x130, 0xf, 0xf, false);
#ifndef TS_NO_ORDER_CHECK // (measuring knob of tools/build_variant.sh: what the check costs, profiles/r03_d_order_check_cost.txt)
            const unsigned long long ka = ((unsigned long long)a[i].x << 32) | a[i].y, kb = ((unsigned long long)bx << 32) | by;
            if (p + 1 < n && lane < 63) bad |= ka >= kb;
#endif
        }
    }
    if (__any(bad) && lane == 0) atomicOr(frame_flags, FRAME_FLAG_ORDER);
}

// The test build (-DSPLAT_TEST_HOOKS, libsplat_hip_hooks.so: tests/ only) gives the kernel three more parameters: the tile whose
// finished list gets two neighbours swapped before the order check, where, and a dispatch order for the workgroups.  The shipped
// kernels do not carry them.
#ifdef SPLAT_TEST_HOOKS
#define TS_HOOK_PARAMS , uint32_t inject_tile, uint32_t inject_pos, const uint32_t *__restrict__ order
#define TS_HOOK_ARGS , inject, inject_pos, ctx->debug_sort_order
#else
#define TS_HOOK_PARAMS
#define TS_HOOK_ARGS
#endif
template <bool RANK_ATOMIC, uint32_t TS_MAX_ITEMS, bool LAST_CLASS>
__global__ __launch_bounds__(TS_THREADS, TS_MAX_ITEMS <= 8 ? 6 : TS_MAX_ITEMS <= 12 ? 5 : TS_MAX_ITEMS <= 16 ? 4 : 3) void k_tile_sort(const uint32_t *__restrict__ offsets, uint32_t tiles,
                                                                                    uint32_t n_above, uint2 *vals, uint2 *scratch,
                                                                                    uint32_t *__restrict__ out_idx,
                                                                                    uint32_t *__restrict__ counts,
                                                                                    uint32_t *__restrict__ frame_flags TS_HOOK_PARAMS) {
    static_assert(TS_MAX_ITEMS % 4 == 0, "items are processed in groups of four");
    constexpr uint32_t TS_CAP = TS_MAX_ITEMS * TS_THREADS < TS_LDS_ELEMS ? TS_MAX_ITEMS * TS_THREADS : TS_LDS_ELEMS;
    __shared__ TileSortShared sh;
    __shared__ uint2 s_el[TS_CAP];
    uint32_t *run_base = reinterpret_cast<uint32_t *>(s_el); // long-list path (s_el unused there): start of each digit's run
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#ifdef SPLAT_TEST_HOOKS
    const uint32_t t = order ? order[blockIdx.x] : blockIdx.x; // (splat_debug_set_tile_sort_order: any permutation gives the same lists)
#else
    const uint32_t t = blockIdx.x;
    constexpr uint32_t inject_tile = 0xffffffffu, inject_pos = 0; // (no tile is the victim: the swap below folds away)
#endif
    const uint32_t base = offsets[t], n = offsets[t + 1] - base;
    if (counts && tid == 0) { // (first launch: the tile counts the composite reads, and how many tiles are beyond this class)
        counts[t] = n;
        if (n > TS_MAX_ITEMS * TS_THREADS) atomicAdd(frame_flags + 2, 1u);
    }
    if (n <= n_above || (!LAST_CLASS && n > TS_CAP)) return; // empty, or another class's tile
    uint2 *src = vals + base, *dst = scratch + base;
    const bool in_lds = !LAST_CLASS || n <= TS_CAP;

    if (tid == 0) {
        sh.kmin = 0xffffffffu;
        sh.kmax = 0;
    }
    __syncthreads();

    // load (LDS path) and the tile's key range.  All of a thread's loads are issued before the first
    // is used: one load per loop trip left every workgroup waiting out up to 23 memory latencies in a
    // row (48 of the kernel's 76 us at C2 went there).  The pairs are loaded in the passes' own layout — position =
    // (wave, item, lane) — and stay in registers: the first pass ranks them from there (no staging write and read).
    uint32_t lo = 0xffffffffu, hi = 0;
    const uint32_t items = ((n + TS_THREADS - 1) / TS_THREADS + 3u) & ~3u;
    const uint32_t wbase = w * items * 64 + lane;
    uint2 el[TS_MAX_ITEMS];
    if (in_lds) {
#pragma unroll
        for (uint32_t g = 0; g < TS_MAX_ITEMS; g += 4) {
            if (g < items) {
#pragma unroll
                for (uint32_t i = g; i < g + 4; ++i) {
                    const uint32_t p = wbase + i * 64;
                    el[i] = src[p < n ? p : n - 1]; // (padding re-reads the last pair: harmless for the key range)
                }
            }
        }
#pragma unroll
        for (uint32_t g = 0; g < TS_MAX_ITEMS; g += 4) {
            if (g < items) {
#pragma unroll
                for (uint32_t i = g; i < g + 4; ++i) {
                    lo = min(lo, el[i].x);
                    hi = max(hi, el[i].x);
                }
            }
        }
    } else {
        for (uint32_t p = tid; p < n; p += TS_THREADS) {
            const uint32_t k = src[p].x;
            lo = min(lo, k);
            hi = max(hi, k);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, d));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, d));
    }
    if (lane == 0) {
        atomicMin(&sh.kmin, lo);
        atomicMax(&sh.kmax, hi);
    }
    __syncthreads();
    const uint32_t kmin = sh.kmin, range = sh.kmax - kmin;
    const uint32_t passes = range == 0 ? 0u : (32u - (uint32_t)__builtin_clz(range) + 7u) / 8u;

    if (in_lds) {
        // per thread, wave-striped: position = (wave, item, lane).  Items go in groups of four so that
        // four LDS reads, then four returning atomics, are in flight together (a branch per item
        // would expose every round trip); positions past n are padding and do nothing.
        if (passes == 0) { // every key equal: the list is in order as it stands
#pragma unroll
            for (uint32_t g = 0; g < TS_MAX_ITEMS; g += 4) {
                if (g < items) {
#pragma unroll
                    for (uint32_t i = g; i < g + 4; ++i)
                        if (wbase + i * 64 < n) s_el[wbase + i * 64] = el[i];
                }
            }
            __syncthreads();
        }
        for (uint32_t pass = 0; pass < passes; ++pass) {
            const uint32_t shift = pass * 8;
            for (uint32_t i = tid; i < TS_WAVES * 256; i += TS_THREADS) (&sh.wave_hist[0][0])[i] = 0;
            __syncthreads();
            uint32_t rank[TS_MAX_ITEMS];
#pragma unroll
            for (uint32_t g = 0; g < TS_MAX_ITEMS; g += 4) {
                if (g < items) {
                    if (pass > 0) { // (the first pass's pairs are in registers)
#pragma unroll
                        for (uint32_t i = g; i < g + 4; ++i) {
                            const uint32_t p = wbase + i * 64;
                            el[i] = s_el[p < n ? p : n - 1];
                        }
                    }
#pragma unroll
                    for (uint32_t i = g; i < g + 4; ++i) {
                        const uint32_t p = wbase + i * 64;
                        rank[i] = 0;
                        if (p < n) rank[i] = wave_rank<RANK_ATOMIC>(sh.wave_hist[w], ((el[i].x - kmin) >> shift) & 255u);
                    }
                }
            }
            __syncthreads();
            // digit d's run starts at the exclusive scan of the digit totals and wave w's elements of digit d follow
            // those of the waves before it: both folded into the wave's own table — ONE lookup per element (the kernel
            // is bound by LDS operations: measured per phase, the passes are 60 % of a long tile's time once its loads
            // are hidden)
            const uint32_t c0 = sh.wave_hist[0][tid], c1 = sh.wave_hist[1][tid], c2 = sh.wave_hist[2][tid], c3 = sh.wave_hist[3][tid];
            const uint32_t excl = ts_scan256(sh.wave_sums, (c0 + c1) + (c2 + c3), tid);
            sh.wave_hist[0][tid] = excl;
            sh.wave_hist[1][tid] = excl + c0;
            sh.wave_hist[2][tid] = excl + c0 + c1;
            sh.wave_hist[3][tid] = excl + c0 + c1 + c2;
            __syncthreads();
#pragma unroll
            for (uint32_t g = 0; g < TS_MAX_ITEMS; g += 4) {
                if (g < items) {
                    uint32_t pos[4];
#pragma unroll
                    for (uint32_t i = g; i < g + 4; ++i) pos[i - g] = sh.wave_hist[w][((el[i].x - kmin) >> shift) & 255u] + rank[i];
#pragma unroll
                    for (uint32_t i = g; i < g + 4; ++i)
                        if (wbase + i * 64 < n) s_el[pos[i - g]] = el[i];
                }
            }
            __syncthreads();
        }
        tile_list_out(s_el, n, out_idx + base, frame_flags, t == inject_tile, inject_pos, tid);
        return;
    }

    // ---- long list: the same passes through global memory -------------------------------------
    // (in chunks of as many pairs per thread as the in-LDS path holds: a short class that runs as the last one must not
    // pay registers for this path)
    constexpr uint32_t CH_ITEMS = TS_MAX_ITEMS < TS_CHUNK_ITEMS ? TS_MAX_ITEMS : TS_CHUNK_ITEMS;
    for (uint32_t pass = 0; pass < passes; ++pass) {
        const uint32_t shift = pass * 8;
        // digit totals of the whole list -> start of every digit's run
        sh.digit_base[tid] = 0;
        __syncthreads();
        for (uint32_t p = tid; p < n; p += TS_THREADS) atomicAdd(&sh.digit_base[((src[p].x - kmin) >> shift) & 255u], 1u);
        __syncthreads();
        const uint32_t tot = sh.digit_base[tid];
        const uint32_t start = ts_scan256(sh.wave_sums, tot, tid);
        run_base[tid] = start;
        __syncthreads();
        for (uint32_t c0 = 0; c0 < n; c0 += CH_ITEMS * TS_THREADS) {
            for (uint32_t i = tid; i < TS_WAVES * 256; i += TS_THREADS) (&sh.wave_hist[0][0])[i] = 0;
            __syncthreads();
            uint2 el[CH_ITEMS];
            uint32_t rank[CH_ITEMS];
#pragma unroll
            for (uint32_t i = 0; i < CH_ITEMS; ++i) {
                const uint32_t p = c0 + w * (CH_ITEMS * 64) + i * 64 + lane;
                el[i] = src[p < n ? p : n - 1];
                rank[i] = 0;
                if (p < n) rank[i] = wave_rank<RANK_ATOMIC>(sh.wave_hist[w], ((el[i].x - kmin) >> shift) & 255u);
            }
            __syncthreads();
            const uint32_t dcount = ts_wave_prefixes(sh.wave_hist, tid);
            const uint32_t rb = run_base[tid];
            sh.digit_base[tid] = rb; // this chunk's elements of digit tid start here
            run_base[tid] = rb + dcount;
            __syncthreads();
#pragma unroll
            for (uint32_t i = 0; i < CH_ITEMS; ++i) {
                const uint32_t p = c0 + w * (CH_ITEMS * 64) + i * 64 + lane;
                if (p < n) {
                    const uint32_t d = ((el[i].x - kmin) >> shift) & 255u;
                    dst[sh.digit_base[d] + sh.wave_hist[w][d] + rank[i]] = el[i];
                }
            }
            __syncthreads();
        }
        uint2 *tmp = src; src = dst; dst = tmp;
        __threadfence_block(); // the next pass reads what other waves of this workgroup just stored
        __syncthreads();
    }
    bool bad = false; // the order check, as above
    for (uint32_t p = tid; p < n; p += TS_THREADS) {
        const bool inj = t == inject_tile && n >= inject_pos + 2u;
        uint2 a = src[p];
        if (inj && (p == inject_pos || p == inject_pos + 1)) a = src[p == inject_pos ? p + 1 : p - 1];
        out_idx[base + p] = a.y;
        if (p + 1 < n) {
            uint2 b = src[p + 1];
            if (inj && (p + 1 == inject_pos || p + 1 == inject_pos + 1)) b = src[p + 1 == inject_pos ? p + 2 : p];
            bad |= a.x > b.x || (a.x == b.x && a.y >= b.y);
        }
    }
    if (__any(bad) && lane == 0) atomicOr(frame_flags, FRAME_FLAG_ORDER);
}

// mean_list: the frame's pairs per tile of its band, as far as the host knows them (the previous frame's in a sync-free frame).
int tile_sort_launch(splat_ctx *ctx, const uint32_t *offsets, uint32_t tiles, uint2 *vals, uint2 *scratch, uint32_t *out_idx,
                     uint32_t *counts, uint32_t *frame_flags, uint32_t mean_list, uint32_t band_tiles, uint32_t long_tiles_hint,
                     uint32_t *short_class_io) {
    {
        int prc = ctx_resolve_rank_mode(ctx); // (probe of the LDS atomics' lane order, once per context)
        if (prc != SPLAT_OK) return prc;
    }
    const bool ra = rank_atomic_ok(ctx, true); // (every list is checked below: atomics are allowed here by default)
#ifdef SPLAT_TEST_HOOKS
    const uint32_t inject = ctx->inject_order_fault ? ctx->inject_order_fault - 1u : 0xffffffffu; // one-shot test hook
    const uint32_t inject_pos = ctx->inject_order_position;
    ctx->inject_order_fault = 0;
#endif
    // A screen — or a multi-GPU rank's band of tile rows — of so few tiles that the long class's kernel takes them in a round or
    // a few gains nothing from a second, denser class: one launch sorts every tile (a dependent launch costs ~5 us whatever it
    // does, and each launch ends with the life of its longest tile).  Measured with virtual ranks (profiles/r05_h_band_tile_sort_
    // one_class.txt): bands of 840 tiles (C2, eight ranks) 32 -> 22 us, of 2040 (four ranks) 35 -> 29, of ~3000 (C3, eight ranks)
    // 48 -> 44; a whole C2 screen (8160 tiles, 4969 of them non-empty) is slower with one launch (77 -> 83 us,
    // profiles/r03_j_tile_sort_one_launch_C2.txt).
    const bool one_class = (band_tiles ? band_tiles : tiles) <= 4200u;
    // The short class's size, by the frame's mean list length (measured, `bin_tile_sort` with 8 | 12 | 16 elements per thread:
    // C1, mean 570: 45.6 | 41.4 | 45.9 us; C3, mean 910: 193 | 180 | 188; C2, mean 1380: 77.0 | 76.6 | 72.9 — and a third
    // class in between loses its launch: profiles/r04_u_tile_sort_classes.txt).  SPLAT_TILE_SORT_SHORT=8 | 12 | 16 forces one.
    static const uint32_t force_short = [] {
        const char *e = getenv("SPLAT_TILE_SORT_SHORT");
        const int v = e ? atoi(e) : 0;
        return (v == 8 || v == 12 || v == 16) ? (uint32_t)v : 0u;
    }();
    const uint32_t short_items = force_short ? force_short : mean_list < 320u ? 8u : mean_list < 1152u ? 12u : 16u;
    const uint32_t short_cap = short_items * TS_THREADS;
    // (two passes of up to 12 bits instead of three of 8 — SPLAT_TILE_SORT_DIGITS=12 in rounds 3 and 4 — measured slower, 76 + 33
    // against 56 + 23 us at C2: profiles/r03_f_tile_sort_wide_digits_C2.txt; removed in round 5)
#define SPLAT_TILE_SORT_(RA, ITEMS, LAST, ABOVE, COUNTS)                                                                       \
    hipLaunchKernelGGL((k_tile_sort<RA, ITEMS, LAST>), dim3(tiles), dim3(TS_THREADS), 0, ctx->stream, offsets, tiles, ABOVE, vals, \
                       scratch, out_idx, COUNTS, frame_flags TS_HOOK_ARGS)
#define SPLAT_TILE_SORT(ITEMS, LAST, ABOVE, COUNTS)                  \
    do {                                                             \
        if (ra) SPLAT_TILE_SORT_(true, ITEMS, LAST, ABOVE, COUNTS);  \
        else SPLAT_TILE_SORT_(false, ITEMS, LAST, ABOVE, COUNTS);    \
    } while (0)
    bool no_long = false;
    if (one_class) {
        SPLAT_TILE_SORT(TS_LONG_ITEMS, true, 0u, counts);
    } else {
        // The frame before (same band, sync-free) had NO tile beyond the short class: the long class's launch would be 5 us of
        // workgroups that look at their tile and leave (C1: every frame).  The short class's kernel then runs as the last
        // class: a tile that has outgrown it since goes through its global-memory passes — slow, correct, and counted, so
        // that the next frame launches both classes again.
        // (the count is of the tiles beyond THAT frame's short class: it says nothing when the class has changed since)
        no_long = long_tiles_hint == 0u && short_class_io && *short_class_io == short_items;
        if (short_class_io) *short_class_io = short_items;
        if (short_items == 8u) { if (no_long) SPLAT_TILE_SORT(8, true, 0u, counts); else SPLAT_TILE_SORT(8, false, 0u, counts); }
        else if (short_items == 12u) { if (no_long) SPLAT_TILE_SORT(12, true, 0u, counts); else SPLAT_TILE_SORT(12, false, 0u, counts); }
        else { if (no_long) SPLAT_TILE_SORT(16, true, 0u, counts); else SPLAT_TILE_SORT(16, false, 0u, counts); }
        if (!no_long) SPLAT_TILE_SORT(TS_LONG_ITEMS, true, short_cap, nullptr);
    }
#undef SPLAT_TILE_SORT_
#undef SPLAT_TILE_SORT
#ifdef SPLAT_TEST_HOOKS
    ctx->tile_sort_launches = one_class || no_long ? 1u : 2u;
#else
    (void)no_long;
#endif
    LAUNCH_CHECK(ctx, "k_tile_sort");
    return SPLAT_OK;
}

// ---------------------------------------------------------------------------------------------
// Second pass of the tile-id sort (the high digit), for the layout k_tf_scatter leaves: every low digit's run starts
// on a partition boundary, so a partition holds pairs of ONE low digit and
//   * a pair carries one byte of tile id (the high digit) instead of four,
//   * the pass writes no tile ids at all: with hist[h][p] = pairs of high digit h in partition p, scanned over p by
//     k_radix_rowscan, tile (h, l) starts at  start(h) + scanned[h][first partition of run l]  — every pair with
//     high digit h in an earlier run has a smaller tile id, every one in run l or later does not.  tf_offsets_block reads
//     the tile offsets straight out of the scanned histogram; the 65-ary search over the sorted tile ids (10 us at
//     C2) and the 45 MB of sorted ids it probed are gone, and so are 3 of every 4 key bytes the pass used to move.
// The price is up to TF2_PART - 1 unused slots at the end of each run of the first pass's output (the second pass's
// output is dense again).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t TF2_ITEMS = 16, TF2_PART = TF2_ITEMS * TF_THREADS; // 4096 pairs per partition
static_assert(TF2_PART == TF_RUN_ALIGN, "k_tf_scatter rounds every run up to whole partitions of this pass");

// partition p: which low digit's run it lies in and how many of its TF2_PART slots hold pairs (0: past the end)
__device__ __forceinline__ uint32_t tf2_valid(const TfRuns &runs, uint32_t p) {
    if (p >= *runs.parts) return 0u;
    const uint32_t d = runs.part_digit[p];
    const uint32_t left = runs.start[d] + runs.total[d] - p * TF2_PART;
    return left < TF2_PART ? left : TF2_PART;
}

__global__ __launch_bounds__(TF_THREADS) void k_tf_upsweep2(const uint8_t *__restrict__ hi, TfRuns runs, uint32_t hmask,
                                                           uint32_t num_parts, uint32_t *__restrict__ hist, uint32_t xcd_per) {
    __shared__ uint32_t lh[TS_WAVES][256];
    const uint32_t tid = threadIdx.x, w = tid >> 6, p = xcd_block_of(blockIdx.x, xcd_per);
    if (p >= num_parts) return;
    const uint32_t valid = tf2_valid(runs, p);
    if (valid == 0) { // (uniform) a column of zeros: the row scans run over all num_parts columns
        if (tid <= hmask) hist[(size_t)tid * num_parts + p] = 0;
        return;
    }
    const uint8_t *src = hi + (size_t)p * TF2_PART;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (valid == TF2_PART) v = reinterpret_cast<const uint4 *>(src)[tid]; // (runs start on 4096-byte boundaries)
    for (uint32_t i = tid; i < TS_WAVES * 256; i += TF_THREADS) (&lh[0][0])[i] = 0;
    __syncthreads();
    if (valid == TF2_PART) {
        const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t a = q[j] & 255u, b = (q[j] >> 8) & 255u, c = (q[j] >> 16) & 255u, d = q[j] >> 24;
            if (a == b && b == c && c == d) { // neighbours in the run often share a tile row
                atomicAdd(&lh[w][a], 4u);
            } else {
                atomicAdd(&lh[w][a], 1u);
                atomicAdd(&lh[w][b], 1u);
                atomicAdd(&lh[w][c], 1u);
                atomicAdd(&lh[w][d], 1u);
            }
        }
    } else {
        for (uint32_t i = tid; i < valid; i += TF_THREADS) atomicAdd(&lh[w][src[i]], 1u);
    }
    __syncthreads();
    if (tid <= hmask) hist[(size_t)tid * num_parts + p] = lh[0][tid] + lh[1][tid] + lh[2][tid] + lh[3][tid];
}

template <uint32_t NW>
struct TfDownsweepShared {
    uint32_t wave_hist[NW][256];
    uint32_t global_base[256]; // (where this partition's pairs of each digit start in the output) - (their local start)
    uint32_t wave_sums[TS_WAVES], wave_gsums[TS_WAVES];
};

// offsets[t] for t = 0 .. tiles (offsets[tiles] = the pair total): tile (h, l) starts at start(h) + scanned[h][first
// partition of run l].  It needs the scanned histogram only, not the second pass's output — so it is not a launch of its own
// (a dependent launch costs ~5 us whatever it does) but the work of k_tf_downsweep2's LAST workgroups, beside the partitions.
// (A screen of at most 256 tiles has no second pass: its dense runs are the tiles, and k_tf_scatter writes the offsets.)
struct TfOffsetsArgs {
    uint32_t tiles, lo_bits;
    uint32_t *offsets;
    const uint32_t *d_total;
};
// (the first 256 threads of the workgroup: one per digit / per tile of the block)
__device__ __forceinline__ void tf_offsets_block(uint32_t block, const TfOffsetsArgs &o, const TfRuns &runs, uint32_t num_parts,
                                                 const uint32_t *__restrict__ scanned, const uint32_t *__restrict__ totals, uint32_t *hstart,
                                                 uint32_t *wsums) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid >= 256u) return; // (whole waves: the barriers below count the waves that are left)
    const uint32_t t = block * 256u + tid;
    const uint32_t total = o.d_total[2];
    // start of every high digit's run in the sorted order: exclusive scan of the second pass's digit totals
    const uint32_t mine = totals[tid];
    uint32_t incl = mine;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t v = __shfl_up(incl, s);
        if ((int)lane >= s) incl += v;
    }
    if (lane == 63) wsums[w] = incl;
    __syncthreads();
    hstart[tid] = (w > 0 ? wsums[0] : 0u) + (w > 1 ? wsums[1] : 0u) + (w > 2 ? wsums[2] : 0u) + incl - mine;
    __syncthreads();
    if (t > o.tiles) return;
    if (t == o.tiles) {
        o.offsets[t] = total;
        return;
    }
    const uint32_t h = t >> o.lo_bits, l = t & ((1u << o.lo_bits) - 1u);
    const uint32_t first_part = runs.start[l] / TF2_PART; // (an empty run starts where the next one does)
    o.offsets[t] = hstart[h] + (first_part < num_parts ? scanned[(size_t)h * num_parts + first_part] : totals[h]);
}

// NW waves per workgroup (4: sixteen pairs per thread; 8: eight)
template <bool RANK_ATOMIC, uint32_t NW>
__global__ __launch_bounds__(NW * 64, 4) void k_tf_downsweep2(const uint8_t *__restrict__ hi_in, const uint2 *__restrict__ val_in,
                                                              uint2 *__restrict__ val_out, TfRuns runs, uint32_t hmask,
                                                              uint32_t num_parts, const uint32_t *__restrict__ scanned,
                                                              const uint32_t *__restrict__ totals, TfOffsetsArgs off, uint32_t part_blocks,
                                                              uint32_t xcd_per) {
    constexpr uint32_t THREADS = NW * 64, ITEMS = TF2_PART / THREADS;
    __shared__ TfDownsweepShared<NW> sh;
    __shared__ uint2 s_val[TF2_PART];
    // the reordered pairs' digits live where the waves' counters were (every position is computed before the first digit is
    // stored): 37.9 instead of 42.3 KB, FOUR workgroups per CU instead of three
    static_assert(sizeof(sh.wave_hist) >= TF2_PART, "s_dig aliases the wave counters");
    uint8_t *const s_dig = reinterpret_cast<uint8_t *>(&sh.wave_hist[0][0]);
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (blockIdx.x >= part_blocks) { // (uniform) not a partition: a block of 256 tile offsets
        tf_offsets_block(blockIdx.x - part_blocks, off, runs, num_parts, scanned, totals, sh.global_base, sh.wave_sums);
        return;
    }
    const uint32_t p = xcd_block_of(blockIdx.x, xcd_per);
    if (p >= num_parts) return;
    const uint32_t valid = tf2_valid(runs, p);
    if (valid == 0) return;
    for (uint32_t i = tid; i < NW * 256; i += THREADS) (&sh.wave_hist[0][0])[i] = 0;
    const bool digit_thread = tid < 256u; // thread tid < 256 owns digit tid in the scans below
    const uint32_t row_prefix = (digit_thread && tid <= hmask) ? scanned[(size_t)tid * num_parts + p] : 0u; // digit tid in earlier partitions
    const uint32_t digit_total = digit_thread ? totals[tid] : 0u;                                              // ... and in the whole frame
    // wave-striped: item i of lane l of wave w is element w * ITEMS * 64 + i * 64 + l, the order the ranking preserves.
    // Slots past `valid` read the last pair and rank as digit 255 behind every real one: they end up past `valid`
    // in the reordered partition and are not written.
    const size_t base = (size_t)p * TF2_PART;
    const uint32_t wbase = w * (ITEMS * 64) + lane;
    uint32_t dig[ITEMS];
    uint2 val[ITEMS];
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) {
        const uint32_t q = wbase + i * 64;
        const uint32_t c = q < valid ? q : valid - 1;
        dig[i] = q < valid ? (uint32_t)hi_in[base + c] : 255u;
        val[i] = val_in[base + c];
    }
    // a partition dominated by one digit (a scene bunched up in a few tile rows) ranks faster with ballots: returning
    // LDS atomics that collide on one counter serialise (see radix_sort.hip)
    bool use_atomic = RANK_ATOMIC;
    if (RANK_ATOMIC) {
        const uint32_t next = (!digit_thread || tid > hmask) ? 0u : (p + 1 < num_parts) ? scanned[(size_t)tid * num_parts + p + 1] : digit_total;
        use_atomic = !__syncthreads_or((next - row_prefix) > TF2_PART / 4);
    } else {
        __syncthreads();
    }
    uint32_t rank[ITEMS];
    if (use_atomic) {
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; ++i) rank[i] = wave_rank<true>(sh.wave_hist[w], dig[i]);
    } else {
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; ++i) rank[i] = wave_rank<false>(sh.wave_hist[w], dig[i]);
    }
    __syncthreads();
    uint32_t cs[NW];
    uint32_t dcount = 0;
    if (digit_thread) {
#pragma unroll
        for (uint32_t v = 0; v < NW; ++v) {
            cs[v] = sh.wave_hist[v][tid];
            dcount += cs[v];
        }
    }
    // exclusive scans over the 256 digits of the partition's counts (local starts) and, in the same shuffles, of the
    // frame's digit totals (where each digit's run starts in the output)
    uint32_t incl = dcount, gincl = digit_total;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t t = __shfl_up(incl, s), g = __shfl_up(gincl, s);
        if ((int)lane >= s) {
            incl += t;
            gincl += g;
        }
    }
    if (digit_thread && lane == 63) {
        sh.wave_sums[w] = incl;
        sh.wave_gsums[w] = gincl;
    }
    __syncthreads();
    if (digit_thread) {
        const uint32_t wprefix = (w > 0 ? sh.wave_sums[0] : 0u) + (w > 1 ? sh.wave_sums[1] : 0u) + (w > 2 ? sh.wave_sums[2] : 0u);
        const uint32_t gprefix = (w > 0 ? sh.wave_gsums[0] : 0u) + (w > 1 ? sh.wave_gsums[1] : 0u) + (w > 2 ? sh.wave_gsums[2] : 0u);
        const uint32_t local_start = wprefix + incl - dcount;
        uint32_t run = local_start;
#pragma unroll
        for (uint32_t v = 0; v < NW; ++v) {
            sh.wave_hist[v][tid] = run;
            run += cs[v];
        }
        sh.global_base[tid] = gprefix + gincl - digit_total + row_prefix - local_start;
    }
    __syncthreads();
    // reorder inside the partition: same-digit pairs become contiguous, stable
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) rank[i] += sh.wave_hist[w][dig[i]]; // (now the pair's position in the partition)
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) {
        s_val[rank[i]] = val[i];
        s_dig[rank[i]] = (uint8_t)dig[i];
    }
    __syncthreads();
    // consecutive lanes write consecutive addresses inside each digit run
#pragma unroll
    for (uint32_t j = 0; j < ITEMS; ++j) {
        const uint32_t pos = j * THREADS + tid;
        if (pos < valid) val_out[sh.global_base[s_dig[pos]] + pos] = s_val[pos];
    }
}

// The second pass and the tile offsets.  hist: 256 * num_parts + 256 words, num_parts = tf2_parts_bound(pairs bound,
// low digits).  hi_bits == 0: only the offsets (the first pass's output is final).
int tf_second_pass_launch(splat_ctx *ctx, const uint8_t *hi, const uint2 *val_in, uint2 *val_out, const TfRuns *runs, uint32_t pairs_bound,
                          uint32_t tiles, uint32_t lo_bits, uint32_t hi_bits, uint32_t *hist, uint32_t *offsets, const uint32_t *d_total) {
    const uint32_t num_parts = tf2_parts_bound(pairs_bound, lo_bits);
    uint32_t *totals = hist + (size_t)256 * num_parts;
    if (hi_bits > 0) {
        const uint32_t hmask = (1u << hi_bits) - 1u;
        // each XCD takes a contiguous eighth of the partitions (common.h: xcd_block_of)
        const uint32_t xcd_per = num_parts >= 64u ? div_up(num_parts, 8u) : 0u;
        const uint32_t part_blocks = xcd_per ? 8u * xcd_per : num_parts;
        hipLaunchKernelGGL(k_tf_upsweep2, dim3(part_blocks), dim3(TF_THREADS), 0, ctx->stream, hi, *runs, hmask, num_parts, hist, xcd_per);
        LAUNCH_CHECK(ctx, "k_tf_upsweep2");
        int rc = radix_rowscan_launch(ctx, hist, num_parts, hmask + 1u);
        if (rc != SPLAT_OK) return rc;
        const TfOffsetsArgs off = {tiles, lo_bits, offsets, d_total};
        const dim3 grid(part_blocks + div_up(tiles + 1, 256)); // the partitions, then the blocks of tile offsets
        const bool ra = rank_atomic_ok(ctx, true); // (checked: k_tile_sort verifies every list this pass contributes to)
        // (512 threads per partition, eight pairs per thread: measured level with this, profiles/r05_b_second_pass_xcd_C3_C2.txt)
        // Workgroups per CU: four fit (37.9 KB each) and are what a pass of 7-bit digits likes (C1 22.4 -> 21.0 us, C2 level);
        // with 8-bit digits (screens beyond 2^14 tiles: C3, where a partition leaves 128-byte fragments) a fourth resident
        // workgroup LOSES — 135 -> 143 us, and two are worse still (154) —, so there 4 KB of dynamic LDS that nobody uses
        // keep it at three (profiles/r05_x_downsweep2_workgroups_per_cu.txt).
        const uint32_t pad = hi_bits >= 8u ? 4096u : 0u;
        if (ra)
            hipLaunchKernelGGL((k_tf_downsweep2<true, 4>), grid, dim3(TF_THREADS), pad, ctx->stream, hi, val_in, val_out, *runs, hmask, num_parts, hist,
                               totals, off, part_blocks, xcd_per);
        else
            hipLaunchKernelGGL((k_tf_downsweep2<false, 4>), grid, dim3(TF_THREADS), pad, ctx->stream, hi, val_in, val_out, *runs, hmask, num_parts, hist,
                               totals, off, part_blocks, xcd_per);
        LAUNCH_CHECK(ctx, "k_tf_downsweep2");
    }
    return SPLAT_OK; // (hi_bits == 0: k_tf_scatter wrote the offsets itself)
}

// hist: the digit histogram the projector (k_project_hist) or the band prepare kernel counted, after
// radix_rowscan_launch (rows scanned in place, digit totals behind them)
int tf_scatter_launch(splat_ctx *ctx, const uint32_t *range32, const uint32_t *depth_keys, uint32_t n, uint32_t ntx, uint32_t mask,
                      const uint32_t *hist, uint32_t *d_total, uint32_t pair_limit, uint32_t *overflow, uint8_t *out_hi,
                      uint2 *out_val, uint32_t block_splats, uint32_t lo_bits, bool second_pass, const TfRuns *runs,
                      uint32_t *offsets_if_final, uint32_t tiles, uint32_t *report, uint32_t seq, const uint32_t *cidx,
                      const uint32_t *kept) {
    const uint32_t parts = div_up(n, block_splats);
    // each XCD takes a contiguous eighth of the blocks (common.h: xcd_block_of; C3: this kernel 112 -> 97 us and the second
    // pass's downsweep 145 -> 135, C2 48.5 -> 47.0 / 44.1 -> 42.6: profiles/r05_c_first_pass_projector_xcd_C3_C2.txt)
    const uint32_t xcd_per = parts >= 64u ? div_up(parts, 8u) : 0u;
    const uint32_t grid_blocks = xcd_per ? 8u * xcd_per : parts;
    if (cidx && block_splats != TF_GROUP) return ctx_fail(ctx, SPLAT_ERR_INVALID, "tf_scatter_launch: compacted input comes in groups of 4096 records");
    const uint32_t *totals = hist + (size_t)256 * parts;
    {
        int prc = ctx_resolve_rank_mode(ctx); // (probe of the LDS atomics' lane order, once per context)
        if (prc != SPLAT_OK) return prc;
    }
    const bool ra = rank_atomic_ok(ctx, true); // (checked: k_tile_sort verifies every list this pass contributes to)
#define SPLAT_TF_SCATTER_(RA, PER, COMPACTED)                                                                                     \
    hipLaunchKernelGGL((k_tf_scatter<RA, PER, COMPACTED>), dim3(grid_blocks), dim3(TF_THREADS), 0, ctx->stream, range32, depth_keys, n, ntx, mask, parts, \
                       hist, totals, d_total, pair_limit, overflow, out_hi, out_val, lo_bits, second_pass ? TF_RUN_ALIGN - 1u : 0u, *runs,         \
                       second_pass ? nullptr : offsets_if_final, tiles, report, seq, cidx, kept, xcd_per)
#define SPLAT_TF_SCATTER(RA, PER) SPLAT_TF_SCATTER_(RA, PER, false)
    if (cidx) { // a multi-GPU band's kept splats, compacted per group of TF_GROUP records
        if (ra) SPLAT_TF_SCATTER_(true, 4, true);
        else SPLAT_TF_SCATTER_(false, 4, true);
    } else if (block_splats == TF_BLOCK_SMALL) {
        if (ra) SPLAT_TF_SCATTER(true, 1);
        else SPLAT_TF_SCATTER(false, 1);
    } else {
        if (ra) SPLAT_TF_SCATTER(true, 4);
        else SPLAT_TF_SCATTER(false, 4);
    }
#undef SPLAT_TF_SCATTER
#undef SPLAT_TF_SCATTER_
    LAUNCH_CHECK(ctx, "k_tf_scatter");
    return SPLAT_OK;
}
