// radix_sort.hip — RadixSorter: stable ascending LSD radix sort of (u32 key, u32 payload) pairs.
//
// Reference contract: /root/reference/src/RadixSorter.ts:39-100,197-271 (8 bits x 4 passes, result
// back in payload_a, stable).  The reference's WGSL (src/shaders/radix-sort.wgsl) emulates 32-wide
// subgroup match/rank through LDS and chains workgroups with a spinning decoupled look-back; this
// file is a wave64 design instead.  Default ("rowscan") mode, per pass:
//
//   k_radix_upsweep    per 4096-key partition a 256-bin digit histogram (per-wave LDS histograms),
//                      written digit-major: row d = counts of digit d over the partitions
//   k_radix_rowscan    256 workgroups, one per digit row: exclusive scan along the partitions in
//                      place + the row total
//   k_radix_downsweep  scans the 256 row totals itself (digit bases); each wave ranks its 16x64 keys
//                      (returning LDS atomics where the lane-order probe allows and the partition is
//                      not dominated by one digit, ballot match masks otherwise); the workgroup
//                      reorders (key, payload) by digit in LDS so that global stores go out in digit
//                      runs, then scatters.
//
// No inter-workgroup waiting: forward progress never depends on dispatch order (the guide's "give
// every wave an exit condition" rule), at the price of reading the keys twice per pass.  Between
// passes the pairs are stored interleaved (uint2) so the scatter is one 8-byte access per element.
// (A onesweep mode — one chained-scan kernel per pass, the reference's structure — was built in round 1 and measured slower
// on MI355X: 5M keys x 4 passes 227 against 176 us; removed in round 5, profiles/EXPERIMENTS.md §4.)
//
// Roofline: HBM.  Algorithmic bytes per key per pass: 4 (upsweep read) + 8 (read key+payload)
// + 8 (write) = 20; 80 B/key for the 4-pass depth sort.
#include "common.h"

#include <cstdlib>

constexpr uint32_t RS_THREADS = 256;
constexpr uint32_t RS_ITEMS = RADIX_PART / RS_THREADS; // 16 keys per thread
constexpr uint32_t RS_WAVES = RS_THREADS / 64;

// ---------------------------------------------------------------------------------------------
// Partition = 16 keys/thread x 256 threads = 4096 keys.  Measured on MI355X (tools/sort_bench.py, 5M
// keys x 4 passes): 4 items 346 us, 6 -> 266, 8 -> 228, 12 -> 193, 14 -> 184, 16 -> 176, 20 -> 174,
// 24 -> 190, 32 -> 177.  Bigger partitions win until ~16 (longer digit runs = better store coalescing,
// fixed per-partition work amortised), then the occupancy loss cancels the gain.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t RS_PART_KEYS = RS_ITEMS * RS_THREADS;

// Number of keys to sort: a host value, or (sync-free callers) a device word clamped to the host
// bound the grid was sized for.  Workgroups past the end exit at once.
__device__ __forceinline__ uint32_t sort_count(uint32_t n_host, const uint32_t *n_dev) {
    if (!n_dev) return n_host;
    const uint32_t v = *n_dev;
    return v < n_host ? v : n_host;
}

// ---------------------------------------------------------------------------------------------
// upsweep: hist[d * num_parts + part] = number of keys of this partition whose digit is d
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void hist_add4(uint32_t *h, uint4 v, uint32_t shift, uint32_t mask) {
    const uint32_t a = (v.x >> shift) & mask, b = (v.y >> shift) & mask, c = (v.z >> shift) & mask, d = (v.w >> shift) & mask;
    // depth keys have a near-constant top byte: 64 lanes x 4 keys on one LDS counter serialise, so
    // four equal digits become one atomic
    if (a == b && b == c && c == d) {
        atomicAdd(&h[a], 4u);
    } else {
        atomicAdd(&h[a], 1u);
        atomicAdd(&h[b], 1u);
        atomicAdd(&h[c], 1u);
        atomicAdd(&h[d], 1u);
    }
}

// IN_PAIRS: the input is the interleaved (key, payload) uint2 array the previous pass wrote (see
// radix_sort_rowscan); otherwise a plain key array.
template <bool IN_PAIRS>
__global__ __launch_bounds__(RS_THREADS) void k_radix_upsweep(const uint32_t *__restrict__ keys, uint32_t n_host,
                                                              const uint32_t *__restrict__ n_dev, uint32_t shift,
                                                              uint32_t mask, uint32_t num_parts, uint32_t part_keys,
                                                              uint32_t *__restrict__ hist, uint32_t xcd_per) {
    __shared__ uint32_t lh[RS_WAVES][256]; // one private histogram per wave: fewer same-bank collisions
    const uint32_t tid = threadIdx.x, w = tid >> 6;
    const uint32_t part = xcd_block_of(blockIdx.x, xcd_per); // (each XCD takes a contiguous eighth of the partitions: common.h)
    if (part >= num_parts) return;
    const uint32_t n = sort_count(n_host, n_dev);
    const uint32_t base = part * part_keys; // multiple of 256 keys = 1 KiB: uint4 loads stay aligned
    if (base >= n) { // partition past the end (device-side n): an all-zero column
        if (tid <= mask) hist[(size_t)tid * num_parts + part] = 0;
        return;
    }
    for (uint32_t i = tid; i < RS_WAVES * 256; i += RS_THREADS) (&lh[0][0])[i] = 0;
    if (base + part_keys <= n && part_keys == RS_ITEMS * RS_THREADS) { // full default partition: all loads in flight at once
        if (IN_PAIRS) {
            const uint4 *k4 = reinterpret_cast<const uint4 *>(keys + (size_t)base * 2); // two (key, payload) pairs per uint4
            uint4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = k4[tid + j * RS_THREADS];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t a = (v[j].x >> shift) & mask, c = (v[j].z >> shift) & mask;
                if (a == c) {
                    atomicAdd(&lh[w][a], 2u);
                } else {
                    atomicAdd(&lh[w][a], 1u);
                    atomicAdd(&lh[w][c], 1u);
                }
            }
        } else {
            const uint4 *k4 = reinterpret_cast<const uint4 *>(keys + base);
            const uint4 v0 = k4[tid], v1 = k4[tid + RS_THREADS], v2 = k4[tid + 2 * RS_THREADS], v3 = k4[tid + 3 * RS_THREADS];
            __syncthreads();
            hist_add4(lh[w], v0, shift, mask);
            hist_add4(lh[w], v1, shift, mask);
            hist_add4(lh[w], v2, shift, mask);
            hist_add4(lh[w], v3, shift, mask);
        }
    } else {
        __syncthreads();
        const uint32_t end = (base + part_keys < n) ? base + part_keys : n;
        for (uint32_t i = base + tid; i < end; i += RS_THREADS)
            atomicAdd(&lh[w][(keys[IN_PAIRS ? (size_t)i * 2 : (size_t)i] >> shift) & mask], 1u);
    }
    __syncthreads();
    // (rows above the mask are never read: a 4-byte store per row is a 64-byte line at the memory side)
    if (tid <= mask) hist[(size_t)tid * num_parts + part] = lh[0][tid] + lh[1][tid] + lh[2][tid] + lh[3][tid];
}

// ---------------------------------------------------------------------------------------------
// rowscan: block d turns row d of hist into its exclusive prefix over partitions; totals[d] = row sum
// ---------------------------------------------------------------------------------------------
constexpr uint32_t ROWSCAN_THREADS = 1024;
// (rows: digits that can occur this pass, mask + 1; the rows above hold zeros already and only need a zero total)
// one round of the row scan: ROWSCAN_THREADS values (one per thread) -> their exclusive prefixes (+ carry), and the carry
__device__ __forceinline__ uint32_t rowscan_round(uint32_t v, uint32_t *wsum, unsigned long long &carry, uint32_t lane, uint32_t w) {
    uint32_t incl = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        uint32_t t = __shfl_up(incl, s);
        if ((int)lane >= s) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    // every wave scans the 16 wave totals itself (lanes 0..15)
    uint32_t ws = (lane < ROWSCAN_THREADS / 64) ? wsum[lane] : 0u;
    uint32_t wincl = ws;
#pragma unroll
    for (int s = 1; s < 16; s <<= 1) {
        uint32_t t = __shfl_up(wincl, s);
        if ((int)lane >= s) wincl += t;
    }
    const uint32_t wprefix = __shfl(wincl - ws, w);
    const uint32_t total = __shfl(wincl, ROWSCAN_THREADS / 64 - 1);
    const uint32_t excl = (uint32_t)carry + wprefix + incl - v;
    carry += total;
    return excl;
}

constexpr uint32_t ROWSCAN_AHEAD = 8; // rounds whose loads are all in flight before the first scan
__global__ __launch_bounds__(ROWSCAN_THREADS) void k_radix_rowscan(uint32_t *__restrict__ hist, uint32_t num_parts,
                                                                   uint32_t *__restrict__ totals, uint32_t rows) {
    __shared__ uint32_t wsum[2][ROWSCAN_THREADS / 64]; // (alternating: a round's totals are read while the next round's are written)
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (blockIdx.x >= rows) {
        if (tid == 0) totals[blockIdx.x] = 0;
        return;
    }
    uint32_t *row = hist + (size_t)blockIdx.x * num_parts;
    unsigned long long carry = 0; // (64 bits: a row total past 2^32 must not come back small — it is saturated below)
    // The rounds depend on each other through the carry only, not through their loads: up to ROWSCAN_AHEAD rounds' values are
    // loaded before the first is scanned (a frame's rows are 3 to 5 rounds long: one memory round trip instead of one per
    // round: 6.1 -> 5.6 us at C2, twice per frame; the rest is the launch and one 16-wave workgroup's barriers).
    for (uint32_t base = 0; base < num_parts; base += ROWSCAN_AHEAD * ROWSCAN_THREADS) {
        uint32_t v[ROWSCAN_AHEAD];
#pragma unroll
        for (uint32_t r = 0; r < ROWSCAN_AHEAD; ++r) {
            const uint32_t i = base + r * ROWSCAN_THREADS + tid;
            v[r] = (i < num_parts) ? row[i] : 0u;
        }
#pragma unroll
        for (uint32_t r = 0; r < ROWSCAN_AHEAD; ++r) {
            const uint32_t i = base + r * ROWSCAN_THREADS + tid;
            if (base + r * ROWSCAN_THREADS >= num_parts) break; // (uniform)
            const uint32_t excl = rowscan_round(v[r], wsum[r & 1u], carry, lane, w);
            if (i < num_parts) row[i] = excl;
        }
    }
    if (tid == 0) totals[blockIdx.x] = carry > 0x40000000ull ? 0x40000000u : (uint32_t)carry; // (every caller's limit is below 2^30)
}

// ---------------------------------------------------------------------------------------------
// downsweep
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t lanemask_lt() {
    const uint32_t lane = threadIdx.x & 63;
    return (lane == 0) ? 0ull : (~0ull >> (64 - lane));
}

struct DownsweepShared {
    uint32_t wave_hist[RS_WAVES][256]; // running per-wave digit counters -> wave offsets
    uint32_t global_base[256];         // (global start of this partition's run of each digit) - (its partition-local start)
    uint32_t wave_sums[RS_WAVES];
    uint32_t wave_gsums[RS_WAVES];
};

template <uint32_t ITEMS, bool FULL, bool RANK_ATOMIC, bool IN_PAIRS = false, bool OUT_PAIRS = false>
__device__ __forceinline__ void downsweep_body(DownsweepShared &sh, uint2 *__restrict__ s_kp, uint32_t part,
                                               const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ pay_in,
                                               uint32_t *__restrict__ keys_out, uint32_t *__restrict__ pay_out, uint32_t n,
                                               uint32_t shift, uint32_t mask, uint32_t num_parts,
                                               const uint32_t *__restrict__ scanned_hist,
                                               const uint32_t *__restrict__ totals) {
    constexpr uint32_t PART_KEYS = ITEMS * RS_THREADS, WAVE_KEYS = ITEMS * 64;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t base = part * PART_KEYS;
    const uint32_t valid = FULL ? PART_KEYS : (n - base);

    for (uint32_t i = tid; i < RS_WAVES * 256; i += RS_THREADS) (&sh.wave_hist[0][0])[i] = 0;
    // digit tid in earlier partitions (digits above the mask do not occur: their rows are neither written nor read)
    const uint32_t row_prefix = tid > mask ? 0u : scanned_hist[(size_t)tid * num_parts + part];
    const uint32_t digit_total = totals[tid]; // global count of digit tid (this pass)

    // striped load: item i of lane l of wave w is element w*WAVE_KEYS + i*64 + l (position order =
    // (wave, item, lane), which is the order the ranking below preserves).  Padding lanes of the
    // last partition read the last real element and are given digit 255 below: they sit after
    // every real key in position order, so they also rank after every real key of digit 255.
    uint32_t key[ITEMS];
    uint32_t pay[ITEMS];
    const uint32_t wbase = w * WAVE_KEYS + lane;
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) {
        const uint32_t p = wbase + i * 64;
        const uint32_t q = FULL ? p : ((p < valid) ? p : (valid - 1));
        if (IN_PAIRS) { // one 8-byte load per element instead of two 4-byte ones
            const uint2 kp = reinterpret_cast<const uint2 *>(keys_in)[base + q];
            key[i] = kp.x;
            pay[i] = kp.y;
        } else {
            key[i] = keys_in[base + q];
            pay[i] = pay_in ? pay_in[base + q] : (base + q); // no payload array: the payload is the element's index
        }
    }
    // (the barrier that makes the zeroed wave_hist visible is folded into the vote below)

    // ---- rank, phase A: per item, the mask of lanes of this wave holding the same digit: 8 ballots,
    // each folded in with one v_bitop3 per mask half (peers &= ~(ballot ^ mybit)).  The lowest lane
    // of every digit group then adds the group's size to the wave's digit counter with a RETURNING
    // LDS atomic.  Within one instruction the leaders have distinct digits (no collisions), and the
    // returning atomics of a wave execute in program order, so the value returned for item i is
    // the number of keys of that digit in items < i: a stable rank, with the LDS round trips
    // pipelined (nothing waits on them until phase B) instead of chained.
    uint32_t rank[ITEMS];
    // Returning LDS atomics that collide on one address serialise (64 lanes on one counter = 64 LDS
    // cycles), so a partition dominated by one digit — the top byte of depth keys — ranks faster
    // with ballots, whose cost does not depend on the digit distribution.  The partition's digit counts
    // are known before it starts (upsweep): pick per workgroup.
    bool use_atomic = RANK_ATOMIC;
    if (RANK_ATOMIC) {
        const uint32_t next = tid > mask ? 0u : (part + 1 < num_parts) ? scanned_hist[(size_t)tid * num_parts + part + 1] : digit_total;
        use_atomic = !__syncthreads_or((next - row_prefix) > PART_KEYS / 4); // some digit holds > 25 % of the partition
    } else {
        __syncthreads(); // wave_hist zeroed
    }
    if (use_atomic) {
        // Measured property of gfx950 (splat_probe_lds_atomic_order, run once per context; the ballot
        // path below is used if it ever fails): the lanes of one returning LDS atomic that hit the same
        // address complete in ascending lane order.  Then old = atomicAdd(&counter[digit], 1) IS the
        // stable rank: same-digit keys of earlier items (program order) + same-digit lanes below me.
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; ++i) {
            uint32_t d = (key[i] >> shift) & mask;
            if (!FULL) d = (wbase + i * 64 < valid) ? d : 255u;
            rank[i] = atomicAdd(&sh.wave_hist[w][d], 1u);
        }
    } else {
        uint32_t prev[ITEMS];
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; ++i) {
            uint32_t d = (key[i] >> shift) & mask;
            if (!FULL) d = (wbase + i * 64 < valid) ? d : 255u;
            uint32_t plo = ~0u, phi = ~0u;
#pragma unroll
            for (uint32_t b = 0; b < 8; ++b) {
                const uint32_t m = (uint32_t)(((int32_t)(d << (31 - b))) >> 31); // all ones if bit b of d is set
                const uint64_t bal = __ballot(m != 0);
                plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)bal, m, 0x90);        // plo & ~(bal ^ m)
                phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(bal >> 32), m, 0x90);
            }
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0)); // same-digit lanes below me
            const uint32_t leader = plo ? (uint32_t)__builtin_ctz(plo) : 32u + (uint32_t)__builtin_ctz(phi);
            rank[i] = below | (leader << 8);
            prev[i] = 0;
            if (below == 0) prev[i] = atomicAdd(&sh.wave_hist[w][d], (uint32_t)(__popc(plo) + __popc(phi)));
        }
        // ---- phase B: every lane fetches its leader's result ---------------------------------------
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; ++i) rank[i] = __shfl(prev[i], rank[i] >> 8) + (rank[i] & 0xffu);
    }
    __syncthreads();

    // thread d: exclusive prefix over waves for digit d, and the partition's count of d
    uint32_t c0 = sh.wave_hist[0][tid], c1 = sh.wave_hist[1][tid], c2 = sh.wave_hist[2][tid], c3 = sh.wave_hist[3][tid];
    uint32_t dcount = c0 + c1 + c2 + c3;
    // exclusive scan of dcount over the 256 digits and, in the same shuffles, of the global digit
    // totals (start of digit d in the output)
    uint32_t incl = dcount, gincl = digit_total;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        uint32_t t = __shfl_up(incl, s), g = __shfl_up(gincl, s);
        if ((int)lane >= s) {
            incl += t;
            gincl += g;
        }
    }
    if (lane == 63) {
        sh.wave_sums[w] = incl;
        sh.wave_gsums[w] = gincl;
    }
    __syncthreads();
    uint32_t wprefix = (w > 0 ? sh.wave_sums[0] : 0u) + (w > 1 ? sh.wave_sums[1] : 0u) + (w > 2 ? sh.wave_sums[2] : 0u);
    uint32_t gprefix = (w > 0 ? sh.wave_gsums[0] : 0u) + (w > 1 ? sh.wave_gsums[1] : 0u) + (w > 2 ? sh.wave_gsums[2] : 0u);
    // one table lookup per element in each of the two loops below instead of two: the wave prefixes
    // absorb the digit's local start, and global_base holds (global start - local start)
    const uint32_t local_start = wprefix + incl - dcount;
    sh.wave_hist[0][tid] = local_start;
    sh.wave_hist[1][tid] = local_start + c0;
    sh.wave_hist[2][tid] = local_start + c0 + c1;
    sh.wave_hist[3][tid] = local_start + c0 + c1 + c2;
    sh.global_base[tid] = gprefix + gincl - digit_total + row_prefix - local_start;
    __syncthreads();

    // reorder inside the partition: same-digit keys become contiguous, stable
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) {
        uint32_t d = (key[i] >> shift) & mask;
        if (!FULL) d = (wbase + i * 64 < valid) ? d : 255u;
        const uint32_t pos = sh.wave_hist[w][d] + rank[i];
        s_kp[pos] = make_uint2(key[i], pay[i]);
    }
    __syncthreads();

    // scatter: thread t handles local positions t, t+256, ...: consecutive lanes write
    // consecutive global addresses inside each digit run
#pragma unroll
    for (uint32_t j = 0; j < ITEMS; ++j) {
        const uint32_t pos = j * RS_THREADS + tid;
        if (FULL || pos < valid) {
            const uint2 kp = s_kp[pos];
            const uint32_t d = (kp.x >> shift) & mask;
            const uint32_t g = sh.global_base[d] + pos;
            if (OUT_PAIRS) {
                reinterpret_cast<uint2 *>(keys_out)[g] = kp; // one 8-byte store, 128-byte digit runs
            } else {
                keys_out[g] = kp.x;
                pay_out[g] = kp.y;
            }
        }
    }
}

// resident workgroups per CU for a given ITEMS: LDS-bound (160 KiB per CU), capped at 8 (32 waves)
constexpr uint32_t downsweep_lds_bytes(uint32_t items) { return items * RS_THREADS * 8 + (uint32_t)sizeof(DownsweepShared); }
// ... and register-bound (~6.5 VGPRs per item + ~20): 4 workgroups per CU for 16 items
constexpr uint32_t downsweep_wg_per_cu(uint32_t items) { return items <= 6 ? 8 : items <= 12 ? 5 : items <= 16 ? 4 : 2; }
static_assert(downsweep_wg_per_cu(RS_ITEMS) * downsweep_lds_bytes(RS_ITEMS) <= 160u * 1024u, "LDS budget");

template <uint32_t ITEMS, bool RANK_ATOMIC, bool IN_PAIRS, bool OUT_PAIRS>
__global__ __launch_bounds__(RS_THREADS, downsweep_wg_per_cu(ITEMS)) void k_radix_downsweep(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ pay_in, uint32_t *__restrict__ keys_out,
    uint32_t *__restrict__ pay_out, uint32_t n_host, const uint32_t *__restrict__ n_dev, uint32_t shift, uint32_t mask,
    uint32_t num_parts, const uint32_t *__restrict__ scanned_hist, const uint32_t *__restrict__ totals, uint32_t xcd_per) {
    __shared__ DownsweepShared sh;
    __shared__ uint2 s_kp[ITEMS * RS_THREADS]; // (key, payload) reordered by digit
    const uint32_t part = xcd_block_of(blockIdx.x, xcd_per);
    if (part >= num_parts) return;
    const uint32_t n = sort_count(n_host, n_dev);
    if (part * ITEMS * RS_THREADS >= n) return; // partition past the end (device-side n)
    // every partition but (possibly) the last is full: it takes the path with no per-key bounds checks
    if ((part + 1) * ITEMS * RS_THREADS <= n)
        downsweep_body<ITEMS, true, RANK_ATOMIC, IN_PAIRS, OUT_PAIRS>(sh, s_kp, part, keys_in, pay_in, keys_out, pay_out, n, shift, mask, num_parts,
                                           scanned_hist, totals);
    else
        downsweep_body<ITEMS, false, RANK_ATOMIC, IN_PAIRS, OUT_PAIRS>(sh, s_kp, part, keys_in, pay_in, keys_out, pay_out, n, shift, mask, num_parts,
                                            scanned_hist, totals);
}

static int radix_sort_rowscan(splat_ctx *ctx, uint32_t *k0, uint32_t *p0, uint32_t *k1, uint32_t *p1, uint32_t *hist, uint32_t n,
                              const uint32_t *n_dev, uint32_t bit_begin, uint32_t bit_end, bool *result_in_primary,
                              bool iota_payload, bool force_ballot_rank, uint32_t cap_stride, uint32_t first_bits) {
    // ranking by returning LDS atomics needs the lane-order property: probed once per context
    {
        int prc = ctx_resolve_rank_mode(ctx); // (probe of the LDS atomics' lane order, once per context)
        if (prc != SPLAT_OK) return prc;
    }
    const bool rank_atomic = rank_atomic_ok(ctx, false) && !force_ballot_rank; // (nothing checks a global sort's result)
    const uint32_t parts = div_up(n, RS_PART_KEYS);
    // Between passes the pairs travel INTERLEAVED: (key, payload) as one uint2 per element, laid over
    // the destination's key+payload storage (the two arrays of a ping-pong side are one allocation:
    // sorter_reserve / the binner's pair buffers).  The scatter is bound by the number of memory
    // instructions and segments, not bytes: an 8-byte access per element halves both.  The first pass
    // reads the caller's separate arrays, the last pass writes separate arrays again.
    const bool can_pair = (p0 == k0 + cap_stride) && (p1 == k1 + cap_stride);
    uint32_t *ki = k0, *pi = p0, *ko = k1, *po = p1;
    bool primary = true;
    // digit widths: first_bits for the first pass (the binner splits its tile-id bits evenly), then 8
    uint32_t npasses = 0;
    for (uint32_t sh = bit_begin, w = first_bits; sh < bit_end; sh += w, w = 8) ++npasses;
    uint32_t shift = bit_begin;
    const uint32_t xcd_per = parts >= 64u ? div_up(parts, 8u) : 0u;
    const uint32_t grid_parts = xcd_per ? 8u * xcd_per : parts;
    for (uint32_t pass = 0; pass < npasses; ++pass) {
        const uint32_t want = pass == 0 ? first_bits : 8u;
        const uint32_t bits = bit_end - shift < want ? bit_end - shift : want;
        const uint32_t mask = (1u << bits) - 1u;
        const bool in_pairs = can_pair && pass > 0, out_pairs = can_pair && pass + 1 < npasses;
        if (in_pairs)
            hipLaunchKernelGGL(k_radix_upsweep<true>, dim3(grid_parts), dim3(RS_THREADS), 0, ctx->stream, ki, n, n_dev, shift, mask, parts,
                               RS_PART_KEYS, hist, xcd_per);
        else
            hipLaunchKernelGGL(k_radix_upsweep<false>, dim3(grid_parts), dim3(RS_THREADS), 0, ctx->stream, ki, n, n_dev, shift, mask, parts,
                               RS_PART_KEYS, hist, xcd_per);
        LAUNCH_CHECK(ctx, "k_radix_upsweep");
        uint32_t *totals = hist + (size_t)256 * parts;
        hipLaunchKernelGGL(k_radix_rowscan, dim3(256), dim3(ROWSCAN_THREADS), 0, ctx->stream, hist, parts, totals, mask + 1);
        LAUNCH_CHECK(ctx, "k_radix_rowscan");
        // iota_payload: the input payload is 0,1,2,... (fresh from the projector): the first pass
        // synthesises it instead of reading 4 B per key that the projector would have had to write
        const uint32_t *pin = (iota_payload && pass == 0) ? nullptr : pi;
#define SPLAT_DS(RA, IP, OP)                                                                                                   \
    hipLaunchKernelGGL((k_radix_downsweep<RS_ITEMS, RA, IP, OP>), dim3(grid_parts), dim3(RS_THREADS), 0, ctx->stream, ki, pin, ko, po, n, \
                       n_dev, shift, mask, parts, hist, totals, xcd_per)
        if (rank_atomic) {
            if (in_pairs && out_pairs) SPLAT_DS(true, true, true);
            else if (in_pairs) SPLAT_DS(true, true, false);
            else if (out_pairs) SPLAT_DS(true, false, true);
            else SPLAT_DS(true, false, false);
        } else {
            if (in_pairs && out_pairs) SPLAT_DS(false, true, true);
            else if (in_pairs) SPLAT_DS(false, true, false);
            else if (out_pairs) SPLAT_DS(false, false, true);
            else SPLAT_DS(false, false, false);
        }
#undef SPLAT_DS
        LAUNCH_CHECK(ctx, "k_radix_downsweep");
        uint32_t *t = ki; ki = ko; ko = t;
        t = pi; pi = po; po = t;
        primary = !primary;
        shift += bits;
    }
    *result_in_primary = primary;
    return SPLAT_OK;
}

int radix_rowscan_launch(splat_ctx *ctx, uint32_t *hist, uint32_t parts, uint32_t rows) {
    hipLaunchKernelGGL(k_radix_rowscan, dim3(256), dim3(ROWSCAN_THREADS), 0, ctx->stream, hist, parts, hist + (size_t)256 * parts, rows);
    LAUNCH_CHECK(ctx, "k_radix_rowscan");
    return SPLAT_OK;
}

// ---------------------------------------------------------------------------------------------
// Probe: do returning LDS atomics of ONE wave instruction that hit the same address complete in
// ascending lane order?  (Then `old = atomicAdd(&counter[digit], 1)` is a stable rank by itself.)
// The ISA does not promise it, so it is measured: random and adversarial address patterns, four
// waves per workgroup on private counters, every returned value compared with the count of lower
// lanes holding the same address.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_probe_lds_atomic_order(uint32_t rounds, uint32_t seed, unsigned long long *mismatches) {
    __shared__ uint32_t cnt[4][256];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint32_t state = seed ^ (blockIdx.x * 2654435761u) ^ (tid * 40503u + 1u);
    unsigned long long bad = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        for (uint32_t i = lane; i < 256; i += 64) cnt[w][i] = 0;
        __builtin_amdgcn_wave_barrier();
        // number of distinct addresses this round: 1, 2, 3, 4, 8, 16, 64, 256 (wave-uniform choice)
        const uint32_t kinds[8] = {1, 2, 3, 4, 8, 16, 64, 256};
        const uint32_t k = kinds[(r + blockIdx.x) & 7];
#pragma unroll
        for (int it = 0; it < 4; ++it) { // four dependent-free instructions back to back, like the sort's items
            state = state * 1664525u + 1013904223u;
            uint32_t d = (state >> 10) % k;
            if (((r >> 3) & 3) == 1) d = (lane / (64 / (k > 64 ? 64 : k))) % k; // runs of equal addresses
            if (((r >> 3) & 3) == 2) d = (lane * 7u) % k;                          // strided
            d = (d * 37u) & 255u;                                                  // spread over banks
            // reference: the set of lanes with the same address, by ballots
            uint32_t plo = ~0u, phi = ~0u;
#pragma unroll
            for (uint32_t b = 0; b < 8; ++b) {
                const uint32_t m = (uint32_t)(((int32_t)(d << (31 - b))) >> 31);
                const uint64_t bal = __ballot(m != 0);
                plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)bal, m, 0x90);
                phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(bal >> 32), m, 0x90);
            }
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0));
            const uint32_t got = atomicAdd(&cnt[w][d], 1u);
            // in lane order every lane of a same-address group reads (count before the instruction)
            // + (same-address lanes below it): got - below must equal the group leader's value
            const uint32_t base_cnt = got - below;
            const uint32_t leader = plo ? (uint32_t)__builtin_ctz(plo) : 32u + (uint32_t)__builtin_ctz(phi);
            if (base_cnt != (uint32_t)__shfl((int)base_cnt, (int)leader)) ++bad;
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) bad += __shfl_xor(bad, d);
    if (lane == 0 && bad) atomicAdd(mismatches, bad);
}

#ifdef SPLAT_TEST_HOOKS
// EXPERIMENT HOOK (tools/lds_atomic_rate.py): what a CU's LDS delivers in RETURNING atomics — the per-tile sort's ranking
// instruction — against plain reads.  WGS workgroups of four waves per CU (the per-tile sort holds three to six), every wave on its
// own 256-counter table, `iters` rounds of four independent instructions at pseudo-random counters (the sort's digits); KIND 0 =
// ds_add_rtn_u32, 1 = ds_read_b32, 2 = non-returning ds_add_u32.
template <int KIND>
__global__ __launch_bounds__(256) void k_probe_lds_rate(uint32_t iters, uint32_t seed, uint32_t *sink) {
    __shared__ uint32_t cnt[4][256];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (uint32_t i = lane; i < 256; i += 64) cnt[w][i] = 0;
    __syncthreads();
    uint32_t state = seed ^ (blockIdx.x * 2654435761u) ^ (tid * 40503u + 1u), acc = 0;
    for (uint32_t r = 0; r < iters; ++r) {
        uint32_t d[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            state = state * 1664525u + 1013904223u;
            d[it] = (state >> 12) & 255u;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if (KIND == 0) acc += atomicAdd(&cnt[w][d[it]], 1u);
            else if (KIND == 1) acc += cnt[w][d[it]];
            else __hip_atomic_fetch_add(&cnt[w][d[it]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), acc += d[it];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc; // (keeps the results alive)
}

int radix_probe_lds_rate(splat_ctx *ctx, int kind, uint32_t wgs_per_cu, uint32_t iters, float *ms_out) {
    int rc = ctx_ensure_scan_ws(ctx, 256);
    if (rc != SPLAT_OK) return rc;
    hipEvent_t a, b;
    HIP_TRY(ctx, hipEventCreate(&a));
    HIP_TRY(ctx, hipEventCreate(&b));
    const dim3 grid(256u * wgs_per_cu);
    for (int rep = 0; rep < 2; ++rep) { // (the second launch is the timed one)
        HIP_TRY(ctx, hipEventRecord(a, ctx->stream));
        if (kind == 0) hipLaunchKernelGGL(k_probe_lds_rate<0>, grid, dim3(256), 0, ctx->stream, iters, 777u, (uint32_t *)ctx->scan_ws);
        else if (kind == 1) hipLaunchKernelGGL(k_probe_lds_rate<1>, grid, dim3(256), 0, ctx->stream, iters, 777u, (uint32_t *)ctx->scan_ws);
        else hipLaunchKernelGGL(k_probe_lds_rate<2>, grid, dim3(256), 0, ctx->stream, iters, 777u, (uint32_t *)ctx->scan_ws);
        HIP_TRY(ctx, hipEventRecord(b, ctx->stream));
    }
    HIP_TRY(ctx, hipEventSynchronize(b));
    HIP_TRY(ctx, hipEventElapsedTime(ms_out, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return SPLAT_OK;
}
#endif

int radix_probe_lds_atomic_order(splat_ctx *ctx, uint64_t *mismatches_host) {
    int rc = ctx_ensure_scan_ws(ctx, 256);
    if (rc != SPLAT_OK) return rc;
    unsigned long long *d = (unsigned long long *)ctx->scan_ws;
    HIP_TRY(ctx, hipMemsetAsync(d, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_probe_lds_atomic_order, dim3(2048), dim3(256), 0, ctx->stream, 256u, 12345u, d);
    LAUNCH_CHECK(ctx, "k_probe_lds_atomic_order");
    unsigned long long v = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&v, d, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *mismatches_host = v;
    return SPLAT_OK;
}

// Which ranking the sort kernels of this context use.  Returning LDS atomics give a stable rank only if the lanes of
// one instruction that collide on an address complete in ascending lane order — observed on gfx950, not an ISA
// promise.  So (the guaranteed ranking measured 12-19 % of a frame, profiles/r03_a_rank_ab.txt):
//   * default: atomics ONLY in the tile-first frame path, where k_tile_sort checks every tile's final list for strict
//     (depth key, index) order — a complete check of every pass before it — and a frame that fails is reported, the
//     context falls back to ballots for good and the frame is rendered again (binner_settle); every other sort (the staged
//     RadixSorter / GPUTileBinner, the sort-first frame order) ranks with ballots, which assume nothing;
//   * SPLAT_RANK=atomic: atomics everywhere the start-up probe (radix_probe_lds_atomic_order) allows;
//   * SPLAT_RANK=ballot: ballots everywhere.
int ctx_resolve_rank_mode(splat_ctx *ctx) {
    if (ctx->lds_atomic_ordered >= 0) return SPLAT_OK;
    const char *e = getenv("SPLAT_RANK");
    if (e && (e[0] == 'b' || e[0] == 'B')) {
        ctx->rank_policy = RANK_BALLOT;
        ctx->lds_atomic_ordered = 0;
        return SPLAT_OK;
    }
    ctx->rank_policy = (e && (e[0] == 'a' || e[0] == 'A')) ? RANK_ATOMIC : RANK_CHECKED;
    uint64_t bad = 1;
    int prc = radix_probe_lds_atomic_order(ctx, &bad);
    if (prc != SPLAT_OK) return prc;
    ctx->lds_atomic_ordered = (bad == 0) ? 1 : 0;
    return SPLAT_OK;
}

int radix_sort_pairs(splat_ctx *ctx, uint32_t *k0, uint32_t *p0, uint32_t *k1, uint32_t *p1, uint32_t *hist, uint32_t n,
                     uint32_t bit_begin, uint32_t bit_end, bool *result_in_primary, int mode, const uint32_t *n_dev,
                     bool iota_payload, uint32_t first_bits) {
    *result_in_primary = true;
    if (n == 0 || bit_end <= bit_begin) return SPLAT_OK;
    if (n >= (1u << 30)) return ctx_fail(ctx, SPLAT_ERR_INVALID, "radix sort: n must be below 2^30");
    if (first_bits < 1 || first_bits > 8) return ctx_fail(ctx, SPLAT_ERR_INVALID, "radix sort: first_bits must be 1..8");
    // mode: -1 / 0 = rank as the context's policy says; 2 = always with ballots
    return radix_sort_rowscan(ctx, k0, p0, k1, p1, hist, n, n_dev, bit_begin, bit_end, result_in_primary, iota_payload, mode == 2,
                              (uint32_t)(p0 - k0), first_bits);
}

// ---------------------------------------------------------------------------------------------
// RadixSorter object
// ---------------------------------------------------------------------------------------------
static void sorter_free(splat_sorter *s) {
    if (s->keys) (void)hipFree(s->keys);
    if (s->keys_b) (void)hipFree(s->keys_b);
    if (s->hist) (void)hipFree(s->hist);
    if (s->d_count) (void)hipFree(s->d_count);
    s->keys = s->keys_b = s->payload = s->payload_b = s->hist = s->d_count = nullptr;
    s->capacity = 0;
}

int sorter_reserve(splat_sorter *s, uint32_t capacity) {
    splat_ctx *ctx = s->ctx;
    uint64_t padded = div_up64(capacity ? capacity : 1, SPLAT_SORT_BLOCK) * SPLAT_SORT_BLOCK; // RadixSorter.ts:46-52
    if (padded > 0xffffff00ull) return ctx_fail(ctx, SPLAT_ERR_INVALID, "sorter capacity too large");
    if (padded <= s->capacity) return SPLAT_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    sorter_free(s);
    size_t bytes = (size_t)padded * 4;
    // 256 rows of one count per partition + the 256 row totals
    const size_t hist_bytes = ((size_t)256 * div_up((uint32_t)padded, RS_PART_KEYS) + 256) * 4;
    // keys|payload of each ping-pong side are ONE allocation (payload = keys + padded): between passes
    // the sort stores interleaved (key, payload) pairs over the whole of it
    if (hipMalloc((void **)&s->keys, 2 * bytes) != hipSuccess || hipMalloc((void **)&s->keys_b, 2 * bytes) != hipSuccess ||
        hipMalloc((void **)&s->hist, hist_bytes) != hipSuccess || hipMalloc((void **)&s->d_count, 16) != hipSuccess) {
        sorter_free(s);
        return ctx_fail(ctx, SPLAT_ERR_OOM, "sorter hipMalloc");
    }
    s->payload = s->keys + padded;
    s->payload_b = s->keys_b + padded;
    // ON THE CONTEXT'S STREAM: that stream is non-blocking, so a fill on the null stream (plain hipMemset returns before the fill
    // has run) is not ordered before the kernels launched next — and the very next ones write this workspace.  The first
    // sort of a fresh sorter could find parts of its scanned histogram zeroed afterwards: a binner's first sort-first frame
    // with its tile ids out of order — seen twice in three rounds of suite runs, both times in the test that renders ONE
    // frame on a fresh Renderer (tests/test_gpu_stages.py: test_tile_lists_equal_the_reference_own_code; round 2's
    // "small300" record; profiles/r04_s_gpu_test_matrix_and_null_stream_memset.txt).
    if (hipMemsetAsync(s->hist, 0, hist_bytes, ctx->stream) != hipSuccess) {
        sorter_free(s);
        return ctx_fail(ctx, SPLAT_ERR_HIP, "sorter workspace hipMemset");
    }
    s->capacity = (uint32_t)padded;
    s->ran = false;
    s->result_in_primary = true;
    return SPLAT_OK;
}

extern "C" {

int splat_sort_create(splat_ctx *ctx, uint32_t capacity, splat_sorter **out) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, out != nullptr);
    *out = nullptr;
    splat_sorter *s = new splat_sorter();
    s->ctx = ctx;
    int rc = sorter_reserve(s, capacity);
    if (rc != SPLAT_OK) {
        delete s;
        return rc;
    }
    *out = s;
    return SPLAT_OK;
}

void splat_sort_destroy(splat_sorter *s) {
    if (!s) return;
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->pinned_count) (void)hipHostFree(s->pinned_count);
    if (s->count_event) (void)hipEventDestroy(s->count_event);
    sorter_free(s);
    delete s;
}

uint32_t splat_sort_capacity(const splat_sorter *s) { return s ? s->capacity : 0; }
void *splat_sort_keys(splat_sorter *s) { return s ? s->keys : nullptr; }
void *splat_sort_payload(splat_sorter *s) { return s ? s->payload : nullptr; }

int splat_sort_run(splat_sorter *s, uint32_t n, uint32_t bit_begin, uint32_t bit_end) {
    if (!s) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "sorter is NULL");
    splat_ctx *ctx = s->ctx;
    ARG_CHECK(ctx, bit_begin <= bit_end && bit_end <= 32);
    if (n > s->capacity) return ctx_fail(ctx, SPLAT_ERR_CAPACITY, "splat_sort_run: n exceeds the sorter's capacity");
    stage_begin(ctx, SPLAT_STAGE_SORT);
    int rc = radix_sort_pairs(ctx, s->keys, s->payload, s->keys_b, s->payload_b, s->hist, n, bit_begin, bit_end,
                              &s->result_in_primary, s->mode);
    stage_end(ctx, SPLAT_STAGE_SORT);
    if (rc == SPLAT_OK) s->ran = true;
    return rc;
}

void *splat_sort_sorted_payload(splat_sorter *s) {
    if (!s) return nullptr;
    return s->result_in_primary ? s->payload : s->payload_b;
}

int splat_sort_set_mode(splat_sorter *s, int mode) {
    if (!s) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "sorter is NULL");
    if (mode != -1 && mode != 0 && mode != 2)
        return ctx_fail(s->ctx, SPLAT_ERR_INVALID, "sort mode must be -1 (default), 0 (rank as the context's policy says) or 2 (always ballot ranking)");
    s->mode = mode;
    return SPLAT_OK;
}

int splat_rank_status(splat_ctx *ctx, int *policy, int *atomics_ordered, uint32_t *order_faults) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    int rc = ctx_resolve_rank_mode(ctx);
    if (rc != SPLAT_OK) return rc;
    if (policy) *policy = ctx->rank_policy;
    if (atomics_ordered) *atomics_ordered = ctx->lds_atomic_ordered;
    if (order_faults) *order_faults = ctx->order_faults;
    return SPLAT_OK;
}

int splat_composite_options(splat_ctx *ctx, int kernel, int ahead, int predict) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, kernel >= -1 && kernel <= 1 && ahead >= 0 && ahead <= 2 && predict >= -1 && predict <= 1);
    ctx->opt_composite_kernel = kernel;
    ctx->opt_px_ahead = ahead;
    ctx->opt_px_predict = predict;
    return splat_composite_forget_history(ctx); // (another schedule: its costs mean something else)
}

int splat_composite_forget_history(splat_ctx *ctx) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    for (auto &h : ctx->px_hist) { // (the arrays stay: a slot's next launch starts a new streak in them)
        h.key = 0;
        h.streak = 0;
        h.last_use = 0;
    }
    return SPLAT_OK;
}

#ifdef SPLAT_TEST_HOOKS // (the test build only: include/splat.h)
int splat_debug_set_tile_sort_order(splat_ctx *ctx, const void *order_dptr) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ctx->debug_sort_order = (const uint32_t *)order_dptr;
    return SPLAT_OK;
}

int splat_debug_set_tile_order(splat_ctx *ctx, const void *order_dptr) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ctx->debug_tile_order = (const uint32_t *)order_dptr;
    return SPLAT_OK;
}

int splat_debug_lds_rate(splat_ctx *ctx, int kind, uint32_t workgroups_per_cu, uint32_t iters, float *ms) {
    if (!ctx || !ms) return ctx_fail(ctx, SPLAT_ERR_INVALID, "ctx/ms is NULL");
    ARG_CHECK(ctx, kind >= 0 && kind <= 2 && workgroups_per_cu >= 1 && workgroups_per_cu <= 8 && iters >= 1 && iters <= (1u << 20));
    return radix_probe_lds_rate(ctx, kind, workgroups_per_cu, iters, ms);
}

int splat_debug_tile_sort_launches(splat_ctx *ctx, uint32_t *launches) {
    if (!ctx || !launches) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx/launches is NULL");
    *launches = ctx->tile_sort_launches;
    return SPLAT_OK;
}

int splat_debug_inject_order_fault(splat_ctx *ctx, uint32_t tile, uint32_t position) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, tile != 0xffffffffu);
    ARG_CHECK(ctx, position < 0x3fffffffu); // (the kernel forms position + 2: no wrap, whatever a test passes)
    ctx->inject_order_fault = tile + 1u;
    ctx->inject_order_position = position;
    return SPLAT_OK;
}
#endif

int splat_probe_lds_atomic_order(splat_ctx *ctx, uint64_t *mismatches) {
    if (!ctx || !mismatches) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx/mismatches is NULL");
    return radix_probe_lds_atomic_order(ctx, mismatches);
}

void *splat_sort_sorted_keys(splat_sorter *s) {
    if (!s) return nullptr;
    return s->result_in_primary ? s->keys : s->keys_b;
}

} // extern "C"
