// radix_sort.hip — RadixSorter: stable ascending LSD radix sort of (u32 key, u32 payload) pairs.
//
// Reference contract: /root/reference/src/RadixSorter.ts:39-100,197-271 (8 bits x 4 passes, result
// back in payload_a, stable).  The reference's WGSL (src/shaders/radix-sort.wgsl) emulates 32-wide
// subgroup match/rank through LDS and chains workgroups with a spinning decoupled look-back; this
// file is a wave64 design instead:
//
//   per pass:  k_radix_upsweep   per-partition 256-bin digit histogram (LDS atomics), written
//                                digit-major: row d = counts of digit d over the partitions
//              k_radix_rowscan   256 workgroups, one per digit row: exclusive scan along the
//                                partitions in place + the row total
//              k_radix_downsweep scans the 256 row totals itself (digit bases), then each wave ranks its 16x64 keys with __ballot match masks
//                                (rank = popcount of same-digit lanes below me + running per-wave
//                                digit counter in LDS), the workgroup reorders keys by digit in
//                                LDS so global stores go out in digit runs, then scatters.
//
// No inter-workgroup spinning: forward progress never depends on dispatch order (the guide's
// "give every wave an exit condition" rule), at the price of reading the keys twice per pass.
//
// Roofline: HBM.  Algorithmic bytes per key per pass: 4 (upsweep read) + 8 (read key+payload)
// + 8 (write) = 20; 80 B/key for the 4-pass depth sort.
#include "common.h"

#include <cstdlib>

constexpr uint32_t RS_THREADS = 256;
constexpr uint32_t RS_ITEMS = RADIX_PART / RS_THREADS; // 16 keys per thread
constexpr uint32_t RS_WAVES = RS_THREADS / 64;

// ---------------------------------------------------------------------------------------------
// Partitioning.  A partition is items*256 keys; the downsweep is instantiated for several sizes
// (SPLAT_RADIX_ITEMS selects one for tuning runs).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t RS_MIN_ITEMS = 4;
static const uint32_t kItemChoices[] = {4, 6, 8, 12, 14, 16, 20, 24, 32};

static uint32_t g_force_items = 0; // tuning hook: SPLAT_RADIX_ITEMS env var (read once)

static uint32_t wg_per_cu_for(uint32_t items) { // mirrors downsweep_wg_per_cu (defined with the kernel below)
    return items <= 6 ? 8 : items <= 12 ? 5 : items <= 16 ? 4 : items <= 20 ? 3 : 2;
}

// Measured on MI355X (tools/sort_bench.py, 5M and 11.3M pairs): 4 items 346 us, 6 -> 266, 8 -> 228,
// 12 -> 193, 14 -> 184, 16 -> 176, 20 -> 174, 24 -> 190, 32 -> 177 (4-pass 5M-key sort).  Bigger
// partitions win until ~16 (longer digit runs = better store coalescing, fixed per-partition work
// amortised) and then occupancy loss cancels the gain, so 16 is the default at every size tried.
static uint32_t choose_items(uint32_t n) {
    (void)n;
    return RS_ITEMS;
}

// ---------------------------------------------------------------------------------------------
// upsweep: hist[d * num_parts + part] = number of keys of this partition whose digit is d
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RS_THREADS) void k_radix_upsweep(const uint32_t *__restrict__ keys, uint32_t n,
                                                              uint32_t shift, uint32_t mask, uint32_t num_parts,
                                                              uint32_t part_keys, uint32_t *__restrict__ hist) {
    __shared__ uint32_t lh[RS_WAVES][256]; // one private histogram per wave: fewer same-bank collisions
    const uint32_t tid = threadIdx.x, w = tid >> 6;
    for (uint32_t i = tid; i < RS_WAVES * 256; i += RS_THREADS) (&lh[0][0])[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * part_keys; // multiple of 256 keys = 1 KiB: uint4 loads stay aligned
    if (base + part_keys <= n) {
        const uint4 *k4 = reinterpret_cast<const uint4 *>(keys + base);
        for (uint32_t j = tid; j < part_keys / 4; j += RS_THREADS) {
            uint4 v = k4[j];
            atomicAdd(&lh[w][(v.x >> shift) & mask], 1u);
            atomicAdd(&lh[w][(v.y >> shift) & mask], 1u);
            atomicAdd(&lh[w][(v.z >> shift) & mask], 1u);
            atomicAdd(&lh[w][(v.w >> shift) & mask], 1u);
        }
    } else {
        for (uint32_t i = base + tid; i < n; i += RS_THREADS) atomicAdd(&lh[w][(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    uint32_t c = lh[0][tid] + lh[1][tid] + lh[2][tid] + lh[3][tid];
    hist[(size_t)tid * num_parts + blockIdx.x] = c;
}

// ---------------------------------------------------------------------------------------------
// rowscan: block d turns row d of hist into its exclusive prefix over partitions; totals[d] = row sum
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RS_THREADS) void k_radix_rowscan(uint32_t *__restrict__ hist, uint32_t num_parts,
                                                              uint32_t *__restrict__ totals) {
    __shared__ uint32_t wsum[RS_WAVES];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint32_t *row = hist + (size_t)blockIdx.x * num_parts;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < num_parts; base += RS_THREADS) {
        const uint32_t i = base + tid;
        const uint32_t v = (i < num_parts) ? row[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            uint32_t t = __shfl_up(incl, s);
            if ((int)lane >= s) incl += t;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        const uint32_t s0 = wsum[0], s1 = wsum[1], s2 = wsum[2], s3 = wsum[3];
        __syncthreads();
        const uint32_t wprefix = (w > 0 ? s0 : 0u) + (w > 1 ? s1 : 0u) + (w > 2 ? s2 : 0u);
        if (i < num_parts) row[i] = carry + wprefix + incl - v;
        carry += s0 + s1 + s2 + s3;
    }
    if (tid == 0) totals[blockIdx.x] = carry;
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
    uint32_t digit_base[256];          // partition-local start of each digit run
    uint32_t global_base[256];         // global start of this partition's run of each digit
    uint32_t wave_sums[RS_WAVES];
    uint32_t wave_gsums[RS_WAVES];
};

template <uint32_t ITEMS, bool FULL>
__device__ __forceinline__ void downsweep_body(DownsweepShared &sh, uint2 *__restrict__ s_kp,
                                               const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ pay_in,
                                               uint32_t *__restrict__ keys_out, uint32_t *__restrict__ pay_out, uint32_t n,
                                               uint32_t shift, uint32_t mask, uint32_t num_parts,
                                               const uint32_t *__restrict__ scanned_hist,
                                               const uint32_t *__restrict__ totals) {
    constexpr uint32_t PART_KEYS = ITEMS * RS_THREADS, WAVE_KEYS = ITEMS * 64;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t part = blockIdx.x;
    const uint32_t base = part * PART_KEYS;
    const uint32_t valid = FULL ? PART_KEYS : (n - base);

    for (uint32_t i = tid; i < RS_WAVES * 256; i += RS_THREADS) (&sh.wave_hist[0][0])[i] = 0;
    const uint32_t row_prefix = scanned_hist[(size_t)tid * num_parts + part]; // digit tid in earlier partitions
    const uint32_t digit_total = totals[tid];

    // striped load: item i of lane l of wave w is element w*WAVE_KEYS + i*64 + l (position order =
    // (wave, item, lane), which is the order the ranking below preserves).  Padding lanes of the
    // last partition read the last real element and are given digit 255 below: they sit after
    // every real key in position order, so they also rank after every real key of digit 255.
    uint32_t key[ITEMS], pay[ITEMS];
    const uint32_t wbase = w * WAVE_KEYS + lane;
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) {
        const uint32_t p = wbase + i * 64;
        const uint32_t q = FULL ? p : ((p < valid) ? p : (valid - 1));
        key[i] = keys_in[base + q];
        pay[i] = pay_in[base + q];
    }
    __syncthreads(); // wave_hist zeroed

    // ---- rank, phase A: per item, the mask of lanes of this wave holding the same digit: 8 ballots,
    // each folded in with one v_bitop3 per mask half (peers &= ~(ballot ^ mybit)).  The lowest lane
    // of every digit group then adds the group's size to the wave's digit counter with a RETURNING
    // LDS atomic.  Within one instruction the leaders have distinct digits (no collisions), and the
    // returning atomics of a wave execute in program order, so the value returned for item i is
    // the number of keys of that digit in items < i: a stable rank, with the LDS round trips
    // pipelined (nothing waits on them until phase B) instead of chained.
    uint32_t rank[ITEMS], prev[ITEMS];
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
    // ---- phase B: every lane fetches its leader's result -------------------------------------------
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) rank[i] = __shfl(prev[i], rank[i] >> 8) + (rank[i] & 0xffu);
    __syncthreads();

    // thread d: exclusive prefix over waves for digit d, and the partition's count of d
    uint32_t c0 = sh.wave_hist[0][tid], c1 = sh.wave_hist[1][tid], c2 = sh.wave_hist[2][tid], c3 = sh.wave_hist[3][tid];
    uint32_t dcount = c0 + c1 + c2 + c3;
    sh.wave_hist[0][tid] = 0;
    sh.wave_hist[1][tid] = c0;
    sh.wave_hist[2][tid] = c0 + c1;
    sh.wave_hist[3][tid] = c0 + c1 + c2;
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
    sh.digit_base[tid] = wprefix + incl - dcount;
    sh.global_base[tid] = gprefix + gincl - digit_total + row_prefix;
    __syncthreads();

    // reorder inside the partition: same-digit keys become contiguous, stable
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) {
        uint32_t d = (key[i] >> shift) & mask;
        if (!FULL) d = (wbase + i * 64 < valid) ? d : 255u;
        const uint32_t pos = sh.digit_base[d] + sh.wave_hist[w][d] + rank[i];
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
            const uint32_t g = sh.global_base[d] + (pos - sh.digit_base[d]);
            keys_out[g] = kp.x;
            pay_out[g] = kp.y;
        }
    }
}

// resident workgroups per CU for a given ITEMS: LDS-bound (160 KiB per CU), capped at 8 (32 waves)
constexpr uint32_t downsweep_lds_bytes(uint32_t items) { return items * RS_THREADS * 8 + (uint32_t)sizeof(DownsweepShared); }
// ... and register-bound: ~6.5 VGPRs per item + ~20, so the LDS count is lowered where asking for it
// would make the compiler spill
constexpr uint32_t downsweep_wg_per_cu(uint32_t items) {
    return items <= 6 ? 8 : items <= 12 ? 5 : items <= 16 ? 4 : items <= 20 ? 3 : 2;
}
static_assert(downsweep_wg_per_cu(16) * downsweep_lds_bytes(16) <= 160u * 1024u, "LDS budget");
static_assert(downsweep_wg_per_cu(20) * downsweep_lds_bytes(20) <= 160u * 1024u, "LDS budget");
static_assert(downsweep_wg_per_cu(32) * downsweep_lds_bytes(32) <= 160u * 1024u, "LDS budget");
static_assert(downsweep_wg_per_cu(12) * downsweep_lds_bytes(12) <= 160u * 1024u, "LDS budget");
static_assert(downsweep_wg_per_cu(8) * downsweep_lds_bytes(8) <= 160u * 1024u, "LDS budget");
static_assert(downsweep_wg_per_cu(6) * downsweep_lds_bytes(6) <= 160u * 1024u, "LDS budget");

template <uint32_t ITEMS>
__global__ __launch_bounds__(RS_THREADS, downsweep_wg_per_cu(ITEMS)) void k_radix_downsweep(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ pay_in, uint32_t *__restrict__ keys_out,
    uint32_t *__restrict__ pay_out, uint32_t n, uint32_t shift, uint32_t mask, uint32_t num_parts,
    const uint32_t *__restrict__ scanned_hist, const uint32_t *__restrict__ totals) {
    __shared__ DownsweepShared sh;
    __shared__ uint2 s_kp[ITEMS * RS_THREADS]; // (key, payload) reordered by digit
    // every partition but (possibly) the last is full: it takes the path with no per-key bounds checks
    if ((blockIdx.x + 1) * ITEMS * RS_THREADS <= n)
        downsweep_body<ITEMS, true>(sh, s_kp, keys_in, pay_in, keys_out, pay_out, n, shift, mask, num_parts, scanned_hist, totals);
    else
        downsweep_body<ITEMS, false>(sh, s_kp, keys_in, pay_in, keys_out, pay_out, n, shift, mask, num_parts, scanned_hist, totals);
}

template <uint32_t ITEMS>
static void launch_downsweep(splat_ctx *ctx, uint32_t parts, const uint32_t *ki, const uint32_t *pi, uint32_t *ko, uint32_t *po,
                             uint32_t n, uint32_t shift, uint32_t mask, const uint32_t *hist, const uint32_t *totals) {
    hipLaunchKernelGGL(k_radix_downsweep<ITEMS>, dim3(parts), dim3(RS_THREADS), 0, ctx->stream, ki, pi, ko, po, n, shift, mask,
                       parts, hist, totals);
}

int radix_sort_pairs(splat_ctx *ctx, uint32_t *k0, uint32_t *p0, uint32_t *k1, uint32_t *p1, uint32_t *hist, uint32_t n,
                     uint32_t bit_begin, uint32_t bit_end, bool *result_in_primary) {
    static bool env_read = false;
    if (!env_read) {
        env_read = true;
        if (const char *e = getenv("SPLAT_RADIX_ITEMS")) {
            uint32_t v = (uint32_t)atoi(e);
            for (uint32_t c : kItemChoices)
                if (c == v) g_force_items = v;
        }
    }
    *result_in_primary = true;
    if (n == 0 || bit_end <= bit_begin) return SPLAT_OK;
    const uint32_t items = g_force_items ? g_force_items : choose_items(n);
    const uint32_t part_keys = items * RS_THREADS;
    const uint32_t parts = div_up(n, part_keys); // <= div_up(n, RS_MIN_ITEMS*256): the hist workspace is sized for that
    uint32_t *ki = k0, *pi = p0, *ko = k1, *po = p1;
    bool primary = true;
    for (uint32_t shift = bit_begin; shift < bit_end; shift += 8) {
        uint32_t bits = bit_end - shift < 8 ? bit_end - shift : 8;
        uint32_t mask = (1u << bits) - 1u;
        hipLaunchKernelGGL(k_radix_upsweep, dim3(parts), dim3(RS_THREADS), 0, ctx->stream, ki, n, shift, mask, parts, part_keys,
                           hist);
        LAUNCH_CHECK(ctx, "k_radix_upsweep");
        uint32_t *totals = hist + (size_t)256 * parts;
        hipLaunchKernelGGL(k_radix_rowscan, dim3(256), dim3(RS_THREADS), 0, ctx->stream, hist, parts, totals);
        LAUNCH_CHECK(ctx, "k_radix_rowscan");
        switch (items) {
        case 4: launch_downsweep<4>(ctx, parts, ki, pi, ko, po, n, shift, mask, hist, totals); break;
        case 6: launch_downsweep<6>(ctx, parts, ki, pi, ko, po, n, shift, mask, hist, totals); break;
        case 8: launch_downsweep<8>(ctx, parts, ki, pi, ko, po, n, shift, mask, hist, totals); break;
        case 12: launch_downsweep<12>(ctx, parts, ki, pi, ko, po, n, shift, mask, hist, totals); break;
        case 14: launch_downsweep<14>(ctx, parts, ki, pi, ko, po, n, shift, mask, hist, totals); break;
        case 20: launch_downsweep<20>(ctx, parts, ki, pi, ko, po, n, shift, mask, hist, totals); break;
        case 24: launch_downsweep<24>(ctx, parts, ki, pi, ko, po, n, shift, mask, hist, totals); break;
        case 32: launch_downsweep<32>(ctx, parts, ki, pi, ko, po, n, shift, mask, hist, totals); break;
        default: launch_downsweep<16>(ctx, parts, ki, pi, ko, po, n, shift, mask, hist, totals); break;
        }
        LAUNCH_CHECK(ctx, "k_radix_downsweep");
        uint32_t *t = ki; ki = ko; ko = t;
        t = pi; pi = po; po = t;
        primary = !primary;
    }
    *result_in_primary = primary;
    return SPLAT_OK;
}

// ---------------------------------------------------------------------------------------------
// RadixSorter object
// ---------------------------------------------------------------------------------------------
static void sorter_free(splat_sorter *s) {
    if (s->keys) (void)hipFree(s->keys);
    if (s->keys_b) (void)hipFree(s->keys_b);
    if (s->payload) (void)hipFree(s->payload);
    if (s->payload_b) (void)hipFree(s->payload_b);
    if (s->hist) (void)hipFree(s->hist);
    s->keys = s->keys_b = s->payload = s->payload_b = s->hist = nullptr;
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
    // rows for the smallest partition size + the 256 row totals
    size_t hist_bytes = ((size_t)256 * div_up((uint32_t)padded, RS_MIN_ITEMS * RS_THREADS) + 256) * 4;
    if (hipMalloc((void **)&s->keys, bytes) != hipSuccess || hipMalloc((void **)&s->keys_b, bytes) != hipSuccess ||
        hipMalloc((void **)&s->payload, bytes) != hipSuccess || hipMalloc((void **)&s->payload_b, bytes) != hipSuccess ||
        hipMalloc((void **)&s->hist, hist_bytes) != hipSuccess) {
        sorter_free(s);
        return ctx_fail(ctx, SPLAT_ERR_OOM, "sorter hipMalloc");
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
                              &s->result_in_primary);
    stage_end(ctx, SPLAT_STAGE_SORT);
    if (rc == SPLAT_OK) s->ran = true;
    return rc;
}

void *splat_sort_sorted_payload(splat_sorter *s) {
    if (!s) return nullptr;
    return s->result_in_primary ? s->payload : s->payload_b;
}

void *splat_sort_sorted_keys(splat_sorter *s) {
    if (!s) return nullptr;
    return s->result_in_primary ? s->keys : s->keys_b;
}

} // extern "C"
