// radix_sort.hip — RadixSorter: stable ascending LSD radix sort of (u32 key, u32 payload) pairs.
//
// Reference contract: /root/reference/src/RadixSorter.ts:39-100,197-271 (8 bits x 4 passes, result
// back in payload_a, stable).  The reference's WGSL (src/shaders/radix-sort.wgsl) emulates 32-wide
// subgroup match/rank through LDS and chains workgroups with a spinning decoupled look-back; this
// file is a wave64 design instead:
//
//   per pass:  k_radix_upsweep   per-partition 256-bin digit histogram (LDS atomics), written
//                                digit-major so ONE exclusive scan yields every (digit, partition)
//                                global base
//              scan_exclusive_u32 (scan.hip)
//              k_radix_downsweep each wave ranks its 16x64 keys with __ballot match masks
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

constexpr uint32_t RS_THREADS = 256;
constexpr uint32_t RS_ITEMS = RADIX_PART / RS_THREADS; // 16 keys per thread
constexpr uint32_t RS_WAVES = RS_THREADS / 64;
constexpr uint32_t RS_WAVE_KEYS = RADIX_PART / RS_WAVES; // 1024 keys per wave

// ---------------------------------------------------------------------------------------------
// upsweep: hist[d * num_parts + part] = number of keys of this partition whose digit is d
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RS_THREADS) void k_radix_upsweep(const uint32_t *__restrict__ keys, uint32_t n,
                                                              uint32_t shift, uint32_t mask, uint32_t num_parts,
                                                              uint32_t *__restrict__ hist) {
    __shared__ uint32_t lh[RS_WAVES][256]; // one private histogram per wave: fewer same-bank collisions
    const uint32_t tid = threadIdx.x, w = tid >> 6;
    for (uint32_t i = tid; i < RS_WAVES * 256; i += RS_THREADS) (&lh[0][0])[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * RADIX_PART;
    if (base + RADIX_PART <= n) {
        const uint4 *k4 = reinterpret_cast<const uint4 *>(keys + base);
#pragma unroll
        for (uint32_t j = 0; j < RS_ITEMS / 4; ++j) {
            uint4 v = k4[j * RS_THREADS + tid];
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
// downsweep
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t lanemask_lt() {
    const uint32_t lane = threadIdx.x & 63;
    return (lane == 0) ? 0ull : (~0ull >> (64 - lane));
}

__global__ __launch_bounds__(RS_THREADS) void k_radix_downsweep(const uint32_t *__restrict__ keys_in,
                                                                const uint32_t *__restrict__ pay_in,
                                                                uint32_t *__restrict__ keys_out,
                                                                uint32_t *__restrict__ pay_out, uint32_t n, uint32_t shift,
                                                                uint32_t mask, uint32_t num_parts,
                                                                const uint32_t *__restrict__ scanned_hist) {
    __shared__ uint32_t wave_hist[RS_WAVES][256]; // running per-wave digit counters -> wave offsets
    __shared__ uint32_t digit_base[256];          // partition-local start of each digit run
    __shared__ uint32_t global_base[256];         // global start of this partition's run of each digit
    __shared__ uint32_t wave_sums[RS_WAVES];
    __shared__ uint32_t s_keys[RADIX_PART];
    __shared__ uint32_t s_pay[RADIX_PART];

    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t part = blockIdx.x;
    const uint32_t base = part * RADIX_PART;
    const uint32_t valid = (n - base < RADIX_PART) ? (n - base) : RADIX_PART;

    for (uint32_t i = tid; i < RS_WAVES * 256; i += RS_THREADS) (&wave_hist[0][0])[i] = 0;
    global_base[tid] = scanned_hist[(size_t)tid * num_parts + part];

    // striped load: item i of lane l of wave w is element w*1024 + i*64 + l (position order =
    // (wave, item, lane), which is the order the ranking below preserves)
    uint32_t key[RS_ITEMS], pay[RS_ITEMS];
    const uint32_t wbase = w * RS_WAVE_KEYS + lane;
#pragma unroll
    for (uint32_t i = 0; i < RS_ITEMS; ++i) {
        uint32_t p = wbase + i * 64;
        bool ok = p < valid;
        key[i] = ok ? keys_in[base + p] : 0xffffffffu;
        pay[i] = ok ? pay_in[base + p] : 0xffffffffu;
    }
    __syncthreads(); // wave_hist zeroed

    const uint64_t lt = lanemask_lt();
    uint32_t rank[RS_ITEMS]; // rank among same-digit keys of this wave (items before + lanes below)
#pragma unroll
    for (uint32_t i = 0; i < RS_ITEMS; ++i) {
        const uint32_t p = wbase + i * 64;
        // padding lanes of the last partition get digit 255: they sit after every real key in
        // position order, so they also rank after every real key of digit 255
        const uint32_t d = (p < valid) ? ((key[i] >> shift) & mask) : 255u;
        uint64_t peers = ~0ull;
#pragma unroll
        for (uint32_t b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t below = __popcll(peers & lt);
        const uint32_t cnt = __popcll(peers);
        // LDS operations of one wave execute in program order: every peer's read is issued
        // before the leader's write, and the next item's read comes after it.  volatile keeps
        // the compiler from reordering or caching these accesses.
        volatile uint32_t *cnt_ptr = &wave_hist[w][d];
        const uint32_t prev = *cnt_ptr; // keys of digit d in earlier items of this wave
        rank[i] = prev + below;
        if (below == 0) *cnt_ptr = prev + cnt;
    }
    __syncthreads();

    // thread d: exclusive prefix over waves for digit d, and the partition's count of d
    uint32_t c0 = wave_hist[0][tid], c1 = wave_hist[1][tid], c2 = wave_hist[2][tid], c3 = wave_hist[3][tid];
    uint32_t dcount = c0 + c1 + c2 + c3;
    wave_hist[0][tid] = 0;
    wave_hist[1][tid] = c0;
    wave_hist[2][tid] = c0 + c1;
    wave_hist[3][tid] = c0 + c1 + c2;
    // exclusive scan of dcount over the 256 digits
    uint32_t incl = dcount;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        uint32_t t = __shfl_up(incl, s);
        if ((int)lane >= s) incl += t;
    }
    if (lane == 63) wave_sums[w] = incl;
    __syncthreads();
    uint32_t wprefix = (w > 0 ? wave_sums[0] : 0u) + (w > 1 ? wave_sums[1] : 0u) + (w > 2 ? wave_sums[2] : 0u);
    digit_base[tid] = wprefix + incl - dcount;
    __syncthreads();

    // reorder inside the partition: same-digit keys become contiguous, stable
#pragma unroll
    for (uint32_t i = 0; i < RS_ITEMS; ++i) {
        const uint32_t p = wbase + i * 64;
        const uint32_t d = (p < valid) ? ((key[i] >> shift) & mask) : 255u;
        const uint32_t pos = digit_base[d] + wave_hist[w][d] + rank[i];
        s_keys[pos] = key[i];
        s_pay[pos] = pay[i];
    }
    __syncthreads();

    // scatter: thread t handles local positions t, t+256, ...: consecutive lanes write
    // consecutive global addresses inside each digit run
#pragma unroll
    for (uint32_t j = 0; j < RS_ITEMS; ++j) {
        const uint32_t pos = j * RS_THREADS + tid;
        if (pos < valid) {
            const uint32_t k = s_keys[pos];
            const uint32_t d = (k >> shift) & mask;
            const uint32_t g = global_base[d] + (pos - digit_base[d]);
            keys_out[g] = k;
            pay_out[g] = s_pay[pos];
        }
    }
}

int radix_sort_pairs(splat_ctx *ctx, uint32_t *k0, uint32_t *p0, uint32_t *k1, uint32_t *p1, uint32_t *hist, uint32_t n,
                     uint32_t bit_begin, uint32_t bit_end, bool *result_in_primary) {
    *result_in_primary = true;
    if (n == 0 || bit_end <= bit_begin) return SPLAT_OK;
    const uint32_t parts = div_up(n, RADIX_PART);
    uint32_t *ki = k0, *pi = p0, *ko = k1, *po = p1;
    bool primary = true;
    for (uint32_t shift = bit_begin; shift < bit_end; shift += 8) {
        uint32_t bits = bit_end - shift < 8 ? bit_end - shift : 8;
        uint32_t mask = (1u << bits) - 1u;
        hipLaunchKernelGGL(k_radix_upsweep, dim3(parts), dim3(RS_THREADS), 0, ctx->stream, ki, n, shift, mask, parts, hist);
        LAUNCH_CHECK(ctx, "k_radix_upsweep");
        int rc = scan_exclusive_u32(ctx, hist, hist, 256u * parts, nullptr);
        if (rc != SPLAT_OK) return rc;
        hipLaunchKernelGGL(k_radix_downsweep, dim3(parts), dim3(RS_THREADS), 0, ctx->stream, ki, pi, ko, po, n, shift, mask,
                           parts, hist);
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
    size_t hist_bytes = (size_t)256 * div_up((uint32_t)padded, RADIX_PART) * 4;
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
