// scan.hip — PrefixSumScanner: exclusive scan of u32 on the device for any length.
//
// Reference: /root/reference/src/PrefixSumScanner.ts:74-87 (scan), src/shaders/prefix-sum.wgsl:28-96
// (single-workgroup Blelloch, <=512 elements) and :131-162 (CPU readback loop above that).  Here
// every length stays on the device: one workgroup for short inputs, reduce / scan-sums / apply
// for long ones.  Wave64 shuffles do the intra-wave scan; LDS carries only the 4 wave totals.
//
// Roofline: HBM, 12 B per element in the 3-phase form (read, read, write), 8 B single-block.
#include "common.h"

constexpr uint32_t SCAN_THREADS = 256;
constexpr uint32_t SCAN_UNIT = SCAN_THREADS * 4;  // one uint4 per thread
constexpr uint32_t SCAN_UNITS_PER_TILE = 8;
constexpr uint32_t SCAN_TILE = SCAN_UNIT * SCAN_UNITS_PER_TILE; // 8192 elements per workgroup
constexpr uint32_t SCAN_SINGLE_MAX = 16384;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread over the 256-thread workgroup; returns this thread's
// exclusive prefix and the workgroup total. wave_sums is 4 u32 of LDS.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *wave_sums, uint32_t &total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t incl = wave_inclusive_scan(v);
    if (lane == 63) wave_sums[w] = incl;
    __syncthreads();
    uint32_t s0 = wave_sums[0], s1 = wave_sums[1], s2 = wave_sums[2], s3 = wave_sums[3];
    __syncthreads(); // wave_sums may be rewritten by the next unit
    uint32_t prefix = (w > 0 ? s0 : 0u) + (w > 1 ? s1 : 0u) + (w > 2 ? s2 : 0u);
    total = s0 + s1 + s2 + s3;
    return prefix + incl - v;
}

__device__ __forceinline__ uint4 load_unit(const uint32_t *in, uint32_t base, uint32_t n) {
    uint32_t i = base + threadIdx.x * 4;
    if (i + 4 <= n) return *reinterpret_cast<const uint4 *>(in + i);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (i < n) v.x = in[i];
    if (i + 1 < n) v.y = in[i + 1];
    if (i + 2 < n) v.z = in[i + 2];
    return v;
}

__device__ __forceinline__ void store_unit(uint32_t *out, uint32_t base, uint32_t n, uint4 v) {
    uint32_t i = base + threadIdx.x * 4;
    if (i + 4 <= n) {
        *reinterpret_cast<uint4 *>(out + i) = v;
        return;
    }
    if (i < n) out[i] = v.x;
    if (i + 1 < n) out[i + 1] = v.y;
    if (i + 2 < n) out[i + 2] = v.z;
}

// scans units [unit0, unit1) of `in` into `out`, starting from `carry`; returns the final carry
__device__ __forceinline__ uint32_t scan_units(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t elem0,
                                               uint32_t elem1, uint32_t carry, uint32_t *wave_sums) {
    for (uint32_t base = elem0; base < elem1; base += SCAN_UNIT) {
        uint4 v = load_unit(in, base, n);
        uint32_t local = v.x + v.y + v.z + v.w, total;
        uint32_t ex = block_exclusive_scan(local, wave_sums, total) + carry;
        uint4 o;
        o.x = ex;
        o.y = ex + v.x;
        o.z = o.y + v.y;
        o.w = o.z + v.z;
        store_unit(out, base, n, o);
        carry += total;
    }
    return carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_single(const uint32_t *in, uint32_t *out, uint32_t n,
                                                              uint32_t *total_out) {
    __shared__ uint32_t wave_sums[4];
    uint32_t carry = scan_units(in, out, n, 0, n, 0u, wave_sums);
    if (total_out && threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const uint32_t *__restrict__ in, uint32_t n,
                                                              uint32_t *__restrict__ sums) {
    __shared__ uint32_t wave_sums[4];
    uint32_t elem0 = blockIdx.x * SCAN_TILE;
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_UNITS_PER_TILE; ++k) {
        uint32_t base = elem0 + k * SCAN_UNIT;
        if (base < n) {
            uint4 v = load_unit(in, base, n);
            acc += v.x + v.y + v.z + v.w;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0) wave_sums[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const uint32_t *in, uint32_t *out, uint32_t n,
                                                             const uint32_t *__restrict__ scanned_sums) {
    __shared__ uint32_t wave_sums[4];
    uint32_t elem0 = blockIdx.x * SCAN_TILE;
    uint32_t elem1 = elem0 + SCAN_TILE < n ? elem0 + SCAN_TILE : n;
    scan_units(in, out, n, elem0, elem1, scanned_sums[blockIdx.x], wave_sums);
}

int scan_exclusive_u32(splat_ctx *ctx, const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total) {
    if (n == 0) {
        if (total) HIP_TRY(ctx, hipMemsetAsync(total, 0, 4, ctx->stream));
        return SPLAT_OK;
    }
    if (n <= SCAN_SINGLE_MAX) {
        hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, in, out, n, total);
        LAUNCH_CHECK(ctx, "k_scan_single");
        return SPLAT_OK;
    }
    uint32_t tiles = div_up(n, SCAN_TILE);
    // workspace: level-1 sums, then recursively level-2 sums ... all carved from ctx->scan_ws
    size_t need = 0;
    for (uint32_t t = tiles;; t = div_up(t, SCAN_TILE)) {
        need += ((size_t)t * 4 + 255) & ~(size_t)255;
        if (t <= SCAN_SINGLE_MAX) break;
    }
    int rc = ctx_ensure_scan_ws(ctx, need);
    if (rc != SPLAT_OK) return rc;
    // iterative descent (tiles > SCAN_SINGLE_MAX only for n > 134M, but handle it)
    uint32_t *level_sums[4];
    uint32_t level_n[4];
    int levels = 0;
    {
        char *p = (char *)ctx->scan_ws;
        for (uint32_t t = tiles;; t = div_up(t, SCAN_TILE)) {
            if (levels >= 4) return ctx_fail(ctx, SPLAT_ERR_INVALID, "scan: input too long");
            level_sums[levels] = (uint32_t *)p;
            level_n[levels] = t;
            p += ((size_t)t * 4 + 255) & ~(size_t)255;
            ++levels;
            if (t <= SCAN_SINGLE_MAX) break;
        }
    }
    // reduce pass per level
    const uint32_t *src = in;
    uint32_t src_n = n;
    for (int l = 0; l < levels; ++l) {
        hipLaunchKernelGGL(k_scan_reduce, dim3(level_n[l]), dim3(SCAN_THREADS), 0, ctx->stream, src, src_n, level_sums[l]);
        LAUNCH_CHECK(ctx, "k_scan_reduce");
        src = level_sums[l];
        src_n = level_n[l];
    }
    // top level: single workgroup, in place; its total is the grand total
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, level_sums[levels - 1],
                       level_sums[levels - 1], level_n[levels - 1], total);
    LAUNCH_CHECK(ctx, "k_scan_single(top)");
    // apply pass per level, top-down
    for (int l = levels - 1; l >= 0; --l) {
        const uint32_t *lin = (l == 0) ? in : level_sums[l - 1];
        uint32_t *lout = (l == 0) ? out : level_sums[l - 1];
        uint32_t ln = (l == 0) ? n : level_n[l - 1];
        hipLaunchKernelGGL(k_scan_apply, dim3(level_n[l]), dim3(SCAN_THREADS), 0, ctx->stream, lin, lout, ln, level_sums[l]);
        LAUNCH_CHECK(ctx, "k_scan_apply");
    }
    return SPLAT_OK;
}

extern "C" int splat_scan_u32(splat_ctx *ctx, const void *in, void *out, uint32_t n, void *total_dptr) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, n == 0 || (in && out));
    ARG_CHECK(ctx, (((uintptr_t)in | (uintptr_t)out) & 15) == 0);
    return scan_exclusive_u32(ctx, (const uint32_t *)in, (uint32_t *)out, n, (uint32_t *)total_dptr);
}
