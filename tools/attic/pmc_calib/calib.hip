// calib.hip — calibration of rocprofv3's FETCH_SIZE for the composite's access pattern (random 16- and
// 32-byte gathers), as MI355X_MICROARCH.md prescribes for anything that is not a wide coalesced stream:
// "calibrate on a known byte count in your own access pattern before trusting an absolute".
//
//   hipcc --offload-arch=gfx950 -O3 calib.hip -o calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o c -- ./calib
//
// Every kernel touches each 64-byte line of a 1 GiB table (4x the 256 MiB Infinity Cache) at most once:
//   k_stream   16 B per lane, coalesced, the whole table            known bytes = table
//   k_gather16 one 16-byte load from each of M distinct random lines    lines x 64 B (32 B if sectors are fetched)
//   k_gather32 one 32-byte record (2 x 16 B) from each of M distinct random lines
// The driver prints M and the byte counts; tools/pmc_calib/report.py divides FETCH_SIZE by them.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            return 1;                                                          \
        }                                                                      \
    } while (0)

__global__ __launch_bounds__(256) void k_stream(const float4 *__restrict__ t, size_t n4, float *sink) {
    float acc = 0.0f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256ull) {
        const float4 v = t[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) *sink = acc; // (never true for a zero table: keeps the loads alive)
}

template <int VECS>
__global__ __launch_bounds__(256) void k_gather(const float4 *__restrict__ t, const uint32_t *__restrict__ line_of, uint32_t m,
                                                float *sink) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const size_t base = (size_t)line_of[i] * 4; // 4 float4 per 64-byte line
    float4 v = t[base];
    float acc = v.x + v.y + v.z + v.w;
    if (VECS == 2) {
        v = t[base + 1];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) *sink = acc;
}

int main() {
    const size_t table_bytes = 1ull << 30, lines = table_bytes / 64;
    const uint32_t m = 4u << 20; // 4M gathers: a quarter of the lines, none twice
    float4 *table = nullptr;
    uint32_t *line_of = nullptr;
    float *sink = nullptr;
    CK(hipMalloc((void **)&table, table_bytes));
    CK(hipMemset(table, 0, table_bytes));
    CK(hipMalloc((void **)&line_of, (size_t)m * 4));
    CK(hipMalloc((void **)&sink, 4));
    std::vector<uint32_t> perm(lines);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937_64 rng(12345);
    for (size_t i = 0; i < m; ++i) { // partial Fisher-Yates: m distinct random lines
        std::uniform_int_distribution<size_t> d(i, lines - 1);
        std::swap(perm[i], perm[d(rng)]);
    }
    CK(hipMemcpy(line_of, perm.data(), (size_t)m * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, table, table_bytes / 16, sink);
        hipLaunchKernelGGL(k_gather<1>, dim3((m + 255) / 256), dim3(256), 0, 0, table, line_of, m, sink);
        hipLaunchKernelGGL(k_gather<2>, dim3((m + 255) / 256), dim3(256), 0, 0, table, line_of, m, sink);
        CK(hipDeviceSynchronize());
    }
    printf("{\"table_bytes\": %zu, \"gathers\": %u, \"index_bytes\": %zu}\n", table_bytes, m, (size_t)m * 4);
    return 0;
}
