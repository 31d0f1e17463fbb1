// gather_bench.hip -- scattered-load throughput on gfx950 by table size and load width: what bounds the hash encoder's
// gathers (L1 tag rate, L2 fill rate, fabric)?  Stand-alone tuning tool, not part of the library:
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o build/gather_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// WIDTH bytes per lane per load (8 or 16); PAIR: lanes 2k and 2k+1 read the two halves of one 16-B chunk
template <int WIDTH, bool PAIR, int UNROLL>
__global__ void __launch_bounds__(512, 2) k(const char *__restrict__ table, uint32_t mask_entries, int iters, float *sink)
{
    uint32_t s = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        float v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            s = s * 1664525u + 1013904223u;
            uint32_t r = s >> 4;
            if (PAIR) r = (__shfl(r, threadIdx.x & ~1, 64) & ~1u) | (threadIdx.x & 1);
            const uint32_t e = r & mask_entries;  // 8-byte entries
            if (WIDTH == 8) {
                const float2 t = reinterpret_cast<const float2 *>(table)[e];
                v[u] = t.x + t.y;
            } else {
                const float4 t = reinterpret_cast<const float4 *>(table)[e >> 1];
                v[u] = t.x + t.y + t.z + t.w;
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    if (acc == 1234567.0f) sink[0] = acc;
}

template <int WIDTH, bool PAIR>
int run(const char *name, const char *table, int log2_entries, double ghz, int ncu, float *sink)
{
    constexpr int UNROLL = 32;
    const int iters = 64, blocks = ncu * 2 * 4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const uint32_t mask = (1u << log2_entries) - 1u;
    hipLaunchKernelGGL((k<WIDTH, PAIR, UNROLL>), dim3(blocks), dim3(512), 0, 0, table, mask, 2, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<WIDTH, PAIR, UNROLL>), dim3(blocks), dim3(512), 0, 0, table, mask, iters, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double loads = (double)blocks * 512 * iters * UNROLL;
    printf("%-22s table %9.0f KB: %7.3f ms  %7.1f Gload/s  %6.2f lane-loads/clk/CU  %6.2f TB/s useful\n", name,
           (double)(8u << log2_entries) / 1024.0, ms, loads / ms * 1e-6, loads / (ms * 1e-3) / (ghz * 1e9) / ncu,
           loads * WIDTH / (ms * 1e-3) * 1e-12);
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", p.name, ncu, ghz);
    char *table;
    float *sink;
    const size_t bytes = (size_t)8 << 26;  // 512 MiB
    CHECK(hipMalloc(&table, bytes));
    CHECK(hipMemset(table, 0, bytes));
    CHECK(hipMalloc(&sink, 4));
    const int sizes[] = { 8, 11, 14, 16, 19, 21, 23, 26 };  // 2 KB (L1) ... 4 MB (one L2) ... 64 MB (MALL) ... 512 MB (HBM)
    for (int l : sizes) {
        if (run<8, false>("8 B random", table, l, ghz, ncu, sink)) return 1;
        if (run<8, true>("8 B lane pairs/16 B", table, l, ghz, ncu, sink)) return 1;
        if (run<16, false>("16 B random", table, l, ghz, ncu, sink)) return 1;
    }
    return 0;
}
