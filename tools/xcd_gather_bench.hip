// xcd_gather_bench.hip -- would a per-level encoder pass lift the training forward off the L2 miss path?
// NL table slices of 4 MB (2^19 8-byte entries, one hashed level each).  "mixed": every load picks a random slice (what the fused
// forward does: each wave walks all levels).  "per-XCD": workgroup b only reads slice (b % 8) + 8 * (phase): workgroups go
// round-robin over the 8 XCDs, so each XCD's 4 MiB L2 sees one slice at a time.  Stand-alone tuning tool:
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_gather_bench.hip -o build/xcd_gather_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kLog2T = 19, UNROLL = 32;

template <bool PER_XCD>
__global__ void __launch_bounds__(512, 2) k(const float2 *__restrict__ table, int nl, int phases, int iters, float *sink)
{
    uint32_t s = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int ph = 0; ph < phases; ++ph) {
        const uint32_t my = (blockIdx.x & 7) + 8 * ph;
        for (int it = 0; it < iters; ++it) {
            float v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                s = s * 1664525u + 1013904223u;
                const uint32_t e = (s >> 4) & ((1u << kLog2T) - 1u);
                const uint32_t lvl = PER_XCD ? my % nl : (s >> 24) % nl;
                const float2 t = table[((size_t)lvl << kLog2T) + e];
                v[u] = t.x + t.y;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u];
        }
    }
    if (acc == 1234567.0f) sink[0] = acc;
}

template <bool PER_XCD>
int run(const char *name, const float2 *table, int nl, int ncu, float *sink)
{
    const int phases = PER_XCD ? (nl + 7) / 8 : 1, iters = PER_XCD ? 64 / phases : 64, blocks = ncu * 2 * 4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<PER_XCD>), dim3(blocks), dim3(512), 0, 0, table, nl, phases, 2, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<PER_XCD>), dim3(blocks), dim3(512), 0, 0, table, nl, phases, iters, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double loads = (double)blocks * 512 * iters * phases * UNROLL;
    printf("%-10s %2d slices of 4 MB: %7.3f ms  %7.1f Gload/s\n", name, nl, ms, loads / ms * 1e-6);
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("%s: %d CUs\n", p.name, ncu);
    float2 *table;
    float *sink;
    const size_t bytes = (size_t)16 << (kLog2T + 3);
    CHECK(hipMalloc(&table, bytes));
    CHECK(hipMemset(table, 0, bytes));
    CHECK(hipMalloc(&sink, 4));
    for (int nl : { 1, 4, 8, 12, 16 }) {
        if (run<false>("mixed", table, nl, ncu, sink)) return 1;
        if (run<true>("per-XCD", table, nl, ncu, sink)) return 1;
    }
    return 0;
}
