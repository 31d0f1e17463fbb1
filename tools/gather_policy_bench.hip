// gather_policy_bench.hip -- does the cache policy of a scattered 8-byte load change what a miss costs beyond the L2?
// A default load that misses L1 and L2 moves a whole 128-byte line from the Infinity Cache for 8 useful bytes; the forward
// kernel's fine levels (10 x 4 MB tables, hashed) do little else.  Stand-alone tuning tool, not part of the library:
//   hipcc --offload-arch=gfx950 -O3 tools/gather_policy_bench.hip -o build/gather_policy_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

// POLICY: 0 default, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc0 sc1 nt
template <int POLICY>
__device__ __forceinline__ v2f load8(const char *table, uint32_t e)
{
    const v2f *p = reinterpret_cast<const v2f *>(table) + e;
    v2f v;
    if (POLICY == 0) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 1) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 3) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 5) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int POLICY, int UNROLL>
__global__ void __launch_bounds__(512, 2) k(const char *__restrict__ table, uint32_t mask_entries, int iters, float *sink)
{
    uint32_t s = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        v2f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            s = s * 1664525u + 1013904223u;
            v[u] = load8<POLICY>(table, (s >> 4) & mask_entries);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y;
    }
    if (acc == 1234567.0f) sink[0] = acc;
}

template <int POLICY>
int run(const char *name, const char *table, int log2_entries, double ghz, int ncu, float *sink)
{
    constexpr int UNROLL = 16;
    const int iters = 64, blocks = ncu * 2 * 4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const uint32_t mask = (1u << log2_entries) - 1u;
    hipLaunchKernelGGL((k<POLICY, UNROLL>), dim3(blocks), dim3(512), 0, 0, table, mask, 2, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<POLICY, UNROLL>), dim3(blocks), dim3(512), 0, 0, table, mask, iters, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double loads = (double)blocks * 512 * iters * UNROLL;
    printf("%-14s table %9.0f KB: %7.3f ms  %7.1f Gload/s  %6.2f lane-loads/clk/CU\n", name, (double)(8u << log2_entries) / 1024.0, ms,
           loads / ms * 1e-6, loads / (ms * 1e-3) / (ghz * 1e9) / ncu);
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
    const int sizes[] = { 16, 19, 22, 23, 26 };  // 512 KB, 4 MB (one L2), 32 MB, 64 MB (Infinity Cache), 512 MB (HBM)
    for (int l : sizes) {
        if (run<0>("default", table, l, ghz, ncu, sink)) return 1;
        if (run<1>("nt", table, l, ghz, ncu, sink)) return 1;
        if (run<2>("sc0", table, l, ghz, ncu, sink)) return 1;
        if (run<3>("sc1", table, l, ghz, ncu, sink)) return 1;
        if (run<4>("sc0 sc1", table, l, ghz, ncu, sink)) return 1;
        if (run<5>("sc0 sc1 nt", table, l, ghz, ncu, sink)) return 1;
    }
    return 0;
}
