// lds_atomic_bench.hip -- LDS atomic throughput on gfx950 (lane-operations per clock per CU) for the primitives the
// table-gradient accumulate could be built on.  Stand-alone tuning tool, not part of the library:
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o gpurun_out/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum Op { U32, U64, F32, F64, U32_RTN, U64_RTN, PK64, RMW64 };

template <int OP, int PATTERN>  // PATTERN 0 random, 1 same address per wave, 2 conflict-free (lane-linear), 3 runs of 4 equal
__global__ void __launch_bounds__(1024) k(int iters, int image_log, uint32_t *sink)
{
    extern __shared__ unsigned long long img[];
    const int n64 = 1 << image_log;
    for (int i = threadIdx.x; i < n64; i += 1024) img[i] = 0;
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        uint32_t a = s >> 8;
        if (PATTERN == 1) a = __builtin_amdgcn_readfirstlane(a);
        if (PATTERN == 2) a = (threadIdx.x & 63) + (it << 6);
        if (PATTERN == 3) a = __shfl(a, (threadIdx.x & 63) & ~3, 64);
        a &= (uint32_t)n64 - 1u;
        if (OP == U32) atomicAdd(reinterpret_cast<uint32_t *>(img) + 2 * a, s);
        if (OP == U64) atomicAdd(img + a, (unsigned long long)s);
        if (OP == F32) unsafeAtomicAdd(reinterpret_cast<float *>(img) + 2 * a, 1.0f);
        if (OP == F64) unsafeAtomicAdd(reinterpret_cast<double *>(img) + a, 1.0);
        if (OP == U32_RTN) acc += atomicAdd(reinterpret_cast<uint32_t *>(img) + 2 * a, 1u);
        if (OP == U64_RTN) acc += (uint32_t)atomicAdd(img + a, 1ull);
        if (OP == PK64) {  // two 32-bit fixed-point fields in one 64-bit add
            const long long v = ((long long)(int)s << 32) + (long long)(int)(s * 3u);
            atomicAdd(img + a, (unsigned long long)v);
        }
        if (OP == RMW64) {  // non-atomic read-modify-write (racy; rate reference only)
            volatile unsigned long long *p = img + a;
            *p = *p + s;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n64; i += 1024) acc += (uint32_t)img[i];
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int OP, int PATTERN>
int run(const char *name, int image_log, double clk_ghz, int ncu, uint32_t *sink)
{
    const int iters = 4096, blocks = ncu * 2 * 4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const size_t lds = (size_t)8 << image_log;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k<OP, PATTERN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k<OP, PATTERN>), dim3(blocks), dim3(1024), lds, 0, 16, image_log, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<OP, PATTERN>), dim3(blocks), dim3(1024), lds, 0, iters, image_log, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)blocks * 1024 * iters;
    printf("%-28s image 2^%d x 8B: %8.3f ms  %7.1f Gop/s  %6.2f lane-ops/clk/CU\n", name, image_log, ms, ops / ms * 1e-6,
           ops / (ms * 1e-3) / (clk_ghz * 1e9) / ncu);
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", p.name, ncu, ghz);
    uint32_t *sink;
    CHECK(hipMalloc(&sink, 4));
#define R(OP, PAT, L) if (run<OP, PAT>(#OP " pattern " #PAT, L, ghz, ncu, sink)) return 1;
    R(U32, 0, 12) R(U64, 0, 12) R(F32, 0, 12) R(F64, 0, 12) R(U32_RTN, 0, 12) R(U64_RTN, 0, 12) R(PK64, 0, 12) R(RMW64, 0, 12)
    R(U32, 1, 12) R(U64, 1, 12) R(U32_RTN, 1, 12)
    R(U32, 2, 12) R(U64, 2, 12) R(F32, 2, 12) R(U32_RTN, 2, 12) R(RMW64, 2, 12)
    R(U64, 3, 12) R(U32, 3, 12)
    R(U64, 0, 10) R(U64, 0, 8)
    return 0;
}
