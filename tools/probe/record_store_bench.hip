// record_store_bench.hip -- how fast can the chip take the t16s backward's record stream ALONE?
// 256 persistent workgroups x 512 threads append REC-byte records to NB x 16 ranges each through LDS cursors, exactly the access
// pattern of emit_pairs12 (scatter_common.h) without any of the kernel's arithmetic: 5.4e8 records = configs[1]'s step.
//   hipcc --offload-arch=gfx950 -O3 -o record_store_bench record_store_bench.hip && ./record_store_bench
// MODE 0: every lane a random range (fine levels); MODE 1: the 16 lanes of a group share the range (coarse levels: consecutive
// samples of a ray in one cell); MODE 2: as the kernel's mix -- 16 levels, level l shares with probability p_l (measured run
// lengths: 86 % at level 0 ... 7 % at level 15).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int REC, int MODE>
__global__ void __launch_bounds__(512) k_append(char *recs, uint32_t per_range, int nranges, int iters)
{
    extern __shared__ uint32_t cursor[];
    for (int i = threadIdx.x; i < nranges; i += 512) cursor[i] = ((uint32_t)blockIdx.x * nranges + i) * per_range;
    __syncthreads();
    const int lane = threadIdx.x & 63, q = lane >> 4;
    uint32_t rng = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        // one "level" per lane group and iteration: 4 records (the four (y,z) pairs) to 4 unrelated buckets of that level
        const int level = (4 * (it & 3) + q) & 15;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            rng = rng * 1664525u + 1013904223u;
            uint32_t r = rng >> 8;
            if (MODE == 1) r = __shfl(r, lane & 48, 64);
            if (MODE == 2) {
                const uint32_t share = (uint32_t)(220 - 13 * level);   // of 256: ~86 % at level 0 ... ~10 % at level 15
                const uint32_t lead = __shfl(r, lane & 48, 64);
                if (((rng >> 4) & 255u) < share) r = lead;
            }
            const int bin = level * (nranges / 16) + (int)(r % (uint32_t)(nranges / 16));
            const uint32_t pos = atomicAdd(&cursor[bin], 1u);
            if (REC == 12) {
                struct __attribute__((aligned(4))) W3 { uint32_t a, b, c; };
                *reinterpret_cast<W3 *>(recs + (size_t)pos * 12) = W3{ r, rng, pos };
            } else if (REC == 16) {
                *reinterpret_cast<uint4 *>(recs + (size_t)pos * 16) = make_uint4(r, rng, pos, 0);
            } else {
                *reinterpret_cast<uint2 *>(recs + (size_t)pos * 8) = make_uint2(r, rng);
            }
        }
    }
}

template <int REC, int MODE>
static void run(char *buf, int nb, int wgs, const char *what)
{
    const int nranges = 16 * nb;
    const size_t total = 536870912ull;                       // 5.4e8 records
    const int iters = (int)(total / ((size_t)wgs * 512 * 4));
    // a range may receive up to ~3x its mean share under MODE 1 / 2: size for 4x
    const uint32_t per_range = (uint32_t)(4 * total / ((size_t)wgs * nranges));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_append<REC, MODE>), dim3(wgs), dim3(512), nranges * 4, 0, buf, per_range, nranges, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-58s rec %2d B  buckets/level %3d  wgs %3d: %7.3f ms  (%.2f G records/s, %.2f TB/s payload)\n", what, REC, nb, wgs, best,
           total / best / 1e6, total * (double)REC / best / 1e9);
}

int main()
{
    char *buf;
    const size_t bytes = 4ull * 536870912ull * 16ull;        // 4x slack x 16 B: 34 GB
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 0, bytes));
    run<12, 0>(buf, 64, 256, "random range per lane (fine levels)");
    run<12, 1>(buf, 64, 256, "16-lane groups share a range (coarse levels)");
    run<12, 2>(buf, 64, 256, "the kernel's mix of levels");
    run<16, 2>(buf, 64, 256, "the kernel's mix, 16-byte records");
    run<8, 2>(buf, 64, 256, "the kernel's mix, 8-byte records");
    run<12, 2>(buf, 16, 256, "the kernel's mix, 16 buckets per level");
    run<12, 2>(buf, 256, 256, "the kernel's mix, 256 buckets per level");
    run<12, 2>(buf, 64, 128, "the kernel's mix, 128 workgroups");
    run<12, 2>(buf, 64, 64, "the kernel's mix, 64 workgroups");
    return 0;
}
