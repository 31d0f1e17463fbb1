// h3_selftest.hip -- device self-test of the split-f16 backward primitives (render_h3.h): the transposed image
// reads (dX = W^T dY) and the staged sample-reduction products (dW = dY X^T).  Test infrastructure entry point
// (tests/test_gpu_parity.py); not on the product path.
#include "render_h3.h"

using namespace scanerf;

namespace {

// One wave.  packed: a decoder workspace (scanerf_pack_decoder); dy [64][32] f32 (unit-major), xin [64][32] f32.
//   out_dx [2 layers][64][32]: layer 0 = Spatial_MLP.mlp.2 (W1^T dy), layer 1 = Directional_MLP.mlp.0 H-part (rows 0..31)
//   out_dw [64][64]: dy x^T
//   out_rs [64]: row sums of dy
__global__ void __launch_bounds__(64) k_h3_selftest(const float *packed, const float *dy, const float *xin, float *out_dx,
                                                    float *out_dw, float *out_rs)
{
    __shared__ __attribute__((aligned(16))) char lds[H3_BYTES + 2 * H3_STAGE_MAT];
    {
        const float4 *src = reinterpret_cast<const float4 *>(packed + PK_TOTAL);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < H3_BYTES / 16; i += 64) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x, sl = lane & 31, h = lane >> 5;
    const H3Lane L = h3_lane(lane);
    char *stY = lds + H3_BYTES, *stX = stY + H3_STAGE_MAT;
    // registers in accumulator layout: block b, register g = unit 32b + nmap(g,h), column = sample sl
    v16f Y[2], X[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            Y[b][g] = dy[(32 * b + nmap(g, h)) * 32 + sl];
            X[b][g] = xin[(32 * b + nmap(g, h)) * 32 + sl];
        }
    const HL2 ys[2] = { split16(Y[0]), split16(Y[1]) }, xs[2] = { split16(X[0]), split16(X[1]) };
    const v16f zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    // ---- transposed products
    for (int layer = 0; layer < 2; ++layer) {
        const int base = layer == 0 ? H3_L1 : H3_D0, ksb = layer == 0 ? 4 : 3, nib = layer == 0 ? 2 : 1;
        for (int ib = 0; ib < nib; ++ib) {
            v16f acc = zero;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int tq = 0; tq < 2; ++tq) {
                    const A2 a = h3_lda_T(lds, base, ksb, nb, tq, ib, L);
                    H3_REGION_BEGIN();
                    mma3(acc, a, ys[nb].t[tq]);
                    H3_REGION_END();
                }
#pragma unroll
            for (int g = 0; g < 16; ++g) out_dx[(layer * 64 + 32 * ib + nmap(g, h)) * 32 + sl] = acc[g];
        }
    }
    // ---- staged products
    h3_stage_put(stY, L, 0, ys[0]);
    h3_stage_put(stY, L, 1, ys[1]);
    h3_stage_put(stX, L, 0, xs[0]);
    h3_stage_put(stX, L, 1, xs[1]);
    __syncthreads();
    for (int nb = 0; nb < 2; ++nb) {
        float rs = 0.0f;
        for (int kb = 0; kb < 2; ++kb) {
            v16f acc = zero;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const HL a = h3_stage_get(stY, L.g0[nb], L.g1[nb], t), b = h3_stage_get(stX, L.g0[kb], L.g1[kb], t);
                if (kb == 0) rs = h3_sum8(a, rs);
                H3_REGION_BEGIN();
                mma3(acc, a, b);
                H3_REGION_END();
            }
            // acc: row n = 32nb + nmap(g,h), column k = 32kb + sl
#pragma unroll
            for (int g = 0; g < 16; ++g) out_dw[(32 * nb + nmap(g, h)) * 64 + 32 * kb + sl] = acc[g];
        }
        rs += __shfl_xor(rs, 32, 64);
        if (h == 0) out_rs[32 * nb + sl] = rs;
    }
}

// ~300 KB of straight-line code on every CU (no LDS, three registers): after it no other kernel's instructions are left in the
// 64 KB instruction caches.  Test infrastructure: launch-to-launch comparisons with this in between are the screen for faults
// that only show when a kernel starts on cold instruction caches (tools/fault_probe.py).
__global__ void __launch_bounds__(512, 2) k_icache_sweep(float *sink, int never)
{
    float a = (float)threadIdx.x, b = 1.0f;
    asm volatile(".rept 75000\n v_add_f32 %0, %0, %1\n .endr\n" : "+v"(a) : "v"(b));
    if (never) sink[threadIdx.x] = a;
}

// Scattered 8-byte loads over a table far larger than the L2s: every lane-load is (nearly) one L2 miss = one 64-byte request to
// the fabric, so loads / second IS the chip's L2 <-> fabric request rate for this access pattern -- the ceiling the forward's
// gathers and the backward's record stream run against.  bench.py times it live next to the step.
__global__ void __launch_bounds__(512) k_gather_rate_probe(const uint2 *__restrict__ table, uint32_t mask, int iters, uint32_t *sink)
{
    uint32_t s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 0x9e3779b9u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it += 8) {
        uint2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {   // eight independent loads in flight per lane
            s = s * 1664525u + 1013904223u;
            v[u] = table[(s >> 4) & mask];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x + v[u].y;
    }
    if (acc == 0x12345678u) sink[0] = acc;   // (keeps the loads; never true for a zero-filled or random table in practice)
}

}  // namespace

// Test / measurement infrastructure: `loads_per_thread` (multiple of 8) scattered 8-byte loads by each of blocks x 512 threads over
// table[0 .. entries) (entries a power of two, 8 bytes each).  The caller times the launch: blocks * 512 * loads_per_thread loads.
SCANERF_API int scanerf_gather_rate_probe(const void *table, long long entries, int blocks, int loads_per_thread, unsigned *sink,
                                          scanerf_stream_t stream)
{
    SCANERF_REQUIRE(table && sink && entries >= 2 && (entries & (entries - 1)) == 0 && entries <= (1ll << 32), "gather_rate_probe: entries=%lld must be a power of two", entries);
    SCANERF_REQUIRE(blocks >= 1 && loads_per_thread >= 8 && loads_per_thread % 8 == 0, "gather_rate_probe: blocks=%d loads_per_thread=%d", blocks, loads_per_thread);
    hipLaunchKernelGGL(k_gather_rate_probe, dim3(blocks), dim3(512), 0, (hipStream_t)stream, static_cast<const uint2 *>(table),
                       (uint32_t)(entries - 1), loads_per_thread, sink);
    return check_launch("gather_rate_probe");
}

SCANERF_API int scanerf_icache_sweep(scanerf_stream_t stream)
{
    hipLaunchKernelGGL(k_icache_sweep, dim3(2 * kNumCU), dim3(512), 0, (hipStream_t)stream, nullptr, 0);
    return check_launch("icache_sweep");
}

SCANERF_API int scanerf_h3_selftest(const float *packed, const float *dy, const float *x, float *out_dx, float *out_dw,
                                    float *out_rs, scanerf_stream_t stream)
{
    SCANERF_REQUIRE(packed && dy && x && out_dx && out_dw && out_rs, "h3_selftest: null pointer");
    hipLaunchKernelGGL(k_h3_selftest, dim3(1), dim3(64), 0, (hipStream_t)stream, packed, dy, x, out_dx, out_dw, out_rs);
    return check_launch("h3_selftest");
}
