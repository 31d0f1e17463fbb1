// pk_hazard_probe.hip -- does a consumer that follows a packed-f32 instruction at the distance LLVM's hazard recogniser allows
// for ANY vector instruction read the packed result correctly on gfx950?  (run ONCE: the loop inside is the soak.)
//   hipcc --offload-arch=gfx950 -O2 -o pk_hazard_probe pk_hazard_probe.hip && ./pk_hazard_probe
// Producer (explicit registers, no compiler in between):  PK: v_pk_mul_f32 v[200:201] / v[202:203];  SC: four v_mul_f32 (control)
// Gap: GAP wait states (s_nop GAP-1; 0 = back to back).  LLVM pads "VALU writes VGPR -> MFMA reads it" to 2 (s_waitcnt / s_nop).
// Consumer: MC = v_mfma_f32_16x16x4_f32 with the products as its C operand and A = B = 0 (D must equal C);
//           MA = the same MFMA with the products as A and B = 1 for k = 0 only ... kept simple: MC, ST (global store), PF (v_pk_fma_f32).
// 2 waves per SIMD (512 threads x 256 blocks, 160 KB LDS requested so that exactly one block sits on a CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define STR2(x) #x
#define STR(x) STR2(x)
#define NOP(G) "s_nop " STR(G) "\n"
#define PROD_PK "v_pk_mul_f32 v[200:201], v[196:197], v[198:199]\n v_pk_mul_f32 v[202:203], v[196:197], v[198:199]\n"
#define PROD_PKB "v_pk_mul_f32 v[200:201], v[196:197], v[198:199] op_sel:[0,1] op_sel_hi:[0,1]\n v_pk_mul_f32 v[202:203], v[196:197], v[198:199] op_sel:[0,1] op_sel_hi:[0,1]\n"
#define PROD_SC "v_mul_f32 v200, v196, v198\n v_mul_f32 v201, v197, v199\n v_mul_f32 v202, v196, v198\n v_mul_f32 v203, v197, v199\n"
#define CONS_MC "v_mfma_f32_16x16x4_f32 v[204:207], v194, v195, v[200:203]\n"
#define CONS_PF "v_pk_fma_f32 v[204:205], v[200:201], v[192:193], v[192:193] op_sel_hi:[1,1,1]\n v_pk_fma_f32 v[206:207], v[202:203], v[192:193], v[192:193] op_sel_hi:[1,1,1]\n"
#define CONS_MV "v_mov_b32 v204, v200\n v_mov_b32 v205, v201\n v_mov_b32 v206, v202\n v_mov_b32 v207, v203\n"
#define BODY(PROD, GAPTXT, CONS)                                                                                                     \
    asm volatile("v_mov_b32 v196, %4\n v_mov_b32 v197, %5\n v_mov_b32 v198, %6\n v_mov_b32 v199, %7\n"                               \
                 "v_mov_b32 v194, 0\n v_mov_b32 v195, 0\n v_mov_b32 v192, 1.0\n v_mov_b32 v193, 0\n s_nop 7\n s_nop 7\n"             \
                 PROD GAPTXT CONS "s_nop 15\n s_nop 15\n"                                                                            \
                 "v_mov_b32 %0, v204\n v_mov_b32 %1, v205\n v_mov_b32 %2, v206\n v_mov_b32 %3, v207\n"                               \
                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1), "v"(b0), "v"(b1)                                       \
                 : "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207")
// PF consumer computes pair * (1, 0) + (1, 0)... expected (p0 * 1 + 1, p1 * 0 + 0): keep the check generic through EXPECT
#define KERNEL(NAME, PROD, GAPTXT, CONS, E0, E1)                                                                                     \
    __global__ void __launch_bounds__(512) NAME(const float *in, unsigned *bad, int iters)                                           \
    {                                                                                                                                \
        extern __shared__ char pad[];                                                                                                \
        const int t = blockIdx.x * 512 + threadIdx.x;                                                                                \
        unsigned nbad = 0, lanes = 0;                                                                                                \
        for (int it = 0; it < iters; ++it) {                                                                                         \
            const float a0 = in[(t * 4 + 0 + it * 7) & 0xfffff], a1 = in[(t * 4 + 1 + it * 13) & 0xfffff];                          \
            const float b0 = in[(t * 4 + 2 + it * 3) & 0xfffff], b1 = in[(t * 4 + 3 + it * 5) & 0xfffff];                           \
            float r0, r1, r2, r3;                                                                                                    \
            BODY(PROD, GAPTXT, CONS);                                                                                                \
            const float p0 = a0 * b0, p1 = a1 * b1;                                                                                  \
            (void)p0; (void)p1;                                                                                                      \
            if (r0 != (E0) || r1 != (E1) || r2 != (E0) || r3 != (E1)) { ++nbad; lanes |= 1u << ((threadIdx.x & 63) >> 4); }          \
        }                                                                                                                            \
        if (nbad) { atomicAdd(&bad[0], nbad); atomicOr(&bad[1], lanes); }                                                            \
        if (pad[threadIdx.x] == 77) bad[2] = 1;                                                                                      \
    }
#define GAPS(X) X(0, "") X(1, NOP(0)) X(2, NOP(1)) X(3, NOP(2)) X(4, NOP(3)) X(6, NOP(5)) X(8, NOP(7))
#define MK(G, TXT)                                                                 \
    KERNEL(k_pk_mc_##G, PROD_PK, TXT, CONS_MC, p0, p1)                             \
    KERNEL(k_pkb_mc_##G, PROD_PKB, TXT, CONS_MC, a0 * b1, a0 * b1)                 \
    KERNEL(k_sc_mc_##G, PROD_SC, TXT, CONS_MC, p0, p1)                             \
    KERNEL(k_pk_pf_##G, PROD_PK, TXT, CONS_PF, p0 + 1.0f, 0.0f)                    \
    KERNEL(k_pkb_pf_##G, PROD_PKB, TXT, CONS_PF, a0 * b1 + 1.0f, 0.0f)             \
    KERNEL(k_pk_mv_##G, PROD_PK, TXT, CONS_MV, p0, p1)
GAPS(MK)

typedef void (*kern_t)(const float *, unsigned *, int);
static void run(const char *name, kern_t k, const float *in, unsigned *bad)
{
    CK(hipMemset(bad, 0, 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k, dim3(1024), dim3(512), 160 * 1024, 0, in, bad, 256);
    CK(hipDeviceSynchronize());
    unsigned h[4];
    CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
    printf("%-14s wrong results %10u of %u  quarter-waves hit (bit q = lanes 16q..16q+15): 0x%x\n", name, h[0], 1024u * 512u * 256u, h[1]);
}
int main()
{
    float *in; unsigned *bad;
    CK(hipMalloc(&in, 4 << 20)); CK(hipMalloc(&bad, 16));
    float *h = (float *)malloc(4 << 20);
    for (int i = 0; i < 1 << 20; ++i) h[i] = 0.5f + (float)((i * 2654435761u) >> 8) * (1.0f / 16777216.0f);
    CK(hipMemcpy(in, h, 4 << 20, hipMemcpyHostToDevice));
#define RUN(G, TXT) run("pk->mfmaC g" #G, k_pk_mc_##G, in, bad); run("pkbc->mfmaC g" #G, k_pkb_mc_##G, in, bad); run("4xmul->mfmaC g" #G, k_sc_mc_##G, in, bad); \
                    run("pk->pk_fma g" #G, k_pk_pf_##G, in, bad); run("pkbc->pk_fma g" #G, k_pkb_pf_##G, in, bad); run("pk->mov g" #G, k_pk_mv_##G, in, bad);
    GAPS(RUN)
    return 0;
}
